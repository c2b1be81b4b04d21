// Internal (non-exported) conv entry points shared between translation units.
#pragma once
#include "common.h"

size_t conv3x3_direct_workspace_bytes(int64_t n_pix, int Cout);
// Rows of per-workgroup partial statistics [rows][2][Cout] a forward launch leaves at the start
// of the workspace (reduced into `stats` unless stats == workspace).
int conv3x3_direct_stats_rows(int64_t n_pix, int B, int H, int W, int Cin, int Cout);
int conv3x3_mfma_stats_rows(int dtype, int B, int H, int W, int Cin, int Cout, int dil = 1);
// `in_aff` / `x_aff`: optional [2][Cin] (scale, shift) of a lazily stored input activation:
// the kernel reads relu(x*scale + shift) instead of x (padding stays zero).
int conv3x3_direct_fwd(const void* x, const float* in_aff, const void* wp, const float* bias, int dtype, int B, int H,
                       int W, int Cin, int Cout, void* z, float* stats, void* workspace, hipStream_t st);
size_t conv3x3_direct_wgrad_workspace_bytes(int64_t n_pix, int Cin, int Cout);
int conv3x3_direct_wgrad(const void* x, const float* x_aff, const void* dz, int dtype, int B, int H, int W, int Cin,
                         int Cout, float* dw, void* workspace, hipStream_t st);

size_t conv3x3_direct_packed_bytes(int dtype, int Cin, int Cout);
int conv3x3_direct_pack(const float* w, int Cw_out, int Cw_in, int transposed, int dtype, void* out, hipStream_t st);

// Input of a conv: one tensor (xb == nullptr, Cb == 0), or the channel concatenation [xa | xb]
// of two that is never materialised.  Each tensor is dense channels-last with its own channel
// count and its own optional lazy-activation affine ([2][C]).
struct ConvIn {
  const void* xa; const float* affa; int Ca;
  const void* xb; const float* affb; int Cb;
  int channels() const { return Ca + Cb; }
};
// Output of a conv: one tensor (zb == nullptr, Cb == 0), or split by channel into two dense
// tensors: channels [0, Ca) -> za, [Ca, Ca+Cb) -> zb.
struct ConvOut {
  void* za; int Ca;
  void* zb; int Cb;
  int channels() const { return Ca + Cb; }
};

// Matrix-core path (conv_mfma.hip).  *_supported tell whether the shape is taken.
bool conv3x3_mfma_supported(int dtype, int B, int H, int W, int Cin, int Cout);
void conv3x3_mfma_set_launch(int fwd_workgroups, int wgrad_workgroups);   // 0 = sized by the planners
bool conv3x3_mfma_cat_supported(int dtype, int Ca, int Cb, int Cout);          // two-tensor input
bool conv3x3_mfma_split_supported(int dtype, int Cin, int Ca, int Cb);         // two-tensor output
size_t conv3x3_mfma_workspace_bytes(int dtype, int B, int H, int W, int Cin, int Cout);
// bnz / bn_aff (optional, single-tensor output only): the conv is a data gradient whose output is the
// gradient of the lazy activation relu(bn(bnz)); `stats` then receives that BatchNorm's backward sums
// (sum dy, sum dy*xhat) instead of (sum z, sum z^2).  bn_aff: [4][Cout] = scale, shift, mean, rstd.
int conv3x3_mfma_fwd(const ConvIn& in, const void* wp, const float* bias, int dtype, int B, int H, int W,
                     const ConvOut& out, float* stats, void* workspace, hipStream_t st, const void* bnz = nullptr,
                     const float* bn_aff = nullptr, int dil = 1);      // dil: tap dilation 1..3 (padding = dil)
size_t conv3x3_mfma_packed_bytes(int dtype, int Cin, int Cout);
int conv3x3_mfma_pack(const float* w, int Cw_out, int Cw_in, int transposed, int dtype, void* out, hipStream_t st);
// both images (forward and transposed = data-gradient operand) of one weight tensor in one launch
int conv3x3_mfma_pack_pair(const float* w, int Cw_out, int Cw_in, int dtype, void* out_fwd, void* out_t, hipStream_t st);
// several weight tensors, both images each, in one launch (jobs: device array)
int conv3x3_mfma_pack_job(const float* w, int Cw_out, int Cw_in, int dtype, void* out_fwd, void* out_t,
                          int first_block, mis_pack_job* job);
int conv3x3_mfma_pack_multi(const mis_pack_job* jobs_dev, int n_jobs, int total_blocks, int dtype, hipStream_t st);
bool conv3x3_wgrad_mfma_supported(int dtype, int B, int H, int W, int Cin, int Cout);
size_t conv3x3_wgrad_mfma_workspace_bytes(int dtype, int B, int H, int W, int Cin, int Cout);
bool conv3x3_wgrad_mfma_cat_supported(int dtype, int B, int H, int W, int Ca, int Cb, int Cout);
// Optional rider of the weight-gradient reduction: dbias[c] = sum over n_rows rows [n_rows][Cout] of
// partial conv-bias gradients (left by the BatchNorm backward), reduced by extra workgroups of
// the same launch.
struct BiasRows {
  const float* rows; int n_rows; float* dbias;
};
int conv3x3_wgrad_mfma(const ConvIn& in, const void* dz, int dtype, int B, int H, int W, int Cout, float* dw,
                       const BiasRows& bias, void* workspace, hipStream_t st, int dil = 1);
// the same without the reduction: slabs [n_slabs][9][Cin][Cout] only, and their reduction for several layers at once
int conv3x3_wgrad_mfma_slabs(const ConvIn& in, const void* dz, int dtype, int B, int H, int W, int Cout, void* slabs,
                             int* n_slabs, hipStream_t st, int dil = 1);
int conv3x3_wgrad_reduce_multi(const mis_wgrad_reduce_job* jobs, int n_jobs, hipStream_t st);
