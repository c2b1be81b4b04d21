// Public 3x3 convolution entry points: pick the matrix-core kernel when the shape
// allows it, else the vector-ALU kernel.  MIS_CONV_FORCE_DIRECT=1 in the environment
// forces the vector-ALU kernels (used by the tests to cross-check the two paths).
#include "conv_internal.h"
#include "reduce.h"
#include <stdlib.h>

static int g_conv_path = -1;   // -1: read MIS_CONV_FORCE_DIRECT once; 0: auto; 1: vector-ALU kernels only

static bool force_direct() {
  if (g_conv_path < 0) {
    const char* e = getenv("MIS_CONV_FORCE_DIRECT");
    g_conv_path = (e && e[0] == '1') ? 1 : 0;
  }
  return g_conv_path == 1;
}

// The packed layout depends only on (dtype, channel counts) so that pack and conv agree.
static bool use_mfma(int dtype, int Cin, int Cout) {
  return !force_direct() && conv3x3_mfma_supported(dtype, 1, 16, 16, Cin, Cout);
}

extern "C" {

int mis_set_conv_path(int mode) {
  if (mode != 0 && mode != 1) {
    mis_set_error("set_conv_path: mode must be 0 (auto) or 1 (vector-ALU only)");
    return MIS_ERR_ARG;
  }
  g_conv_path = mode;
  return MIS_OK;
}

int mis_set_conv_launch(int fwd_workgroups, int wgrad_workgroups) {
  if (fwd_workgroups < 0 || fwd_workgroups > 8192 || wgrad_workgroups < 0 || wgrad_workgroups > 8192) {
    mis_set_error("set_conv_launch: workgroup counts must lie in 0..8192 (0 = sized by the library)");
    return MIS_ERR_ARG;
  }
  conv3x3_mfma_set_launch(fwd_workgroups, wgrad_workgroups);
  return MIS_OK;
}

size_t mis_conv3x3_packed_weight_bytes(int Cout_w, int Cin_w, int transposed, int dtype) {
  if (Cout_w < 1 || Cin_w < 1 || (dtype != MIS_F32 && dtype != MIS_BF16)) return 0;
  const int Cin = transposed ? Cout_w : Cin_w, Cout = transposed ? Cin_w : Cout_w;
  return use_mfma(dtype, Cin, Cout) ? conv3x3_mfma_packed_bytes(dtype, Cin, Cout)
                                    : conv3x3_direct_packed_bytes(dtype, Cin, Cout);
}

int mis_pack_conv3x3_weights(const float* w, int Cout_w, int Cin_w, int transposed, int dtype, void* out,
                             mis_stream_t stream) {
  MIS_REQUIRE(w && out && Cout_w >= 1 && Cin_w >= 1, "pack_conv3x3_weights: bad arguments");
  MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, "pack_conv3x3_weights: bad dtype %d", dtype);
  const int Cin = transposed ? Cout_w : Cin_w, Cout = transposed ? Cin_w : Cout_w;
  if (use_mfma(dtype, Cin, Cout))
    return conv3x3_mfma_pack(w, Cout_w, Cin_w, transposed ? 1 : 0, dtype, out, (hipStream_t)stream);
  return conv3x3_direct_pack(w, Cout_w, Cin_w, transposed ? 1 : 0, dtype, out, (hipStream_t)stream);
}

int mis_pack_conv3x3_weights_pair(const float* w, int Cout_w, int Cin_w, int dtype, void* out_fwd, void* out_t,
                                  mis_stream_t stream) {
  MIS_REQUIRE(w && out_fwd && out_t && Cout_w >= 1 && Cin_w >= 1, "pack_conv3x3_weights_pair: bad arguments");
  MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, "pack_conv3x3_weights_pair: bad dtype %d", dtype);
  if (use_mfma(dtype, Cin_w, Cout_w) && use_mfma(dtype, Cout_w, Cin_w))
    return conv3x3_mfma_pack_pair(w, Cout_w, Cin_w, dtype, out_fwd, out_t, (hipStream_t)stream);
  const int rc = mis_pack_conv3x3_weights(w, Cout_w, Cin_w, 0, dtype, out_fwd, stream);
  return rc != MIS_OK ? rc : mis_pack_conv3x3_weights(w, Cout_w, Cin_w, 1, dtype, out_t, stream);
}

int mis_pack_conv3x3_job(const float* w, int Cout_w, int Cin_w, int dtype, void* out_fwd, void* out_t, int first_block,
                         mis_pack_job* job) {
  if (!w || !out_fwd || !out_t || !job || Cout_w < 1 || Cin_w < 1 || first_block < 0 ||
      (dtype != MIS_F32 && dtype != MIS_BF16))
    return 0;
  if (!(use_mfma(dtype, Cin_w, Cout_w) && use_mfma(dtype, Cout_w, Cin_w))) return 0;
  return conv3x3_mfma_pack_job(w, Cout_w, Cin_w, dtype, out_fwd, out_t, first_block, job);
}

int mis_pack_conv3x3_weights_multi(const mis_pack_job* jobs, int n_jobs, int total_blocks, int dtype,
                                   mis_stream_t stream) {
  MIS_REQUIRE(jobs && n_jobs >= 1 && total_blocks >= 1, "pack_conv3x3_weights_multi: bad arguments");
  MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, "pack_conv3x3_weights_multi: bad dtype %d", dtype);
  return conv3x3_mfma_pack_multi(jobs, n_jobs, total_blocks, dtype, (hipStream_t)stream);
}

int mis_conv3x3_stats_rows(int dtype, int B, int H, int W, int Cin, int Cout) {
  if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1 || (dtype != MIS_F32 && dtype != MIS_BF16)) return 0;
  return use_mfma(dtype, Cin, Cout) ? conv3x3_mfma_stats_rows(dtype, B, H, W, Cin, Cout)
                                    : conv3x3_direct_stats_rows((int64_t)B * H * W, B, H, W, Cin, Cout);
}

size_t mis_conv3x3_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
  if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return 0;
  size_t a = conv3x3_direct_workspace_bytes((int64_t)B * H * W, Cout);
  size_t b = conv3x3_mfma_workspace_bytes(MIS_BF16, B, H, W, Cin, Cout);
  size_t c = conv3x3_mfma_workspace_bytes(MIS_F32, B, H, W, Cin, Cout);
  size_t m = a > b ? a : b;
  return (m > c ? m : c) + 256;
}

int mis_conv3x3_fwd(const void* x, const float* x_affine, const void* w_packed, const float* bias, int dtype, int B,
                    int H, int W, int Cin, int Cout, void* z, float* stats, void* workspace, mis_stream_t stream) {
  MIS_REQUIRE(x && w_packed && z, "conv3x3_fwd: null pointer");
  MIS_REQUIRE(!stats || workspace, "conv3x3_fwd: stats requested without a workspace");
  MIS_REQUIRE(B >= 1 && H >= 1 && W >= 1 && Cin >= 1 && Cout >= 1, "conv3x3_fwd: bad shape");
  MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, "conv3x3_fwd: bad dtype %d", dtype);
  MIS_REQUIRE((int64_t)B * H * W * (int64_t)(Cin > Cout ? Cin : Cout) < ((int64_t)1 << 40), "conv3x3_fwd: tensor too large");
  MIS_REQUIRE((int64_t)B * H * W < ((int64_t)1 << 31), "conv3x3_fwd: B*H*W must be below 2^31");
  if (use_mfma(dtype, Cin, Cout))
    return conv3x3_mfma_fwd(ConvIn{x, x_affine, Cin, nullptr, nullptr, 0}, w_packed, bias, dtype, B, H, W,
                            ConvOut{z, Cout, nullptr, 0}, stats, workspace, (hipStream_t)stream);
  return conv3x3_direct_fwd(x, x_affine, w_packed, bias, dtype, B, H, W, Cin, Cout, z, stats, workspace,
                            (hipStream_t)stream);
}

size_t mis_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
  if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return 0;
  size_t a = conv3x3_direct_wgrad_workspace_bytes((int64_t)B * H * W, Cin, Cout);
  size_t b = conv3x3_wgrad_mfma_workspace_bytes(MIS_BF16, B, H, W, Cin, Cout);
  size_t c = conv3x3_wgrad_mfma_workspace_bytes(MIS_F32, B, H, W, Cin, Cout);
  size_t m = a > b ? a : b;
  return (m > c ? m : c) + 256;
}

int mis_conv3x3_dgrad_bn_sums_supported(int dtype, int B, int H, int W, int Cin, int Cout) {
  if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1 || (dtype != MIS_F32 && dtype != MIS_BF16)) return 0;
  return (use_mfma(dtype, Cin, Cout) && conv3x3_mfma_supported(dtype, B, H, W, Cin, Cout)) ? 1 : 0;
}

int mis_conv3x3_dgrad_bn_sums(const void* dz, const void* w_packed_t, int dtype, int B, int H, int W, int Cin, int Cout,
                              void* da, const void* z_prev, const float* prev_affine4, float* bn_sums,
                              void* workspace, mis_stream_t stream) {
  MIS_REQUIRE(dz && w_packed_t && da && z_prev && prev_affine4 && bn_sums && workspace,
              "conv3x3_dgrad_bn_sums: null pointer");
  MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, "conv3x3_dgrad_bn_sums: bad dtype %d", dtype);
  if (!mis_conv3x3_dgrad_bn_sums_supported(dtype, B, H, W, Cin, Cout)) {
    mis_set_error("conv3x3_dgrad_bn_sums: shape %d -> %d not taken by the matrix-core kernels", Cin, Cout);
    return MIS_ERR_UNSUPPORTED;
  }
  return conv3x3_mfma_fwd(ConvIn{dz, nullptr, Cin, nullptr, nullptr, 0}, w_packed_t, nullptr, dtype, B, H, W,
                          ConvOut{da, Cout, nullptr, 0}, bn_sums, workspace, (hipStream_t)stream, z_prev,
                          prev_affine4);
}

int mis_conv3x3_wgrad(const void* x, const float* x_affine, const void* dz, int dtype, int B, int H, int W, int Cin,
                      int Cout, float* dw, const float* bias_rows, int n_bias_rows, float* dbias, void* workspace,
                      mis_stream_t stream) {
  MIS_REQUIRE(x && dz && dw && workspace, "conv3x3_wgrad: null pointer");
  MIS_REQUIRE(!dbias || (bias_rows && n_bias_rows >= 1), "conv3x3_wgrad: dbias needs its partial rows");
  MIS_REQUIRE(B >= 1 && H >= 1 && W >= 1 && Cin >= 1 && Cout >= 1, "conv3x3_wgrad: bad shape");
  MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, "conv3x3_wgrad: bad dtype %d", dtype);
  if (!force_direct() && workspace && conv3x3_wgrad_mfma_supported(dtype, B, H, W, Cin, Cout))
    return conv3x3_wgrad_mfma(ConvIn{x, x_affine, Cin, nullptr, nullptr, 0}, dz, dtype, B, H, W, Cout, dw,
                              BiasRows{bias_rows, n_bias_rows, dbias}, workspace, (hipStream_t)stream);
  if (dbias) {
    launch_reduce_partials(bias_rows, n_bias_rows, Cout, dbias, (hipStream_t)stream);
    MIS_CHECK_LAUNCH("reduce_partials_kernel");
  }
  return conv3x3_direct_wgrad(x, x_affine, dz, dtype, B, H, W, Cin, Cout, dw, workspace, (hipStream_t)stream);
}

int mis_conv3x3_wgrad_slabs_supported(int dtype, int B, int H, int W, int Ca, int Cb, int Cout, int dilation) {
  if (B < 1 || H < 1 || W < 1 || Ca < 1 || Cb < 0 || Cout < 1 || (dtype != MIS_F32 && dtype != MIS_BF16)) return 0;
  if (dilation < 0 || dilation > 3 || (Cb > 0 && dilation != 1) || force_direct()) return 0;
  return (Cb > 0 ? conv3x3_wgrad_mfma_cat_supported(dtype, B, H, W, Ca, Cb, Cout)
                 : conv3x3_wgrad_mfma_supported(dtype, B, H, W, Ca, Cout)) ? 1 : 0;
}

int mis_conv3x3_wgrad_slabs(const void* xa, const float* xa_affine, const void* xb, const float* xb_affine,
                            const void* dz, int dtype, int B, int H, int W, int Ca, int Cb, int Cout, int dilation,
                            void* slabs, int* n_slabs, mis_stream_t stream) {
  MIS_REQUIRE(xa && dz && slabs && n_slabs, "conv3x3_wgrad_slabs: null pointer");
  MIS_REQUIRE((xb != nullptr) == (Cb > 0), "conv3x3_wgrad_slabs: xb and Cb go together");
  MIS_REQUIRE(B >= 1 && H >= 1 && W >= 1 && Ca >= 1 && Cb >= 0 && Cout >= 1, "conv3x3_wgrad_slabs: bad shape");
  MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, "conv3x3_wgrad_slabs: bad dtype %d", dtype);
  if (!mis_conv3x3_wgrad_slabs_supported(dtype, B, H, W, Ca, Cb, Cout, dilation)) {
    mis_set_error("conv3x3_wgrad_slabs: %d+%d -> %d channels, dilation %d is not taken by the matrix-core kernels",
                  Ca, Cb, Cout, dilation);
    return MIS_ERR_UNSUPPORTED;
  }
  return conv3x3_wgrad_mfma_slabs(ConvIn{xa, xa_affine, Ca, xb, xb_affine, Cb}, dz, dtype, B, H, W, Cout, slabs,
                                  n_slabs, (hipStream_t)stream, dilation);
}

int mis_wgrad_reduce_multi(const mis_wgrad_reduce_job* jobs, int n_jobs, mis_stream_t stream) {
  MIS_REQUIRE(n_jobs >= 0 && (jobs || n_jobs == 0), "wgrad_reduce_multi: null job array");
  for (int j = 0; j < n_jobs; j++) {
    const mis_wgrad_reduce_job& jb = jobs[j];
    MIS_REQUIRE(jb.slabs && jb.dw && jb.n_slabs >= 1 && jb.Cin >= 1 && jb.Cout >= 1, "wgrad_reduce_multi: bad job %d", j);
    MIS_REQUIRE(!jb.dbias || (jb.bias_rows && jb.n_bias_rows >= 1), "wgrad_reduce_multi: job %d: dbias needs its partial rows", j);
  }
  if (n_jobs == 0) return MIS_OK;
  return conv3x3_wgrad_reduce_multi(jobs, n_jobs, (hipStream_t)stream);
}

int mis_conv3x3_dilated_supported(int dtype, int B, int H, int W, int Cin, int Cout, int dilation) {
  if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1 || (dtype != MIS_F32 && dtype != MIS_BF16)) return 0;
  if (dilation < 0 || dilation > 3 || force_direct()) return 0;      // 0: only the centre tap is non-zero (a 1x1 conv)
  return (use_mfma(dtype, Cin, Cout) && use_mfma(dtype, Cout, Cin) && conv3x3_mfma_supported(dtype, B, H, W, Cin, Cout) &&
          conv3x3_mfma_supported(dtype, B, H, W, Cout, Cin) && conv3x3_wgrad_mfma_supported(dtype, B, H, W, Cin, Cout))
             ? 1 : 0;
}

int mis_conv3x3_dilated_stats_rows(int dtype, int B, int H, int W, int Cin, int Cout, int dilation) {
  if (!mis_conv3x3_dilated_supported(dtype, B, H, W, Cin, Cout, dilation)) return 0;
  return conv3x3_mfma_stats_rows(dtype, B, H, W, Cin, Cout, dilation);
}

int mis_conv3x3_dilated_fwd(const void* x, const float* x_affine, const void* w_packed, const float* bias, int dtype,
                            int B, int H, int W, int Cin, int Cout, int dilation, void* z, float* stats,
                            void* workspace, mis_stream_t stream) {
  MIS_REQUIRE(x && w_packed && z, "conv3x3_dilated_fwd: null pointer");
  MIS_REQUIRE(!stats || workspace, "conv3x3_dilated_fwd: stats requested without a workspace");
  MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, "conv3x3_dilated_fwd: bad dtype %d", dtype);
  MIS_REQUIRE(B >= 1 && H >= 1 && W >= 1 && Cin >= 1 && Cout >= 1 && (int64_t)B * H * W < ((int64_t)1 << 31),
              "conv3x3_dilated_fwd: bad shape");
  if (dilation < 0 || dilation > 3 || force_direct() || !use_mfma(dtype, Cin, Cout)) {
    mis_set_error("conv3x3_dilated_fwd: %d -> %d channels with dilation %d is not taken by the matrix-core kernels",
                  Cin, Cout, dilation);
    return MIS_ERR_UNSUPPORTED;
  }
  return conv3x3_mfma_fwd(ConvIn{x, x_affine, Cin, nullptr, nullptr, 0}, w_packed, bias, dtype, B, H, W,
                          ConvOut{z, Cout, nullptr, 0}, stats, workspace, (hipStream_t)stream, nullptr, nullptr,
                          dilation);
}

int mis_conv3x3_dilated_wgrad(const void* x, const float* x_affine, const void* dz, int dtype, int B, int H, int W,
                              int Cin, int Cout, int dilation, float* dw, const float* bias_rows, int n_bias_rows,
                              float* dbias, void* workspace, mis_stream_t stream) {
  MIS_REQUIRE(x && dz && dw && workspace, "conv3x3_dilated_wgrad: null pointer");
  MIS_REQUIRE(!dbias || (bias_rows && n_bias_rows >= 1), "conv3x3_dilated_wgrad: dbias needs its partial rows");
  MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, "conv3x3_dilated_wgrad: bad dtype %d", dtype);
  if (dilation < 0 || dilation > 3 || force_direct() || !conv3x3_wgrad_mfma_supported(dtype, B, H, W, Cin, Cout)) {
    mis_set_error("conv3x3_dilated_wgrad: %d -> %d channels with dilation %d is not taken by the matrix-core kernels",
                  Cin, Cout, dilation);
    return MIS_ERR_UNSUPPORTED;
  }
  return conv3x3_wgrad_mfma(ConvIn{x, x_affine, Cin, nullptr, nullptr, 0}, dz, dtype, B, H, W, Cout, dw,
                            BiasRows{bias_rows, n_bias_rows, dbias}, workspace, (hipStream_t)stream, dilation);
}

int mis_conv3x3_cat_supported(int dtype, int B, int H, int W, int Ca, int Cb, int Cout) {
  if (B < 1 || H < 1 || W < 1 || Ca < 1 || Cb < 1 || Cout < 1 || (dtype != MIS_F32 && dtype != MIS_BF16)) return 0;
  if (force_direct()) return 0;
  return (conv3x3_mfma_cat_supported(dtype, Ca, Cb, Cout) &&            // forward: two-tensor input
          conv3x3_mfma_split_supported(dtype, Cout, Ca, Cb) &&          // data gradient: two-tensor output
          conv3x3_wgrad_mfma_cat_supported(dtype, B, H, W, Ca, Cb, Cout)) ? 1 : 0;
}

int mis_conv3x3_cat_fwd(const void* xa, const float* xa_affine, const void* xb, const float* xb_affine,
                        const void* w_packed, const float* bias, int dtype, int B, int H, int W, int Ca, int Cb,
                        int Cout, void* z, float* stats, void* workspace, mis_stream_t stream) {
  MIS_REQUIRE(xa && xb && w_packed && z, "conv3x3_cat_fwd: null pointer");
  MIS_REQUIRE(!stats || workspace, "conv3x3_cat_fwd: stats requested without a workspace");
  MIS_REQUIRE(B >= 1 && H >= 1 && W >= 1 && Ca >= 1 && Cb >= 1 && Cout >= 1, "conv3x3_cat_fwd: bad shape");
  MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, "conv3x3_cat_fwd: bad dtype %d", dtype);
  if (force_direct() || !conv3x3_mfma_cat_supported(dtype, Ca, Cb, Cout)) {
    mis_set_error("conv3x3_cat_fwd: unsupported channel split %d+%d -> %d (concatenate and use conv3x3_fwd)", Ca, Cb, Cout);
    return MIS_ERR_UNSUPPORTED;
  }
  return conv3x3_mfma_fwd(ConvIn{xa, xa_affine, Ca, xb, xb_affine, Cb}, w_packed, bias, dtype, B, H, W,
                          ConvOut{z, Cout, nullptr, 0}, stats, workspace, (hipStream_t)stream);
}

int mis_conv3x3_split_fwd(const void* x, const float* x_affine, const void* w_packed, const float* bias, int dtype,
                          int B, int H, int W, int Cin, int Ca, int Cb, void* za, void* zb, void* workspace,
                          mis_stream_t stream) {
  MIS_REQUIRE(x && w_packed && za && zb, "conv3x3_split_fwd: null pointer");
  MIS_REQUIRE(B >= 1 && H >= 1 && W >= 1 && Cin >= 1 && Ca >= 1 && Cb >= 1, "conv3x3_split_fwd: bad shape");
  MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, "conv3x3_split_fwd: bad dtype %d", dtype);
  if (force_direct() || !conv3x3_mfma_split_supported(dtype, Cin, Ca, Cb)) {
    mis_set_error("conv3x3_split_fwd: unsupported channel split %d -> %d+%d", Cin, Ca, Cb);
    return MIS_ERR_UNSUPPORTED;
  }
  return conv3x3_mfma_fwd(ConvIn{x, x_affine, Cin, nullptr, nullptr, 0}, w_packed, bias, dtype, B, H, W,
                          ConvOut{za, Ca, zb, Cb}, nullptr, workspace, (hipStream_t)stream);
}

int mis_conv3x3_cat_wgrad(const void* xa, const float* xa_affine, const void* xb, const float* xb_affine,
                          const void* dz, int dtype, int B, int H, int W, int Ca, int Cb, int Cout, float* dw,
                          const float* bias_rows, int n_bias_rows, float* dbias, void* workspace,
                          mis_stream_t stream) {
  MIS_REQUIRE(xa && xb && dz && dw && workspace, "conv3x3_cat_wgrad: null pointer");
  MIS_REQUIRE(!dbias || (bias_rows && n_bias_rows >= 1), "conv3x3_cat_wgrad: dbias needs its partial rows");
  MIS_REQUIRE(B >= 1 && H >= 1 && W >= 1 && Ca >= 1 && Cb >= 1 && Cout >= 1, "conv3x3_cat_wgrad: bad shape");
  MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, "conv3x3_cat_wgrad: bad dtype %d", dtype);
  if (force_direct() || !conv3x3_wgrad_mfma_cat_supported(dtype, B, H, W, Ca, Cb, Cout)) {
    mis_set_error("conv3x3_cat_wgrad: unsupported channel split %d+%d -> %d", Ca, Cb, Cout);
    return MIS_ERR_UNSUPPORTED;
  }
  return conv3x3_wgrad_mfma(ConvIn{xa, xa_affine, Ca, xb, xb_affine, Cb}, dz, dtype, B, H, W, Cout, dw,
                            BiasRows{bias_rows, n_bias_rows, dbias}, workspace, (hipStream_t)stream);
}

}  // extern "C"
