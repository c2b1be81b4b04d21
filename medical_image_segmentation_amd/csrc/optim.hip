// clip_grad_norm_ + AdamW over every parameter tensor in two launches (the reference runs
// torch.nn.utils.clip_grad_norm_ and torch.optim.AdamW.step: utils/train_and_eval.py:115-118,
// train.py:221-226 — a dozen multi-tensor launches over 74 small tensors).
//
// The tensors are described by a device table of mis_param_desc; work is cut into chunks of
// OPT_CHUNK elements, one workgroup per chunk, chunk -> tensor through a prefix array.
//   launch 1: per-chunk sum of squares of the gradients (double partial per chunk)
//   launch 2: every workgroup sums the partials in the same fixed order (so all see the bitwise
//             same norm), forms the clip coefficient, and updates its chunk.
// HBM-bound: reads g (x2), p, m, v; writes p, m, v (and the clipped g) once.
#include "common.h"

namespace {

constexpr int OPT_THREADS = 256;
constexpr int OPT_CHUNK = 4096;     // elements per workgroup

struct Desc {
  float* p;
  float* g;
  float* m;
  float* v;
  int64_t n;
};

__device__ __forceinline__ int find_tensor(const int* __restrict__ prefix, int n_tensors, int chunk) {
  int lo = 0, hi = n_tensors;            // prefix[lo] <= chunk < prefix[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (prefix[mid] <= chunk) lo = mid; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ double block_sum_double(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) red[wid] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < OPT_THREADS / 64; w++) s += red[w];
  __syncthreads();
  return s;
}

__global__ __launch_bounds__(OPT_THREADS) void grad_sqsum_kernel(const Desc* __restrict__ descs,
                                                                 const int* __restrict__ prefix, int n_tensors,
                                                                 double* __restrict__ partials) {
  __shared__ double red[OPT_THREADS / 64];
  const int chunk = blockIdx.x;
  const int t = find_tensor(prefix, n_tensors, chunk);
  const Desc d = descs[t];
  const int64_t e0 = (int64_t)(chunk - prefix[t]) * OPT_CHUNK;
  int64_t e1 = e0 + OPT_CHUNK;
  if (e1 > d.n) e1 = d.n;
  float s = 0.f;
  for (int64_t i = e0 + threadIdx.x; i < e1; i += OPT_THREADS) {
    const float g = d.g[i];
    s += g * g;
  }
  const double tot = block_sum_double((double)s, red);
  if (threadIdx.x == 0) partials[chunk] = tot;
}

__global__ __launch_bounds__(OPT_THREADS) void clip_adamw_kernel(const Desc* __restrict__ descs,
                                                                 const int* __restrict__ prefix, int n_tensors,
                                                                 int chunk0, const double* __restrict__ partials,
                                                                 int n_partials, float max_norm, float decay,
                                                                 float beta1, float omb1, float beta2, float omb2,
                                                                 float eps, float step_size,
                                                                 float bc2_sqrt, int write_grads,
                                                                 float* __restrict__ total_norm_out,
                                                                 const float* __restrict__ dyn) {
  __shared__ double red[OPT_THREADS / 64];
  if (dyn) {            // step-dependent constants from device memory (a captured launch replayed every step)
    decay = dyn[0];
    step_size = dyn[1];
    bc2_sqrt = dyn[2];
  }
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_partials; i += OPT_THREADS) acc += partials[i];
  const double sq = block_sum_double(acc, red);
  const float total_norm = (float)sqrt(sq);
  float coef = 1.f;
  if (max_norm > 0.f) {
    coef = max_norm / (total_norm + 1e-6f);       // clip_grad_norm_: clamp(max_norm / (norm + 1e-6), max=1)
    if (coef > 1.f) coef = 1.f;
  }
  const int chunk = chunk0 + blockIdx.x;
  if (chunk == 0 && threadIdx.x == 0 && total_norm_out) total_norm_out[0] = total_norm;
  const int t = find_tensor(prefix, n_tensors, chunk);
  const Desc d = descs[t];
  const int64_t e0 = (int64_t)(chunk - prefix[t]) * OPT_CHUNK;
  int64_t e1 = e0 + OPT_CHUNK;
  if (e1 > d.n) e1 = d.n;
  for (int64_t i = e0 + threadIdx.x; i < e1; i += OPT_THREADS) {
    const float g = d.g[i] * coef;
    float p = d.p[i], m = d.m[i], v = d.v[i];
    p = p * decay;                                       // param.mul_(1 - lr*weight_decay)
    m = m + (g - m) * omb1;                              // exp_avg.lerp_(grad, 1 - beta1)
    v = v * beta2 + omb2 * g * g;                        // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - step_size * (m / denom);                     // param.addcdiv_(exp_avg, denom, value=-step_size)
    d.p[i] = p;
    d.m[i] = m;
    d.v[i] = v;
    if (write_grads) d.g[i] = g;
  }
}

}  // namespace

extern "C" {

size_t mis_clip_adamw_workspace_bytes(int n_chunks) {
  return n_chunks < 1 ? 0 : align256((size_t)n_chunks * sizeof(double));
}

int mis_grad_sqsum(const void* descs, const int* chunk_prefix, int n_tensors, int n_chunks, void* workspace,
                   mis_stream_t stream) {
  MIS_REQUIRE(descs && chunk_prefix && workspace && n_tensors >= 1 && n_chunks >= 1, "grad_sqsum: bad arguments");
  grad_sqsum_kernel<<<n_chunks, OPT_THREADS, 0, (hipStream_t)stream>>>((const Desc*)descs, chunk_prefix, n_tensors,
                                                                       (double*)workspace);
  MIS_CHECK_LAUNCH("grad_sqsum_kernel");
  return MIS_OK;
}

int mis_clip_adamw_step(const void* descs, const int* chunk_prefix, int n_tensors, int chunk_begin, int chunk_end,
                        int n_chunks, const void* workspace, float max_norm, double lr, double beta1, double beta2,
                        double eps, double weight_decay, int64_t step, int write_clipped_grads, float* total_norm,
                        mis_stream_t stream) {
  MIS_REQUIRE(descs && chunk_prefix && workspace && n_tensors >= 1, "clip_adamw_step: null pointer");
  MIS_REQUIRE(0 <= chunk_begin && chunk_begin < chunk_end && chunk_end <= n_chunks, "clip_adamw_step: bad chunk range");
  MIS_REQUIRE(step >= 1 && beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1., "clip_adamw_step: bad step/betas");
  // hyper-parameters arrive as the doubles Python holds; every derived constant is formed in
  // double and rounded once, as torch does when it passes Python scalars to its kernels
  const double b1d = beta1, b2d = beta2;
  const double bc1 = 1.0 - pow(b1d, (double)step), bc2 = 1.0 - pow(b2d, (double)step);
  clip_adamw_kernel<<<chunk_end - chunk_begin, OPT_THREADS, 0, (hipStream_t)stream>>>(
      (const Desc*)descs, chunk_prefix, n_tensors, chunk_begin, (const double*)workspace, n_chunks, max_norm,
      (float)(1.0 - (double)lr * (double)weight_decay), (float)beta1, (float)(1.0 - b1d), (float)beta2, (float)(1.0 - b2d), (float)eps,
      (float)((double)lr / bc1), (float)sqrt(bc2), write_clipped_grads, total_norm, nullptr);
  MIS_CHECK_LAUNCH("clip_adamw_kernel");
  return MIS_OK;
}

void mis_adamw_step_constants(double lr, double beta1, double beta2, double weight_decay, int64_t step, float* out3) {
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  out3[0] = (float)(1.0 - lr * weight_decay);
  out3[1] = (float)(lr / bc1);
  out3[2] = (float)sqrt(bc2);
}

int mis_clip_adamw_step_dyn(const void* descs, const int* chunk_prefix, int n_tensors, int chunk_begin, int chunk_end,
                            int n_chunks, const void* workspace, float max_norm, double beta1, double beta2, double eps,
                            const float* step_constants, int write_clipped_grads, float* total_norm,
                            mis_stream_t stream) {
  MIS_REQUIRE(descs && chunk_prefix && workspace && step_constants && n_tensors >= 1, "clip_adamw_step_dyn: null pointer");
  MIS_REQUIRE(0 <= chunk_begin && chunk_begin < chunk_end && chunk_end <= n_chunks, "clip_adamw_step_dyn: bad chunk range");
  MIS_REQUIRE(beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1., "clip_adamw_step_dyn: bad betas");
  clip_adamw_kernel<<<chunk_end - chunk_begin, OPT_THREADS, 0, (hipStream_t)stream>>>(
      (const Desc*)descs, chunk_prefix, n_tensors, chunk_begin, (const double*)workspace, n_chunks, max_norm, 1.f,
      (float)beta1, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, 0.f, 1.f, write_clipped_grads,
      total_norm, step_constants);
  MIS_CHECK_LAUNCH("clip_adamw_kernel");
  return MIS_OK;
}

}  // extern "C"
