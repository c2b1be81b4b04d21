// BatchNorm2d (training statistics) + ReLU around the conv kernels, and the per-channel
// reductions they need.  The conv kernel emits sum(z) and sum(z^2); bn_finalize turns them
// into a per-channel affine (scale, shift) plus the saved (mean, rstd) and updates the
// running statistics the way nn.BatchNorm2d does (momentum, unbiased variance).
//
// Roofline: HBM for the elementwise passes.  Algorithmic bytes per element:
//   bn_relu_apply     read z + write a            = 2*sizeof(T)
//   bn_relu_bwd_reduce read z + da                = 2*sizeof(T)
//   bn_relu_bwd_dz    read z + da, write dz       = 3*sizeof(T)
#include "common.h"
#include "vec.h"
#include "reduce.h"

// 1024 threads = 32 row lanes x 32 columns: coalesced 128-byte row reads, 4 independent loads
// in flight per thread, double accumulation, fixed combination order (bitwise reproducible).
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float* __restrict__ partials, int nblk, int n,
                                                               int64_t stride, float* __restrict__ out) {
  __shared__ double red[32][33];
  reduce_rows_block(partials, nblk, n, stride, out, blockIdx.x, red);
}

__global__ __launch_bounds__(1024) void reduce_partials3_kernel(const float* __restrict__ partials, int nblk, int n,
                                                                int n0, int n1, float* __restrict__ out0,
                                                                float* __restrict__ out1, float* __restrict__ out2) {
  __shared__ double red[32][33];
  const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + c;
  double s = 0.0;
  if (j < n) {
    int i = r;
    for (; i + 96 < nblk; i += 128) {
      const float a0 = partials[(int64_t)i * n + j], a1 = partials[(int64_t)(i + 32) * n + j];
      const float a2 = partials[(int64_t)(i + 64) * n + j], a3 = partials[(int64_t)(i + 96) * n + j];
      s += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
    }
    for (; i < nblk; i += 32) s += (double)partials[(int64_t)i * n + j];
  }
  red[r][c] = s;
  __syncthreads();
  if (r == 0 && j < n) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 32; k++) t += red[k][c];
    if (j < n0) out0[j] = (float)t;
    else if (j < n0 + n1) out1[j - n0] = (float)t;
    else out2[j - n0 - n1] = (float)t;
  }
}

namespace {

constexpr int EW_THREADS = 256;

// stats: [rows][2][C] partial (sum, sum of squares); rows == 1 is the reduced form.  A workgroup
// of 1024 threads owns CL channels (8/16/32: fewer channels -> more row lanes, so narrow layers
// still spread their rows over the whole workgroup); 4 independent loads in flight per thread,
// double accumulation, fixed-order tree (bitwise reproducible).
template <int CL>
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ stats, int rows, int C,
                                                           double inv_n, double unbias,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, float momentum,
                                                           float* __restrict__ running_mean,
                                                           float* __restrict__ running_var, int64_t* __restrict__ nbt,
                                                           float* __restrict__ scale_shift,
                                                           float* __restrict__ mean_rstd) {
  constexpr int RL = 1024 / CL;
  __shared__ double red[2][RL][CL + 1];
  const int cl = threadIdx.x % CL, r = threadIdx.x / CL;
  const int c = blockIdx.x * CL + cl;
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) nbt[0] += 1;
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    const int64_t st = 2 * (int64_t)C;
    int i = r;
    for (; i + 3 * RL < rows; i += 4 * RL) {
      const float a0 = stats[i * st + c], a1 = stats[(i + RL) * st + c];
      const float a2 = stats[(i + 2 * RL) * st + c], a3 = stats[(i + 3 * RL) * st + c];
      const float b0 = stats[i * st + C + c], b1 = stats[(i + RL) * st + C + c];
      const float b2 = stats[(i + 2 * RL) * st + C + c], b3 = stats[(i + 3 * RL) * st + C + c];
      s1 += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
      s2 += ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
    }
    for (; i < rows; i += RL) {
      s1 += (double)stats[i * st + c];
      s2 += (double)stats[i * st + C + c];
    }
  }
  red[0][r][cl] = s1;
  red[1][r][cl] = s2;
  __syncthreads();
  for (int h = RL / 2; h >= 1; h >>= 1) {
    if (r < h) {
      red[0][r][cl] += red[0][r + h][cl];
      red[1][r][cl] += red[1][r + h][cl];
    }
    __syncthreads();
  }
  if (r != 0 || c >= C) return;
  s1 = red[0][0][cl];
  s2 = red[1][0][cl];
  // the conv kernels hand over float32 sums; keep the reduced sums' float32 rounding so that the
  // one-row and the many-row forms give identical results
  const double mean = (double)(float)s1 * inv_n;
  double var = (double)(float)s2 * inv_n - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[c] * rstd;
  scale_shift[c] = sc;
  scale_shift[C + c] = beta[c] - (float)mean * sc;
  if (mean_rstd) {
    mean_rstd[c] = (float)mean;
    mean_rstd[C + c] = rstd;
  }
  if (running_mean) running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
  if (running_var) running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)(var * unbias);
}

__global__ void bn_eval_affine_kernel(const float* __restrict__ rm, const float* __restrict__ rv,
                                      const float* __restrict__ gamma, const float* __restrict__ beta, int C, float eps,
                                      float* __restrict__ scale_shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] / sqrtf(rv[c] + eps);
  scale_shift[c] = sc;
  scale_shift[C + c] = beta[c] - rm[c] * sc;
}

// The launcher sizes the grid so that grid*256 is a multiple of C/V: a thread then always works
// on the same channel vector and keeps its scale/shift in registers.
template <typename T, int V>
__global__ __launch_bounds__(EW_THREADS) void bn_relu_apply_kernel(const T* __restrict__ z, T* __restrict__ a,
                                                                   int64_t n_vec, int C,
                                                                   const float* __restrict__ scale_shift) {
  const int CV = C / V;
  const int64_t i0 = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  const int c0 = (int)(i0 % CV) * V;
  float sc[V], sh[V];
#pragma unroll
  for (int k = 0; k < V; k++) {
    sc[k] = scale_shift[c0 + k];
    sh[k] = scale_shift[C + c0 + k];
  }
  for (int64_t i = i0; i < n_vec; i += (int64_t)gridDim.x * EW_THREADS) {
    float v[V];
    vload<T, V>(z + i * V, v);
#pragma unroll
    for (int k = 0; k < V; k++) v[k] = fmaxf(v[k] * sc[k] + sh[k], 0.f);
    vstore<T, V>(a + i * V, v);
  }
}

template <typename T, int V>
struct BnBwdReduce {
  static constexpr int NOUT = 2;
  const T* z;
  const T* da;
  const float* mean_rstd;
  const float* gamma;
  const float* beta;
  int C;
  int relu;                                    // 0: BatchNorm alone (no ReLU mask on the gradient)
  float mean[V], rstd[V], gam[V], bet[V];      // this thread's channel constants
  __device__ __forceinline__ void prepare(int c0) {
#pragma unroll
    for (int k = 0; k < V; k++) {
      mean[k] = mean_rstd[c0 + k];
      rstd[k] = mean_rstd[C + c0 + k];
      gam[k] = gamma[c0 + k];
      bet[k] = beta[c0 + k];
    }
  }
  __device__ __forceinline__ void load(int64_t pix, int c0, float (&acc)[2][V]) const {
    float zv[V], gv[V];
    vload<T, V>(z + pix * C + c0, zv);
    vload<T, V>(da + pix * C + c0, gv);
#pragma unroll
    for (int k = 0; k < V; k++) {
      const float xh = (zv[k] - mean[k]) * rstd[k];
      const float y = xh * gam[k] + bet[k];
      const float dy = (!relu || y > 0.f) ? gv[k] : 0.f;
      acc[0][k] += dy;
      acc[1][k] += dy * xh;
    }
  }
};

template <typename T, int V>
struct ChannelSum {
  static constexpr int NOUT = 1;
  const T* x;
  int C;
  __device__ __forceinline__ void prepare(int) {}
  __device__ __forceinline__ void load(int64_t pix, int c0, float (&acc)[1][V]) const {
    float v[V];
    vload<T, V>(x + pix * C + c0, v);
#pragma unroll
    for (int k = 0; k < V; k++) acc[0][k] += v[k];
  }
};

// dz = gamma*rstd*(dy - mean(dy) - xhat*mean(dy*xhat)).  When `dz_partials` is given the grid is
// sized so that a thread always works on the same channel vector (grid*256 is a multiple of
// C/V): it then also accumulates sum(dz) per channel — the conv-bias gradient — from the
// values it stores, and the block writes one partial row [C].
// HEAD: the incoming gradient `da` is not a tensor but the data gradient of a following 1x1 conv,
// formed here from that conv's output gradient head_dy [pixel][head_co] and its weights head_w
// [head_co][C] — da[c] = sum_j dy[j]*w[j][c], rounded to T exactly as the 1x1 backward kernel
// would have stored it — so that tensor is neither written nor read.
// SCALED: the incoming gradient is da = g * g_scale[b][c] + g_shift[b][c] with per-(sample, channel) tables
// (the backward of a per-sample weighted sum plus a global average pool: the AMSFN fusion): formed on load
// in float32, so no per-branch gradient tensor exists.  `head_w` carries g_scale.
constexpr int HEAD_MAX_CO = 8;
// (block, n_blocks: the workgroup's index and count inside ITS tensor's launch — a launch may carry several tensors)
template <typename T, int V, bool HEAD, int HCO, bool SCALED>
__device__ __forceinline__ void bn_relu_bwd_dz_body(const T* __restrict__ z, const T* __restrict__ da,
                                                    T* __restrict__ dz, int64_t n_vec, int C, float inv_n,
                                                    const float* __restrict__ mean_rstd,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const float* __restrict__ sums, float* __restrict__ dz_partials,
                                                    int relu, const float* __restrict__ head_w, int head_co,
                                                    const float* __restrict__ g_shift, int coef_ld,
                                                    unsigned px_per_sample, unsigned block, unsigned n_blocks) {
  extern __shared__ float prm[];   // scratch for the optional bias partial: [256][V]
  const int CV = C / V;
  const int64_t i0 = (int64_t)block * EW_THREADS + threadIdx.x;
  const int c0 = (int)(i0 % CV) * V;
  // (HCO: compile-time bound of head_co, so the 4-class head keeps 4 weight rows in registers, not 8)
  float hw[HEAD ? HCO : 1][V];
  if constexpr (HEAD) {
#pragma unroll
    for (int j = 0; j < HCO; j++)
#pragma unroll
      for (int k = 0; k < V; k++) hw[j][k] = j < head_co ? head_w[j * C + c0 + k] : 0.f;
  }
  float mean[V], rstd[V], gam[V], bet[V], m1[V], m2[V];     // this thread's channel constants
#pragma unroll
  for (int k = 0; k < V; k++) {
    mean[k] = mean_rstd[c0 + k];
    rstd[k] = mean_rstd[C + c0 + k];
    gam[k] = gamma[c0 + k];
    bet[k] = beta[c0 + k];
    m1[k] = sums[c0 + k] * inv_n;
    m2[k] = sums[C + c0 + k] * inv_n;
  }
  float bsum[V];
#pragma unroll
  for (int k = 0; k < V; k++) bsum[k] = 0.f;
  // HEAD: the grid keeps a thread on one channel vector (its stride is a multiple of CV), so the pixel
  // index advances by a constant: no division in the loop
  int64_t pix = i0 / CV;
  const int64_t pix_step = ((int64_t)n_blocks * EW_THREADS) / CV;
  // SCALED: the tables' rows of the current sample stay in registers; the pixel index inside the sample is
  // tracked by additions (no division in the loop), and the rows are reloaded when it wraps
  float gsc[SCALED ? V : 1], gsh[SCALED ? V : 1];
  unsigned s_rem = 0, s_row = 0;
  const unsigned n_samples = SCALED ? (unsigned)((n_vec / CV) / px_per_sample) : 0;
  if constexpr (SCALED) {
    s_row = (unsigned)pix / px_per_sample;
    s_rem = (unsigned)pix - s_row * px_per_sample;
    const int64_t row = (int64_t)(s_row < n_samples ? s_row : 0) * coef_ld + c0;
#pragma unroll
    for (int k = 0; k < V; k++) {
      gsc[k] = head_w[row + k];
      gsh[k] = g_shift[row + k];
    }
  }
  for (int64_t i = i0; i < n_vec; i += (int64_t)n_blocks * EW_THREADS, pix += pix_step) {
    float zv[V], gv[V], o[V];
    vload<T, V>(z + i * V, zv);
    if constexpr (HEAD) {
      // `da` holds the 1x1 conv's output gradient: head_co values per pixel, one vector load where
      // the count allows it
      const T* g_ = da + pix * head_co;
      float gj[HCO];
      if (head_co == HCO && (HCO == 4 || HCO == 8)) {
        vload<T, HCO>(g_, gj);
      } else {
#pragma unroll
        for (int j = 0; j < HCO; j++) gj[j] = j < head_co ? to_f32(g_[j]) : 0.f;
      }
#pragma unroll
      for (int k = 0; k < V; k++) gv[k] = 0.f;
#pragma unroll
      for (int j = 0; j < HCO; j++) {
        if (j < head_co) {
#pragma unroll
          for (int k = 0; k < V; k++) gv[k] += gj[j] * hw[j][k];
        }
      }
#pragma unroll
      for (int k = 0; k < V; k++) gv[k] = to_f32(from_f32<T>(gv[k]));
    } else {
      vload<T, V>(da + i * V, gv);
      if constexpr (SCALED) {
#pragma unroll
        for (int k = 0; k < V; k++) gv[k] = gv[k] * gsc[k] + gsh[k];
        s_rem += (unsigned)pix_step;                 // (32-bit: the host checks the pixel count)
        if (s_rem >= px_per_sample) {
          do {
            s_rem -= px_per_sample;
            ++s_row;
          } while (s_rem >= px_per_sample);
          if (s_row < n_samples) {
            const int64_t row = (int64_t)s_row * coef_ld + c0;
#pragma unroll
            for (int k = 0; k < V; k++) {
              gsc[k] = head_w[row + k];
              gsh[k] = g_shift[row + k];
            }
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < V; k++) {
      const float xh = (zv[k] - mean[k]) * rstd[k];
      const float y = xh * gam[k] + bet[k];
      const float dy = (!relu || y > 0.f) ? gv[k] : 0.f;
      o[k] = gam[k] * rstd[k] * (dy - m1[k] - xh * m2[k]);
      bsum[k] += to_f32(from_f32<T>(o[k]));
    }
    vstore<T, V>(dz + i * V, o);
  }
  if (dz_partials) {
    // fixed-order block reduction: thread t holds channel vector (block*256 + t) % CV
    float* red = prm;                          // [256][V]
#pragma unroll
    for (int k = 0; k < V; k++) red[threadIdx.x * V + k] = bsum[k];
    __syncthreads();
    const int base = (int)(((int64_t)block * EW_THREADS) % CV);
    for (int j = threadIdx.x; j < C; j += EW_THREADS) {
      const int cv = j / V, k = j % V;
      float acc = 0.f;
      for (int t = (cv - base + CV) % CV; t < EW_THREADS; t += CV) acc += red[t * V + k];
      dz_partials[(int64_t)block * C + j] = acc;
    }
  }
}


template <typename T, int V, bool HEAD = false, int HCO = HEAD_MAX_CO, bool SCALED = false>
__global__ __launch_bounds__(EW_THREADS) void bn_relu_bwd_dz_kernel(const T* __restrict__ z, const T* __restrict__ da,
                                                                    T* __restrict__ dz, int64_t n_vec, int C,
                                                                    float inv_n, const float* __restrict__ mean_rstd,
                                                                    const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta,
                                                                    const float* __restrict__ sums,
                                                                    float* __restrict__ dz_partials, int relu,
                                                                    const float* __restrict__ head_w = nullptr,
                                                                    int head_co = 0,
                                                                    const float* __restrict__ g_shift = nullptr,
                                                                    int coef_ld = 0, unsigned px_per_sample = 1) {
  bn_relu_bwd_dz_body<T, V, HEAD, HCO, SCALED>(z, da, dz, n_vec, C, inv_n, mean_rstd, gamma, beta, sums, dz_partials, relu,
                                               head_w, head_co, g_shift, coef_ld, px_per_sample, blockIdx.x, gridDim.x);
}

// The four branch layers of one AMSFN block in ONE launch: every branch's gradient is formed from the same tensor
// g (scaled mode), so the four workgroups that walk the same positions of g get ids 8 apart (consecutive ids go
// round the 8 XCDs): they run on one XCD at about the same time and g is fetched from memory once, not four times.
struct DzBranches {
  const void* z[4];
  void* dz[4];
  const float* mean_rstd[4];
  const float* gamma[4];
  const float* beta[4];
  const float* sums[4];         // [2][C] each
  float* partials[4];           // [blocks per branch][C] each, or null
};
template <typename T, int V>
__global__ __launch_bounds__(EW_THREADS) void bn_relu_bwd_dz_scaled4_kernel(DzBranches br, const T* __restrict__ g,
                                                                            int64_t n_vec, int C, float inv_n,
                                                                            const float* __restrict__ g_scale,
                                                                            const float* __restrict__ g_shift,
                                                                            int coef_ld, unsigned px_per_sample) {
  const unsigned per_branch = gridDim.x >> 2;
  unsigned k = blockIdx.x / per_branch, block = blockIdx.x % per_branch;
  if ((per_branch & 7) == 0) {
    k = (blockIdx.x >> 3) & 3;
    block = ((blockIdx.x >> 5) << 3) | (blockIdx.x & 7);
  }
  bn_relu_bwd_dz_body<T, V, false, HEAD_MAX_CO, true>((const T*)br.z[k], g, (T*)br.dz[k], n_vec, C, inv_n, br.mean_rstd[k],
                                                      br.gamma[k], br.beta[k], br.sums[k], br.partials[k], 1,
                                                      g_scale + k * C, 0, g_shift + k * C, coef_ld, px_per_sample, block,
                                                      per_branch);
}

template <typename T>
int apply_t(const void* z, void* a, int64_t n_pix, int C, const float* ss, hipStream_t st) {
  constexpr int N = VecOf<T>::N;
  const int64_t n = n_pix * C;
  const bool vec = pick_vec<T>(C, z, a) == N;
  const int V = vec ? N : 1;
  const int grid = grid_for_fixed_channels(ew_grid(n / V, EW_THREADS), C / V, 4096);
  if (vec)
    bn_relu_apply_kernel<T, N><<<grid, EW_THREADS, 0, st>>>((const T*)z, (T*)a, n / N, C, ss);
  else
    bn_relu_apply_kernel<T, 1><<<grid, EW_THREADS, 0, st>>>((const T*)z, (T*)a, n, C, ss);
  return 0;
}

template <typename T>
int bwd_reduce_t(const void* z, const void* da, int64_t n_pix, int C, const float* mean_rstd, const float* gamma,
                 const float* beta, float* sums, void* ws, hipStream_t st, int relu) {
  constexpr int N = VecOf<T>::N;
  if (pick_vec<T>(C, z, da) == N) {
    BnBwdReduce<T, N> f{(const T*)z, (const T*)da, mean_rstd, gamma, beta, C, relu, {}, {}, {}, {}};
    launch_channel_reduce<N>(f, n_pix, C, (float*)ws, sums, st);
  } else {
    BnBwdReduce<T, 1> f{(const T*)z, (const T*)da, mean_rstd, gamma, beta, C, relu, {}, {}, {}, {}};
    launch_channel_reduce<1>(f, n_pix, C, (float*)ws, sums, st);
  }
  return 0;
}

static inline int bwd_dz_grid(int64_t n, int C, int V) {
  // two workgroups per CU, each thread streaming a long range, measured fastest (384: -10 %, 1024: -2 %,
  // 2048: -6 %), and the conv-bias rows the weight-gradient reduction sums afterwards are fewer
  return grid_for_fixed_channels(ew_grid(n / V, EW_THREADS), C / V, 512);
}

template <typename T>
int bwd_dz_t(const void* z, const void* da, void* dz, int64_t n_pix, int C, const float* mean_rstd,
             const float* gamma, const float* beta, const float* sums, float* dbias, void* ws, hipStream_t st,
             int relu, const float* head_w = nullptr, int head_co = 0, const float* g_shift = nullptr,
             int coef_ld = 0, int64_t hw = 0) {
  constexpr int N = VecOf<T>::N;
  const int64_t n = n_pix * C;
  const size_t lds = (size_t)EW_THREADS * N * sizeof(float);
  const float inv_n = (float)(1.0 / (double)n_pix);
  const bool vec = ((head_w && !g_shift) ? pick_vec<T>(C, z, dz) : pick_vec<T>(C, z, da, dz)) == N;
  const int V = vec ? N : 1;
  const int grid = bwd_dz_grid(n, C, V);
  float* partials = dbias ? (float*)ws : nullptr;
  if (g_shift && vec)
    bn_relu_bwd_dz_kernel<T, N, false, HEAD_MAX_CO, true><<<grid, EW_THREADS, lds, st>>>(
        (const T*)z, (const T*)da, (T*)dz, n / N, C, inv_n, mean_rstd, gamma, beta, sums, partials, relu, head_w, 0,
        g_shift, coef_ld, (unsigned)hw);
  else if (g_shift)
    bn_relu_bwd_dz_kernel<T, 1, false, HEAD_MAX_CO, true><<<grid, EW_THREADS, lds, st>>>(
        (const T*)z, (const T*)da, (T*)dz, n, C, inv_n, mean_rstd, gamma, beta, sums, partials, relu, head_w, 0, g_shift,
        coef_ld, (unsigned)hw);
  else if (head_w && vec && head_co <= 4)
    bn_relu_bwd_dz_kernel<T, N, true, 4><<<grid, EW_THREADS, lds, st>>>((const T*)z, (const T*)da, (T*)dz, n / N, C,
                                                                          inv_n, mean_rstd, gamma, beta, sums, partials,
                                                                          relu, head_w, head_co);
  else if (head_w && vec)
    bn_relu_bwd_dz_kernel<T, N, true><<<grid, EW_THREADS, lds, st>>>((const T*)z, (const T*)da, (T*)dz, n / N, C, inv_n,
                                                                       mean_rstd, gamma, beta, sums, partials, relu,
                                                                       head_w, head_co);
  else if (head_w)
    bn_relu_bwd_dz_kernel<T, 1, true><<<grid, EW_THREADS, lds, st>>>((const T*)z, (const T*)da, (T*)dz, n, C, inv_n,
                                                                       mean_rstd, gamma, beta, sums, partials, relu,
                                                                       head_w, head_co);
  else if (vec)
    bn_relu_bwd_dz_kernel<T, N><<<grid, EW_THREADS, lds, st>>>((const T*)z, (const T*)da, (T*)dz, n / N, C, inv_n,
                                                                 mean_rstd, gamma, beta, sums, partials, relu);
  else
    bn_relu_bwd_dz_kernel<T, 1><<<grid, EW_THREADS, lds, st>>>((const T*)z, (const T*)da, (T*)dz, n, C, inv_n,
                                                                 mean_rstd, gamma, beta, sums, partials, relu);
  // dbias == workspace: the caller takes the partial rows [grid][C] as they are (and folds
  // their reduction into a later launch)
  if (dbias && dbias != partials) launch_reduce_partials(partials, grid, C, dbias, st);
  return 0;
}

template <typename T>
int chan_sum_t(const void* x, int64_t n_pix, int C, float* out, void* ws, hipStream_t st) {
  constexpr int N = VecOf<T>::N;
  if (pick_vec<T>(C, x) == N) {
    ChannelSum<T, N> f{(const T*)x, C};
    launch_channel_reduce<N>(f, n_pix, C, (float*)ws, out, st);
  } else {
    ChannelSum<T, 1> f{(const T*)x, C};
    launch_channel_reduce<1>(f, n_pix, C, (float*)ws, out, st);
  }
  return 0;
}

}  // namespace

extern "C" {

int mis_bn_finalize(const float* stats, int stats_rows, int C, int64_t n, const float* gamma, const float* beta,
                    float eps, float momentum, float* running_mean, float* running_var,
                    int64_t* num_batches_tracked, float* scale_shift, float* mean_rstd, mis_stream_t stream) {
  MIS_REQUIRE(stats && gamma && beta && scale_shift, "bn_finalize: null pointer");
  MIS_REQUIRE(C >= 1 && n >= 1 && stats_rows >= 1, "bn_finalize: bad shape C=%d n=%lld rows=%d", C, (long long)n,
              stats_rows);
  const double unbias = n > 1 ? (double)n / (double)(n - 1) : 1.0;
#define FIN_LAUNCH(CL)                                                                                       \
  bn_finalize_kernel<CL><<<cdiv(C, CL), 1024, 0, (hipStream_t)stream>>>(stats, stats_rows, C, 1.0 / (double)n, unbias, \
                                                                        gamma, beta, eps, momentum, running_mean,   \
                                                                        running_var, num_batches_tracked,           \
                                                                        scale_shift, mean_rstd)
  if (C <= 64) FIN_LAUNCH(8);
  else if (C <= 256) FIN_LAUNCH(16);
  else FIN_LAUNCH(32);
#undef FIN_LAUNCH
  MIS_CHECK_LAUNCH("bn_finalize_kernel");
  return MIS_OK;
}

int mis_bn_eval_affine(const float* running_mean, const float* running_var, const float* gamma, const float* beta,
                       int C, float eps, float* scale_shift, mis_stream_t stream) {
  MIS_REQUIRE(running_mean && running_var && gamma && beta && scale_shift && C >= 1, "bn_eval_affine: bad arguments");
  bn_eval_affine_kernel<<<cdiv(C, 128), 128, 0, (hipStream_t)stream>>>(running_mean, running_var, gamma, beta, C, eps,
                                                                       scale_shift);
  MIS_CHECK_LAUNCH("bn_eval_affine_kernel");
  return MIS_OK;
}

int mis_bn_relu_apply(const void* z, int dtype, int64_t n_pix, int C, const float* scale_shift, void* a,
                      mis_stream_t stream) {
  MIS_REQUIRE(z && a && scale_shift && n_pix >= 1 && C >= 1 && C <= 4096, "bn_relu_apply: bad arguments");
  if (dtype == MIS_F32) apply_t<float>(z, a, n_pix, C, scale_shift, (hipStream_t)stream);
  else if (dtype == MIS_BF16) apply_t<bf16>(z, a, n_pix, C, scale_shift, (hipStream_t)stream);
  else MIS_REQUIRE(false, "bn_relu_apply: bad dtype %d", dtype);
  MIS_CHECK_LAUNCH("bn_relu_apply_kernel");
  return MIS_OK;
}

size_t mis_bn_relu_bwd_workspace_bytes(int64_t n_pix, int C) {
  const size_t a = red_workspace_bytes(2, C), b = align256((size_t)1024 * C * sizeof(float));
  return a > b ? a : b;
}

static int bn_bwd_reduce_any(const void* z, const void* da, int dtype, int64_t n_pix, int C, const float* mean_rstd,
                             const float* gamma, const float* beta, float* sums, void* workspace, mis_stream_t stream,
                             int relu) {
  MIS_REQUIRE(z && da && mean_rstd && gamma && beta && sums && workspace, "bn_relu_bwd_reduce: null pointer");
  MIS_REQUIRE(n_pix >= 1 && C >= 1 && C <= 2048, "bn_relu_bwd_reduce: bad shape");
  if (dtype == MIS_F32) {
    MIS_REQUIRE(C % 4 == 0 ? C / 4 <= 256 : C <= 256, "bn_relu_bwd_reduce: C=%d too large for this layout", C);
    bwd_reduce_t<float>(z, da, n_pix, C, mean_rstd, gamma, beta, sums, workspace, (hipStream_t)stream, relu);
  } else if (dtype == MIS_BF16) {
    MIS_REQUIRE(C % 8 == 0 ? C / 8 <= 256 : C <= 256, "bn_relu_bwd_reduce: C=%d too large for this layout", C);
    bwd_reduce_t<bf16>(z, da, n_pix, C, mean_rstd, gamma, beta, sums, workspace, (hipStream_t)stream, relu);
  } else MIS_REQUIRE(false, "bn_relu_bwd_reduce: bad dtype %d", dtype);
  MIS_CHECK_LAUNCH("bn_relu_bwd_reduce");
  return MIS_OK;
}

int mis_bn_relu_bwd_reduce(const void* z, const void* da, int dtype, int64_t n_pix, int C, const float* mean_rstd,
                           const float* gamma, const float* beta, float* sums, void* workspace, mis_stream_t stream) {
  return bn_bwd_reduce_any(z, da, dtype, n_pix, C, mean_rstd, gamma, beta, sums, workspace, stream, 1);
}

int mis_bn_bwd_reduce(const void* z, const void* dy, int dtype, int64_t n_pix, int C, const float* mean_rstd,
                      const float* gamma, const float* beta, float* sums, void* workspace, mis_stream_t stream) {
  return bn_bwd_reduce_any(z, dy, dtype, n_pix, C, mean_rstd, gamma, beta, sums, workspace, stream, 0);
}

static int bn_bwd_dz_any(const void* z, const void* da, int dtype, int64_t n_pix, int C, const float* mean_rstd,
                         const float* gamma, const float* beta, const float* sums, void* dz, float* dbias,
                         void* workspace, mis_stream_t stream, int relu, const float* head_w = nullptr,
                         int head_co = 0, const float* g_shift = nullptr, int coef_ld = 0, int64_t hw = 0) {
  MIS_REQUIRE(z && da && mean_rstd && gamma && beta && sums && dz, "bn_relu_bwd_dz: null pointer");
  MIS_REQUIRE(n_pix >= 1 && C >= 1 && C <= 2048, "bn_relu_bwd_dz: bad shape");
  MIS_REQUIRE(!dbias || workspace, "bn_relu_bwd_dz: dbias requested without a workspace");
  MIS_REQUIRE((const void*)dbias != workspace ||
                  ((((uintptr_t)z | (uintptr_t)da | (uintptr_t)dz) & 15) == 0),
              "bn_relu_bwd_dz: partial rows need 16-byte aligned tensors");
  if (dtype == MIS_F32)
    bwd_dz_t<float>(z, da, dz, n_pix, C, mean_rstd, gamma, beta, sums, dbias, workspace, (hipStream_t)stream, relu,
                    head_w, head_co, g_shift, coef_ld, hw);
  else if (dtype == MIS_BF16)
    bwd_dz_t<bf16>(z, da, dz, n_pix, C, mean_rstd, gamma, beta, sums, dbias, workspace, (hipStream_t)stream, relu,
                   head_w, head_co, g_shift, coef_ld, hw);
  else MIS_REQUIRE(false, "bn_relu_bwd_dz: bad dtype %d", dtype);
  MIS_CHECK_LAUNCH("bn_relu_bwd_dz_kernel");
  return MIS_OK;
}

int mis_bn_relu_bwd_dz(const void* z, const void* da, int dtype, int64_t n_pix, int C, const float* mean_rstd,
                       const float* gamma, const float* beta, const float* sums, void* dz, float* dbias,
                       void* workspace, mis_stream_t stream) {
  return bn_bwd_dz_any(z, da, dtype, n_pix, C, mean_rstd, gamma, beta, sums, dz, dbias, workspace, stream, 1);
}

int mis_bn_relu_bwd_dz_head(const void* z, const void* head_dy, const float* head_w, int head_cout, int dtype,
                            int64_t n_pix, int C, const float* mean_rstd, const float* gamma, const float* beta,
                            const float* sums, void* dz, float* dbias, void* workspace, mis_stream_t stream) {
  MIS_REQUIRE(head_w && head_cout >= 1 && head_cout <= HEAD_MAX_CO, "bn_relu_bwd_dz_head: 1 <= head_cout <= %d required",
              HEAD_MAX_CO);
  return bn_bwd_dz_any(z, head_dy, dtype, n_pix, C, mean_rstd, gamma, beta, sums, dz, dbias, workspace, stream, 1, head_w,
                       head_cout);
}

int mis_bn_relu_bwd_dz_scaled(const void* z, const void* g, const float* g_scale, const float* g_shift, int coef_ld,
                              int64_t hw, int dtype, int64_t n_pix, int C, const float* mean_rstd, const float* gamma,
                              const float* beta, const float* sums, void* dz, float* dbias, void* workspace,
                              mis_stream_t stream) {
  MIS_REQUIRE(g_scale && g_shift && coef_ld >= C, "bn_relu_bwd_dz_scaled: need both tables, row stride >= C");
  MIS_REQUIRE(hw >= 1 && n_pix % hw == 0 && n_pix < ((int64_t)1 << 31),
              "bn_relu_bwd_dz_scaled: n_pix must be a whole number of samples of hw pixels, below 2^31");
  return bn_bwd_dz_any(z, g, dtype, n_pix, C, mean_rstd, gamma, beta, sums, dz, dbias, workspace, stream, 1, g_scale, 0,
                       g_shift, coef_ld, hw);
}

int mis_bn_relu_bwd_dz_scaled4(const void* const* z, const void* g, const float* g_scale, const float* g_shift,
                               int coef_ld, int64_t hw, int dtype, int64_t n_pix, int C,
                               const float* const* mean_rstd, const float* const* gamma, const float* const* beta,
                               const float* sums, void* const* dz, float* const* rows, mis_stream_t stream) {
  MIS_REQUIRE(z && g && g_scale && g_shift && mean_rstd && gamma && beta && sums && dz && rows,
              "bn_relu_bwd_dz_scaled4: null pointer");
  MIS_REQUIRE(coef_ld >= 4 * C, "bn_relu_bwd_dz_scaled4: the tables hold four branches per row, row stride >= 4*C");
  MIS_REQUIRE(n_pix >= 1 && C >= 1 && C <= 2048, "bn_relu_bwd_dz_scaled4: bad shape");
  MIS_REQUIRE(hw >= 1 && n_pix % hw == 0 && n_pix < ((int64_t)1 << 31),
              "bn_relu_bwd_dz_scaled4: n_pix must be a whole number of samples of hw pixels, below 2^31");
  MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, "bn_relu_bwd_dz_scaled4: bad dtype %d", dtype);
  DzBranches br;
  uintptr_t align = (uintptr_t)g;
  for (int k = 0; k < 4; k++) {
    MIS_REQUIRE(z[k] && dz[k] && mean_rstd[k] && gamma[k] && beta[k], "bn_relu_bwd_dz_scaled4: null pointer (branch %d)", k);
    br.z[k] = z[k];
    br.dz[k] = dz[k];
    br.mean_rstd[k] = mean_rstd[k];
    br.gamma[k] = gamma[k];
    br.beta[k] = beta[k];
    br.sums[k] = sums + (size_t)k * 2 * C;
    br.partials[k] = rows[k];
    align |= (uintptr_t)z[k] | (uintptr_t)dz[k];
  }
  const int N = dtype == MIS_F32 ? VecOf<float>::N : VecOf<bf16>::N;
  const bool vec = C % N == 0 && (align % 16) == 0;
  // (the partial rows are [mis_bn_relu_bwd_dz_rows][C]: that count assumes full vectors whenever C allows them)
  MIS_REQUIRE(vec || C % N != 0, "bn_relu_bwd_dz_scaled4: tensors must be 16-byte aligned");
  const int V = vec ? N : 1;
  const int64_t n = n_pix * C;
  const int per_branch = bwd_dz_grid(n, C, V);
  const size_t lds = (size_t)EW_THREADS * N * sizeof(float);
  const float inv_n = (float)(1.0 / (double)n_pix);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIS_BF16 && vec)
    bn_relu_bwd_dz_scaled4_kernel<bf16, 8><<<4 * per_branch, EW_THREADS, lds, st>>>(br, (const bf16*)g, n / 8, C, inv_n, g_scale,
                                                                                    g_shift, coef_ld, (unsigned)hw);
  else if (dtype == MIS_BF16)
    bn_relu_bwd_dz_scaled4_kernel<bf16, 1><<<4 * per_branch, EW_THREADS, lds, st>>>(br, (const bf16*)g, n, C, inv_n, g_scale,
                                                                                    g_shift, coef_ld, (unsigned)hw);
  else if (vec)
    bn_relu_bwd_dz_scaled4_kernel<float, 4><<<4 * per_branch, EW_THREADS, lds, st>>>(br, (const float*)g, n / 4, C, inv_n, g_scale,
                                                                                     g_shift, coef_ld, (unsigned)hw);
  else
    bn_relu_bwd_dz_scaled4_kernel<float, 1><<<4 * per_branch, EW_THREADS, lds, st>>>(br, (const float*)g, n, C, inv_n, g_scale,
                                                                                     g_shift, coef_ld, (unsigned)hw);
  MIS_CHECK_LAUNCH("bn_relu_bwd_dz_scaled4_kernel");
  return MIS_OK;
}

int mis_bn_bwd_dz(const void* z, const void* dy, int dtype, int64_t n_pix, int C, const float* mean_rstd,
                  const float* gamma, const float* beta, const float* sums, void* dz, float* dbias, void* workspace,
                  mis_stream_t stream) {
  return bn_bwd_dz_any(z, dy, dtype, n_pix, C, mean_rstd, gamma, beta, sums, dz, dbias, workspace, stream, 0);
}

int mis_bn_relu_bwd_dz_rows(int dtype, int64_t n_pix, int C) {
  if (n_pix < 1 || C < 1 || (dtype != MIS_F32 && dtype != MIS_BF16)) return 0;
  const int N = dtype == MIS_F32 ? VecOf<float>::N : VecOf<bf16>::N;
  const int V = (C % N == 0) ? N : 1;
  return bwd_dz_grid(n_pix * C, C, V);
}

size_t mis_channel_sum_workspace_bytes(int64_t n_pix, int C) { return red_workspace_bytes(1, C); }

int mis_channel_sum(const void* x, int dtype, int64_t n_pix, int C, float* out, void* workspace,
                    mis_stream_t stream) {
  MIS_REQUIRE(x && out && workspace && n_pix >= 1 && C >= 1, "channel_sum: bad arguments");
  if (dtype == MIS_F32) {
    MIS_REQUIRE(C % 4 == 0 ? C / 4 <= 256 : C <= 256, "channel_sum: C=%d too large for this layout", C);
    chan_sum_t<float>(x, n_pix, C, out, workspace, (hipStream_t)stream);
  } else if (dtype == MIS_BF16) {
    MIS_REQUIRE(C % 8 == 0 ? C / 8 <= 256 : C <= 256, "channel_sum: C=%d too large for this layout", C);
    chan_sum_t<bf16>(x, n_pix, C, out, workspace, (hipStream_t)stream);
  } else MIS_REQUIRE(false, "channel_sum: bad dtype %d", dtype);
  MIS_CHECK_LAUNCH("channel_sum");
  return MIS_OK;
}

}  // extern "C"
