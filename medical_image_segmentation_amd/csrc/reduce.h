// Per-channel reductions over the pixels of a channels-last tensor.
//
// Thread layout inside a 256-thread block: channel-vector index cv = tid % CV (CV = C/V
// 16-byte vectors per pixel) and pixel row pr = tid / CV; consecutive lanes read consecutive
// 16-byte vectors, so a wave reads contiguous memory.  Each block owns a contiguous pixel
// range, reduces it to one value per (output, channel) through LDS and writes a partial
// row; reduce_partials sums the rows in a fixed order (double accumulator), which makes
// every result bitwise reproducible run to run.
#pragma once
#include "common.h"
#include "vec.h"

constexpr int RED_THREADS = 256;
constexpr int RED_MAX_BLOCKS = 1024;

struct RedPlan {
  int CV, PR, nblk;
  int64_t pix_per_blk;
};

static inline RedPlan red_plan(int64_t n_pix, int C, int V) {
  RedPlan p;
  p.CV = C / V;
  p.PR = RED_THREADS / p.CV;
  if (p.PR < 1) p.PR = 1;
  int64_t want = (n_pix + (int64_t)p.PR * 8 - 1) / ((int64_t)p.PR * 8);   // >= 8 pixels per thread
  if (want > RED_MAX_BLOCKS) want = RED_MAX_BLOCKS;
  if (want < 1) want = 1;
  p.nblk = (int)want;
  p.pix_per_blk = (n_pix + p.nblk - 1) / p.nblk;
  return p;
}

// out[j] = sum_i partials[i*stride + j] for the 32 columns j of block `blk` (1024 threads: 32
// columns x 32 row lanes, double accumulators, fixed order).  `red` is [32][33] doubles of LDS.
__device__ __forceinline__ void reduce_rows_block(const float* __restrict__ partials, int nblk, int n,
                                                  int64_t stride, float* __restrict__ out, int blk,
                                                  double (*red)[33]) {
  const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
  const int j = blk * 32 + c;
  double s = 0.0;
  if (j < n) {
    int i = r;
    for (; i + 96 < nblk; i += 128) {
      const float a0 = partials[(int64_t)i * stride + j], a1 = partials[(int64_t)(i + 32) * stride + j];
      const float a2 = partials[(int64_t)(i + 64) * stride + j], a3 = partials[(int64_t)(i + 96) * stride + j];
      s += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
    }
    for (; i < nblk; i += 32) s += (double)partials[(int64_t)i * stride + j];
  }
  red[r][c] = s;
  __syncthreads();
  if (r == 0 && j < n) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 32; k++) t += red[k][c];
    out[j] = (float)t;
  }
}

// out[j] = sum_i partials[i*stride + j], j < n, i < nblk, fixed order.
__global__ void reduce_partials_kernel(const float* __restrict__ partials, int nblk, int n, int64_t stride,
                                       float* __restrict__ out);

static inline void launch_reduce_partials(const float* partials, int nblk, int n, float* out, hipStream_t st,
                                          int64_t stride = 0) {
  reduce_partials_kernel<<<cdiv(n, 32), 1024, 0, st>>>(partials, nblk, n, stride ? stride : n, out);
}

// Same reduction with the n outputs split over up to three destinations (n0, n1, rest).
__global__ void reduce_partials3_kernel(const float* __restrict__ partials, int nblk, int n, int n0, int n1,
                                        float* __restrict__ out0, float* __restrict__ out1,
                                        float* __restrict__ out2);

static inline void launch_reduce_partials3(const float* partials, int nblk, int n, int n0, int n1, float* out0,
                                           float* out1, float* out2, hipStream_t st) {
  reduce_partials3_kernel<<<cdiv(n, 32), 1024, 0, st>>>(partials, nblk, n, n0, n1, out0, out1, out2);
}

// Largest grid <= min(want, cap) such that grid*256 is a multiple of CV (so that a grid-stride
// thread of a 256-thread block keeps one channel vector); at least the smallest such grid.
static inline int grid_for_fixed_channels(int want, int CV, int cap) {
  int g = 1;
  while (((int64_t)g * 256) % CV != 0) g++;
  if (want > cap) want = cap;
  int grid = (want / g) * g;
  return grid < g ? g : grid;
}

// F provides:  static constexpr int NOUT;
//              __device__ void prepare(int c0)          (per-thread channel constants -> registers)
//              __device__ void load(int64_t pix, int c0, float (&acc)[NOUT][V]) const   (accumulates)
template <int V, typename F>
__global__ __launch_bounds__(RED_THREADS) void channel_reduce_kernel(F f, int64_t n_pix, int C, int CV, int PR,
                                                                     int64_t pix_per_blk,
                                                                     float* __restrict__ partials) {
  constexpr int NOUT = F::NOUT;
  extern __shared__ __attribute__((aligned(16))) float red_lds[];   // [PR][NOUT][C]
  const int tid = threadIdx.x;
  const int cv = tid % CV, pr = tid / CV;
  float acc[NOUT][V];
#pragma unroll
  for (int o = 0; o < NOUT; o++)
#pragma unroll
    for (int k = 0; k < V; k++) acc[o][k] = 0.f;
  const int64_t p0 = (int64_t)blockIdx.x * pix_per_blk;
  int64_t p1 = p0 + pix_per_blk;
  if (p1 > n_pix) p1 = n_pix;
  if (pr < PR) {
    f.prepare(cv * V);
#pragma unroll 4
    for (int64_t p = p0 + pr; p < p1; p += PR) f.load(p, cv * V, acc);
#pragma unroll
    for (int o = 0; o < NOUT; o++)
#pragma unroll
      for (int k = 0; k < V; k++) red_lds[((int64_t)pr * NOUT + o) * C + cv * V + k] = acc[o][k];
  }
  __syncthreads();
  for (int j = tid; j < NOUT * C; j += RED_THREADS) {
    float s = 0.f;
    for (int r = 0; r < PR; r++) s += red_lds[(int64_t)r * NOUT * C + j];
    partials[(int64_t)blockIdx.x * NOUT * C + j] = s;
  }
}

template <int V, typename F>
static inline void launch_channel_reduce(const F& f, int64_t n_pix, int C, float* partials, float* out,
                                         hipStream_t st) {
  const RedPlan p = red_plan(n_pix, C, V);
  const size_t lds = (size_t)p.PR * F::NOUT * C * sizeof(float);
  channel_reduce_kernel<V, F><<<p.nblk, RED_THREADS, lds, st>>>(f, n_pix, C, p.CV, p.PR, p.pix_per_blk, partials);
  launch_reduce_partials(partials, p.nblk, F::NOUT * C, out, st);
}

static inline size_t red_workspace_bytes(int nout, int C) {
  return align256((size_t)RED_MAX_BLOCKS * nout * C * sizeof(float));
}
