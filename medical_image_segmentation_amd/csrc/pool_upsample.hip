// Down (2x2 max-pool) and Up (bilinear x2, align_corners=True, + skip concat) on
// channels-last tensors.  All four kernels are pure streaming: one lane moves a
// 16-byte channel vector, consecutive lanes take consecutive channel vectors of a
// pixel and then the next pixel, so every wave access is a contiguous 1 KiB.
//
// Roofline: HBM.  Algorithmic bytes (elements * sizeof(T)):
//   pool fwd   : read 4*n_out*C, write n_out*C
//   pool bwd   : read 4*n_out*C (x) + n_out*C (dy), write 4*n_out*C
//   up+cat fwd : read n_low*Cup + 4*n_low*Cskip, write 4*n_low*(Cup+Cskip)
//   up+cat bwd : read 4*n_low*(Cup+Cskip), write n_low*Cup + 4*n_low*Cskip
#include "common.h"
#include "vec.h"
#include "reduce.h"

namespace {

constexpr int EW_THREADS = 256;
constexpr int POOL_MAX_ROWS = 4096;      // partial rows (= workgroups) of the pooling backward's fused sums

template <typename T, int V>
__global__ __launch_bounds__(EW_THREADS) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                                 int B, int H, int W, int C,
                                                                 const float* __restrict__ in_aff,
                                                                 const float* __restrict__ noise) {
  extern __shared__ float aff_lds[];
  const float* aff = nullptr;
  if (in_aff) {
    for (int i = threadIdx.x; i < 2 * C; i += EW_THREADS) aff_lds[i] = in_aff[i];
    __syncthreads();
    aff = aff_lds;
  }
  const int Ho = H >> 1, Wo = W >> 1, CV = C / V;
  const int64_t total = (int64_t)B * Ho * Wo * CV;
  for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_THREADS) {
    const int cv = (int)(i % CV);
    int64_t p = i / CV;
    const int xo = (int)(p % Wo);
    p /= Wo;
    const int yo = (int)(p % Ho);
    const int b = (int)(p / Ho);
    const T* src = x + (((int64_t)b * H + 2 * yo) * W + 2 * xo) * C + cv * V;
    float v00[V], v01[V], v10[V], v11[V], m[V];
    vload<T, V>(src, v00);
    vload<T, V>(src + C, v01);
    vload<T, V>(src + (int64_t)W * C, v10);
    vload<T, V>(src + (int64_t)W * C + C, v11);
    affine_relu<V>(v00, aff, C, cv * V);
    affine_relu<V>(v01, aff, C, cv * V);
    affine_relu<V>(v10, aff, C, cv * V);
    affine_relu<V>(v11, aff, C, cv * V);
#pragma unroll
    for (int k = 0; k < V; k++) {
      float mm = v00[k];
      mm = (v01[k] > mm || v01[k] != v01[k]) ? v01[k] : mm;
      mm = (v10[k] > mm || v10[k] != v10[k]) ? v10[k] : mm;
      mm = (v11[k] > mm || v11[k] != v11[k]) ? v11[k] : mm;
      m[k] = mm;
    }
    if (noise) {      // Dropout2d on the pooled tensor: one multiplier per (sample, channel)
#pragma unroll
      for (int k = 0; k < V; k++) m[k] = to_f32(from_f32<T>(m[k])) * noise[(int64_t)b * C + cv * V + k];
    }
    vstore<T, V>(y + i * V, m);
  }
}

// Backward of the pooling (+ optional second gradient `add` of the same tensor).  When the
// pooled tensor is a lazy activation whose producer is conv+BatchNorm(train)+ReLU, the very
// next step of the backward pass is that BatchNorm's reduction over (z, dx):
//     sum(dy), sum(dy * xhat)  with  dy = dx where the activation is positive, xhat = (z-mean)*rstd.
// Both z and dx are in registers here, so with `bn_partials` the kernel accumulates these sums
// too (per-thread, fixed channel vector; fixed-order block reduction; one partial row
// [2][C] per workgroup) and that layer's separate reduction pass over 2 tensors disappears.
// `in_aff` is then [4][C]: scale, shift, mean, rstd.
template <typename T, int V>
__global__ __launch_bounds__(EW_THREADS) void maxpool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                 const T* add, T* dx, int B, int H, int W, int C,
                                                                 const float* __restrict__ in_aff,
                                                                 const float* __restrict__ noise,
                                                                 float* __restrict__ bn_partials) {
  extern __shared__ float aff_lds[];          // [2][C] affine table, reused as [256][2V] reduction scratch
  const float* aff = nullptr;
  if (in_aff) {
    for (int i = threadIdx.x; i < 2 * C; i += EW_THREADS) aff_lds[i] = in_aff[i];
    __syncthreads();
    aff = aff_lds;
  }
  const int Ho = H >> 1, Wo = W >> 1, CV = C / V;
  const int64_t total = (int64_t)B * Ho * Wo * CV;
  const int64_t i0 = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  float bmean[V], brstd[V], bs[2][V];
  if (bn_partials) {
    const int c0 = (int)(i0 % CV) * V;        // fixed for this thread (the grid is sized for that)
#pragma unroll
    for (int k = 0; k < V; k++) {
      bmean[k] = in_aff[2 * C + c0 + k];
      brstd[k] = in_aff[3 * C + c0 + k];
      bs[0][k] = bs[1][k] = 0.f;
    }
  }
  for (int64_t i = i0; i < total; i += (int64_t)gridDim.x * EW_THREADS) {
    const int cv = (int)(i % CV);
    int64_t p = i / CV;
    const int xo = (int)(p % Wo);
    p /= Wo;
    const int yo = (int)(p % Ho);
    const int b = (int)(p / Ho);
    const int64_t off = (((int64_t)b * H + 2 * yo) * W + 2 * xo) * C + cv * V;
    const T* src = x + off;
    float zr[4][V], v[4][V], g[V], o[4][V];
    vload<T, V>(src, zr[0]);
    vload<T, V>(src + C, zr[1]);
    vload<T, V>(src + (int64_t)W * C, zr[2]);
    vload<T, V>(src + (int64_t)W * C + C, zr[3]);
#pragma unroll
    for (int j = 0; j < 4; j++) {
#pragma unroll
      for (int k = 0; k < V; k++) v[j][k] = zr[j][k];
      affine_relu<V>(v[j], aff, C, cv * V);
    }
    vload<T, V>(dy + i * V, g);
    if (noise) {      // gradient through the Dropout2d that followed the pooling
#pragma unroll
      for (int k = 0; k < V; k++) g[k] = to_f32(from_f32<T>(g[k] * noise[(int64_t)b * C + cv * V + k]));
    }
#pragma unroll
    for (int k = 0; k < V; k++) {
      float mm = v[0][k];
      int am = 0;
#pragma unroll
      for (int j = 1; j < 4; j++) {
        const bool take = (v[j][k] > mm) || (v[j][k] != v[j][k]);
        mm = take ? v[j][k] : mm;
        am = take ? j : am;
      }
#pragma unroll
      for (int j = 0; j < 4; j++) o[j][k] = (am == j) ? g[k] : 0.f;
    }
    if (add) {
      // a second gradient of the same tensor (its use as a skip connection), summed here instead
      // of in a separate pass; `add` may alias dx (each lane reads exactly what it overwrites)
      const T* asrc = add + off;
      float a4[4][V];
      vload<T, V>(asrc, a4[0]);
      vload<T, V>(asrc + C, a4[1]);
      vload<T, V>(asrc + (int64_t)W * C, a4[2]);
      vload<T, V>(asrc + (int64_t)W * C + C, a4[3]);
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int k = 0; k < V; k++) o[j][k] += a4[j][k];
    }
    if (bn_partials) {
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int k = 0; k < V; k++) {
          const float d = (v[j][k] > 0.f) ? to_f32(from_f32<T>(o[j][k])) : 0.f;   // the value the next pass would read
          bs[0][k] += d;
          bs[1][k] += d * ((zr[j][k] - bmean[k]) * brstd[k]);
        }
    }
    if (dx) {                                  // (null: only the BatchNorm sums are wanted)
      T* dst = dx + off;
      vstore<T, V>(dst, o[0]);
      vstore<T, V>(dst + C, o[1]);
      vstore<T, V>(dst + (int64_t)W * C, o[2]);
      vstore<T, V>(dst + (int64_t)W * C + C, o[3]);
    }
  }
  if (bn_partials) {
    // fixed-order block reduction: thread t holds channel vector (blockIdx*256 + t) % CV
    __syncthreads();
    float* red = aff_lds;                      // [256][2V]
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
      for (int k = 0; k < V; k++) red[(threadIdx.x * 2 + q) * V + k] = bs[q][k];
    __syncthreads();
    const int base = (int)(((int64_t)blockIdx.x * EW_THREADS) % CV);
    for (int j = threadIdx.x; j < 2 * C; j += EW_THREADS) {
      const int q = j / C, c = j % C, cv = c / V, k = c % V;
      float acc = 0.f;
      for (int t = (cv - base + CV) % CV; t < EW_THREADS; t += CV) acc += red[(t * 2 + q) * V + k];
      bn_partials[(int64_t)blockIdx.x * 2 * C + j] = acc;
    }
  }
}

// align_corners=True source coordinate, computed as torch does (float scale, float product).
__device__ __forceinline__ void up_taps(int dst, float scale, int n_in, int& i0, int& i1, float& l0, float& l1) {
  const float src = scale * (float)dst;
  i0 = (int)src;
  if (i0 > n_in - 1) i0 = n_in - 1;
  i1 = i0 + ((i0 < n_in - 1) ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.0f - l1;
}

// ---- bilinear x2 upsampling, tiled through LDS ------------------------------------------
// Forward: a workgroup owns a 16x16 output tile and a chunk of UP_CC channels.  The <= 10x10
// low-resolution pixels the tile interpolates from are loaded once, the producer's lazy
// affine + ReLU is applied to each of them ONCE (not once per output tap), and they are kept in
// LDS in the storage type; every output then blends 4 LDS vectors.  `Cdst` is the channel count
// of the destination pixel (== C, or C + Cskip when the result is the head of a concatenation).
#ifndef MIS_UP_DEBUG
#define MIS_UP_DEBUG 0
#endif
#if MIS_UP_DEBUG
__device__ unsigned int up_dbg[8];
#endif
constexpr int UP_T = 16;          // output tile edge
constexpr int UP_SRC = 10;        // max low-res rows/cols under a tile: floor(15*s)+2 with s < 1/2

template <typename T, int CC>
__global__ __launch_bounds__(EW_THREADS) void up_fwd_tiled_kernel(const T* __restrict__ low, T* __restrict__ out,
                                                                  int B, int h, int w, int C, int Cdst, float sy,
                                                                  float sx, int tiles_x, int tiles_y,
                                                                  const float* __restrict__ low_aff,
                                                                  const float* __restrict__ noise) {
  constexpr int G = 16 / (int)sizeof(T);        // channels per 16-byte vector
  constexpr int VC = CC / G;                    // vectors per pixel in this chunk
  constexpr int PITCH = VC * 16 + 16;           // LDS bytes per staged pixel (padded: odd number of 16-byte slots)
  __shared__ __attribute__((aligned(16))) unsigned char tile[UP_SRC * UP_SRC * PITCH];
  __shared__ float affl[2 * CC];
  const int tid = threadIdx.x;
  const int c0 = blockIdx.y * CC;
  int t = blockIdx.x;
  const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
  const int Ho = 2 * h, Wo = 2 * w, Y0 = ty * UP_T, X0 = tx * UP_T;
  int ya, xa, i1;
  float l0, l1;
  up_taps(Y0, sy, h, ya, i1, l0, l1);
  up_taps(X0, sx, w, xa, i1, l0, l1);
  if (low_aff) {
    for (int i = tid; i < 2 * CC; i += EW_THREADS) {
      const int c = c0 + i % CC;
      affl[i] = (c < C) ? low_aff[(i / CC) * C + c] : 0.f;
    }
    __syncthreads();
  }
  // stage the source pixels (rows ya.., cols xa.., clipped to the image)
  const T* src = low + (int64_t)b * h * w * C;
  // (all of a thread's loads are issued before the first use: one memory latency per tile)
  constexpr int NV = UP_SRC * UP_SRC * VC, IT = (NV + EW_THREADS - 1) / EW_THREADS;
  uint4 stg[IT];
#pragma unroll
  for (int i = 0; i < IT; i++) {
    const int e = tid + i * EW_THREADS;
    const int cv = e % VC, pos = e / VC;
    const int gy = ya + pos / UP_SRC, gx = xa + pos % UP_SRC, ch = c0 + cv * G;
    const bool ok = e < NV && gy < h && gx < w && ch < C;
    stg[i] = *reinterpret_cast<const uint4*>(src + (ok ? ((int64_t)gy * w + gx) * C + ch : 0));
  }
#pragma unroll
  for (int i = 0; i < IT; i++) {
    const int e = tid + i * EW_THREADS;
    const int cv = e % VC, pos = e / VC;
    const int gy = ya + pos / UP_SRC, gx = xa + pos % UP_SRC, ch = c0 + cv * G;
    if (e < NV && gy < h && gx < w && ch < C) {
      float v[G];
      vload<T, G>(reinterpret_cast<const T*>(&stg[i]), v);
      affine_relu<G>(v, low_aff ? affl : nullptr, CC, cv * G);
      if (noise) {    // Dropout2d between the producer and the upsampling
#pragma unroll
        for (int k = 0; k < G; k++) v[k] = to_f32(from_f32<T>(v[k])) * noise[(int64_t)b * C + ch + k];
      }
      vstore<T, G>(reinterpret_cast<T*>(tile + pos * PITCH + cv * 16), v);
    }
  }
  __syncthreads();
  // 256 threads = (256 / VC) pixels per pass; a thread keeps its column and channel vector
  constexpr int PPP = EW_THREADS / VC;          // pixels per pass
  static_assert(PPP % UP_T == 0 || UP_T % PPP == 0, "pass shape");
  const int cv = tid % VC, pp = tid / VC;
  for (int p0 = 0; p0 < UP_T * UP_T; p0 += PPP) {
    const int pix = p0 + pp, oy = pix / UP_T, ox = pix % UP_T;
    const int Y = Y0 + oy, X = X0 + ox, ch = c0 + cv * G;
    if (Y >= Ho || X >= Wo || ch >= C) continue;
    int y0, y1, x0, x1;
    float hy0, hy1, wx0, wx1;
    up_taps(Y, sy, h, y0, y1, hy0, hy1);
    up_taps(X, sx, w, x0, x1, wx0, wx1);
    const unsigned char* tb = tile + cv * 16;
    float a[G], bq[G], c[G], d[G], r[G];
    typedef __attribute__((ext_vector_type(4))) unsigned int raw4;
    raw4 ra = *reinterpret_cast<const raw4*>(tb + ((y0 - ya) * UP_SRC + (x0 - xa)) * PITCH);
    raw4 rb = *reinterpret_cast<const raw4*>(tb + ((y0 - ya) * UP_SRC + (x1 - xa)) * PITCH);
    raw4 rc = *reinterpret_cast<const raw4*>(tb + ((y1 - ya) * UP_SRC + (x0 - xa)) * PITCH);
    raw4 rd = *reinterpret_cast<const raw4*>(tb + ((y1 - ya) * UP_SRC + (x1 - xa)) * PITCH);
    vload<T, G>(reinterpret_cast<const T*>(&ra), a);
    vload<T, G>(reinterpret_cast<const T*>(&rb), bq);
    vload<T, G>(reinterpret_cast<const T*>(&rc), c);
    vload<T, G>(reinterpret_cast<const T*>(&rd), d);
#pragma unroll
    for (int k = 0; k < G; k++) r[k] = hy0 * (wx0 * a[k] + wx1 * bq[k]) + hy1 * (wx0 * c[k] + wx1 * d[k]);
    T* dstp = out + (((int64_t)b * Ho + Y) * Wo + X) * Cdst + ch;
    vstore<T, G>(dstp, r);
#if MIS_UP_DEBUG
    {
      // Instrumentation (tools/up_repro.sh, built with -DMIS_UP_DEBUG=1): where does a wrong value come from?
      // Counters: [0..3] an LDS operand re-read after use differs from the value used; [4] the blended
      // registers differ from a scalar recomputation; [5] memory differs from the registers; [7] lanes checked.
      __builtin_amdgcn_sched_barrier(0);
      raw4 ra2 = *reinterpret_cast<const volatile raw4*>(tb + ((y0 - ya) * UP_SRC + (x0 - xa)) * PITCH);
      raw4 rb2 = *reinterpret_cast<const volatile raw4*>(tb + ((y0 - ya) * UP_SRC + (x1 - xa)) * PITCH);
      raw4 rc2 = *reinterpret_cast<const volatile raw4*>(tb + ((y1 - ya) * UP_SRC + (x0 - xa)) * PITCH);
      raw4 rd2 = *reinterpret_cast<const volatile raw4*>(tb + ((y1 - ya) * UP_SRC + (x1 - xa)) * PITCH);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra2), "+v"(rb2), "+v"(rc2), "+v"(rd2));
      bool da = false, db = false, dc = false, dd = false;
      for (int k = 0; k < 4; k++) {
        da |= ra2[k] != ra[k];
        db |= rb2[k] != rb[k];
        dc |= rc2[k] != rc[k];
        dd |= rd2[k] != rd[k];
      }
      if (da) atomicAdd(&up_dbg[0], 1u);
      if (db) atomicAdd(&up_dbg[1], 1u);
      if (dc) atomicAdd(&up_dbg[2], 1u);
      if (dd) atomicAdd(&up_dbg[3], 1u);
      // the same blend again, one element at a time, from the re-read values
      float a2[G], b2[G], c2[G], d2[G];
      vload<T, G>(reinterpret_cast<const T*>(&ra2), a2);
      vload<T, G>(reinterpret_cast<const T*>(&rb2), b2);
      vload<T, G>(reinterpret_cast<const T*>(&rc2), c2);
      vload<T, G>(reinterpret_cast<const T*>(&rd2), d2);
      union { raw4 u; T t[G]; } want, got;
      for (int k = 0; k < G; k++) {
        float top = wx0 * a2[k], bot = wx0 * c2[k];
        asm volatile("" : "+v"(top), "+v"(bot));
        top = __builtin_fmaf(wx1, b2[k], top);
        bot = __builtin_fmaf(wx1, d2[k], bot);
        asm volatile("" : "+v"(top), "+v"(bot));
        want.t[k] = from_f32<T>(__builtin_fmaf(hy1, bot, hy0 * top));
        got.t[k] = from_f32<T>(r[k]);
      }
      // values in registers differ by more than rounding?  (compare as floats with a tolerance of 2 bf16 steps)
      bool dreg = false;
      for (int k = 0; k < G; k++) {
        const float w_ = to_f32(want.t[k]), g_ = to_f32(got.t[k]);
        dreg |= fabsf(w_ - g_) > 0.02f * fmaxf(fabsf(w_), 1e-3f);
      }
      if (dreg) atomicAdd(&up_dbg[4], 1u);
      // what did memory receive?
      __threadfence();
      raw4 mem = *reinterpret_cast<const volatile raw4*>(dstp);
      bool dmem = false;
      for (int k = 0; k < 4; k++) dmem |= mem[k] != got.u[k];
      if (dmem) atomicAdd(&up_dbg[5], 1u);
      atomicAdd(&up_dbg[7], 1u);
    }
#endif
  }
}

// Backward (transposed interpolation) in gather form.  With scale (n-1)/(2n-1) < 1/2 the taps
// of output row Y are floor(Y*s) and the row after it, so low-resolution row r only receives
// from output rows 2r-1 .. 2r+2 (4 candidates; checked exhaustively for n <= 4096).  A workgroup
// owns an 8x8 low-resolution tile and a chunk of channels, stages the 18x18 high-resolution
// gradient pixels that can touch it (rows/cols 2*first-1 .. 2*first+16) in LDS, and every
// low-resolution pixel sums its <= 16 contributions in a fixed order (bitwise reproducible).
// `Csrc` is the channel count of a source pixel (== C, or C + Cskip when the gradient is the head
// of a concatenation's).
constexpr int UPB_T = 8;
constexpr int UPB_K = 4;                       // candidate rows/cols per low-resolution pixel
constexpr int UPB_SRC = 2 * UPB_T + UPB_K - 2;

template <typename T, int CC>
__global__ __launch_bounds__(EW_THREADS) void up_bwd_tiled_kernel(const T* __restrict__ dout, T* __restrict__ dlow,
                                                                  int B, int h, int w, int C, int Csrc, float sy,
                                                                  float sx, int tiles_x, int tiles_y,
                                                                  const float* __restrict__ noise) {
  constexpr int G = 16 / (int)sizeof(T);
  constexpr int VC = CC / G;
  constexpr int PITCH = VC * 16 + 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char btile[];      // [UPB_SRC*UPB_SRC][PITCH]
  const int tid = threadIdx.x;
  const int c0 = blockIdx.y * CC;
  const int t = blockIdx.x;
  const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
  const int Ho = 2 * h, Wo = 2 * w;
  const int yl0 = ty * UPB_T, xl0 = tx * UPB_T, YA = 2 * yl0 - 1, XA = 2 * xl0 - 1;
  const T* src = dout + (int64_t)b * Ho * Wo * Csrc;
  // all of a thread's staging loads are issued before the first LDS store (clamped address, zero
  // selected outside the image): one memory latency per tile instead of one per element
  constexpr int NV = UPB_SRC * UPB_SRC * VC, IT = (NV + EW_THREADS - 1) / EW_THREADS;
  uint4 stg[IT];
#pragma unroll
  for (int i = 0; i < IT; i++) {
    const int e = tid + i * EW_THREADS;
    const int cv = e % VC, pos = e / VC;
    const int Y = YA + pos / UPB_SRC, X = XA + pos % UPB_SRC, ch = c0 + cv * G;
    const bool ok = e < NV && Y >= 0 && Y < Ho && X >= 0 && X < Wo && ch < C;
    const uint4 v = *reinterpret_cast<const uint4*>(src + (ok ? ((int64_t)Y * Wo + X) * Csrc + ch : 0));
    stg[i] = ok ? v : uint4{0u, 0u, 0u, 0u};
  }
#pragma unroll
  for (int i = 0; i < IT; i++) {
    const int e = tid + i * EW_THREADS;
    if (e < NV) *reinterpret_cast<uint4*>(btile + (e / VC) * PITCH + (e % VC) * 16) = stg[i];
  }
  __syncthreads();
  for (int i = tid; i < UPB_T * UPB_T * VC; i += EW_THREADS) {
    const int cv = i % VC, pix = i / VC;
    const int yl = yl0 + pix / UPB_T, xl = xl0 + pix % UPB_T, ch = c0 + cv * G;
    if (yl >= h || xl >= w || ch >= C) continue;
    float wy[UPB_K], wxv[UPB_K];
#pragma unroll
    for (int k = 0; k < UPB_K; k++) {
      int i0, i1;
      float l0, l1;
      const int Y = 2 * yl - 1 + k;
      float wgt = 0.f;
      if (Y >= 0 && Y < Ho) {
        up_taps(Y, sy, h, i0, i1, l0, l1);
        wgt = ((i0 == yl) ? l0 : 0.f) + ((i1 == yl) ? l1 : 0.f);
      }
      wy[k] = wgt;
      const int X = 2 * xl - 1 + k;
      wgt = 0.f;
      if (X >= 0 && X < Wo) {
        up_taps(X, sx, w, i0, i1, l0, l1);
        wgt = ((i0 == xl) ? l0 : 0.f) + ((i1 == xl) ? l1 : 0.f);
      }
      wxv[k] = wgt;
    }
    float acc[G];
#pragma unroll
    for (int k = 0; k < G; k++) acc[k] = 0.f;
    // staged position of candidate (ky, kx): (2*(yl-yl0) + ky, 2*(xl-xl0) + kx)
    const unsigned char* tb = btile + ((2 * (yl - yl0)) * UPB_SRC + 2 * (xl - xl0)) * PITCH + cv * 16;
#pragma unroll
    for (int ky = 0; ky < UPB_K; ky++) {
      if (wy[ky] == 0.f) continue;            // also skips rows outside the image (never staged)
      float row[G];
#pragma unroll
      for (int k = 0; k < G; k++) row[k] = 0.f;
#pragma unroll
      for (int kx = 0; kx < UPB_K; kx++) {
        if (wxv[kx] == 0.f) continue;
        float g[G];
        vload<T, G>(reinterpret_cast<const T*>(tb + (ky * UPB_SRC + kx) * PITCH), g);
#pragma unroll
        for (int k = 0; k < G; k++) row[k] += wxv[kx] * g[k];
      }
#pragma unroll
      for (int k = 0; k < G; k++) acc[k] += wy[ky] * row[k];
    }
    if (noise) {
#pragma unroll
      for (int k = 0; k < G; k++) acc[k] = to_f32(from_f32<T>(acc[k])) * noise[(int64_t)b * C + ch + k];
    }
    vstore<T, G>(dlow + (((int64_t)b * h + yl) * w + xl) * C + ch, acc);
  }
}

// Scalar fallbacks (channel counts that are not a multiple of the 16-byte vector).
template <typename T>
__global__ __launch_bounds__(EW_THREADS) void up_fwd_scalar_kernel(const T* __restrict__ low, T* __restrict__ out,
                                                                   int B, int h, int w, int C, int Cdst, float sy,
                                                                   float sx, const float* __restrict__ low_aff,
                                                                   const float* __restrict__ noise) {
  const int Ho = 2 * h, Wo = 2 * w;
  const int64_t total = (int64_t)B * Ho * Wo * C;
  for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_THREADS) {
    const int c = (int)(i % C);
    int64_t p = i / C;
    const int xo = (int)(p % Wo);
    p /= Wo;
    const int yo = (int)(p % Ho);
    const int b = (int)(p / Ho);
    int y0, y1, x0, x1;
    float hy0, hy1, wx0, wx1;
    up_taps(yo, sy, h, y0, y1, hy0, hy1);
    up_taps(xo, sx, w, x0, x1, wx0, wx1);
    const T* base = low + (int64_t)b * h * w * C + c;
    float v[4] = {to_f32(base[((int64_t)y0 * w + x0) * C]), to_f32(base[((int64_t)y0 * w + x1) * C]),
                  to_f32(base[((int64_t)y1 * w + x0) * C]), to_f32(base[((int64_t)y1 * w + x1) * C])};
    if (low_aff) {
#pragma unroll
      for (int k = 0; k < 4; k++) v[k] = to_f32(from_f32<T>(fmaxf(v[k] * low_aff[c] + low_aff[C + c], 0.f)));
    }
    if (noise) {
#pragma unroll
      for (int k = 0; k < 4; k++) v[k] = to_f32(from_f32<T>(v[k] * noise[(int64_t)b * C + c]));
    }
    out[(((int64_t)b * Ho + yo) * Wo + xo) * Cdst + c] =
        from_f32<T>(hy0 * (wx0 * v[0] + wx1 * v[1]) + hy1 * (wx0 * v[2] + wx1 * v[3]));
  }
}

// Gather form of the transposed interpolation: a low-res pixel collects from every
// high-res pixel whose taps touch it (|offset| <= 4 rows/cols), with the forward's weights.
template <typename T, int V>
__global__ __launch_bounds__(EW_THREADS) void up_bwd_kernel(const T* __restrict__ dout, T* __restrict__ dlow, int B,
                                                            int h, int w, int Cup, int Ct, float sy, float sx,
                                                            const float* __restrict__ noise) {
  const int Ho = 2 * h, Wo = 2 * w, CV = Cup / V;
  const int64_t total = (int64_t)B * h * w * CV;
  for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_THREADS) {
    const int cv = (int)(i % CV);
    int64_t p = i / CV;
    const int xl = (int)(p % w);
    p /= w;
    const int yl = (int)(p % h);
    const int b = (int)(p / h);
    float wy[8], wxv[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      int i0, i1;
      float l0, l1;
      const int Y = 2 * yl - 3 + k;
      float wgt = 0.f;
      if (Y >= 0 && Y < Ho) {
        up_taps(Y, sy, h, i0, i1, l0, l1);
        wgt = ((i0 == yl) ? l0 : 0.f) + ((i1 == yl) ? l1 : 0.f);
      }
      wy[k] = wgt;
      const int X = 2 * xl - 3 + k;
      wgt = 0.f;
      if (X >= 0 && X < Wo) {
        up_taps(X, sx, w, i0, i1, l0, l1);
        wgt = ((i0 == xl) ? l0 : 0.f) + ((i1 == xl) ? l1 : 0.f);
      }
      wxv[k] = wgt;
    }
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; k++) acc[k] = 0.f;
    const T* base = dout + (int64_t)b * Ho * Wo * Ct + cv * V;
#pragma unroll
    for (int ky = 0; ky < 8; ky++) {
      if (wy[ky] == 0.f) continue;
      const int Y = 2 * yl - 3 + ky;
      float row[V];
#pragma unroll
      for (int k = 0; k < V; k++) row[k] = 0.f;
#pragma unroll
      for (int kx = 0; kx < 8; kx++) {
        if (wxv[kx] == 0.f) continue;
        const int X = 2 * xl - 3 + kx;
        float g[V];
        vload<T, V>(base + ((int64_t)Y * Wo + X) * Ct, g);
#pragma unroll
        for (int k = 0; k < V; k++) row[k] += wxv[kx] * g[k];
      }
#pragma unroll
      for (int k = 0; k < V; k++) acc[k] += wy[ky] * row[k];
    }
    if (noise) {
#pragma unroll
      for (int k = 0; k < V; k++) acc[k] = to_f32(from_f32<T>(acc[k])) * noise[(int64_t)b * Cup + cv * V + k];
    }
    vstore<T, V>(dlow + i * V, acc);
  }
}

// dst[pix][d0 : d0+Cs] = act(src[pix][s0 : s0+Cs]) between tensors of Cdst / Csrc channels per
// pixel (the skip half of a concatenation and the slice of its gradient).
template <typename T, int V>
__global__ __launch_bounds__(EW_THREADS) void copy_channels_kernel(const T* __restrict__ src, T* __restrict__ dst,
                                                                   int64_t n_pix, int Csrc, int s0, int Cdst, int d0,
                                                                   int Cs, const float* __restrict__ aff) {
  const int CV = Cs / V;
  const int64_t total = n_pix * CV;
  for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_THREADS) {
    const int cv = (int)(i % CV);
    const int64_t p = i / CV;
    float v[V];
    vload<T, V>(src + p * Csrc + s0 + cv * V, v);
    affine_relu<V>(v, aff, Cs, cv * V);
    vstore<T, V>(dst + p * Cdst + d0 + cv * V, v);
  }
}

template <typename T>
int pool_fwd(const void* x, void* y, int B, int H, int W, int C, const float* aff, const float* noise,
             hipStream_t st) {
  constexpr int N = VecOf<T>::N;
  const int64_t n = (int64_t)B * (H / 2) * (W / 2) * C;
  const size_t lds = aff ? 2 * (size_t)C * sizeof(float) : 0;
  if (pick_vec<T>(C, x, y) == N)
    maxpool_fwd_kernel<T, N><<<ew_grid(n / N, EW_THREADS), EW_THREADS, lds, st>>>((const T*)x, (T*)y, B, H, W, C, aff, noise);
  else
    maxpool_fwd_kernel<T, 1><<<ew_grid(n, EW_THREADS), EW_THREADS, lds, st>>>((const T*)x, (T*)y, B, H, W, C, aff, noise);
  return 0;
}

template <typename T>
int pool_bwd(const void* x, const void* dy, const void* add, void* dx, int B, int H, int W, int C, const float* aff,
             const float* noise, float* bn_sums, void* ws, hipStream_t st) {
  constexpr int N = VecOf<T>::N;
  const int64_t n = (int64_t)B * (H / 2) * (W / 2) * C;
  const bool vec = pick_vec<T>(C, x, dy, dx, add) == N;
  const int V = vec ? N : 1;
  size_t lds = aff ? 2 * (size_t)C * sizeof(float) : 0;
  int grid = ew_grid(n / V, EW_THREADS);
  float* partials = nullptr;
  if (bn_sums) {
    // fixed channel vector per thread; one partial row per workgroup
    // (6 workgroups per CU measured best at the sizes this path is used for: 1024 -> 1536 rows took
    // 8 % off the kernel, 2048+ gave it back in the row reduction)
    const int cap = 1536;
    grid = grid_for_fixed_channels(grid, C / V, cap);
    const size_t red = (size_t)EW_THREADS * 2 * V * sizeof(float);
    if (red > lds) lds = red;
    partials = (float*)ws;
  }
  if (vec)
    maxpool_bwd_kernel<T, N><<<grid, EW_THREADS, lds, st>>>((const T*)x, (const T*)dy, (const T*)add, (T*)dx, B, H, W,
                                                             C, aff, noise, partials);
  else
    maxpool_bwd_kernel<T, 1><<<grid, EW_THREADS, lds, st>>>((const T*)x, (const T*)dy, (const T*)add, (T*)dx, B, H, W,
                                                             C, aff, noise, partials);
  if (bn_sums) launch_reduce_partials(partials, grid, 2 * C, bn_sums, st);
  return 0;
}

inline float up_scale(int n_in) { return (2 * n_in > 1) ? (float)(n_in - 1) / (float)(2 * n_in - 1) : 0.f; }

template <typename T>
int upcat_fwd(const void* low, const void* skip, void* out, int B, int h, int w, int Cup, int Cskip,
              const float* low_aff, const float* skip_aff, const float* noise, hipStream_t st) {
  constexpr int N = VecOf<T>::N;
  const int Ct = Cup + Cskip;
  const bool vec = (Cup % N == 0) && (Cskip % N == 0) && pick_vec<T>(Ct, low, skip, out) == N;
  if (vec) {
    const int tiles_x = (2 * w + UP_T - 1) / UP_T, tiles_y = (2 * h + UP_T - 1) / UP_T;
    // channel chunk per workgroup: 128 bytes of a pixel, or 64 when that leaves no idle lanes
    if (Cup % (8 * N) == 0) {
      constexpr int CC = 8 * N;
      dim3 grid(B * tiles_x * tiles_y, Cup / CC);
      up_fwd_tiled_kernel<T, CC><<<grid, EW_THREADS, 0, st>>>((const T*)low, (T*)out, B, h, w, Cup, Ct, up_scale(h),
                                                              up_scale(w), tiles_x, tiles_y, low_aff, noise);
    } else {
      constexpr int CC = 4 * N;
      dim3 grid(B * tiles_x * tiles_y, (Cup + CC - 1) / CC);
      up_fwd_tiled_kernel<T, CC><<<grid, EW_THREADS, 0, st>>>((const T*)low, (T*)out, B, h, w, Cup, Ct, up_scale(h),
                                                              up_scale(w), tiles_x, tiles_y, low_aff, noise);
    }
  } else {
    const int64_t n = (int64_t)B * 4 * h * w * Cup;
    up_fwd_scalar_kernel<T><<<ew_grid(n, EW_THREADS), EW_THREADS, 0, st>>>((const T*)low, (T*)out, B, h, w, Cup, Ct,
                                                                           up_scale(h), up_scale(w), low_aff, noise);
  }
  if (Cskip > 0) {
    const int64_t npix = (int64_t)B * 4 * h * w;
    const size_t lds = 0;
    if (vec)
      copy_channels_kernel<T, N><<<ew_grid(npix * Cskip / N, EW_THREADS), EW_THREADS, lds, st>>>(
          (const T*)skip, (T*)out, npix, Cskip, 0, Ct, Cup, Cskip, skip_aff);
    else
      copy_channels_kernel<T, 1><<<ew_grid(npix * Cskip, EW_THREADS), EW_THREADS, lds, st>>>(
          (const T*)skip, (T*)out, npix, Cskip, 0, Ct, Cup, Cskip, skip_aff);
  }
  return 0;
}

template <typename T>
int upcat_bwd(const void* dout, void* dlow, void* dskip, int B, int h, int w, int Cup, int Cskip, const float* noise,
              hipStream_t st) {
  constexpr int N = VecOf<T>::N;
  const int Ct = Cup + Cskip;
  const bool vec = (Cup % N == 0) && (Cskip % N == 0) && pick_vec<T>(Ct, dout, dlow, dskip) == N;
  const int64_t nl = (int64_t)B * h * w * Cup;
  const int64_t npix = (int64_t)B * 4 * h * w;
  if (vec) {
    const int tiles_x = (w + UPB_T - 1) / UPB_T, tiles_y = (h + UPB_T - 1) / UPB_T;
    // channel chunk per workgroup: 128 bytes of a pixel when the channel count allows, else 64
    if (Cup % (8 * N) == 0) {
      constexpr int CC = 8 * N;
      const size_t lds = (size_t)UPB_SRC * UPB_SRC * (CC / N * 16 + 16);
      up_bwd_tiled_kernel<T, CC><<<dim3(B * tiles_x * tiles_y, Cup / CC), EW_THREADS, lds, st>>>(
          (const T*)dout, (T*)dlow, B, h, w, Cup, Ct, up_scale(h), up_scale(w), tiles_x, tiles_y, noise);
    } else {
      constexpr int CC = 4 * N;
      const size_t lds = (size_t)UPB_SRC * UPB_SRC * (CC / N * 16 + 16);
      up_bwd_tiled_kernel<T, CC><<<dim3(B * tiles_x * tiles_y, (Cup + CC - 1) / CC), EW_THREADS, lds, st>>>(
          (const T*)dout, (T*)dlow, B, h, w, Cup, Ct, up_scale(h), up_scale(w), tiles_x, tiles_y, noise);
    }
    if (dskip && Cskip > 0)
      copy_channels_kernel<T, N><<<ew_grid(npix * Cskip / N, EW_THREADS), EW_THREADS, 0, st>>>(
          (const T*)dout, (T*)dskip, npix, Ct, Cup, Cskip, 0, Cskip, nullptr);
  } else {
    up_bwd_kernel<T, 1><<<ew_grid(nl, EW_THREADS), EW_THREADS, 0, st>>>((const T*)dout, (T*)dlow, B, h, w, Cup, Ct,
                                                                          up_scale(h), up_scale(w), noise);
    if (dskip && Cskip > 0)
      copy_channels_kernel<T, 1><<<ew_grid(npix * Cskip, EW_THREADS), EW_THREADS, 0, st>>>(
          (const T*)dout, (T*)dskip, npix, Ct, Cup, Cskip, 0, Cskip, nullptr);
  }
  return 0;
}

}  // namespace

// ---------------------------------------------------------------------------
// nn.Dropout2d multipliers from a counter-based generator (Philox4x32-10, Salmon et al. 2011):
// element i = b*C + c of call `offset` under `seed` is word 0 of philox(counter = (i, 0, offset_lo,
// offset_hi), key = (seed_lo, seed_hi)); kept when its top 24 bits, as a fraction, are below 1-p.
// No state and no order: the value of an element depends on (seed, offset, b, c) only, whatever
// the launch shape, so a captured graph or a second rank can reproduce it.
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned philox4x32_10_word0(unsigned c0, unsigned c1, unsigned c2, unsigned c3,
                                                        unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    c1 = (unsigned)p1;
    c3 = (unsigned)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c0;
}

__global__ void dropout2d_noise_kernel(unsigned long long seed, unsigned long long offset, int n, float keep,
                                       float scale, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned w = philox4x32_10_word0((unsigned)i, 0u, (unsigned)offset, (unsigned)(offset >> 32), (unsigned)seed,
                                         (unsigned)(seed >> 32));
  const float u = (float)(w >> 8) * (1.0f / 16777216.0f);       // [0, 1) in steps of 2^-24, exact
  out[i] = u < keep ? scale : 0.f;
}

extern "C" {

int mis_maxpool2x2_fwd(const void* x, const float* x_affine, const float* noise, int dtype, int B, int H, int W, int C,
                       void* y, mis_stream_t stream) {
  MIS_REQUIRE(x && y, "maxpool2x2_fwd: null pointer");
  MIS_REQUIRE(B >= 1 && C >= 1 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0,
              "maxpool2x2_fwd: H and W must be even and >= 2 (got %dx%d)", H, W);
  MIS_REQUIRE(C <= 4096, "maxpool2x2_fwd: C too large");
  if (dtype == MIS_F32) pool_fwd<float>(x, y, B, H, W, C, x_affine, noise, (hipStream_t)stream);
  else if (dtype == MIS_BF16) pool_fwd<bf16>(x, y, B, H, W, C, x_affine, noise, (hipStream_t)stream);
  else MIS_REQUIRE(false, "maxpool2x2_fwd: bad dtype %d", dtype);
  MIS_CHECK_LAUNCH("maxpool_fwd_kernel");
  return MIS_OK;
}

size_t mis_maxpool2x2_bwd_workspace_bytes(int C) {
  return C < 1 ? 0 : align256((size_t)POOL_MAX_ROWS * 2 * C * sizeof(float));
}

int mis_maxpool2x2_bwd(const void* x, const float* x_affine, const float* noise, const void* dy, const void* dx_add,
                       int dtype, int B, int H, int W, int C, void* dx, float* bn_sums, void* workspace,
                       mis_stream_t stream) {
  MIS_REQUIRE(x && dy && (dx || bn_sums), "maxpool2x2_bwd: null pointer");
  MIS_REQUIRE(!bn_sums || (x_affine && workspace), "maxpool2x2_bwd: bn_sums needs the [4][C] affine and a workspace");
  MIS_REQUIRE(B >= 1 && C >= 1 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0,
              "maxpool2x2_bwd: H and W must be even and >= 2 (got %dx%d)", H, W);
  MIS_REQUIRE(C <= 4096, "maxpool2x2_bwd: C too large");
  if (dtype == MIS_F32) pool_bwd<float>(x, dy, dx_add, dx, B, H, W, C, x_affine, noise, bn_sums, workspace, (hipStream_t)stream);
  else if (dtype == MIS_BF16) pool_bwd<bf16>(x, dy, dx_add, dx, B, H, W, C, x_affine, noise, bn_sums, workspace, (hipStream_t)stream);
  else MIS_REQUIRE(false, "maxpool2x2_bwd: bad dtype %d", dtype);
  MIS_CHECK_LAUNCH("maxpool_bwd_kernel");
  return MIS_OK;
}

int mis_upsample2x_cat_fwd(const void* low, const float* low_affine, const float* low_noise, const void* skip,
                           const float* skip_affine, int dtype, int B, int h, int w, int Cup, int Cskip, void* out,
                           mis_stream_t stream) {
  MIS_REQUIRE(low && out && (skip || Cskip == 0), "upsample2x_cat_fwd: null pointer");
  MIS_REQUIRE(B >= 1 && h >= 1 && w >= 1 && Cup >= 1 && Cskip >= 0, "upsample2x_cat_fwd: bad shape");
  MIS_REQUIRE(Cup + Cskip <= 4096, "upsample2x_cat_fwd: too many channels");
  if (dtype == MIS_F32)
    upcat_fwd<float>(low, skip, out, B, h, w, Cup, Cskip, low_affine, skip_affine, low_noise, (hipStream_t)stream);
  else if (dtype == MIS_BF16)
    upcat_fwd<bf16>(low, skip, out, B, h, w, Cup, Cskip, low_affine, skip_affine, low_noise, (hipStream_t)stream);
  else MIS_REQUIRE(false, "upsample2x_cat_fwd: bad dtype %d", dtype);
  MIS_CHECK_LAUNCH("up_fwd_tiled_kernel");
  return MIS_OK;
}

int mis_upsample2x_cat_bwd(const void* dout, const float* low_noise, int dtype, int B, int h, int w, int Cup, int Cskip,
                           void* dlow, void* dskip, mis_stream_t stream) {
  MIS_REQUIRE(dout && dlow && (dskip || Cskip == 0), "upsample2x_cat_bwd: null pointer");
  MIS_REQUIRE(B >= 1 && h >= 1 && w >= 1 && Cup >= 1 && Cskip >= 0, "upsample2x_cat_bwd: bad shape");
  if (dtype == MIS_F32) upcat_bwd<float>(dout, dlow, dskip, B, h, w, Cup, Cskip, low_noise, (hipStream_t)stream);
  else if (dtype == MIS_BF16) upcat_bwd<bf16>(dout, dlow, dskip, B, h, w, Cup, Cskip, low_noise, (hipStream_t)stream);
  else MIS_REQUIRE(false, "upsample2x_cat_bwd: bad dtype %d", dtype);
  MIS_CHECK_LAUNCH("up_bwd_tiled_kernel");
  return MIS_OK;
}

int mis_dropout2d_noise(uint64_t seed, uint64_t offset, int B, int C, float p, float* noise, mis_stream_t stream) {
  MIS_REQUIRE(noise, "dropout2d_noise: null pointer");
  MIS_REQUIRE(B >= 1 && C >= 1 && (int64_t)B * C < ((int64_t)1 << 31), "dropout2d_noise: bad shape");
  MIS_REQUIRE(p >= 0.f && p <= 1.f, "dropout2d_noise: p = %g is not a probability", (double)p);
  const int n = B * C;
  const float keep = 1.f - p, scale = p < 1.f ? 1.f / (1.f - p) : 0.f;
  dropout2d_noise_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(seed, offset, n, keep, scale, noise);
  MIS_CHECK_LAUNCH("dropout2d_noise_kernel");
  return MIS_OK;
}

}  // extern "C"

#if MIS_UP_DEBUG
extern "C" int mis_debug_up_counters(unsigned int* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(up_dbg), sizeof(up_dbg));
}
#endif
