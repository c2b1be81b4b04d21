// Vector-ALU convolution kernels: the image-facing first layer (Cin = 1 or 3, where the
// GEMM K dimension is too short for the matrix cores), the 1x1 output layer, the weight
// packers, and a shape-agnostic 3x3 fallback used for channel counts the MFMA kernels do
// not take.  Channels-last activations, float32 accumulation.
//
// Roofline: HBM for the production uses (first layer: read Cin + write Cout per pixel;
// 1x1: read Cin + write Cout per pixel).
#include <type_traits>
#include "common.h"
#include "vec.h"
#include "reduce.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));


constexpr int CD_THREADS = 256;
constexpr int CO_T = 8;          // output channels per thread
constexpr int CI_CHUNK = 32;     // input channels staged in LDS per pass

// ---------------------------------------------------------------------------------
// weight packers
// ---------------------------------------------------------------------------------
template <typename T>
__global__ void pack_w3x3_kernel(const float* __restrict__ w, int Cout, int Cin, T* __restrict__ w_fwd,
                                 T* __restrict__ w_dgrad) {
  const int64_t n = (int64_t)Cout * Cin * 9;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int tap = (int)(i % 9);
    const int ci = (int)((i / 9) % Cin);
    const int co = (int)(i / (9 * (int64_t)Cin));
    const float v = w[i];
    if (w_fwd) w_fwd[((int64_t)tap * Cout + co) * Cin + ci] = from_f32<T>(v);
    if (w_dgrad) w_dgrad[((int64_t)(8 - tap) * Cin + ci) * Cout + co] = from_f32<T>(v);
  }
}

// ---------------------------------------------------------------------------------
// direct 3x3 forward: one thread = one pixel x CO_T output channels
// ---------------------------------------------------------------------------------
// COT: output channels per thread (8, or fewer for narrow layers: with 8 a 1-channel output spends 7/8 of its
// multiply-adds and LDS weight reads on padding)
template <typename T, int V, int COT = CO_T>
__global__ __launch_bounds__(CD_THREADS) void conv3x3_direct_kernel(
    const T* __restrict__ x, const T* __restrict__ wp /*[9][Cout][Cin]*/, const float* __restrict__ bias,
    T* __restrict__ z, float* __restrict__ partials /*[gridDim.x][2][Cout] or null*/, int B, int H, int W, int Cin,
    int Cout, const float* __restrict__ in_aff) {
  __shared__ float wl[9][CI_CHUNK][COT];
  __shared__ float red[2 * COT * (CD_THREADS / 64)];
  const int tid = threadIdx.x;
  const int co0 = blockIdx.y * COT;
  const int64_t n_pix = (int64_t)B * H * W;
  const bool one_chunk = Cin <= CI_CHUNK;           // weights staged once for the whole pixel walk
  float st[2 * COT];
#pragma unroll
  for (int j = 0; j < 2 * COT; j++) st[j] = 0.f;
  float bv[COT];
#pragma unroll
  for (int j = 0; j < COT; j++) bv[j] = (bias && co0 + j < Cout) ? bias[co0 + j] : 0.f;

  auto stage_w = [&](int c0, int cn) {
    for (int i = tid; i < 9 * CI_CHUNK * COT; i += CD_THREADS) {
      const int j = i % COT, ci = (i / COT) % CI_CHUNK, tap = i / (COT * CI_CHUNK);
      float v = 0.f;
      if (ci < cn && co0 + j < Cout) v = to_f32(wp[((int64_t)tap * Cout + co0 + j) * Cin + c0 + ci]);
      wl[tap][ci][j] = v;
    }
  };
  if (one_chunk) {
    stage_w(0, Cin);
    __syncthreads();
  }
  // persistent walk over 256-pixel groups (trip count is block-uniform: barriers inside are safe)
  for (int64_t pix0 = (int64_t)blockIdx.x * CD_THREADS; pix0 < n_pix; pix0 += (int64_t)gridDim.x * CD_THREADS) {
    const int64_t pix = pix0 + tid;
    const bool valid = pix < n_pix;
    int px = 0, py = 0;
    if (valid) {
      // 32-bit coordinates (the host checks B*H*W < 2^31): 64-bit division is emulated and slow
      const unsigned p32 = (unsigned)pix, row = p32 / (unsigned)W;
      px = (int)(p32 - row * (unsigned)W);
      py = (int)(row % (unsigned)H);
    }
    float acc[COT];
#pragma unroll
    for (int j = 0; j < COT; j++) acc[j] = 0.f;
    for (int c0 = 0; c0 < Cin; c0 += CI_CHUNK) {
      const int cn = (Cin - c0 < CI_CHUNK) ? (Cin - c0) : CI_CHUNK;
      if (!one_chunk) {
        __syncthreads();
        stage_w(c0, cn);
        __syncthreads();
      }
      if (valid) {
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
          const int yy = py + tap / 3 - 1, xx = px + tap % 3 - 1;
          if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
          const T* xp = x + (pix + (int64_t)(tap / 3 - 1) * W + (tap % 3 - 1)) * Cin + c0;
          for (int ci = 0; ci < cn; ci += V) {
            float xv[V];
            vload<T, V>(xp + ci, xv);
            affine_relu_rounded<T, V>(xv, in_aff, Cin, c0 + ci);
#pragma unroll
            for (int k = 0; k < V; k++)
#pragma unroll
              for (int j = 0; j < COT; j++) acc[j] += xv[k] * wl[tap][ci + k][j];
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < COT; j++) {
      acc[j] += bv[j];
      if (valid) {
        st[j] += acc[j];
        st[COT + j] += acc[j] * acc[j];
      }
    }
    if (valid) {
      T* zp = z + pix * Cout + co0;
      bool stored = false;
      if constexpr (sizeof(T) * COT >= 16) {
        if (co0 + COT <= Cout && (Cout % COT) == 0) {
          stored = true;
          if constexpr (sizeof(T) * COT == 16) {
            vstore<T, COT>(zp, acc);
          } else {
            float lo[4] = {acc[0], acc[1], acc[2], acc[3]}, hi[4] = {acc[4], acc[5], acc[6], acc[7]};
            vstore<T, 4>(zp, lo);
            vstore<T, 4>(zp + 4, hi);
          }
        }
      }
      if (!stored) {
#pragma unroll
        for (int j = 0; j < COT; j++)
          if (co0 + j < Cout) zp[j] = from_f32<T>(acc[j]);
      }
    }
  }
  if (partials) {
    block_sum<2 * COT>(st, red);
    if (tid < 2 * COT) {
      float v = 0.f;
#pragma unroll
      for (int i = 0; i < 2 * COT; i++) v = (tid == i) ? st[i] : v;
      const int which = tid / COT, j = tid % COT;
      if (co0 + j < Cout) partials[((int64_t)blockIdx.x * 2 + which) * Cout + co0 + j] = v;
    }
  }
}

// ---------------------------------------------------------------------------------
// image-facing first layer (CIN = 1 or 3 input channels).  A workgroup stages the (16+2)x(64+2)
// halo of a 16x64-pixel tile in LDS as float32 (lazy-input affine applied, padding = 0), so the
// compute loop has no bounds checks and no global address arithmetic: a lane owns one column and
// 4 consecutive rows, reads its 6x3 window once (4.5 LDS reads per pixel instead of 9 global
// loads), and stores 16 bytes per pixel with consecutive lanes on consecutive pixels.
// HBM-bound on the output write.
// ---------------------------------------------------------------------------------
constexpr int F_TW = 64, F_TH = 16;

// NSL > 1 (no statistics wanted, Cout <= NSL * CO_T): one workgroup computes all NSL channel slices of its
// tiles from one staged halo, slice after slice, so a pixel's 16-byte pieces are stored back to back by the
// same lane (with one slice per workgroup, grid.y, they arrive from different workgroups at different times:
// measured 339 us for 3 -> 32 channels at 16 x 512^2, a quarter of the write rate).
template <typename T, int CIN, int NSL = 1>
__global__ __launch_bounds__(CD_THREADS) void conv3x3_first_kernel(
    const T* __restrict__ x, const T* __restrict__ wp /*[9][Cout][CIN]*/, const float* __restrict__ bias,
    T* __restrict__ z, float* __restrict__ partials /*[gridDim.x][2][Cout] or null*/, int B, int H, int W, int Cout,
    const float* __restrict__ in_aff, int tiles_x, int tiles_y, int n_tiles) {
  __shared__ __attribute__((aligned(16))) float wl[NSL][9 * CIN][CO_T];
  __shared__ float red[2 * CO_T * (CD_THREADS / 64)];
  __shared__ float halo[CIN][F_TH + 2][F_TW + 2];
  __shared__ float bl[NSL][CO_T];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int co_base = NSL > 1 ? 0 : blockIdx.y * CO_T;
  for (int i = tid; i < NSL * 9 * CIN * CO_T; i += CD_THREADS) {
    const int j = i % CO_T, k = (i / CO_T) % (9 * CIN), sl = i / (CO_T * 9 * CIN), ci = k % CIN, tap = k / CIN;
    const int co = co_base + sl * CO_T + j;
    wl[sl][k][j] = (co < Cout) ? to_f32(wp[((int64_t)tap * Cout + co) * CIN + ci]) : 0.f;
  }
  for (int i = tid; i < NSL * CO_T; i += CD_THREADS) {
    const int co = co_base + i;
    bl[i / CO_T][i % CO_T] = (bias && co < Cout) ? bias[co] : 0.f;
  }
  float st[2 * CO_T];
#pragma unroll
  for (int j = 0; j < CO_T; j++) st[j] = st[CO_T + j] = 0.f;
  float asc[CIN], ash[CIN];
#pragma unroll
  for (int c = 0; c < CIN; c++) {
    asc[c] = in_aff ? in_aff[c] : 1.f;
    ash[c] = in_aff ? in_aff[CIN + c] : 0.f;
  }
  const bool vec_store = (Cout % CO_T == 0) && (((uintptr_t)z & 15) == 0);
  // The halo of the workgroup's next tile is requested (raw, clamped address, validity kept as a
  // bit mask) before the current tile is computed, so its latency hides under the FMAs and stores.
  constexpr int F_HALO = (F_TH + 2) * (F_TW + 2), F_IT = (F_HALO + CD_THREADS - 1) / CD_THREADS;
  T raw[F_IT][CIN];
  unsigned okm = 0;
  auto request = [&](int t) {
    const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
    okm = 0;
#pragma unroll
    for (int i = 0; i < F_IT; i++) {
      const int e = tid + i * CD_THREADS;
      const int hy = e / (F_TW + 2), hx = e % (F_TW + 2);
      const int gy = ty * F_TH - 1 + hy, gx = tx * F_TW - 1 + hx;
      const bool ok = e < F_HALO && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      okm |= ok ? (1u << i) : 0u;
      const T* xp = x + (ok ? ((int64_t)(b * H + gy) * W + gx) * CIN : 0);
#pragma unroll
      for (int c = 0; c < CIN; c++) raw[i][c] = xp[c];
    }
  };
  if ((int)blockIdx.x < n_tiles) request(blockIdx.x);
  for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
    const int x0 = tx * F_TW, y0 = ty * F_TH;
    __syncthreads();                                   // previous tile's readers (and the weight staging)
#pragma unroll
    for (int i = 0; i < F_IT; i++) {
      const int e = tid + i * CD_THREADS;
      if (e < F_HALO) {
        const int hy = e / (F_TW + 2), hx = e % (F_TW + 2);
#pragma unroll
        for (int c = 0; c < CIN; c++) {
          float v = to_f32(raw[i][c]);
          if (in_aff) v = to_f32(from_f32<T>(fmaxf(v * asc[c] + ash[c], 0.f)));
          halo[c][hy][hx] = ((okm >> i) & 1u) ? v : 0.f;
        }
      }
    }
    __syncthreads();
    if (t + (int)gridDim.x < n_tiles) request(t + gridDim.x);
    float xv[CIN][6][3];
#pragma unroll
    for (int c = 0; c < CIN; c++)
#pragma unroll
      for (int rr = 0; rr < 6; rr++)
#pragma unroll
        for (int kx = 0; kx < 3; kx++) xv[c][rr][kx] = halo[c][4 * wv + rr][lane + kx];
#pragma unroll 1
    for (int sl = 0; sl < NSL; sl++) {
    const int co0 = co_base + sl * CO_T;
    if (NSL > 1 && co0 >= Cout) break;                 // (3 slices run in the 4-slice instantiation)
    // two output channels per multiply-add instruction (v_pk_fma_f32, the input value broadcast to both halves):
    // the layer is bound by the vector ALU — 9*CIN*CO_T multiply-adds per pixel against 16 stored bytes
    f32x2 acc2[4][CO_T / 2];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int j = 0; j < CO_T / 2; j++) acc2[r][j] = f32x2{bl[sl][2 * j], bl[sl][2 * j + 1]};
    // The weights of one tap (CIN x CO_T floats) are read from LDS while the previous tap is multiplied; the
    // scheduling barrier keeps the compiler from hoisting all 9*CIN*CO_T of them into registers at once (with
    // three input channels that took 216 registers and spilled 200).
    f32x4 wq[2][CIN][2];
#pragma unroll
    for (int c = 0; c < CIN; c++) {
      wq[0][c][0] = *reinterpret_cast<const f32x4*>(&wl[sl][c][0]);
      wq[0][c][1] = *reinterpret_cast<const f32x4*>(&wl[sl][c][4]);
    }
#pragma unroll
    for (int tap = 0; tap < 9; tap++) {
      if (tap + 1 < 9) {
#pragma unroll
        for (int c = 0; c < CIN; c++) {
          wq[(tap + 1) & 1][c][0] = *reinterpret_cast<const f32x4*>(&wl[sl][(tap + 1) * CIN + c][0]);
          wq[(tap + 1) & 1][c][1] = *reinterpret_cast<const f32x4*>(&wl[sl][(tap + 1) * CIN + c][4]);
        }
      }
#pragma unroll
      for (int c = 0; c < CIN; c++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const float v = xv[c][r + tap / 3][tap % 3];
          const f32x2 vv = f32x2{v, v};
          const f32x4 wa = wq[tap & 1][c][0], wb = wq[tap & 1][c][1];
          acc2[r][0] = __builtin_elementwise_fma(vv, f32x2{wa[0], wa[1]}, acc2[r][0]);
          acc2[r][1] = __builtin_elementwise_fma(vv, f32x2{wa[2], wa[3]}, acc2[r][1]);
          acc2[r][2] = __builtin_elementwise_fma(vv, f32x2{wb[0], wb[1]}, acc2[r][2]);
          acc2[r][3] = __builtin_elementwise_fma(vv, f32x2{wb[2], wb[3]}, acc2[r][3]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // (the sums are complete HERE: without this the compiler sinks each row's multiplications into the branch
    // that stores the row, and keeps every weight alive until the last of them)
    float acc[4][CO_T];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int j = 0; j < CO_T; j++) {
        acc[r][j] = acc2[r][j / 2][j % 2];
        asm volatile("" : "+v"(acc[r][j]));
      }
    const int gx = x0 + lane;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int gy = y0 + 4 * wv + r;
      if (gy < H && gx < W) {
        if constexpr (NSL == 1) {
#pragma unroll
          for (int j = 0; j < CO_T; j++) {
            st[j] += acc[r][j];
            st[CO_T + j] += acc[r][j] * acc[r][j];
          }
        }
        T* zp = z + ((int64_t)(b * H + gy) * W + gx) * Cout + co0;
        if (vec_store) {
          if constexpr (sizeof(T) == 2) {
            vstore<T, CO_T>(zp, acc[r]);
          } else {
            float lo[4] = {acc[r][0], acc[r][1], acc[r][2], acc[r][3]};
            float hi[4] = {acc[r][4], acc[r][5], acc[r][6], acc[r][7]};
            vstore<T, 4>(zp, lo);
            vstore<T, 4>(zp + 4, hi);
          }
        } else {
#pragma unroll
          for (int j = 0; j < CO_T; j++)
            if (co0 + j < Cout) zp[j] = from_f32<T>(acc[r][j]);
        }
      }
    }
    }   // slices
  }
  if (partials && NSL == 1) {
    const int co0 = co_base;
    __syncthreads();
    block_sum<2 * CO_T>(st, red);
    if (tid < 2 * CO_T) {
      float v = 0.f;
#pragma unroll
      for (int i = 0; i < 2 * CO_T; i++) v = (tid == i) ? st[i] : v;
      const int which = tid / CO_T, j = tid % CO_T;
      if (co0 + j < Cout) partials[((int64_t)blockIdx.x * 2 + which) * Cout + co0 + j] = v;
    }
  }
}

// ---------------------------------------------------------------------------------
// direct 3x3 weight gradient: grid (tile walkers, Cout/CO_T, Cin).  Same tiling as the first-layer
// forward: the halo of input channel `ci` is staged in LDS (affine applied, padding = 0), a lane
// owns one column and 4 rows of the 16x64 tile, reads its 6x3 window and its 4 gradient vectors
// (16 bytes each, coalesced) and keeps 9 x CO_T accumulators; each workgroup writes its 72 sums
// to a row of partials [blk][Cout][Cin][9] that a second launch adds in a fixed order (no
// atomics: bitwise reproducible).
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(CD_THREADS) void conv3x3_wgrad_direct_kernel(const T* __restrict__ x,
                                                                          const T* __restrict__ dz,
                                                                          float* __restrict__ partials, int B, int H,
                                                                          int W, int Cin, int Cout, int tiles_x,
                                                                          int tiles_y, int n_tiles,
                                                                          const float* __restrict__ x_aff) {
  __shared__ float red[9 * CO_T * (CD_THREADS / 64)];
  __shared__ float halo[F_TH + 2][F_TW + 2];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int co0 = blockIdx.y * CO_T, ci = blockIdx.z;
  f32x2 acc2[9][CO_T / 2];                 // (two output channels per v_pk_fma_f32: the kernel is bound by the vector ALU)
#pragma unroll
  for (int t = 0; t < 9; t++)
#pragma unroll
    for (int j = 0; j < CO_T / 2; j++) acc2[t][j] = f32x2{0.f, 0.f};
  const bool vec_dz = (Cout % CO_T == 0) && (sizeof(T) * CO_T == 16) && (((uintptr_t)dz & 15) == 0);
  float asc = 1.f, ash = 0.f;
  if (x_aff) {
    asc = x_aff[ci];
    ash = x_aff[Cin + ci];
  }
  // next tile's halo values and gradient vectors are requested (raw, clamped addresses, validity
  // as bit masks) before the current tile's FMAs: one tile of look-ahead hides the memory latency
  constexpr int F_HALO = (F_TH + 2) * (F_TW + 2), F_IT = (F_HALO + CD_THREADS - 1) / CD_THREADS;
  T raw[F_IT];
  T graw[4][CO_T];
  unsigned okm = 0, gokm = 0;
  auto request = [&](int t) {
    const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
    okm = gokm = 0;
#pragma unroll
    for (int i = 0; i < F_IT; i++) {
      const int e = tid + i * CD_THREADS;
      const int hy = e / (F_TW + 2), hx = e % (F_TW + 2);
      const int gy = ty * F_TH - 1 + hy, gx = tx * F_TW - 1 + hx;
      const bool ok = e < F_HALO && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      okm |= ok ? (1u << i) : 0u;
      raw[i] = x[ok ? ((int64_t)(b * H + gy) * W + gx) * Cin + ci : 0];
    }
    const int gx = tx * F_TW + lane;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int gy = ty * F_TH + 4 * wv + r;
      const bool ok = gy < H && gx < W;
      gokm |= ok ? (1u << r) : 0u;
      const T* gp = dz + (ok ? ((int64_t)(b * H + gy) * W + gx) * Cout + co0 : 0);
      if (vec_dz) {
        *reinterpret_cast<uint4*>(&graw[r][0]) = *reinterpret_cast<const uint4*>(gp);
      } else {
#pragma unroll
        for (int j = 0; j < CO_T; j++) graw[r][j] = (ok && co0 + j < Cout) ? gp[j] : from_f32<T>(0.f);
      }
    }
  };
  if ((int)blockIdx.x < n_tiles) request(blockIdx.x);
  for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < F_IT; i++) {
      const int e = tid + i * CD_THREADS;
      if (e < F_HALO) {
        float v = to_f32(raw[i]);
        if (x_aff) v = to_f32(from_f32<T>(fmaxf(v * asc + ash, 0.f)));
        halo[e / (F_TW + 2)][e % (F_TW + 2)] = ((okm >> i) & 1u) ? v : 0.f;
      }
    }
    float g[4][CO_T];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int j = 0; j < CO_T; j++) g[r][j] = ((gokm >> r) & 1u) ? to_f32(graw[r][j]) : 0.f;
    __syncthreads();
    if (t + (int)gridDim.x < n_tiles) request(t + gridDim.x);
    float xv[6][3];
#pragma unroll
    for (int rr = 0; rr < 6; rr++)
#pragma unroll
      for (int kx = 0; kx < 3; kx++) xv[rr][kx] = halo[4 * wv + rr][lane + kx];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int tp = 0; tp < 9; tp++) {
        const float v = xv[r + tp / 3][tp % 3];
#pragma unroll
        for (int j = 0; j < CO_T / 2; j++)
          acc2[tp][j] = __builtin_elementwise_fma(f32x2{v, v}, f32x2{g[r][2 * j], g[r][2 * j + 1]}, acc2[tp][j]);
      }
  }
#pragma unroll
  for (int t = 0; t < 9; t++)
#pragma unroll
    for (int j = 0; j < CO_T; j++) {
      const float s = wave_sum(acc2[t][j / 2][j % 2]);
      if (lane == 0) red[wv * 72 + t * CO_T + j] = s;
    }
  __syncthreads();
  if (tid < 72) {
    float s = 0.f;
    for (int w = 0; w < CD_THREADS / 64; w++) s += red[w * 72 + tid];
    const int t = tid / CO_T, j = tid % CO_T;
    if (co0 + j < Cout)
      partials[(int64_t)blockIdx.x * 9 * Cin * Cout + ((int64_t)(co0 + j) * Cin + ci) * 9 + t] = s;
  }
}

// ---------------------------------------------------------------------------------
// Narrow layers: ONE output channel, or ONE input channel with a wide output.  DoubleConv(base, n_classes) has
// n_classes / 4 middle channels (model/utils/modules.py:13-14) — one for the 4-class output block of A_UNet
// (model/a_unet.py:52) — so its first conv is 32 -> 1 at full resolution, and the data gradient of that conv
// is 1 -> 32.  Thread = (pixel column, 16-byte channel vector cv of the wide side): the CV lanes of a pixel
// read / write one contiguous pixel, the weights of a lane's channels (9 taps x V) stay in registers, and a
// lane walks NR consecutive rows of its column so that an input row it has loaded (and converted, for a lazy
// input) serves three output rows.  HBM-bound on the wide tensor.
// ---------------------------------------------------------------------------------
constexpr int NR = 8;                    // rows per thread and item

template <int CV>
__device__ __forceinline__ float sum_cv(float a) {      // total over the CV adjacent lanes of a pixel, on every one
  if constexpr (CV >= 2) a += dpp_move<0xB1>(a);
  if constexpr (CV >= 4) a += dpp_move<0x4E>(a);
  if constexpr (CV >= 8) a += dpp_move<0x141>(a);
  return a;
}

// z[p] = bias + sum_{tap, ci} act(x)[p + off(tap)][ci] * w[tap][ci]      (Cin = CV*V -> 1 channel)
template <typename T, int V, int CV>
__global__ __launch_bounds__(CD_THREADS, 2) void conv3x3_to1_kernel(const T* __restrict__ x,
                                                                 const T* __restrict__ wp /*[9][1][Cin]*/,
                                                                 const float* __restrict__ bias, T* __restrict__ z,
                                                                 float* __restrict__ partials /*[gridDim.x][2] or null*/,
                                                                 int B, int H, int W,
                                                                 const float* __restrict__ in_aff /*[2][Cin] or null*/) {
  constexpr int Cin = CV * V, PXB = CD_THREADS / CV;
  __shared__ float red[2 * (CD_THREADS / 64)];
  const int tid = threadIdx.x, cv = tid % CV, px = tid / CV, c0 = cv * V;
  float wr[9][V], asc[V], ash[V];
#pragma unroll
  for (int t = 0; t < 9; t++)
#pragma unroll
    for (int k = 0; k < V; k++) wr[t][k] = to_f32(wp[t * Cin + c0 + k]);
#pragma unroll
  for (int k = 0; k < V; k++) {
    asc[k] = in_aff ? in_aff[c0 + k] : 1.f;
    ash[k] = in_aff ? in_aff[Cin + c0 + k] : 0.f;
  }
  const float b0 = bias ? bias[0] : 0.f;
  const int strips = (H + NR - 1) / NR, xblocks = (W + PXB - 1) / PXB, items = B * strips * xblocks;
  float st[2] = {0.f, 0.f};
  // (lazy or plain input decided ONCE: a test per element splits the strip into hundreds of basic blocks)
  auto run = [&](auto lazy_tag) {
  constexpr bool LAZY = decltype(lazy_tag)::value;
  for (int item = blockIdx.x; item < items; item += gridDim.x) {
    const int xb = item % xblocks, sy = (item / xblocks) % strips, b = item / (xblocks * strips);
    const int gx = xb * PXB + px, y0 = sy * NR;
    float acc[NR];
#pragma unroll
    for (int o = 0; o < NR; o++) acc[o] = 0.f;
    // every load is unconditional (coordinates clamped into the image, the value masked afterwards): no branch
    // inside the strip, so its 3*(NR+2) loads are in flight together
    int xc[3];
    unsigned mx[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
      const int xx = gx + d - 1;
      mx[d] = (xx >= 0 && xx < W) ? ~0u : 0u;
      xc[d] = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
    }
    // (one row of loads ahead of the row being multiplied; the scheduling barrier keeps the compiler from
    // hoisting all 3*(NR+2) loads and their conversions, which takes every register and one wave per SIMD)
    float nx[2][3][V];
    auto load_row = [&](int ri, float (&dst)[3][V]) {
      const int yy = y0 - 1 + ri, yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
#pragma unroll
      for (int d = 0; d < 3; d++) vload<T, V>(x + ((int64_t)(b * H + yc) * W + xc[d]) * Cin + c0, dst[d]);
    };
    load_row(0, nx[0]);
#pragma unroll
    for (int ri = 0; ri < NR + 2; ri++) {
      if (ri + 1 < NR + 2) load_row(ri + 1, nx[(ri + 1) & 1]);
      const int yy = y0 - 1 + ri;
      const unsigned my = (yy >= 0 && yy < H) ? ~0u : 0u;
#pragma unroll
      for (int d = 0; d < 3; d++) {
        float (&av)[V] = nx[ri & 1][d];
        const unsigned m = my & mx[d];
#pragma unroll
        for (int k = 0; k < V; k++) {
          if constexpr (LAZY) av[k] = to_f32(from_f32<T>(fmaxf(av[k] * asc[k] + ash[k], 0.f)));
          av[k] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, av[k]) & m);
        }
#pragma unroll
        for (int ky = 0; ky < 3; ky++) {
          const int o = ri - ky;                       // output row that sees this input row through tap row ky
          if (o < 0 || o >= NR) continue;
#pragma unroll
          for (int k = 0; k < V; k++) acc[o] += av[k] * wr[ky * 3 + d][k];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int o = 0; o < NR; o++) {
      const float t = sum_cv<CV>(acc[o]) + b0;
      const int gy = y0 + o;
      if (cv == 0 && gy < H && gx < W) {
        z[(int64_t)(b * H + gy) * W + gx] = from_f32<T>(t);
        st[0] += t;
        st[1] += t * t;
      }
    }
  }
  };
  if (in_aff) run(std::true_type{}); else run(std::false_type{});
  if (partials) {
    block_sum<2>(st, red);
    if (tid < 2) partials[(int64_t)blockIdx.x * 2 + tid] = tid == 0 ? st[0] : st[1];
  }
}

// z[p][co] = bias[co] + sum_tap act(x)[p + off(tap)] * w[tap][co]        (1 channel -> Cout = CV*V)
// The column's 3x3 window slides down the strip: three new scalars per output row.
template <typename T, int V, int CV, bool STATS>
__global__ __launch_bounds__(CD_THREADS, STATS ? 1 : 3) void conv3x3_from1_kernel(const T* __restrict__ x,
                                                                      const T* __restrict__ wp /*[9][Cout][1]*/,
                                                                      const float* __restrict__ bias, T* __restrict__ z,
                                                                      float* __restrict__ partials /*[gridDim.x][2][Cout]*/,
                                                                      int B, int H, int W,
                                                                      const float* __restrict__ in_aff /*[2][1] or null*/) {
  constexpr int Cout = CV * V, PXB = CD_THREADS / CV;
  __shared__ float red[STATS ? CD_THREADS : 1][2 * V];
  const int tid = threadIdx.x, cv = tid % CV, px = tid / CV, c0 = cv * V;
  float wr[9][V], bv[V], st[2][STATS ? V : 1];
#pragma unroll
  for (int t = 0; t < 9; t++)
#pragma unroll
    for (int k = 0; k < V; k++) wr[t][k] = to_f32(wp[t * Cout + c0 + k]);
#pragma unroll
  for (int k = 0; k < V; k++) bv[k] = bias ? bias[c0 + k] : 0.f;
  if constexpr (STATS) {
#pragma unroll
    for (int k = 0; k < V; k++) st[0][k] = st[1][k] = 0.f;
  }
  const float asc = in_aff ? in_aff[0] : 1.f, ash = in_aff ? in_aff[1] : 0.f;
  const bool lazy = in_aff != nullptr;               // (a select per scalar, not a branch: three scalars per output row)
  const int strips = (H + NR - 1) / NR, xblocks = (W + PXB - 1) / PXB, items = B * strips * xblocks;
  for (int item = blockIdx.x; item < items; item += gridDim.x) {
    const int xb = item % xblocks, sy = (item / xblocks) % strips, b = item / (xblocks * strips);
    const int gx = xb * PXB + px, y0 = sy * NR;
    auto row = [&](int yy, float (&r)[3]) {            // window row yy (padding = 0); unconditional loads
      const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
#pragma unroll
      for (int dx = -1; dx <= 1; dx++) {
        const int xx = gx + dx, xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
        const unsigned m = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? ~0u : 0u;
        float v = to_f32(x[(int64_t)(b * H + yc) * W + xc]);
        const float a = to_f32(from_f32<T>(fmaxf(v * asc + ash, 0.f)));
        v = lazy ? a : v;
        r[dx + 1] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & m);
      }
    };
    float xr[3][3];
    row(y0 - 1, xr[0]);
    row(y0, xr[1]);
#pragma unroll
    for (int o = 0; o < NR; o++) {
      const int gy = y0 + o;
      if (gy >= H) continue;                           // (uniform over the workgroup)
      row(gy + 1, xr[(o + 2) % 3]);
      float acc[V];
#pragma unroll
      for (int k = 0; k < V; k++) acc[k] = bv[k];
#pragma unroll
      for (int t = 0; t < 9; t++) {
        const float v = xr[(o + t / 3) % 3][t % 3];
#pragma unroll
        for (int k = 0; k < V; k++) acc[k] += v * wr[t][k];
      }
      if (gx < W) {
        vstore<T, V>(z + ((int64_t)(b * H + gy) * W + gx) * Cout + c0, acc);
        if constexpr (STATS) {
#pragma unroll
          for (int k = 0; k < V; k++) {
            st[0][k] += acc[k];
            st[1][k] += acc[k] * acc[k];
          }
        }
      }
    }
  }
  if constexpr (STATS) {
    // fixed-order sum over the pixel lanes of each channel vector
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
      for (int k = 0; k < V; k++) red[tid][q * V + k] = st[q][k];
    __syncthreads();
    for (int j = tid; j < 2 * Cout; j += CD_THREADS) {
      const int q = j / Cout, c = j % Cout, cvj = c / V, k = c % V;
      float a = 0.f;
      for (int t = cvj; t < CD_THREADS; t += CV) a += red[t][q * V + k];
      partials[((int64_t)blockIdx.x * 2 + q) * Cout + c] = a;
    }
  }
}

// dW[0][ci][tap] = sum_p dz[p] * act(x)[p + off(tap)][ci]               (Cin = CV*V, one output channel)
// A lane walks the INPUT pixels of its column: each is loaded (and converted) once and multiplied by the 9
// gradients around it; partials [gridDim.x][Cin][9] (the layout of the generic direct kernel for Cout = 1).
template <typename T, int V, int CV>
__global__ __launch_bounds__(CD_THREADS) void conv3x3_wgrad_to1_kernel(const T* __restrict__ x, const T* __restrict__ dz,
                                                                       float* __restrict__ partials, int B, int H,
                                                                       int W, const float* __restrict__ x_aff) {
  constexpr int Cin = CV * V, PXB = CD_THREADS / CV, NW = CD_THREADS / 64;
  __shared__ float red[NW][CV][9 * V];
  const int tid = threadIdx.x, cv = tid % CV, px = tid / CV, c0 = cv * V;
  float asc[V], ash[V], acc[9][V];
#pragma unroll
  for (int k = 0; k < V; k++) {
    asc[k] = x_aff ? x_aff[c0 + k] : 1.f;
    ash[k] = x_aff ? x_aff[Cin + c0 + k] : 0.f;
  }
#pragma unroll
  for (int t = 0; t < 9; t++)
#pragma unroll
    for (int k = 0; k < V; k++) acc[t][k] = 0.f;
  const int strips = (H + NR - 1) / NR, xblocks = (W + PXB - 1) / PXB, items = B * strips * xblocks;
  auto run = [&](auto lazy_tag) {
  constexpr bool LAZY = decltype(lazy_tag)::value;
  for (int item = blockIdx.x; item < items; item += gridDim.x) {
    const int xb = item % xblocks, sy = (item / xblocks) % strips, b = item / (xblocks * strips);
    const int gx = xb * PXB + px, y0 = sy * NR;
    // unconditional loads (clamped coordinates, masked values): the strip is one basic block
    float g[NR + 2][3];                                // gradients around the column (0 outside the image)
#pragma unroll
    for (int ri = 0; ri < NR + 2; ri++) {
      const int yy = y0 - 1 + ri, yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
#pragma unroll
      for (int dx = -1; dx <= 1; dx++) {
        const int xx = gx + dx, xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
        const unsigned m = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? ~0u : 0u;
        g[ri][dx + 1] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, to_f32(dz[(int64_t)(b * H + yc) * W + xc])) & m);
      }
    }
    const int gxc = gx < W ? gx : W - 1;
#pragma unroll
    for (int i = 0; i < NR; i++) {
      const int qy = y0 + i, qc = qy < H ? qy : H - 1;
      const unsigned m = (qy < H && gx < W) ? ~0u : 0u;
      float av[V];
      vload<T, V>(x + ((int64_t)(b * H + qc) * W + gxc) * Cin + c0, av);
#pragma unroll
      for (int k = 0; k < V; k++) {
        if constexpr (LAZY) av[k] = to_f32(from_f32<T>(fmaxf(av[k] * asc[k] + ash[k], 0.f)));
        av[k] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, av[k]) & m);
      }
      // tap (ky, kx) pairs this input pixel with the output pixel (qy - ky + 1, gx - kx + 1)
#pragma unroll
      for (int t = 0; t < 9; t++) {
        const float gv = g[i + 2 - t / 3][2 - t % 3];
#pragma unroll
        for (int k = 0; k < V; k++) acc[t][k] += gv * av[k];
      }
    }
  }
  };
  if (x_aff) run(std::true_type{}); else run(std::false_type{});
  // lanes with the same channel vector: xor butterflies over the pixel bits of the lane index, then the waves
  const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
  for (int t = 0; t < 9; t++)
#pragma unroll
    for (int k = 0; k < V; k++) {
      float v = acc[t][k];
#pragma unroll
      for (int m = CV; m < 64; m <<= 1) v += __shfl_xor(v, m);
      if (lane < CV) red[wv][lane][t * V + k] = v;
    }
  __syncthreads();
  for (int j = tid; j < CV * 9 * V; j += CD_THREADS) {
    const int cvj = j / (9 * V), e = j % (9 * V), t = e / V, k = e % V;
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w++) a += red[w][cvj][e];
    partials[(int64_t)blockIdx.x * 9 * Cin + (cvj * V + k) * 9 + t] = a;
  }
}

// ---------------------------------------------------------------------------------
// 1x1 output convolution
// ---------------------------------------------------------------------------------
template <typename T, int V, int CO>
__global__ __launch_bounds__(CD_THREADS) void conv1x1_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ bias, T* __restrict__ y,
                                                                 int64_t n_pix, int Cin,
                                                                 const float* __restrict__ x_aff) {
  extern __shared__ __attribute__((aligned(16))) float wl[];   // [Cin][CO]  (transposed: ci major), then [2][Cin]
  for (int i = threadIdx.x; i < Cin * CO; i += CD_THREADS) wl[i] = w[(i % CO) * Cin + i / CO];
  const float* aff = nullptr;
  if (x_aff) {
    for (int i = threadIdx.x; i < 2 * Cin; i += CD_THREADS) wl[Cin * CO + i] = x_aff[i];
    aff = wl + Cin * CO;
  }
  __syncthreads();
  float bv[CO];
#pragma unroll
  for (int j = 0; j < CO; j++) bv[j] = bias ? bias[j] : 0.f;
  for (int64_t p = (int64_t)blockIdx.x * CD_THREADS + threadIdx.x; p < n_pix; p += (int64_t)gridDim.x * CD_THREADS) {
    float acc[CO];
#pragma unroll
    for (int j = 0; j < CO; j++) acc[j] = bv[j];
    for (int c = 0; c < Cin; c += V) {
      float xv[V];
      vload<T, V>(x + p * Cin + c, xv);
      affine_relu<V>(xv, aff, Cin, c);
#pragma unroll
      for (int k = 0; k < V; k++)
#pragma unroll
        for (int j = 0; j < CO; j++) acc[j] += xv[k] * wl[(c + k) * CO + j];
    }
    vstore<T, CO>(y + p * CO, acc);
  }
}

// The same conv with thread = (pixel, channel vector): consecutive lanes read the consecutive
// 16-byte vectors of a pixel (a wave reads 1 KiB contiguous instead of 64 strided vectors), each
// lane forms the partial dot products of its channels against register-resident weights, and the
// CV = Cin/V lanes of a pixel are summed on the DPP path.  Lane 0 of the pixel stores.
template <typename T, int V, int CO, int CV>
__global__ __launch_bounds__(CD_THREADS) void conv1x1_fwd_cv_kernel(const T* __restrict__ x,
                                                                    const float* __restrict__ w,
                                                                    const float* __restrict__ bias,
                                                                    T* __restrict__ y, int64_t n_pix,
                                                                    const float* __restrict__ x_aff) {
  constexpr int Cin = CV * V, PR = CD_THREADS / CV;
  const int tid = threadIdx.x, cv = tid % CV, pr = tid / CV, c0 = cv * V;
  float wr[CO][V], asc[V], ash[V], bv[CO];
#pragma unroll
  for (int j = 0; j < CO; j++) {
    bv[j] = bias ? bias[j] : 0.f;
#pragma unroll
    for (int k = 0; k < V; k++) wr[j][k] = w[j * Cin + c0 + k];
  }
#pragma unroll
  for (int k = 0; k < V; k++) {
    asc[k] = x_aff ? x_aff[c0 + k] : 1.f;
    ash[k] = x_aff ? x_aff[Cin + c0 + k] : 0.f;
  }
  for (int64_t p = (int64_t)blockIdx.x * PR + pr; p < n_pix; p += (int64_t)gridDim.x * PR) {
    float xv[V], acc[CO];
    vload<T, V>(x + p * Cin + c0, xv);
    if (x_aff) {
#pragma unroll
      for (int k = 0; k < V; k++) xv[k] = fmaxf(xv[k] * asc[k] + ash[k], 0.f);
    }
#pragma unroll
    for (int j = 0; j < CO; j++) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < V; k++) a += xv[k] * wr[j][k];
      if constexpr (CV >= 2) a += dpp_move<0xB1>(a);
      if constexpr (CV >= 4) a += dpp_move<0x4E>(a);
      if constexpr (CV >= 8) a += dpp_move<0x141>(a);
      if constexpr (CV >= 16) a += dpp_move<0x140>(a);
      acc[j] = a + bv[j];
    }
    if (cv == 0) vstore<T, CO>(y + p * CO, acc);
  }
}

// Backward of the 1x1 conv.  Thread = (pixel, channel vector): consecutive lanes take the
// consecutive channel vectors of a pixel, then the next pixel, so x / dx accesses of a wave are
// contiguous.  Each thread writes its dx vector and accumulates dW[:, its channels] (and dbias
// on the cv==0 lanes); the block reduces in a fixed order through LDS and writes one partial
// row [CO*Cin + CO].  Requires 256 % (Cin/V) == 0 (a thread keeps its channel vector).
template <typename T, int V, int CO>
__global__ __launch_bounds__(CD_THREADS) void conv1x1_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                 const float* __restrict__ w, T* __restrict__ dx,
                                                                 float* __restrict__ partials, int64_t n_pix, int Cin,
                                                                 int64_t pix_per_blk,
                                                                 const float* __restrict__ x_aff, int bn) {
  // bn != 0: x is the lazy output of conv+BatchNorm(train)+ReLU and x_aff is [4][Cin] (scale,
  // shift, mean, rstd): the sums that BatchNorm's backward needs over (z, dx) — sum(dy) and
  // sum(dy*xhat), dy = dx where the activation is positive — are accumulated here, where both
  // are in registers, and appended to the partial row (2*Cin values).
  extern __shared__ float red[];                 // [256][CO*V + CO], reused as [256][2V] for the BatchNorm sums
  constexpr int NR = CO * V + CO;
  const int tid = threadIdx.x;
  const int CV = Cin / V, PR = CD_THREADS / CV;
  const int cv = tid % CV, pr = tid / CV;
  const int c0 = cv * V;
  float wr[CO][V];
#pragma unroll
  for (int j = 0; j < CO; j++)
#pragma unroll
    for (int k = 0; k < V; k++) wr[j][k] = w[j * Cin + c0 + k];
  float asc[V], ash[V];                          // lazy-activation affine of this thread's channels
#pragma unroll
  for (int k = 0; k < V; k++) {
    asc[k] = x_aff ? x_aff[c0 + k] : 1.f;
    ash[k] = x_aff ? x_aff[Cin + c0 + k] : 0.f;
  }
  float bmean[V], brstd[V], bs[2][V];
#pragma unroll
  for (int k = 0; k < V; k++) {
    bmean[k] = bn ? x_aff[2 * Cin + c0 + k] : 0.f;
    brstd[k] = bn ? x_aff[3 * Cin + c0 + k] : 0.f;
    bs[0][k] = bs[1][k] = 0.f;
  }
  float aw[CO][V], ab[CO];
#pragma unroll
  for (int j = 0; j < CO; j++) {
    ab[j] = 0.f;
#pragma unroll
    for (int k = 0; k < V; k++) aw[j][k] = 0.f;
  }
  const int64_t p0 = (int64_t)blockIdx.x * pix_per_blk;
  int64_t p1 = p0 + pix_per_blk;
  if (p1 > n_pix) p1 = n_pix;
  // Lazy / BatchNorm sums are decided once (not per pixel), and UNR pixels are loaded before the first is
  // multiplied: the loop is otherwise one pixel's two loads deep, with ~150 vector instructions behind them.
  constexpr int UNR = 4;
  auto run = [&](auto lazy_tag, auto bn_tag) {
    constexpr bool LAZY = decltype(lazy_tag)::value, BN = decltype(bn_tag)::value;
    for (int64_t pb = p0 + pr; pr < PR && pb < p1; pb += (int64_t)UNR * PR) {
      float g[UNR][CO], xq[UNR][V];
#pragma unroll
      for (int u = 0; u < UNR; u++) {
        const int64_t p = pb + (int64_t)u * PR, pc = p < p1 ? p : p1 - 1;      // (clamped address; the tail is skipped below)
        vload<T, CO>(dy + pc * CO, g[u]);
        vload<T, V>(x + pc * Cin + c0, xq[u]);
      }
#pragma unroll
      for (int u = 0; u < UNR; u++) {
        const int64_t p = pb + (int64_t)u * PR;
        if (p >= p1) break;
        float xv[V], d[V], zr[V];
#pragma unroll
        for (int k = 0; k < V; k++) {
          zr[k] = xq[u][k];
          xv[k] = LAZY ? fmaxf(zr[k] * asc[k] + ash[k], 0.f) : zr[k];
          d[k] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < CO; j++) {
          ab[j] += g[u][j];
#pragma unroll
          for (int k = 0; k < V; k++) {
            d[k] += g[u][j] * wr[j][k];
            aw[j][k] += g[u][j] * xv[k];
          }
        }
        if (dx) vstore<T, V>(dx + p * Cin + c0, d);
        if constexpr (BN) {
#pragma unroll
          for (int k = 0; k < V; k++) {
            const float dd = (xv[k] > 0.f) ? to_f32(from_f32<T>(d[k])) : 0.f;     // the value the next pass would read
            bs[0][k] += dd;
            bs[1][k] += dd * ((zr[k] - bmean[k]) * brstd[k]);
          }
        }
      }
    }
  };
  if (bn) run(std::true_type{}, std::true_type{});
  else if (x_aff) run(std::true_type{}, std::false_type{});
  else run(std::false_type{}, std::false_type{});
#pragma unroll
  for (int j = 0; j < CO; j++) {
#pragma unroll
    for (int k = 0; k < V; k++) red[tid * NR + j * V + k] = aw[j][k];
    red[tid * NR + CO * V + j] = ab[j];
  }
  __syncthreads();
  const int row_n = CO * Cin + CO + (bn ? 2 * Cin : 0);
  float* prow = partials + (int64_t)blockIdx.x * row_n;
  for (int o = tid; o < CO * Cin + CO; o += CD_THREADS) {
    float acc = 0.f;
    if (o < CO * Cin) {
      const int j = o / Cin, c = o % Cin, ocv = c / V, k = c % V;
      for (int r = 0; r < PR; r++) acc += red[(r * CV + ocv) * NR + j * V + k];
    } else {
      const int j = o - CO * Cin;
      for (int r = 0; r < PR; r++) acc += red[(r * CV) * NR + CO * V + j];     // cv == 0 lanes
    }
    prow[o] = acc;
  }
  if (bn) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
      for (int k = 0; k < V; k++) red[(tid * 2 + q) * V + k] = bs[q][k];
    __syncthreads();
    for (int e = tid; e < 2 * Cin; e += CD_THREADS) {
      const int q = e / Cin, c = e % Cin, ocv = c / V, k = c % V;
      float acc = 0.f;
      for (int r = 0; r < PR; r++) acc += red[((r * CV + ocv) * 2 + q) * V + k];
      prow[CO * Cin + CO + e] = acc;
    }
  }
}

template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int B, int C, int64_t HW) {
  const int64_t n = (int64_t)B * HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / HW, p = i % HW;
    for (int c = 0; c < C; c++) y[i * C + c] = from_f32<T>(x[(b * C + c) * HW + p]);
  }
}

constexpr int C1_MAX_BLOCKS = 2048;

template <typename T, int CO>
int conv1x1_fwd_t(const void* x, const float* x_aff, int64_t n_pix, int Cin, const float* w, const float* bias,
                  void* y, hipStream_t st) {
  constexpr int N = VecOf<T>::N;
  const size_t lds = ((size_t)Cin * CO + 2 * (size_t)Cin) * sizeof(float);
  const int grid = ew_grid(n_pix, CD_THREADS);
  const bool vec = pick_vec<T>(Cin, x) == N;
  const int cvn = vec ? Cin / N : 0;
  if (cvn == 2 || cvn == 4 || cvn == 8 || cvn == 16) {
    const int g = ew_grid(n_pix * cvn, CD_THREADS);
#define C1_CV(CVN) conv1x1_fwd_cv_kernel<T, N, CO, CVN><<<g, CD_THREADS, 0, st>>>((const T*)x, w, bias, (T*)y, n_pix, x_aff)
    if (cvn == 2) C1_CV(2); else if (cvn == 4) C1_CV(4); else if (cvn == 8) C1_CV(8); else C1_CV(16);
#undef C1_CV
  } else if (vec)
    conv1x1_fwd_kernel<T, N, CO><<<grid, CD_THREADS, lds, st>>>((const T*)x, w, bias, (T*)y, n_pix, Cin, x_aff);
  else
    conv1x1_fwd_kernel<T, 1, CO><<<grid, CD_THREADS, lds, st>>>((const T*)x, w, bias, (T*)y, n_pix, Cin, x_aff);
  return 0;
}

template <typename T, int CO>
int conv1x1_bwd_t(const void* x, const float* x_aff, const void* dy, int64_t n_pix, int Cin, const float* w, void* dx,
                  float* dw, float* dbias, float* bn_sums, void* ws, hipStream_t st) {
  constexpr int N = VecOf<T>::N;
  const bool vec = pick_vec<T>(Cin, x, dx) == N && (CD_THREADS % (Cin / N)) == 0;
  const int V = vec ? N : 1;
  const int PR = CD_THREADS / (Cin / V);
  int nblk = (int)((n_pix + (int64_t)PR * 16 - 1) / ((int64_t)PR * 16));
  // with dx written the pass is a streaming one and many short workgroups measured best (2048); without it
  // (reductions only) three per CU: 768 -> 105 us, 2048 -> 120 us, 1024 -> 128 us at 16 x 512^2, 32 -> 4 channels
  const int cap = dx ? C1_MAX_BLOCKS : 768;
  if (nblk > cap) nblk = cap;
  if (nblk < 1) nblk = 1;
  const int64_t ppb = (n_pix + nblk - 1) / nblk;
  float* partials = (float*)ws;
  const size_t lds = (size_t)CD_THREADS * (CO * V + CO > 2 * V ? CO * V + CO : 2 * V) * sizeof(float);
  const int bn = bn_sums ? 1 : 0;
  if (vec) {
    auto k = conv1x1_bwd_kernel<T, N, CO>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k<<<nblk, CD_THREADS, lds, st>>>((const T*)x, (const T*)dy, w, (T*)dx, partials, n_pix, Cin, ppb, x_aff, bn);
  } else {
    conv1x1_bwd_kernel<T, 1, CO><<<nblk, CD_THREADS, lds, st>>>((const T*)x, (const T*)dy, w, (T*)dx, partials, n_pix,
                                                                 Cin, ppb, x_aff, bn);
  }
  // one launch sums the partial rows into dw, dbias and (optionally) the BatchNorm sums
  launch_reduce_partials3(partials, nblk, CO * Cin + CO + (bn ? 2 * Cin : 0), CO * Cin, CO, dw, dbias, bn_sums, st);
  return 0;
}

#define DISPATCH_CO(FN, T, CO, ...)               \
  switch (CO) {                                   \
    case 1: FN<T, 1>(__VA_ARGS__); break;         \
    case 2: FN<T, 2>(__VA_ARGS__); break;         \
    case 3: FN<T, 3>(__VA_ARGS__); break;         \
    case 4: FN<T, 4>(__VA_ARGS__); break;         \
    case 5: FN<T, 5>(__VA_ARGS__); break;         \
    case 6: FN<T, 6>(__VA_ARGS__); break;         \
    case 7: FN<T, 7>(__VA_ARGS__); break;         \
    case 8: FN<T, 8>(__VA_ARGS__); break;         \
    case 9: FN<T, 9>(__VA_ARGS__); break;         \
    case 10: FN<T, 10>(__VA_ARGS__); break;       \
    case 11: FN<T, 11>(__VA_ARGS__); break;       \
    case 12: FN<T, 12>(__VA_ARGS__); break;       \
    case 13: FN<T, 13>(__VA_ARGS__); break;       \
    case 14: FN<T, 14>(__VA_ARGS__); break;       \
    case 15: FN<T, 15>(__VA_ARGS__); break;       \
    case 16: FN<T, 16>(__VA_ARGS__); break;       \
  }

}  // namespace

// Entry points shared with the MFMA file (declared in conv_internal.h).
#include "conv_internal.h"

constexpr int CD_MAX_BLOCKS = 2048;      // 8 workgroups per CU: the direct kernels hide latency with waves

size_t conv3x3_direct_workspace_bytes(int64_t n_pix, int Cout) {
  return align256((size_t)CD_MAX_BLOCKS * 2 * Cout * sizeof(float));
}

// Workgroups of the image-facing kernel: two per CU, each with the same whole number of tiles where the
// tile count allows it (measured at 16 x 512^2: 512 workgroups of 8 tiles 30 us; 1024 x 4: 36; 2048 x 2: 45;
// 640 uneven: 39) — a longer tile loop amortises the weight staging and keeps the next tile's halo in flight.
static int first_layer_grid(int B, int H, int W, int Cin) {
  const int n_tiles = B * cdiv(W, F_TW) * cdiv(H, F_TH);
  // workgroups resident at once: four per CU with one input channel (112 registers), two with three
  // (1 -> 8 at 16 x 512^2: 512 workgroups 27.7 us, 1024: 26.0, 2048: 30.7)
  const int per = cdiv(n_tiles, Cin == 1 ? 1024 : 512);
  return cdiv(n_tiles, per);
}

// Workgroups (= rows of statistics partials) of the forward kernels; every kernel walks its work with a grid
// stride, so the count depends on the shape alone (not on which kernel the pointers' alignment allows).
int conv3x3_direct_stats_rows(int64_t n_pix, int B, int H, int W, int Cin, int Cout) {
  // one input channel into whole channel vectors: the strip kernel wants many short workgroups
  if ((Cin == 1 || Cin == 3) && !(Cin == 1 && Cout >= 16 && Cout % 4 == 0)) return first_layer_grid(B, H, W, Cin);
  int gx = cdiv(n_pix, CD_THREADS);
  return gx > CD_MAX_BLOCKS ? CD_MAX_BLOCKS : gx;
}

int conv3x3_direct_fwd(const void* x, const float* in_aff, const void* wp, const float* bias, int dtype, int B, int H,
                       int W, int Cin, int Cout, void* z, float* stats, void* workspace, hipStream_t st) {
  const int64_t n_pix = (int64_t)B * H * W;
  int gx = cdiv(n_pix, CD_THREADS);
  if (gx > CD_MAX_BLOCKS) gx = CD_MAX_BLOCKS;
  dim3 grid(gx, cdiv(Cout, CO_T));
  float* partials = stats ? (float*)workspace : nullptr;
  const int NV = dtype == MIS_F32 ? 4 : 8;             // channels per 16-byte vector
  const bool small32 = n_pix < ((int64_t)1 << 31) / 64;      // (the narrow kernels index pixels in 32 bits)
  // narrow layers: wide side in whole vectors, 1 / 2 / 4 / 8 of them (or one 8-byte bf16 vector of 4 channels)
  auto vecs = [&](int C, const void* p) {
    if (((uintptr_t)p % 16) != 0 || C % NV != 0) return 0;
    const int cv = C / NV;
    return (cv == 1 || cv == 2 || cv == 4 || cv == 8) ? cv : 0;
  };
#define NARROW_CV(KERNEL, TT, VV, ...)                                                      \
  switch (cv) {                                                                             \
    case 1: KERNEL<TT, VV, 1><<<grid.x, CD_THREADS, 0, st>>>(__VA_ARGS__); break;           \
    case 2: KERNEL<TT, VV, 2><<<grid.x, CD_THREADS, 0, st>>>(__VA_ARGS__); break;           \
    case 4: KERNEL<TT, VV, 4><<<grid.x, CD_THREADS, 0, st>>>(__VA_ARGS__); break;           \
    default: KERNEL<TT, VV, 8><<<grid.x, CD_THREADS, 0, st>>>(__VA_ARGS__); break;          \
  }
  if (Cin == 1 && Cout >= 16 && small32 && vecs(Cout, z)) {      // (8 channels: the LDS-tiled kernel below is faster, 32 vs 46 us)
    const int cv = vecs(Cout, z);
    grid.x = conv3x3_direct_stats_rows(n_pix, B, H, W, Cin, Cout);
#define FROM1(TT, VV, CVV)                                                                                              \
  if (partials)                                                                                                         \
    conv3x3_from1_kernel<TT, VV, CVV, true><<<grid.x, CD_THREADS, 0, st>>>((const TT*)x, (const TT*)wp, bias, (TT*)z,   \
                                                                             partials, B, H, W, in_aff);               \
  else                                                                                                                  \
    conv3x3_from1_kernel<TT, VV, CVV, false><<<grid.x, CD_THREADS, 0, st>>>((const TT*)x, (const TT*)wp, bias, (TT*)z,  \
                                                                              partials, B, H, W, in_aff);
#define FROM1_CV(TT, VV)                                                                                                \
  switch (cv) {                                                                                                         \
    case 1: FROM1(TT, VV, 1) break;                                                                                     \
    case 2: FROM1(TT, VV, 2) break;                                                                                     \
    case 4: FROM1(TT, VV, 4) break;                                                                                     \
    default: FROM1(TT, VV, 8) break;                                                                                    \
  }
    if (dtype == MIS_F32) FROM1_CV(float, 4) else FROM1_CV(bf16, 8)
#undef FROM1_CV
#undef FROM1
  } else if (Cout == 1 && small32 && (vecs(Cin, x) || (dtype == MIS_BF16 && Cin == 4 && ((uintptr_t)x % 8) == 0))) {
    const int cv = vecs(Cin, x);
    if (dtype == MIS_F32)
      NARROW_CV(conv3x3_to1_kernel, float, 4, (const float*)x, (const float*)wp, bias, (float*)z, partials, B, H, W, in_aff)
    else if (cv)
      NARROW_CV(conv3x3_to1_kernel, bf16, 8, (const bf16*)x, (const bf16*)wp, bias, (bf16*)z, partials, B, H, W, in_aff)
    else
      conv3x3_to1_kernel<bf16, 4, 1><<<grid.x, CD_THREADS, 0, st>>>((const bf16*)x, (const bf16*)wp, bias, (bf16*)z,
                                                                     partials, B, H, W, in_aff);
  } else if (Cin == 1 || Cin == 3) {
    const int ftx = cdiv(W, F_TW), fty = cdiv(H, F_TH);
    grid.x = conv3x3_direct_stats_rows(n_pix, B, H, W, Cin, Cout);
#define FIRST_LAUNCH(TT, CIN)                                                                               \
  conv3x3_first_kernel<TT, CIN><<<grid, CD_THREADS, 0, st>>>((const TT*)x, (const TT*)wp, bias, (TT*)z, partials, B, \
                                                             H, W, Cout, in_aff, ftx, fty, B * ftx * fty)
#define FIRST_ALL(TT, CIN, NSL_)                                                                            \
  conv3x3_first_kernel<TT, CIN, NSL_><<<grid.x, CD_THREADS, 0, st>>>((const TT*)x, (const TT*)wp, bias, (TT*)z,    \
                                                                     nullptr, B, H, W, Cout, in_aff, ftx, fty,      \
                                                                     B * ftx * fty)
    // no statistics and 2..4 channel slices (RDAM_UNet.encoder1, a bare 3 -> 32 conv, model/rdam_unet.py:21):
    // all slices from one workgroup
    const int nsl = cdiv(Cout, CO_T);
    if (!partials && Cin == 3 && nsl >= 2 && nsl <= 4) {
      if (dtype == MIS_F32) {
        if (nsl == 2) FIRST_ALL(float, 3, 2); else FIRST_ALL(float, 3, 4);
      } else {
        if (nsl == 2) FIRST_ALL(bf16, 3, 2); else FIRST_ALL(bf16, 3, 4);
      }
    } else if (dtype == MIS_F32) {
      if (Cin == 1) FIRST_LAUNCH(float, 1); else FIRST_LAUNCH(float, 3);
    } else {
      if (Cin == 1) FIRST_LAUNCH(bf16, 1); else FIRST_LAUNCH(bf16, 3);
    }
#undef FIRST_ALL
#undef FIRST_LAUNCH
  } else {
    // narrow outputs take fewer channels per thread
    const int cot = Cout <= 1 ? 1 : Cout <= 2 ? 2 : Cout <= 4 ? 4 : CO_T;
    grid.y = cdiv(Cout, cot);
#define DIRECT_LAUNCH(TT, VV, COT_)                                                                          \
  conv3x3_direct_kernel<TT, VV, COT_><<<grid, CD_THREADS, 0, st>>>((const TT*)x, (const TT*)wp, bias, (TT*)z, \
                                                                  partials, B, H, W, Cin, Cout, in_aff)
#define DIRECT_COT(TT, VV)                                                                                   \
  switch (cot) {                                                                                             \
    case 1: DIRECT_LAUNCH(TT, VV, 1); break;                                                                 \
    case 2: DIRECT_LAUNCH(TT, VV, 2); break;                                                                 \
    case 4: DIRECT_LAUNCH(TT, VV, 4); break;                                                                 \
    default: DIRECT_LAUNCH(TT, VV, CO_T); break;                                                             \
  }
    if (dtype == MIS_F32) {
      if (Cin % 4 == 0 && ((uintptr_t)x % 16) == 0) DIRECT_COT(float, 4) else DIRECT_COT(float, 1)
    } else {
      if (Cin % 8 == 0 && ((uintptr_t)x % 16) == 0) DIRECT_COT(bf16, 8) else DIRECT_COT(bf16, 1)
    }
#undef DIRECT_COT
#undef DIRECT_LAUNCH
  }
#undef NARROW_CV
  MIS_CHECK_LAUNCH("conv3x3_direct_kernel");
  if (stats && (void*)stats != workspace) {      // stats == workspace: the caller reduces the partial rows itself
    launch_reduce_partials(partials, grid.x, 2 * Cout, stats, st);
    MIS_CHECK_LAUNCH("reduce_partials_kernel");
  }
  return MIS_OK;
}

size_t conv3x3_direct_packed_bytes(int dtype, int Cin, int Cout) {
  return (size_t)9 * Cin * Cout * (dtype == MIS_BF16 ? 2 : 4);
}

int conv3x3_direct_pack(const float* w, int Cw_out, int Cw_in, int transposed, int dtype, void* out, hipStream_t st) {
  const int64_t n = (int64_t)Cw_out * Cw_in * 9;
  const int grid = ew_grid(n, 256);
  if (dtype == MIS_F32)
    pack_w3x3_kernel<float><<<grid, 256, 0, st>>>(w, Cw_out, Cw_in, transposed ? nullptr : (float*)out,
                                                  transposed ? (float*)out : nullptr);
  else
    pack_w3x3_kernel<bf16><<<grid, 256, 0, st>>>(w, Cw_out, Cw_in, transposed ? nullptr : (bf16*)out,
                                                 transposed ? (bf16*)out : nullptr);
  MIS_CHECK_LAUNCH("pack_w3x3_kernel");
  return MIS_OK;
}

namespace {
// pixel blocks of the direct weight gradient: enough workgroups to fill the chip across the
// (Cout/CO_T, Cin) grid dimensions, at most 1024 rows of partials
int wgrad_direct_blocks(int64_t n_pix, int Cin, int Cout) {
  int nblk = (int)((n_pix + F_TW * F_TH - 1) / (F_TW * F_TH));      // never more walkers than whole tiles
  const int per = cdiv(Cout, CO_T) * Cin;
  // two workgroups per CU over all (co-slice, ci) planes, each walking the same number of tiles where possible
  // (measured for 1 -> 8 channels at 16 x 512^2: 512 walkers 37 us, 1024: 45, 384: 43, 256: 57)
  const int cap = per >= 512 ? 1 : 512 / per;
  if (nblk > cap) nblk = cdiv(nblk, cdiv(nblk, cap));
  return nblk < 1 ? 1 : nblk;
}
}  // namespace

constexpr int TO1_WGRAD_BLOCKS = 512;    // workgroups (= partial rows) of the one-output-channel weight gradient

size_t conv3x3_direct_wgrad_workspace_bytes(int64_t n_pix, int Cin, int Cout) {
  const int rows = Cout == 1 && wgrad_direct_blocks(n_pix, Cin, Cout) < TO1_WGRAD_BLOCKS ? TO1_WGRAD_BLOCKS
                                                                                         : wgrad_direct_blocks(n_pix, Cin, Cout);
  return align256((size_t)rows * 9 * Cin * Cout * sizeof(float));
}

int conv3x3_direct_wgrad(const void* x, const float* x_aff, const void* dz, int dtype, int B, int H, int W, int Cin,
                         int Cout, float* dw, void* workspace, hipStream_t st) {
  const int64_t n_pix = (int64_t)B * H * W;
  if (n_pix >= ((int64_t)1 << 31) || !workspace) {
    mis_set_error("conv3x3_wgrad: needs a workspace and B*H*W < 2^31");
    return MIS_ERR_ARG;
  }
  float* partials = (float*)workspace;
  const int NV = dtype == MIS_F32 ? 4 : 8;
  const int cv = (Cin % NV == 0 && ((uintptr_t)x % 16) == 0) ? Cin / NV : 0;
  if (Cout == 1 && n_pix < ((int64_t)1 << 31) / 64 && (cv == 1 || cv == 2 || cv == 4 || cv == 8)) {
    const int items = B * cdiv(H, NR) * cdiv(W, CD_THREADS / cv);
    const int nb = items < TO1_WGRAD_BLOCKS ? items : TO1_WGRAD_BLOCKS;
#define TO1_WGRAD(TT, VV)                                                                                          \
  switch (cv) {                                                                                                    \
    case 1: conv3x3_wgrad_to1_kernel<TT, VV, 1><<<nb, CD_THREADS, 0, st>>>((const TT*)x, (const TT*)dz, partials, B, H, W, x_aff); break; \
    case 2: conv3x3_wgrad_to1_kernel<TT, VV, 2><<<nb, CD_THREADS, 0, st>>>((const TT*)x, (const TT*)dz, partials, B, H, W, x_aff); break; \
    case 4: conv3x3_wgrad_to1_kernel<TT, VV, 4><<<nb, CD_THREADS, 0, st>>>((const TT*)x, (const TT*)dz, partials, B, H, W, x_aff); break; \
    default: conv3x3_wgrad_to1_kernel<TT, VV, 8><<<nb, CD_THREADS, 0, st>>>((const TT*)x, (const TT*)dz, partials, B, H, W, x_aff); break; \
  }
    if (dtype == MIS_F32) TO1_WGRAD(float, 4) else TO1_WGRAD(bf16, 8)
#undef TO1_WGRAD
    MIS_CHECK_LAUNCH("conv3x3_wgrad_to1_kernel");
    launch_reduce_partials(partials, nb, 9 * Cin, dw, st);
    MIS_CHECK_LAUNCH("reduce_partials_kernel");
    return MIS_OK;
  }
  const int nblk = wgrad_direct_blocks(n_pix, Cin, Cout);
  const int ftx = cdiv(W, F_TW), fty = cdiv(H, F_TH), n_tiles = B * ftx * fty;
  dim3 grid(nblk, cdiv(Cout, CO_T), Cin);
  if (dtype == MIS_F32)
    conv3x3_wgrad_direct_kernel<float><<<grid, CD_THREADS, 0, st>>>((const float*)x, (const float*)dz, partials, B, H,
                                                                     W, Cin, Cout, ftx, fty, n_tiles, x_aff);
  else
    conv3x3_wgrad_direct_kernel<bf16><<<grid, CD_THREADS, 0, st>>>((const bf16*)x, (const bf16*)dz, partials, B, H, W,
                                                                    Cin, Cout, ftx, fty, n_tiles, x_aff);
  MIS_CHECK_LAUNCH("conv3x3_wgrad_direct_kernel");
  launch_reduce_partials(partials, nblk, 9 * Cin * Cout, dw, st);
  MIS_CHECK_LAUNCH("reduce_partials_kernel");
  return MIS_OK;
}

extern "C" {

int mis_conv1x1_fwd(const void* x, const float* x_affine, int dtype, int64_t n_pix, int Cin, int Cout, const float* w,
                    const float* bias, void* y, mis_stream_t stream) {
  MIS_REQUIRE(x && w && y && n_pix >= 1 && Cin >= 1, "conv1x1_fwd: bad arguments");
  MIS_REQUIRE(Cout >= 1 && Cout <= 16, "conv1x1_fwd: 1 <= Cout <= 16 required (got %d)", Cout);
  MIS_REQUIRE((size_t)Cin * Cout * 4 <= 48 * 1024, "conv1x1_fwd: Cin*Cout too large");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIS_F32) { DISPATCH_CO(conv1x1_fwd_t, float, Cout, x, x_affine, n_pix, Cin, w, bias, y, st) }
  else if (dtype == MIS_BF16) { DISPATCH_CO(conv1x1_fwd_t, bf16, Cout, x, x_affine, n_pix, Cin, w, bias, y, st) }
  else MIS_REQUIRE(false, "conv1x1_fwd: bad dtype %d", dtype);
  MIS_CHECK_LAUNCH("conv1x1_fwd_kernel");
  return MIS_OK;
}

size_t mis_conv1x1_bwd_workspace_bytes(int64_t n_pix, int Cin, int Cout) {
  return align256((size_t)C1_MAX_BLOCKS * ((size_t)Cout * Cin + Cout + 2 * (size_t)Cin) * sizeof(float));
}

int mis_conv1x1_bwd(const void* x, const float* x_affine, const void* dy, int dtype, int64_t n_pix, int Cin, int Cout,
                    const float* w, void* dx, float* dw, float* dbias, float* bn_sums, void* workspace,
                    mis_stream_t stream) {
  MIS_REQUIRE(!bn_sums || x_affine, "conv1x1_bwd: bn_sums needs the [4][Cin] affine");
  MIS_REQUIRE(x && dy && w && dw && dbias && workspace && n_pix >= 1 && Cin >= 1, "conv1x1_bwd: bad arguments");
  MIS_REQUIRE(Cout >= 1 && Cout <= 16, "conv1x1_bwd: 1 <= Cout <= 16 required (got %d)", Cout);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIS_F32) { DISPATCH_CO(conv1x1_bwd_t, float, Cout, x, x_affine, dy, n_pix, Cin, w, dx, dw, dbias, bn_sums, workspace, st) }
  else if (dtype == MIS_BF16) { DISPATCH_CO(conv1x1_bwd_t, bf16, Cout, x, x_affine, dy, n_pix, Cin, w, dx, dw, dbias, bn_sums, workspace, st) }
  else MIS_REQUIRE(false, "conv1x1_bwd: bad dtype %d", dtype);
  MIS_CHECK_LAUNCH("conv1x1_bwd_kernel");
  return MIS_OK;
}

int mis_nchw_f32_to_nhwc(const float* x, int B, int C, int H, int W, int dtype, void* y, mis_stream_t stream) {
  MIS_REQUIRE(x && y && B >= 1 && C >= 1 && H >= 1 && W >= 1, "nchw_f32_to_nhwc: bad arguments");
  const int64_t HW = (int64_t)H * W;
  const int grid = ew_grid((int64_t)B * HW, 256);
  if (dtype == MIS_F32) nchw_to_nhwc_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(x, (float*)y, B, C, HW);
  else if (dtype == MIS_BF16) nchw_to_nhwc_kernel<bf16><<<grid, 256, 0, (hipStream_t)stream>>>(x, (bf16*)y, B, C, HW);
  else MIS_REQUIRE(false, "nchw_f32_to_nhwc: bad dtype %d", dtype);
  MIS_CHECK_LAUNCH("nchw_to_nhwc_kernel");
  return MIS_OK;
}

}  // extern "C"
