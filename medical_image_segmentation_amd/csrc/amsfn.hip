// The non-convolution parts of the AMSFN multi-scale fusion block (model/utils/modules.py:791-871
// of the reference; used by A_UNet, model/a_unet.py:27-30,48-51):
//
//   * nn.MaxPool2d(kernel_size=3, stride=1, padding=1)            (modules.py:799)
//   * nn.AdaptiveAvgPool2d(1) of the four branch outputs          (modules.py:846-856)
//   * the squeeze MLP  conv1x1(4C->C) + ReLU + conv1x1(C->4C) + sigmoid   (modules.py:833-837,858-860)
//   * the weighted sum of the four branches                       (modules.py:862-868)
// and their backward passes.  The branch outputs arrive as lazy activations (conv output z +
// BatchNorm affine): every kernel applies relu(z*scale + shift) while loading.  All reductions
// run in a fixed order (no float atomics).  Every kernel is a streaming pass: HBM-bound.
#include <type_traits>
#include "common.h"
#include "vec.h"
#include "reduce.h"

namespace {

constexpr int TPB = 256;

// ---- max-pool 3x3, stride 1, padding 1 ------------------------------------------------
// Forward also records, per output element, which of the 9 window positions (row-major, first
// maximum, NaN maximal — the scan order of torch's kernel) supplied the value; the backward is
// then a gather: input pixel p receives dy[q] from each of the <= 9 windows q whose recorded
// position is p.
// A thread owns MP3_R consecutive output rows of one column and channel vector.  The maximum is taken in two
// steps that keep the scan order: first along each input row (3 taps, first maximum), then down the three
// rows of a window (first maximum again) — a row's result serves up to three output rows, so an output costs
// (MP3_R+2)*3/MP3_R loads and lazy-input conversions instead of 9.
constexpr int MP3_R = 4;

// Every load of a strip is unconditional (coordinates clamped into the image, the value masked afterwards): the
// loop body is one basic block, so all (MP3_R+2)*3 loads are in flight together instead of one per branch.
template <typename T, int V>
__global__ __launch_bounds__(TPB) void maxpool3_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                           uint8_t* __restrict__ arg, int B, int H, int W, int C,
                                                           const float* __restrict__ in_aff) {
  extern __shared__ float aff_lds[];
  const float* aff = nullptr;
  if (in_aff) {
    for (int i = threadIdx.x; i < 2 * C; i += TPB) aff_lds[i] = in_aff[i];
    __syncthreads();
    aff = aff_lds;
  }
  // 32-bit item arithmetic (the host checks the item count): 64-bit division is emulated
  const unsigned CV = (unsigned)(C / V), HS = (unsigned)((H + MP3_R - 1) / MP3_R);
  const unsigned total = (unsigned)B * HS * (unsigned)W * CV;
  for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < total; i += gridDim.x * TPB) {
    const unsigned cv = i % CV;
    unsigned p = i / CV;
    const int xo = (int)(p % (unsigned)W);
    p /= (unsigned)W;
    const int y0 = (int)(p % HS) * MP3_R;
    const int b = (int)(p / HS);
    const T* xb = x + (int64_t)b * H * W * C + cv * V;
    const bool okx[3] = {xo > 0, true, xo + 1 < W};
    const int xc[3] = {xo > 0 ? xo - 1 : 0, xo, xo + 1 < W ? xo + 1 : W - 1};
    float v[MP3_R + 2][3][V];
#pragma unroll
    for (int r = 0; r < MP3_R + 2; r++) {
      const int yy = y0 - 1 + r, yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
#pragma unroll
      for (int d = 0; d < 3; d++) vload<T, V>(xb + ((int64_t)yc * W + xc[d]) * C, v[r][d]);
    }
    float hm[3][V];          // per input row: maximum of its 3 taps ...
    int ha[3][V];            // ... and which tap (0..2) supplied it
    bool hv[3];              // row inside the image
#pragma unroll
    for (int r = 0; r < MP3_R + 2; r++) {
      const int s = r % 3, yy = y0 - 1 + r;
      hv[s] = yy >= 0 && yy < H;
      // start below everything at the first tap inside the image: a later tap wins only if greater, or NaN
      // over a number (`!(v <= m)` holds for both; `m == m` keeps the first NaN); padding never wins
#pragma unroll
      for (int k = 0; k < V; k++) { hm[s][k] = -INFINITY; ha[s][k] = okx[0] ? 0 : 1; }
#pragma unroll
      for (int d = 0; d < 3; d++) {
        affine_relu_rounded<T, V>(v[r][d], aff, C, cv * V);
#pragma unroll
        for (int k = 0; k < V; k++) {
          const bool take = okx[d] && !(v[r][d][k] <= hm[s][k]) && (hm[s][k] == hm[s][k]);
          hm[s][k] = take ? v[r][d][k] : hm[s][k];
          ha[s][k] = take ? d : ha[s][k];
        }
      }
      if (r < 2) continue;
      const int yo = y0 + r - 2;                                       // window rows: slots (r-2)%3, (r-1)%3, r%3
      if (yo >= H) continue;
      float m[V];
      int am[V];
      const bool top = hv[(r - 2) % 3];                                // else the window starts at its centre row
#pragma unroll
      for (int k = 0; k < V; k++) {
        m[k] = -INFINITY;
        am[k] = top ? ha[(r - 2) % 3][k] : 3 + ha[(r - 1) % 3][k];
      }
#pragma unroll
      for (int d = 0; d < 3; d++) {
        const int q = (r - 2 + d) % 3;
#pragma unroll
        for (int k = 0; k < V; k++) {
          const bool take = hv[q] && !(hm[q][k] <= m[k]) && (m[k] == m[k]);
          m[k] = take ? hm[q][k] : m[k];
          am[k] = take ? d * 3 + ha[q][k] : am[k];
        }
      }
      const int64_t o = ((((int64_t)b * H + yo) * W + xo) * CV + cv) * V;
      vstore<T, V>(y + o, m);
      // the V window positions of this vector as one store (V single-byte stores cost V store instructions)
      if constexpr (V == 8) {
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          lo |= (unsigned)am[k] << (8 * k);
          hi |= (unsigned)am[4 + k] << (8 * k);
        }
        *reinterpret_cast<uint2*>(arg + o) = uint2{lo, hi};
      } else if constexpr (V == 4) {
        unsigned lo = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) lo |= (unsigned)am[k] << (8 * k);
        *reinterpret_cast<unsigned*>(arg + o) = lo;
      } else {
#pragma unroll
        for (int k = 0; k < V; k++) arg[o + k] = (uint8_t)am[k];
      }
    }
  }
}

// The recorded positions of a vector as one load.
template <int V>
__device__ __forceinline__ void load_positions(const uint8_t* __restrict__ p, int (&a)[V]) {
  if constexpr (V == 8) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      a[k] = (u.x >> (8 * k)) & 255;
      a[4 + k] = (u.y >> (8 * k)) & 255;
    }
  } else if constexpr (V == 4) {
    const unsigned u = *reinterpret_cast<const unsigned*>(p);
#pragma unroll
    for (int k = 0; k < 4; k++) a[k] = (u >> (8 * k)) & 255;
  } else {
#pragma unroll
    for (int k = 0; k < V; k++) a[k] = p[k];
  }
}

// Backward as a gather over the same strips: a thread owns MP3_R rows of one column and channel vector and
// reads the gradient and the recorded positions of the (MP3_R+2) x 3 windows that can point into the strip,
// each once (3*(MP3_R+2)/MP3_R loads per output instead of 9), all unconditional.  The window centred at
// (yq, xq) gives its gradient to pixel (yq + t - 1, xq + u - 1) when its recorded position is 3t + u.
template <typename T, int V>
__global__ __launch_bounds__(TPB) void maxpool3_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ arg,
                                                           T* __restrict__ dx, int B, int H, int W, int C) {
  const unsigned CV = (unsigned)(C / V), HS = (unsigned)((H + MP3_R - 1) / MP3_R);
  const unsigned total = (unsigned)B * HS * (unsigned)W * CV;
  for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < total; i += gridDim.x * TPB) {
    const unsigned cv = i % CV;
    unsigned p = i / CV;
    const int xi = (int)(p % (unsigned)W);
    p /= (unsigned)W;
    const int y0 = (int)(p % HS) * MP3_R;
    const int b = (int)(p / HS);
    const int64_t base = (int64_t)b * H * W * C + cv * V;
    // window column xq = xi - ox sees this column at u = ox + 1: d = 0, 1, 2 <-> ox = -1, 0, 1 <-> xq = xi+1, xi, xi-1
    const bool okx[3] = {xi + 1 < W, true, xi > 0};
    const int xq[3] = {xi + 1 < W ? xi + 1 : W - 1, xi, xi > 0 ? xi - 1 : 0};
    float g[MP3_R + 2][3][V];
    int a[MP3_R + 2][3][V];
#pragma unroll
    for (int r = 0; r < MP3_R + 2; r++) {
      const int yq = y0 - 1 + r, yc = yq < 0 ? 0 : (yq >= H ? H - 1 : yq);
#pragma unroll
      for (int d = 0; d < 3; d++) {
        const int64_t off = base + ((int64_t)yc * W + xq[d]) * C;
        vload<T, V>(dy + off, g[r][d]);
        load_positions<V>(arg + off, a[r][d]);
      }
    }
    float acc[MP3_R][V];
#pragma unroll
    for (int j = 0; j < MP3_R; j++)
#pragma unroll
      for (int k = 0; k < V; k++) acc[j][k] = 0.f;
#pragma unroll
    for (int r = 0; r < MP3_R + 2; r++) {
      const int yq = y0 - 1 + r;
      const bool oky = yq >= 0 && yq < H;
#pragma unroll
      for (int d = 0; d < 3; d++) {
        const bool ok = oky && okx[d];
#pragma unroll
        for (int t = 0; t < 3; t++) {
          const int j = r + t - 2;                       // the strip row this window's row t is
          if (j < 0 || j >= MP3_R) continue;
#pragma unroll
          for (int k = 0; k < V; k++) acc[j][k] += (ok && a[r][d][k] == 3 * t + d) ? g[r][d][k] : 0.f;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < MP3_R; j++) {
      if (y0 + j >= H) continue;
      vstore<T, V>(dx + base + ((int64_t)(y0 + j) * W + xi) * C, acc[j]);
    }
  }
}

// ---- the same two passes for bf16 tensors of 8-channel vectors, on integer keys -------------------------
// Both strip kernels above are bound by the vector ALU (a wave instruction takes four cycles: about 8 lane
// instructions per byte of memory traffic at the streaming rate), not by memory.  These forms cut the
// instruction count per output from about 80 to about 20:
//   * a thread owns MP3_COLS adjacent columns and walks down `run` rows, so an input row is loaded and converted
//     once per thread (plus one halo column each side and two halo rows per run);
//   * forward, lazy input only: relu(z*scale + shift) rounded to bf16 is a non-negative number, its float32 bit
//     pattern orders like the number, and the low 16 bits are free: key = bits | stamp, where the stamp
//     MP3_STAMP - (4*row + column) (row, column counted inside the thread's patch) falls in scan order, makes
//     `v_max3_u32` return the first maximum and where it was at once — one instruction per 3 taps horizontally,
//     one vertically, nothing to add in between; padding has key 0 and never wins (every stamp is >= 4).
//     (A -0 activation — the affine must produce -0 exactly — would order above every number: it is numerically
//     a maximum whenever it wins against zeros only, so at worst a tie among zeros goes to it.)
constexpr int MP3_COLS = 2;
constexpr int MP3_RUN_MAX = 12;      // (4 .. 16 measure within 4 % of each other)
constexpr unsigned MP3_STAMP = 4 * (MP3_RUN_MAX + 1) + 3 + 4;
int g_mp3_strip_kernels = 0;      // test hook: 1 = the strip kernels for every input

__device__ __forceinline__ unsigned umax3(unsigned a, unsigned b, unsigned c) {
  unsigned d;
  asm("v_max3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(TPB) void maxpool3_fwd_keys_kernel(const bf16* __restrict__ x, bf16* __restrict__ y,
                                                                uint8_t* __restrict__ arg, int B, int H, int W, int C,
                                                                const float* __restrict__ in_aff, int run) {
  constexpr int V = 8, NC = MP3_COLS + 2;
  static_assert(NC <= 4, "the stamp gives the column two bits");
  extern __shared__ float aff_lds[];
  for (int i = threadIdx.x; i < 2 * C; i += TPB) aff_lds[i] = in_aff[i];
  __syncthreads();
  const unsigned CV = (unsigned)(C / V), WG = (unsigned)((W + MP3_COLS - 1) / MP3_COLS), HR = (unsigned)((H + run - 1) / run);
  const unsigned total = (unsigned)B * HR * WG * CV;
  for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < total; i += gridDim.x * TPB) {
    const unsigned cv = i % CV;
    unsigned p = i / CV;
    const int x0 = (int)(p % WG) * MP3_COLS;
    p /= WG;
    const int ys = (int)(p % HR) * run, ye = ys + run < H ? ys + run : H;
    const int b = (int)(p / HR);
    float sc[V], sh[V];
#pragma unroll
    for (int k = 0; k < V; k++) {
      sc[k] = aff_lds[cv * V + k];
      sh[k] = aff_lds[C + cv * V + k];
    }
    const bf16* xb = x + (int64_t)b * H * W * C + cv * V;
    int64_t coff[NC];
#pragma unroll
    for (int d = 0; d < NC; d++) {
      const int xx = x0 - 1 + d;
      coff[d] = (int64_t)(xx < 0 ? 0 : (xx >= W ? W - 1 : xx)) * C;
    }
    // the two halo columns (the others: see the stores); masks, not selects: a select on a per-thread condition
    // invites the compiler to branch around the conversions, which splits the row into a dozen basic blocks
    const unsigned m_first = x0 > 0 ? ~0u : 0u, m_last = x0 + MP3_COLS < W ? ~0u : 0u;
    uint4 raw[NC];
    auto issue = [&](int yy) {
      const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
#pragma unroll
      for (int d = 0; d < NC; d++) raw[d] = *reinterpret_cast<const uint4*>(xb + (int64_t)yc * W * C + coff[d]);
    };
    // raw -> the horizontal maxima of patch row r (image row ys - 1 + r); all 0 for a row outside the image
    auto finish = [&](int r, unsigned (&hk)[MP3_COLS][V]) {
      const int yy = ys - 1 + r;
      const unsigned m_row = (yy >= 0 && yy < H) ? ~0u : 0u;
      unsigned key[NC][V];
#pragma unroll
      for (int d = 0; d < NC; d++) {
        const unsigned stamp = MP3_STAMP - (unsigned)(4 * r + d);
        const unsigned w4[4] = {raw[d].x, raw[d].y, raw[d].z, raw[d].w};
#pragma unroll
        for (int k = 0; k < V; k += 2) {
          f32x2 a;
          a.x = fmaxf(__builtin_bit_cast(float, w4[k >> 1] << 16) * sc[k] + sh[k], 0.f);
          a.y = fmaxf(__builtin_bit_cast(float, w4[k >> 1] & 0xFFFF0000u) * sc[k + 1] + sh[k + 1], 0.f);
          const unsigned pr = __builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2));   // two rounded values
          key[d][k] = (pr << 16) | stamp;
          key[d][k + 1] = (pr & 0xFFFF0000u) | stamp;
        }
        if (d == 0 || d == NC - 1) {
          const unsigned m = d == 0 ? m_first : m_last;
#pragma unroll
          for (int k = 0; k < V; k++) key[d][k] &= m;
        }
      }
#pragma unroll
      for (int j = 0; j < MP3_COLS; j++)
#pragma unroll
        for (int k = 0; k < V; k++) hk[j][k] = umax3(key[j][k], key[j + 1][k], key[j + 2][k]) & m_row;
    };
    // output row ys + q from patch rows q, q+1, q+2: window position = 3*(row - q) + (column - j), and
    // 3*row + column = n - (n >> 2) for n = 4*row + column = MP3_STAMP - stamp, on four packed bytes at once
    auto emit = [&](int q, const unsigned (&ha)[MP3_COLS][V], const unsigned (&hb)[MP3_COLS][V],
                    const unsigned (&hc)[MP3_COLS][V]) {
#pragma unroll
      for (int j = 0; j < MP3_COLS; j++) {
        unsigned key[V];
#pragma unroll
        for (int k = 0; k < V; k++) key[k] = umax3(ha[j][k], hb[j][k], hc[j][k]);
        if (x0 + j >= W) continue;
        const int64_t o = (((int64_t)b * H + ys + q) * W + x0 + j) * C + cv * V;
        uint4 val;   // the upper halves of two keys are two bf16 values
        val.x = __builtin_amdgcn_perm(key[1], key[0], 0x07060302u);
        val.y = __builtin_amdgcn_perm(key[3], key[2], 0x07060302u);
        val.z = __builtin_amdgcn_perm(key[5], key[4], 0x07060302u);
        val.w = __builtin_amdgcn_perm(key[7], key[6], 0x07060302u);
        *reinterpret_cast<uint4*>(y + o) = val;
        const unsigned first = (unsigned)(3 * q + j) * 0x01010101u;
        uint2 pos;
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const unsigned lo = __builtin_amdgcn_perm(key[4 * h + 1], key[4 * h], 0x0c0c0400u);
          const unsigned hi = __builtin_amdgcn_perm(key[4 * h + 3], key[4 * h + 2], 0x0c0c0400u);
          const unsigned n4 = MP3_STAMP * 0x01010101u - __builtin_amdgcn_perm(hi, lo, 0x05040100u);
          const unsigned p4 = n4 - ((n4 >> 2) & 0x3F3F3F3Fu) - first;
          if (h == 0) pos.x = p4; else pos.y = p4;
        }
        *reinterpret_cast<uint2*>(arg + o) = pos;
      }
    };
    unsigned h0[MP3_COLS][V], h1[MP3_COLS][V], h2[MP3_COLS][V];
    issue(ys - 1);
    finish(0, h0);
    issue(ys);
    finish(1, h1);
    issue(ys + 1);
    const int nq = ye - ys;
    int q = 0;
    while (true) {                                   // (each step loads the row after the one it converts)
      finish(q + 2, h2); issue(ys + q + 2); emit(q, h0, h1, h2); if (++q >= nq) break;
      finish(q + 2, h0); issue(ys + q + 2); emit(q, h1, h2, h0); if (++q >= nq) break;
      finish(q + 2, h1); issue(ys + q + 2); emit(q, h2, h0, h1); if (++q >= nq) break;
    }
  }
}

// Backward on the same thread layout: the windows of row yq give to pixel rows yq-1, yq, yq+1 — three rolling
// accumulator rows per thread; after window row yq pixel row yq-1 is complete.  A window outside the image gets
// position 15, which matches nothing.
__global__ __launch_bounds__(TPB) void maxpool3_bwd_run_kernel(const bf16* __restrict__ dy, const uint8_t* __restrict__ arg,
                                                               bf16* __restrict__ dx, int B, int H, int W, int C, int run) {
  constexpr int V = 8, NC = MP3_COLS + 2;
  const unsigned CV = (unsigned)(C / V), WG = (unsigned)((W + MP3_COLS - 1) / MP3_COLS), HR = (unsigned)((H + run - 1) / run);
  const unsigned total = (unsigned)B * HR * WG * CV;
  for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < total; i += gridDim.x * TPB) {
    const unsigned cv = i % CV;
    unsigned p = i / CV;
    const int x0 = (int)(p % WG) * MP3_COLS;
    p /= WG;
    const int ys = (int)(p % HR) * run, ye = ys + run < H ? ys + run : H;
    const int b = (int)(p / HR);
    const int64_t base = (int64_t)b * H * W * C + cv * V;
    int64_t coff[NC];
    bool okc[NC];
#pragma unroll
    for (int d = 0; d < NC; d++) {
      const int xx = x0 - 1 + d;
      okc[d] = xx >= 0 && xx < W;
      coff[d] = (int64_t)(xx < 0 ? 0 : (xx >= W ? W - 1 : xx)) * C;
    }
    uint4 graw[NC];
    uint2 araw[NC];
    auto issue = [&](int yq) {
      const int yc = yq < 0 ? 0 : (yq >= H ? H - 1 : yq);
#pragma unroll
      for (int d = 0; d < NC; d++) {
        const int64_t off = base + (int64_t)yc * W * C + coff[d];
        graw[d] = *reinterpret_cast<const uint4*>(dy + off);
        araw[d] = *reinterpret_cast<const uint2*>(arg + off);
      }
    };
    // window row yq: pixel row yq-1 is the windows' row 0 (`up`), yq their row 1 (`mid`), yq+1 their row 2 (`down`)
    auto add_row = [&](int yq, float (&up)[MP3_COLS][V], float (&mid)[MP3_COLS][V], float (&down)[MP3_COLS][V]) {
      const bool oky = yq >= 0 && yq < H;
#pragma unroll
      for (int d = 0; d < NC; d++) {
        const bool ok = oky && okc[d];
        const unsigned g4[4] = {graw[d].x, graw[d].y, graw[d].z, graw[d].w};
        const unsigned a2[2] = {ok ? araw[d].x : 0x0F0F0F0Fu, ok ? araw[d].y : 0x0F0F0F0Fu};
#pragma unroll
        for (int k = 0; k < V; k++) {
          const float g = __builtin_bit_cast(float, (k & 1) ? (g4[k >> 1] & 0xFFFF0000u) : (g4[k >> 1] << 16));
          const unsigned a = (a2[k >> 2] >> (8 * (k & 3))) & 255u;
#pragma unroll
          for (int j = 0; j < MP3_COLS; j++) {
            const int u = j + 2 - d;                 // the window's column that pixel column x0 + j is
            if (u < 0 || u > 2) continue;
            up[j][k] += a == (unsigned)u ? g : 0.f;
            mid[j][k] += a == (unsigned)(3 + u) ? g : 0.f;
            down[j][k] += a == (unsigned)(6 + u) ? g : 0.f;
          }
        }
      }
    };
    auto store = [&](int yp, float (&acc)[MP3_COLS][V]) {
#pragma unroll
      for (int j = 0; j < MP3_COLS; j++) {
        if (yp >= ys && yp < ye && x0 + j < W) vstore<bf16, V>(dx + base + ((int64_t)yp * W + x0 + j) * C, acc[j]);
#pragma unroll
        for (int k = 0; k < V; k++) acc[j][k] = 0.f;        // the slot's next use: the row two below
      }
    };
    float a0[MP3_COLS][V], a1[MP3_COLS][V], a2[MP3_COLS][V];
#pragma unroll
    for (int j = 0; j < MP3_COLS; j++)
#pragma unroll
      for (int k = 0; k < V; k++) a0[j][k] = a1[j][k] = a2[j][k] = 0.f;
    issue(ys - 1);
    int yq = ys - 1;
    while (true) {
      add_row(yq, a0, a1, a2); issue(yq + 1); store(yq - 1, a0); if (++yq > ye) break;
      add_row(yq, a1, a2, a0); issue(yq + 1); store(yq - 1, a1); if (++yq > ye) break;
      add_row(yq, a2, a0, a1); issue(yq + 1); store(yq - 1, a2); if (++yq > ye) break;
    }
  }
}

// rows per thread of the two kernels above: long runs (few halo rows) while the launch still has ~256 K threads
static inline int mp3_run(int B, int H, int W, int C) {
  const int64_t per_row = (int64_t)B * ((W + MP3_COLS - 1) / MP3_COLS) * (C / 8);
  int run = (int)((per_row * H) / 262144);
  if (run > MP3_RUN_MAX) run = MP3_RUN_MAX;
  if (run < 4) run = 4;
  return run < H ? run : H;
}

// ---- the four branch tensors ----------------------------------------------------------------
struct Branches {
  const void* z[4];        // [B][HW][C] each
  const float* aff[4];     // [2][C] (scale, shift) or null (plain tensor)
};

// activation vector of branch k at element offset idx (channel c0..c0+V)
template <typename T, int V>
__device__ __forceinline__ void act_vec(const Branches& br, int k, int C, int64_t idx, int c0, float (&v)[V]) {
  vload<T, V>((const T*)br.z[k] + idx, v);
  const float* aff = br.aff[k];
  if (aff) {
#pragma unroll
    for (int j = 0; j < V; j++) v[j] = to_f32(from_f32<T>(fmaxf(v[j] * aff[c0 + j] + aff[C + c0 + j], 0.f)));
  }
}

// partials[r][b][k*C + c] = scale * sum over the pixels p = r*PL + pl, step rows*PL, of
// act_k[b][p][c] * (g ? g[b][p][c] : 1).  With g = null and scale = 1/HW this is the global
// average pool of the four branches (modules.py:846-856); with g = dout it is the gradient of
// the fusion weights.  Thread = (channel vector cv, pixel lane pl); fixed-order LDS tree.
template <typename T, int V>
__global__ __launch_bounds__(TPB) void branch_dot_kernel(Branches br, const T* __restrict__ g, int B, int HW, int C,
                                                         int rows, float scale, float* __restrict__ partials) {
  extern __shared__ float red[];             // [PL][4][C]
  const int CV = C / V, PL = TPB / CV;
  const int r = blockIdx.x % rows, b = blockIdx.x / rows;
  const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
  float acc[4][V];
#pragma unroll
  for (int k = 0; k < 4; k++)
#pragma unroll
    for (int j = 0; j < V; j++) acc[k][j] = 0.f;
  if (pl < PL) {
    for (int p = r * PL + pl; p < HW; p += rows * PL) {
      const int64_t idx = ((int64_t)b * HW + p) * C + cv * V;
      float gv[V];
      if (g) vload<T, V>(g + idx, gv);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        float a[V];
        act_vec<T, V>(br, k, C, idx, cv * V, a);
#pragma unroll
        for (int j = 0; j < V; j++) acc[k][j] += g ? a[j] * gv[j] : a[j];
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int j = 0; j < V; j++) red[(pl * 4 + k) * C + cv * V + j] = acc[k][j];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 4 * C; e += TPB) {
    float t = 0.f;
    for (int q = 0; q < PL; q++) t += red[q * 4 * C + e];
    partials[((int64_t)r * B + b) * 4 * C + e] = t * scale;
  }
}

// Squeeze MLP, one workgroup per sample:  h = relu(W1 s + b1) [C],  w = sigmoid(W2 h + b2) [4C].
// W1 [C][4C], W2 [4C][C] (the 1x1 conv weights, modules.py:833-837).
__global__ __launch_bounds__(TPB) void mlp_fwd_kernel(const float* __restrict__ s, const float* __restrict__ W1,
                                                      const float* __restrict__ b1, const float* __restrict__ W2,
                                                      const float* __restrict__ b2, int C, float* __restrict__ h,
                                                      float* __restrict__ w) {
  extern __shared__ float sm[];              // s [4C] then h [C]
  const int b = blockIdx.x, C4 = 4 * C;
  float* ss = sm;
  float* hh = sm + C4;
  for (int i = threadIdx.x; i < C4; i += TPB) ss[i] = s[(int64_t)b * C4 + i];
  __syncthreads();
  for (int o = threadIdx.x; o < C; o += TPB) {
    float acc = b1[o];
    for (int i = 0; i < C4; i++) acc += W1[(int64_t)o * C4 + i] * ss[i];
    acc = fmaxf(acc, 0.f);
    hh[o] = acc;
    h[(int64_t)b * C + o] = acc;
  }
  __syncthreads();
  for (int o = threadIdx.x; o < C4; o += TPB) {
    float acc = b2[o];
    for (int i = 0; i < C; i++) acc += W2[(int64_t)o * C + i] * hh[i];
    w[(int64_t)b * C4 + o] = 1.0f / (1.0f + expf(-acc));
  }
}

// Backward, per sample: g2 = dw * w * (1-w) [4C];  dh = W2^T g2;  g1 = dh * [h > 0] [C];
// ds = W1^T g1 [4C].  g1 and g2 are kept for the parameter gradients.  One workgroup of MLP_BWD_THREADS per
// sample; the sum over the 4C rows of W2 is split over blockDim/C groups of C threads (each thread keeps its
// column, so the loads stay coalesced) and the groups are added in a fixed order.
constexpr int MLP_BWD_THREADS = 1024;
__global__ __launch_bounds__(MLP_BWD_THREADS) void mlp_bwd_sample_kernel(const float* __restrict__ dw,
                                                                         const float* __restrict__ w,
                                                                         const float* __restrict__ h,
                                                                         const float* __restrict__ W1,
                                                                         const float* __restrict__ W2, int C,
                                                                         float* __restrict__ g1, float* __restrict__ g2,
                                                                         float* __restrict__ ds) {
  extern __shared__ float sm[];              // g2 [4C] | g1 [C] | partial sums [blockDim/C][C]
  const int b = blockIdx.x, C4 = 4 * C, nt = blockDim.x;
  float* s2 = sm;
  float* s1 = sm + C4;
  float* part = s1 + C;
  for (int o = threadIdx.x; o < C4; o += nt) {
    const float y = w[(int64_t)b * C4 + o];
    const float g = dw[(int64_t)b * C4 + o] * y * (1.0f - y);
    s2[o] = g;
    g2[(int64_t)b * C4 + o] = g;
  }
  __syncthreads();
  const int groups = nt / C, i = threadIdx.x % C, grp = threadIdx.x / C;
  if (grp < groups) {
    float acc = 0.f;
#pragma unroll 8
    for (int o = grp; o < C4; o += groups) acc += W2[(int64_t)o * C + i] * s2[o];
    part[grp * C + i] = acc;
  }
  __syncthreads();
  if (threadIdx.x < C) {
    float acc = 0.f;
    for (int q = 0; q < groups; q++) acc += part[q * C + threadIdx.x];
    acc = (h[(int64_t)b * C + threadIdx.x] > 0.f) ? acc : 0.f;
    s1[threadIdx.x] = acc;
    g1[(int64_t)b * C + threadIdx.x] = acc;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < C4; j += nt) {
    float acc = 0.f;
#pragma unroll 8
    for (int q = 0; q < C; q++) acc += W1[(int64_t)q * C4 + j] * s1[q];
    ds[(int64_t)b * C4 + j] = acc;
  }
}

// Parameter gradients (sums over the batch in sample order):
//   dW1[i][j] = sum_b g1[b][i] s[b][j],  db1[i] = sum_b g1[b][i],
//   dW2[o][i] = sum_b g2[b][o] h[b][i],  db2[o] = sum_b g2[b][o].
__global__ __launch_bounds__(TPB) void mlp_bwd_param_kernel(const float* __restrict__ g1, const float* __restrict__ g2,
                                                            const float* __restrict__ s, const float* __restrict__ h,
                                                            int B, int C, float* __restrict__ dW1,
                                                            float* __restrict__ db1, float* __restrict__ dW2,
                                                            float* __restrict__ db2) {
  const int C4 = 4 * C;
  const int64_t nW = (int64_t)C * C4;
  const int64_t total = 2 * nW + C + C4;
  for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < total; e += (int64_t)gridDim.x * TPB) {
    float acc = 0.f;
    if (e < nW) {
      const int i = (int)(e / C4), j = (int)(e % C4);
      for (int b = 0; b < B; b++) acc += g1[(int64_t)b * C + i] * s[(int64_t)b * C4 + j];
      dW1[e] = acc;
    } else if (e < 2 * nW) {
      const int64_t f = e - nW;
      const int o = (int)(f / C), i = (int)(f % C);
      for (int b = 0; b < B; b++) acc += g2[(int64_t)b * C4 + o] * h[(int64_t)b * C + i];
      dW2[f] = acc;
    } else if (e < 2 * nW + C) {
      const int i = (int)(e - 2 * nW);
      for (int b = 0; b < B; b++) acc += g1[(int64_t)b * C + i];
      db1[i] = acc;
    } else {
      const int o = (int)(e - 2 * nW - C);
      for (int b = 0; b < B; b++) acc += g2[(int64_t)b * C4 + o];
      db2[o] = acc;
    }
  }
}

// out[b][p][c] = sum_k act_k[b][p][c] * w[b][k*C + c]
// Workgroup = (sample b, pixel chunk r); thread = (channel vector, pixel lane): the sample's four weight
// vectors and the branches' affines stay in registers, the loop holds four tensor loads and one store.
template <typename T, int V>
__global__ __launch_bounds__(TPB) void wsum_fwd_kernel(Branches br, const float* __restrict__ w, int B, int HW, int C,
                                                       int rows, int chunk, T* __restrict__ out) {
  const int CV = C / V, PL = TPB / CV;
  const int r = blockIdx.x % rows, b = blockIdx.x / rows;
  const int cv = threadIdx.x % CV, pl = threadIdx.x / CV, c0 = cv * V;
  if (pl >= PL) return;
  float wk[4][V], sc[4][V], sh[4][V];
#pragma unroll
  for (int k = 0; k < 4; k++)
#pragma unroll
    for (int j = 0; j < V; j++) {
      wk[k][j] = w[(int64_t)b * 4 * C + k * C + c0 + j];
      sc[k][j] = br.aff[k] ? br.aff[k][c0 + j] : 1.f;
      sh[k][j] = br.aff[k] ? br.aff[k][C + c0 + j] : 0.f;
    }
  const int p1 = (r + 1) * chunk < HW ? (r + 1) * chunk : HW;
  // (all four branches lazy — the case of the models — is decided once: a test per branch inside the loop puts
  // each of the four loads into a basic block of its own)
  auto run = [&](auto all_lazy_tag) {
    constexpr bool ALL_LAZY = decltype(all_lazy_tag)::value;
#pragma unroll 2
    for (int p = r * chunk + pl; p < p1; p += PL) {
      const int64_t idx = ((int64_t)b * HW + p) * C + c0;
      float a[4][V];
#pragma unroll
      for (int k = 0; k < 4; k++) vload<T, V>((const T*)br.z[k] + idx, a[k]);
      float acc[V];
#pragma unroll
      for (int j = 0; j < V; j++) acc[j] = 0.f;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (ALL_LAZY || br.aff[k]) {
#pragma unroll
          for (int j = 0; j < V; j++) a[k][j] = to_f32(from_f32<T>(fmaxf(a[k][j] * sc[k][j] + sh[k][j], 0.f)));
        }
#pragma unroll
        for (int j = 0; j < V; j++) acc[j] += a[k][j] * wk[k][j];
      }
      vstore<T, V>(out + idx, acc);
    }
  };
  if (br.aff[0] && br.aff[1] && br.aff[2] && br.aff[3]) run(std::true_type{}); else run(std::false_type{});
}

// dact_k[b][p][c] = dout[b][p][c] * w[b][k*C + c] + ds[b][k*C + c] / HW   (weighted sum + average pool)
struct Grads {
  void* d[4];
};
template <typename T, int V>
__global__ __launch_bounds__(TPB) void wsum_bwd_dact_kernel(const T* __restrict__ dout, const float* __restrict__ w,
                                                            const float* __restrict__ ds, int B, int HW, int C,
                                                            Grads out) {
  const int CV = C / V;
  const int64_t total = (int64_t)B * HW * CV;
  const float inv = 1.0f / (float)HW;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TPB) {
    const int cv = (int)(i % CV);
    const int b = (int)(i / ((int64_t)HW * CV));
    float g[V];
    vload<T, V>(dout + i * V, g);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int64_t j0 = (int64_t)b * 4 * C + k * C + cv * V;
      float o[V];
#pragma unroll
      for (int j = 0; j < V; j++) o[j] = g[j] * w[j0 + j] + ds[j0 + j] * inv;
      vstore<T, V>((T*)out.d[k] + i * V, o);
    }
  }
}

// ---- backward without the four branch gradients in memory -----------------------------------------
// The gradient of branch k is da_k = dout * w[b][k][c] + ds[b][k][c] / HW: one tensor (dout) and two small
// per-(sample, channel) tables.  The BatchNorm layer that produced the branch needs sum(dy) and sum(dy*xhat)
// of dy = da_k*[y > 0]; both are linear in the tables, so ONE pass over (dout, z_k) that runs BEFORE ds is
// known collects, per (sample, branch, channel),
//   q0 = sum a*g (the gradient of the fusion weights),  q1 = sum [y>0]*g,  q2 = sum [y>0]*g*xhat,
//   q3 = sum [y>0],  q4 = sum [y>0]*xhat,
// and a small kernel combines them after the squeeze MLP's backward:
//   sum dy = sum_b (w*q1 + ds/HW*q3),   sum dy*xhat = sum_b (w*q2 + ds/HW*q4).
// The branch layers then form da_k on the fly in their dz kernel (bn.hip, scaled mode); the four gradient
// tensors are neither written nor read, and their four reduction passes disappear.
struct BranchBn {
  const float* gamma[4];   // [C] each; mean and rstd are rows 2 and 3 of the branch's [4][C] affine
  const float* beta[4];
};
constexpr int ST_N_BWD = 5;

// Workgroup = (branch k, sample b, pixel chunk r); thread = (channel vector cv, pixel lane pl) walking a
// contiguous chunk.  partials[r][q][b][k*C + c]; fixed-order LDS tree.
// POOL: only sum a (g unused, one sum) — the global average pool of the forward pass, scaled by `scale`.
// Backward: the pass is bound by the vector ALU, not by memory (a wave instruction takes four cycles), so it
// accumulates q1..q4 only — with the mask as a float factor, 12 instructions per element — and forms
// q0 = sum a*g per workgroup as gamma*q2 + beta*q1: a = [y>0]*(gamma*xhat + beta) before its rounding to the
// storage type, so in float32 this is q0 up to the order of the additions, and in bf16 it is the gradient with
// respect to the unrounded activation (the rounding error of a is unbiased and relative 2^-9 per element).
template <typename T, int V, bool POOL = false>
__global__ __launch_bounds__(TPB) void branch_stats_kernel(Branches br, BranchBn bn, const T* __restrict__ g, int B,
                                                           int HW, int C, int rows, int chunk,
                                                           float* __restrict__ partials, float scale = 1.f) {
  constexpr int NQ = POOL ? 1 : ST_N_BWD - 1;      // sums accumulated per element
  extern __shared__ float red[];             // [PL][NQ][C]
  const int CV = C / V, PL = TPB / CV;
  // The four branches of one pixel chunk read the same gradient chunk: when the chunk count allows it they get
  // workgroup ids 8 apart (consecutive ids go round the 8 XCDs), so all four run on one XCD at about the same
  // time and its L2 serves three of the four reads.
  int k = blockIdx.x & 3, ch = blockIdx.x >> 2;
  if (!POOL && ((B * rows) & 7) == 0) {
    k = (blockIdx.x >> 3) & 3;
    ch = ((blockIdx.x >> 5) << 3) | (blockIdx.x & 7);
  }
  const int r = ch % rows, b = ch / rows;
  const int cv = threadIdx.x % CV, pl = threadIdx.x / CV, c0 = cv * V;
  const float* aff = br.aff[k];              // [4][C]: scale, shift, mean, rstd
  float acc[NQ][V];
#pragma unroll
  for (int q = 0; q < NQ; q++)
#pragma unroll
    for (int j = 0; j < V; j++) acc[q][j] = 0.f;
  if (pl < PL) {
    const T* zk = (const T*)br.z[k];
    const int p1 = (r + 1) * chunk < HW ? (r + 1) * chunk : HW;
    if constexpr (POOL) {
      float sc[V], sh[V];
#pragma unroll
      for (int j = 0; j < V; j++) {
        sc[j] = aff ? aff[c0 + j] : 1.f;
        sh[j] = aff ? aff[C + c0 + j] : 0.f;
      }
      // the activation as consumers read it (a plain branch tensor has no affine: decided once, not per element)
      auto run = [&](auto lazy_tag) {
        constexpr bool LAZY = decltype(lazy_tag)::value;
#pragma unroll 4
        for (int p = r * chunk + pl; p < p1; p += PL) {
          float zv[V];
          vload<T, V>(zk + ((int64_t)b * HW + p) * C + c0, zv);
#pragma unroll
          for (int j = 0; j < V; j++)
            acc[0][j] += LAZY ? to_f32(from_f32<T>(fmaxf(zv[j] * sc[j] + sh[j], 0.f))) : zv[j];
        }
      };
      if (aff) run(std::true_type{}); else run(std::false_type{});
    } else {
      float mu[V], rs[V], ga[V], be[V];
#pragma unroll
      for (int j = 0; j < V; j++) {
        mu[j] = aff[2 * C + c0 + j];
        rs[j] = aff[3 * C + c0 + j];
        ga[j] = bn.gamma[k][c0 + j];
        be[j] = bn.beta[k][c0 + j];
      }
#pragma unroll 2
      for (int p = r * chunk + pl; p < p1; p += PL) {
        const int64_t idx = ((int64_t)b * HW + p) * C + c0;
        float gv[V], zv[V];
        vload<T, V>(g + idx, gv);
        vload<T, V>(zk + idx, zv);
#pragma unroll
        for (int j = 0; j < V; j++) {
          const float xh = (zv[j] - mu[j]) * rs[j];
          const float on = xh * ga[j] + be[j] > 0.f ? 1.f : 0.f;          // the mask of the layer's dz kernel
          const float dy = on * gv[j];
          acc[0][j] += dy;
          acc[1][j] = fmaf(dy, xh, acc[1][j]);
          acc[2][j] += on;
          acc[3][j] = fmaf(on, xh, acc[3][j]);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NQ; q++)
#pragma unroll
      for (int j = 0; j < V; j++) red[(pl * NQ + q) * C + c0 + j] = acc[q][j];
  }
  __syncthreads();
  const int q0 = POOL ? 0 : 1;               // plane of the first accumulated sum
  const int planes = POOL ? 1 : ST_N_BWD;
  for (int e = threadIdx.x; e < NQ * C; e += TPB) {
    float t = 0.f;
    for (int q = 0; q < PL; q++) t += red[q * NQ * C + e];
    const int qq = e / C, c = e % C;
    partials[(((int64_t)r * planes + q0 + qq) * B + b) * 4 * C + k * C + c] = t * scale;
  }
  if constexpr (!POOL) {
    for (int c = threadIdx.x; c < C; c += TPB) {
      float t1 = 0.f, t2 = 0.f;
      for (int q = 0; q < PL; q++) {
        t1 += red[q * NQ * C + c];
        t2 += red[(q * NQ + 1) * C + c];
      }
      partials[(((int64_t)r * planes) * B + b) * 4 * C + k * C + c] = bn.gamma[k][c] * t2 + bn.beta[k][c] * t1;
    }
  }
}

// stats [ST_N_BWD][B][4C] (+ w, ds [B][4C]) -> bn_sums [4][2][C] and the shift table ds/HW [B][4C]
__global__ __launch_bounds__(TPB) void branch_bn_sums_kernel(const float* __restrict__ stats, const float* __restrict__ w,
                                                             const float* __restrict__ ds, float inv, int B, int C,
                                                             float* __restrict__ bn_sums, float* __restrict__ shift) {
  const int C4 = 4 * C;
  const int64_t plane = (int64_t)B * C4;
  for (int j = blockIdx.x * TPB + threadIdx.x; j < C4; j += gridDim.x * TPB) {
    double s1 = 0.0, s2 = 0.0;
    for (int b = 0; b < B; b++) {
      const int64_t i = (int64_t)b * C4 + j;
      const float wk = w[i], sk = ds[i] * inv;
      shift[i] = sk;
      s1 += (double)wk * stats[plane + i] + (double)sk * stats[3 * plane + i];
      s2 += (double)wk * stats[2 * plane + i] + (double)sk * stats[4 * plane + i];
    }
    const int k = j / C, c = j % C;
    bn_sums[(k * 2 + 0) * C + c] = (float)s1;
    bn_sums[(k * 2 + 1) * C + c] = (float)s2;
  }
}

// pixel chunks per (sample, branch): >= 16 pixels per thread, about 8192 workgroups at most
inline int stats_rows(int B, int HW, int C, int V) {
  const int PL = TPB / (C / V);
  int rows = HW / (PL * 16);
  const int cap = (8192 + 4 * B - 1) / (4 * B);
  if (rows > cap) rows = cap;
  if (rows < 1) rows = 1;
  return rows;
}

// pixel rows of the two reductions: enough workgroups to fill the chip, >= 8 pixels per thread
inline int dot_rows(int B, int HW, int C, int V) {
  const int PL = TPB / (C / V);
  int rows = HW / (PL * 8);
  const int cap = (4096 + B - 1) / B;
  if (rows > cap) rows = cap;
  if (rows < 1) rows = 1;
  return rows;
}

// ---- DWConv = pointwise(depthwise(x)) as one dense 3x3 conv ---------------------------------
// W[o][i][t] = P[o][i] * D[i][t],  b[o] = bp[o] + sum_i P[o][i] * bd[i]   (modules.py:52-60)
__global__ __launch_bounds__(TPB) void dw_compose_fwd_kernel(const float* __restrict__ P, const float* __restrict__ D,
                                                             const float* __restrict__ bd,
                                                             const float* __restrict__ bp, int Cout, int Cin,
                                                             float* __restrict__ W, float* __restrict__ b) {
  const int nW = Cout * Cin * 9;
  for (int e = blockIdx.x * TPB + threadIdx.x; e < nW + Cout; e += gridDim.x * TPB) {
    if (e < nW) {
      const int t = e % 9, i = (e / 9) % Cin, o = e / (9 * Cin);
      W[e] = P[o * Cin + i] * D[i * 9 + t];
    } else {
      const int o = e - nW;
      float acc = 0.f;
      for (int i = 0; i < Cin; i++) acc += P[o * Cin + i] * bd[i];
      b[o] = bp[o] + acc;
    }
  }
}

// the same composition for many layers: job j owns workgroups [first_block, first_block + n_blocks)
__global__ __launch_bounds__(TPB) void dw_compose_multi_kernel(const mis_compose_job* __restrict__ jobs, int n_jobs) {
  int j = 0;
  while (j + 1 < n_jobs && (int)blockIdx.x >= jobs[j + 1].first_block) j++;
  const mis_compose_job jb = jobs[j];
  const int Cin = jb.Cin, Cout = jb.Cout, nW = Cout * Cin * 9;
  const int total = nW + (jb.D ? Cout : 0);
  for (int e = ((int)blockIdx.x - jb.first_block) * TPB + threadIdx.x; e < total; e += jb.n_blocks * TPB) {
    if (e < nW) {
      const int t = e % 9, i = (e / 9) % Cin, o = e / (9 * Cin);
      jb.W[e] = jb.D ? jb.P[o * Cin + i] * jb.D[i * 9 + t] : (t == 4 ? jb.P[o * Cin + i] : 0.f);
    } else {
      const int o = e - nW;
      float acc = 0.f;
      for (int i = 0; i < Cin; i++) acc += jb.P[o * Cin + i] * jb.bd[i];
      jb.b[o] = jb.bp[o] + acc;
    }
  }
}

// dP[o][i] = sum_t dW[o][i][t] D[i][t] + db[o] bd[i];  dD[i][t] = sum_o dW[o][i][t] P[o][i];
// dbd[i] = sum_o db[o] P[o][i]   (dbp = db).  Fixed summation order.
__device__ __forceinline__ void dw_compose_bwd_body(const float* __restrict__ P, const float* __restrict__ D,
                                                    const float* __restrict__ bd, const float* __restrict__ dW,
                                                    const float* __restrict__ db, int Cout, int Cin,
                                                    float* __restrict__ dP, float* __restrict__ dD,
                                                    float* __restrict__ dbd, int blk, int nblk) {
  // dP: one thread per element (9 terms).  dD and dbd sum over Cout: 8 consecutive lanes share an
  // element (lane j takes o = j, j+8, ...) and are summed on the DPP path, so the dependent chain is
  // Cout/8 long instead of Cout.  TPB is a multiple of 8 and the grid-stride keeps lane groups whole.
  // (the 8-lane groups start at a multiple of 8, so that a group is one aligned lane octet of a wave)
  const int nP = Cout * Cin, nP8 = (nP + 7) & ~7, nD = Cin * 9;
  const int total = nP8 + 8 * (nD + Cin);
  for (int e = blk * TPB + threadIdx.x; e < total; e += nblk * TPB) {
    float acc = 0.f;
    if (e < nP) {
      const int i = e % Cin, o = e / Cin;
      for (int t = 0; t < 9; t++) acc += dW[(o * Cin + i) * 9 + t] * D[i * 9 + t];
      dP[e] = acc + db[o] * bd[i];
    } else if (e >= nP8) {
      const int g = (e - nP8) >> 3, j = (e - nP8) & 7;
      if (g < nD) {
        const int t = g % 9, i = g / 9;
        for (int o = j; o < Cout; o += 8) acc += dW[(o * Cin + i) * 9 + t] * P[o * Cin + i];
      } else {
        const int i = g - nD;
        for (int o = j; o < Cout; o += 8) acc += db[o] * P[o * Cin + i];
      }
      acc += dpp_move<0xB1>(acc);
      acc += dpp_move<0x4E>(acc);
      acc += dpp_move<0x141>(acc);
      if (j == 0) {
        if (g < nD) dD[g] = acc;
        else dbd[g - nD] = acc;
      }
    }
  }
}

__global__ __launch_bounds__(TPB) void dw_compose_bwd_kernel(const float* __restrict__ P, const float* __restrict__ D,
                                                             const float* __restrict__ bd,
                                                             const float* __restrict__ dW,
                                                             const float* __restrict__ db, int Cout, int Cin,
                                                             float* __restrict__ dP, float* __restrict__ dD,
                                                             float* __restrict__ dbd) {
  dw_compose_bwd_body(P, D, bd, dW, db, Cout, Cin, dP, dD, dbd, blockIdx.x, gridDim.x);
}

// The chain rule of MANY layers in one launch (the jobs travel in the kernel arguments): layer j owns workgroups
// [first[j], first[j+1]).  D == NULL: a 1x1 conv carried as a 3x3 — dP is the centre tap of dW.
constexpr int COMPOSE_BWD_MAX = 24;
struct ComposeBwdArgs {
  mis_compose_bwd_job job[COMPOSE_BWD_MAX];
  int first[COMPOSE_BWD_MAX + 1];
  int n;
};
__global__ __launch_bounds__(TPB) void dw_compose_bwd_multi_kernel(const ComposeBwdArgs a) {
  int j = 0;
  while (j + 1 < a.n && (int)blockIdx.x >= a.first[j + 1]) j++;
  const mis_compose_bwd_job& jb = a.job[j];
  const int blk = (int)blockIdx.x - a.first[j], nblk = a.first[j + 1] - a.first[j];
  if (jb.D) {
    dw_compose_bwd_body(jb.P, jb.D, jb.bd, jb.dW, jb.db, jb.Cout, jb.Cin, jb.dP, jb.dD, jb.dbd, blk, nblk);
  } else {
    const int nP = jb.Cout * jb.Cin;
    for (int e = blk * TPB + threadIdx.x; e < nP; e += nblk * TPB) jb.dP[e] = jb.dW[(int64_t)e * 9 + 4];
  }
}

}  // namespace

extern "C" {

int mis_set_maxpool3_path(int strip_kernels) {
  const int before = g_mp3_strip_kernels;
  g_mp3_strip_kernels = strip_kernels ? 1 : 0;
  return before;
}

int mis_maxpool3x3s1_fwd(const void* x, const float* x_affine, int dtype, int B, int H, int W, int C, void* y,
                         uint8_t* argmax, mis_stream_t stream) {
  MIS_REQUIRE(x && y && argmax, "maxpool3x3s1_fwd: null pointer");
  MIS_REQUIRE(B >= 1 && H >= 1 && W >= 1 && C >= 1 && C <= 4096, "maxpool3x3s1_fwd: bad shape");
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = x_affine ? 2 * (size_t)C * sizeof(float) : 0;
  // items: one per (image, strip of MP3_R rows, column, channel vector)
  const int64_t n = (int64_t)B * ((H + MP3_R - 1) / MP3_R) * W * C;
  MIS_REQUIRE(n < ((int64_t)1 << 31), "maxpool3x3s1_fwd: tensor too large for 32-bit item arithmetic");
  if (dtype == MIS_BF16) {
    if (x_affine && pick_vec<bf16>(C, x, y, argmax) == 8 && !g_mp3_strip_kernels) {
      const int run = mp3_run(B, H, W, C);
      const int64_t items = (int64_t)B * ((H + run - 1) / run) * ((W + MP3_COLS - 1) / MP3_COLS) * (C / 8);
      maxpool3_fwd_keys_kernel<<<ew_grid(items, TPB), TPB, lds, st>>>((const bf16*)x, (bf16*)y, argmax, B, H, W, C, x_affine, run);
    } else if (pick_vec<bf16>(C, x, y, argmax) == 8)
      maxpool3_fwd_kernel<bf16, 8><<<ew_grid(n / 8, TPB), TPB, lds, st>>>((const bf16*)x, (bf16*)y, argmax, B, H, W, C, x_affine);
    else
      maxpool3_fwd_kernel<bf16, 1><<<ew_grid(n, TPB), TPB, lds, st>>>((const bf16*)x, (bf16*)y, argmax, B, H, W, C, x_affine);
  } else if (dtype == MIS_F32) {
    if (pick_vec<float>(C, x, y, argmax) == 4)
      maxpool3_fwd_kernel<float, 4><<<ew_grid(n / 4, TPB), TPB, lds, st>>>((const float*)x, (float*)y, argmax, B, H, W, C, x_affine);
    else
      maxpool3_fwd_kernel<float, 1><<<ew_grid(n, TPB), TPB, lds, st>>>((const float*)x, (float*)y, argmax, B, H, W, C, x_affine);
  } else {
    MIS_REQUIRE(false, "maxpool3x3s1_fwd: bad dtype %d", dtype);
  }
  MIS_CHECK_LAUNCH("maxpool3_fwd_kernel");
  return MIS_OK;
}

int mis_maxpool3x3s1_bwd(const void* dy, const uint8_t* argmax, int dtype, int B, int H, int W, int C, void* dx,
                         mis_stream_t stream) {
  MIS_REQUIRE(dy && argmax && dx, "maxpool3x3s1_bwd: null pointer");
  MIS_REQUIRE(B >= 1 && H >= 1 && W >= 1 && C >= 1, "maxpool3x3s1_bwd: bad shape");
  hipStream_t st = (hipStream_t)stream;
  // items as in the forward pass
  const int64_t n = (int64_t)B * ((H + MP3_R - 1) / MP3_R) * W * C;
  MIS_REQUIRE(n < ((int64_t)1 << 31), "maxpool3x3s1_bwd: tensor too large for 32-bit item arithmetic");
  if (dtype == MIS_BF16) {
    if (pick_vec<bf16>(C, dy, dx, argmax) == 8 && !g_mp3_strip_kernels) {
      const int run = mp3_run(B, H, W, C);
      const int64_t items = (int64_t)B * ((H + run - 1) / run) * ((W + MP3_COLS - 1) / MP3_COLS) * (C / 8);
      maxpool3_bwd_run_kernel<<<ew_grid(items, TPB), TPB, 0, st>>>((const bf16*)dy, argmax, (bf16*)dx, B, H, W, C, run);
    } else if (pick_vec<bf16>(C, dy, dx, argmax) == 8)
      maxpool3_bwd_kernel<bf16, 8><<<ew_grid(n / 8, TPB), TPB, 0, st>>>((const bf16*)dy, argmax, (bf16*)dx, B, H, W, C);
    else
      maxpool3_bwd_kernel<bf16, 1><<<ew_grid(n, TPB), TPB, 0, st>>>((const bf16*)dy, argmax, (bf16*)dx, B, H, W, C);
  } else if (dtype == MIS_F32) {
    if (pick_vec<float>(C, dy, dx, argmax) == 4)
      maxpool3_bwd_kernel<float, 4><<<ew_grid(n / 4, TPB), TPB, 0, st>>>((const float*)dy, argmax, (float*)dx, B, H, W, C);
    else
      maxpool3_bwd_kernel<float, 1><<<ew_grid(n, TPB), TPB, 0, st>>>((const float*)dy, argmax, (float*)dx, B, H, W, C);
  } else {
    MIS_REQUIRE(false, "maxpool3x3s1_bwd: bad dtype %d", dtype);
  }
  MIS_CHECK_LAUNCH("maxpool3_bwd_kernel");
  return MIS_OK;
}

}  // extern "C"

namespace {

// V = full 16-byte channel vectors when C allows (and at most 256 vectors per pixel), else scalar
template <typename T>
inline int fuse_vec(int C, const Branches& br, const void* a, const void* b) {
  const int N = VecOf<T>::N;
  uintptr_t m = (uintptr_t)a | (uintptr_t)b;
  for (int k = 0; k < 4; k++) m |= (uintptr_t)br.z[k];
  return (C % N == 0 && C / N <= TPB && (m % 16) == 0) ? N : 1;
}

template <typename T>
int launch_dot(const Branches& br, const void* g, int B, int HW, int C, int V, float scale, float* partials, int& rows,
               hipStream_t st) {
  rows = dot_rows(B, HW, C, V);
  const size_t lds = (size_t)(TPB / (C / V)) * 4 * C * sizeof(float);
  if (V == VecOf<T>::N)
    branch_dot_kernel<T, VecOf<T>::N><<<B * rows, TPB, lds, st>>>(br, (const T*)g, B, HW, C, rows, scale, partials);
  else
    branch_dot_kernel<T, 1><<<B * rows, TPB, lds, st>>>(br, (const T*)g, B, HW, C, rows, scale, partials);
  return 0;
}

template <typename T>
int fuse_fwd_t(const Branches& br, int B, int HW, int C, const float* W1, const float* b1, const float* W2,
               const float* b2, float* s, float* h, float* w, void* out, void* workspace, hipStream_t st) {
  const int V = fuse_vec<T>(C, br, out, nullptr);
  const int rows = stats_rows(B, HW, C, V), chunk = (HW + rows - 1) / rows;
  const size_t lds = (size_t)(TPB / (C / V)) * C * sizeof(float);
  BranchBn none{};
  if (V == VecOf<T>::N)
    branch_stats_kernel<T, VecOf<T>::N, true><<<4 * B * rows, TPB, lds, st>>>(br, none, nullptr, B, HW, C, rows, chunk,
                                                                            (float*)workspace, 1.0f / (float)HW);
  else
    branch_stats_kernel<T, 1, true><<<4 * B * rows, TPB, lds, st>>>(br, none, nullptr, B, HW, C, rows, chunk,
                                                                    (float*)workspace, 1.0f / (float)HW);
  MIS_CHECK_LAUNCH("branch_stats_kernel");
  launch_reduce_partials((const float*)workspace, rows, B * 4 * C, s, st);
  MIS_CHECK_LAUNCH("reduce_partials_kernel");
  mlp_fwd_kernel<<<B, TPB, (size_t)5 * C * sizeof(float), st>>>(s, W1, b1, W2, b2, C, h, w);
  MIS_CHECK_LAUNCH("mlp_fwd_kernel");
  // pixel chunks of the weighted sum: about 4096 workgroups, >= 8 pixels per thread
  int wrows = HW / ((TPB / (C / V)) * 8);
  if (wrows > (4096 + B - 1) / B) wrows = (4096 + B - 1) / B;
  if (wrows < 1) wrows = 1;
  const int wchunk = (HW + wrows - 1) / wrows;
  if (V == VecOf<T>::N)
    wsum_fwd_kernel<T, VecOf<T>::N><<<B * wrows, TPB, 0, st>>>(br, w, B, HW, C, wrows, wchunk, (T*)out);
  else
    wsum_fwd_kernel<T, 1><<<B * wrows, TPB, 0, st>>>(br, w, B, HW, C, wrows, wchunk, (T*)out);
  MIS_CHECK_LAUNCH("wsum_fwd_kernel");
  return MIS_OK;
}

template <typename T>
int fuse_bwd_t(const Branches& br, const Grads& gr, const void* dout, int B, int HW, int C, const float* W1,
               const float* W2, const float* s, const float* h, const float* w, float* dW1, float* db1, float* dW2,
               float* db2, void* workspace, hipStream_t st) {
  const int C4 = 4 * C;
  uintptr_t m = 0;
  for (int k = 0; k < 4; k++) m |= (uintptr_t)gr.d[k];
  const int V = fuse_vec<T>(C, br, dout, (const void*)m);
  // workspace: partial rows [rows][B][4C] | dw [B][4C] | g2 [B][4C] | ds [B][4C] | g1 [B][C]
  const size_t vec = align256((size_t)B * C4 * sizeof(float));
  float* partials = (float*)workspace;
  float* dw = (float*)((char*)workspace + align256((size_t)dot_rows(B, HW, C, V) * B * C4 * sizeof(float)));
  float* g2 = (float*)((char*)dw + vec);
  float* ds = (float*)((char*)g2 + vec);
  float* g1 = (float*)((char*)ds + vec);
  int rows;
  launch_dot<T>(br, dout, B, HW, C, V, 1.0f, partials, rows, st);
  MIS_CHECK_LAUNCH("branch_dot_kernel");
  launch_reduce_partials(partials, rows, B * C4, dw, st);
  MIS_CHECK_LAUNCH("reduce_partials_kernel");
  mlp_bwd_sample_kernel<<<B, MLP_BWD_THREADS, (size_t)(5 * C + MLP_BWD_THREADS) * sizeof(float), st>>>(dw, w, h, W1, W2, C, g1, g2, ds);
  MIS_CHECK_LAUNCH("mlp_bwd_sample_kernel");
  const int64_t n = (int64_t)B * HW * C;
  if (V == VecOf<T>::N)
    wsum_bwd_dact_kernel<T, VecOf<T>::N><<<ew_grid(n / V, TPB), TPB, 0, st>>>((const T*)dout, w, ds, B, HW, C, gr);
  else
    wsum_bwd_dact_kernel<T, 1><<<ew_grid(n, TPB), TPB, 0, st>>>((const T*)dout, w, ds, B, HW, C, gr);
  MIS_CHECK_LAUNCH("wsum_bwd_dact_kernel");
  const int64_t np = 2 * (int64_t)C * C4 + C + C4;
  mlp_bwd_param_kernel<<<ew_grid(np, TPB), TPB, 0, st>>>(g1, g2, s, h, B, C, dW1, db1, dW2, db2);
  MIS_CHECK_LAUNCH("mlp_bwd_param_kernel");
  return MIS_OK;
}


template <typename T>
int fuse_bwd_lazy_t(const Branches& br, const BranchBn& bn, const void* dout, int B, int HW, int C, const float* W1,
                    const float* W2, const float* s, const float* h, const float* w, float* bn_sums, float* shift,
                    float* dW1, float* db1, float* dW2, float* db2, void* workspace, hipStream_t st) {
  const int C4 = 4 * C;
  const int V = fuse_vec<T>(C, br, dout, nullptr);
  const int rows = stats_rows(B, HW, C, V), chunk = (HW + rows - 1) / rows;
  // workspace: partial rows [rows][ST_N_BWD][B][4C] | stats [ST_N_BWD][B][4C] (plane 0 = dw) | g2 | ds [B][4C] | g1 [B][C]
  const size_t vec = align256((size_t)B * C4 * sizeof(float));
  float* partials = (float*)workspace;
  float* stats = (float*)((char*)workspace + align256((size_t)rows * ST_N_BWD * B * C4 * sizeof(float)));
  float* g2 = (float*)((char*)stats + align256((size_t)ST_N_BWD * B * C4 * sizeof(float)));
  float* ds = (float*)((char*)g2 + vec);
  float* g1 = (float*)((char*)ds + vec);
  const size_t lds = (size_t)(TPB / (C / V)) * ST_N_BWD * C * sizeof(float);
  if (V == VecOf<T>::N)
    branch_stats_kernel<T, VecOf<T>::N><<<4 * B * rows, TPB, lds, st>>>(br, bn, (const T*)dout, B, HW, C, rows, chunk, partials);
  else
    branch_stats_kernel<T, 1><<<4 * B * rows, TPB, lds, st>>>(br, bn, (const T*)dout, B, HW, C, rows, chunk, partials);
  MIS_CHECK_LAUNCH("branch_stats_kernel");
  launch_reduce_partials(partials, rows, ST_N_BWD * B * C4, stats, st);
  MIS_CHECK_LAUNCH("reduce_partials_kernel");
  mlp_bwd_sample_kernel<<<B, MLP_BWD_THREADS, (size_t)(5 * C + MLP_BWD_THREADS) * sizeof(float), st>>>(stats, w, h, W1, W2, C, g1, g2, ds);
  MIS_CHECK_LAUNCH("mlp_bwd_sample_kernel");
  branch_bn_sums_kernel<<<cdiv(C4, TPB), TPB, 0, st>>>(stats, w, ds, 1.0f / (float)HW, B, C, bn_sums, shift);
  MIS_CHECK_LAUNCH("branch_bn_sums_kernel");
  const int64_t np = 2 * (int64_t)C * C4 + C + C4;
  mlp_bwd_param_kernel<<<ew_grid(np, TPB), TPB, 0, st>>>(g1, g2, s, h, B, C, dW1, db1, dW2, db2);
  MIS_CHECK_LAUNCH("mlp_bwd_param_kernel");
  return MIS_OK;
}

// ---- several consumers of one activation: their gradients summed in ONE pass ----------------------
// out = g0 + g1 (+ g2 + g3 + g4), added in float32 in slot order and rounded once.  When the activation is
// the lazy output relu(bn(z)) of a training-mode layer (z and its [4][C] affine given), the kernel also
// accumulates that layer's BatchNorm-backward sums of the summed gradient as it is stored —
// sum out*[y>0] and sum out*[y>0]*xhat — as one partial row per workgroup (fixed channel vector per thread,
// fixed-order block reduction), so the layer's own reduction pass over (z, gradient) disappears.
constexpr int SUM_MAX = 5;
struct SumPtrs {
  const void* g[SUM_MAX];
};
template <typename T, int V, int N>
__global__ __launch_bounds__(TPB) void sum_gradients_kernel(SumPtrs src, T* __restrict__ out, int64_t n_vec, int C,
                                                            const T* __restrict__ z, const float* __restrict__ aff4,
                                                            float* __restrict__ bn_partials) {
  __shared__ float red[TPB * 2 * V];
  const int CV = C / V;
  const int64_t i0 = (int64_t)blockIdx.x * TPB + threadIdx.x;
  const int c0 = (int)(i0 % CV) * V;             // fixed for this thread when sums are wanted (the grid is sized for that)
  float sc[V], sh[V], mu[V], rs[V], bs[2][V];
#pragma unroll
  for (int k = 0; k < V; k++) {
    sc[k] = bn_partials ? aff4[c0 + k] : 0.f;
    sh[k] = bn_partials ? aff4[C + c0 + k] : 0.f;
    mu[k] = bn_partials ? aff4[2 * C + c0 + k] : 0.f;
    rs[k] = bn_partials ? aff4[3 * C + c0 + k] : 0.f;
    bs[0][k] = bs[1][k] = 0.f;
  }
  for (int64_t i = i0; i < n_vec; i += (int64_t)gridDim.x * TPB) {
    float acc[V], t[V];
    vload<T, V>((const T*)src.g[0] + i * V, acc);
#pragma unroll
    for (int j = 1; j < N; j++) {
      vload<T, V>((const T*)src.g[j] + i * V, t);
#pragma unroll
      for (int k = 0; k < V; k++) acc[k] += t[k];
    }
    vstore<T, V>(out + i * V, acc);
    if (bn_partials) {
      float zv[V];
      vload<T, V>(z + i * V, zv);
#pragma unroll
      for (int k = 0; k < V; k++) {
        const bool on = zv[k] * sc[k] + sh[k] > 0.f;
        const float d = on ? to_f32(from_f32<T>(acc[k])) : 0.f;      // the value the next pass would read
        bs[0][k] += d;
        bs[1][k] = on ? __builtin_fmaf(d, (zv[k] - mu[k]) * rs[k], bs[1][k]) : bs[1][k];
      }
    }
  }
  if (bn_partials) {
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
      for (int k = 0; k < V; k++) red[(threadIdx.x * 2 + q) * V + k] = bs[q][k];
    __syncthreads();
    const int base = (int)(((int64_t)blockIdx.x * TPB) % CV);
    for (int j = threadIdx.x; j < 2 * C; j += TPB) {
      const int q = j / C, c = j % C, cv = c / V, k = c % V;
      float a = 0.f;
      for (int t = (cv - base + CV) % CV; t < TPB; t += CV) a += red[(t * 2 + q) * V + k];
      bn_partials[(int64_t)blockIdx.x * 2 * C + j] = a;
    }
  }
}

constexpr int SUM_MAX_ROWS = 1536;

template <typename T, int V, int N>
void launch_sum_n(const SumPtrs& src, void* out, int64_t n, int C, const void* z, const float* aff4, float* bn_sums,
                  void* ws, hipStream_t st) {
  int grid = ew_grid(n / V, TPB);
  float* partials = nullptr;
  if (bn_sums) {
    grid = grid_for_fixed_channels(grid, C / V, SUM_MAX_ROWS);
    partials = (float*)ws;
  }
  sum_gradients_kernel<T, V, N><<<grid, TPB, 0, st>>>(src, (T*)out, n / V, C, (const T*)z, aff4, partials);
  if (bn_sums) launch_reduce_partials(partials, grid, 2 * C, bn_sums, st);
}

template <typename T>
int sum_gradients_t(const SumPtrs& src, int n_src, void* out, int64_t n_pix, int C, const void* z, const float* aff4,
                    float* bn_sums, void* ws, hipStream_t st) {
  constexpr int NV = VecOf<T>::N;
  bool vec = pick_vec<T>(C, out, z) == NV && C / NV <= TPB;
  for (int j = 0; j < n_src; j++) vec = vec && ((uintptr_t)src.g[j] % 16) == 0;
  const int64_t n = n_pix * C;
#define SUM_DISPATCH(V_)                                                                            \
  switch (n_src) {                                                                                  \
    case 2: launch_sum_n<T, V_, 2>(src, out, n, C, z, aff4, bn_sums, ws, st); break;                \
    case 3: launch_sum_n<T, V_, 3>(src, out, n, C, z, aff4, bn_sums, ws, st); break;                \
    case 4: launch_sum_n<T, V_, 4>(src, out, n, C, z, aff4, bn_sums, ws, st); break;                \
    default: launch_sum_n<T, V_, 5>(src, out, n, C, z, aff4, bn_sums, ws, st); break;               \
  }
  if (vec) {
    SUM_DISPATCH(NV)
  } else {
    SUM_DISPATCH(1)
  }
#undef SUM_DISPATCH
  return 0;
}

}  // namespace

extern "C" {

size_t mis_amsfn_workspace_bytes(int B, int HW, int C) {
  if (B < 1 || HW < 1 || C < 1) return 0;
  // the scalar layout has the most pixel rows
  int rows = dot_rows(B, HW, C, 1) > dot_rows(B, HW, C, 4) ? dot_rows(B, HW, C, 1) : dot_rows(B, HW, C, 4);
  if (stats_rows(B, HW, C, 1) > rows) rows = stats_rows(B, HW, C, 1);
  return align256((size_t)rows * B * 4 * C * sizeof(float)) + 4 * align256((size_t)B * 4 * C * sizeof(float));
}

int mis_amsfn_fuse_fwd(const void* const* z, const float* const* z_affine, int dtype, int B, int HW, int C,
                       const float* W1, const float* b1, const float* W2, const float* b2, float* s, float* h,
                       float* w, void* out, void* workspace, mis_stream_t stream) {
  MIS_REQUIRE(z && z_affine && W1 && b1 && W2 && b2 && s && h && w && out && workspace, "amsfn_fuse_fwd: null pointer");
  MIS_REQUIRE(B >= 1 && HW >= 1 && C >= 1 && C <= TPB, "amsfn_fuse_fwd: need 1 <= C <= %d (C=%d)", TPB, C);
  MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, "amsfn_fuse_fwd: bad dtype %d", dtype);
  Branches br;
  for (int k = 0; k < 4; k++) {
    MIS_REQUIRE(z[k], "amsfn_fuse_fwd: null branch %d", k);
    br.z[k] = z[k];
    br.aff[k] = z_affine[k];
  }
  if (dtype == MIS_BF16)
    return fuse_fwd_t<bf16>(br, B, HW, C, W1, b1, W2, b2, s, h, w, out, workspace, (hipStream_t)stream);
  return fuse_fwd_t<float>(br, B, HW, C, W1, b1, W2, b2, s, h, w, out, workspace, (hipStream_t)stream);
}

int mis_amsfn_fuse_bwd(const void* const* z, const float* const* z_affine, const void* dout, int dtype, int B, int HW,
                       int C, const float* W1, const float* W2, const float* s, const float* h, const float* w,
                       void* const* dact, float* dW1, float* db1, float* dW2, float* db2, void* workspace,
                       mis_stream_t stream) {
  MIS_REQUIRE(z && z_affine && dout && W1 && W2 && s && h && w && dact && dW1 && db1 && dW2 && db2 && workspace,
              "amsfn_fuse_bwd: null pointer");
  MIS_REQUIRE(B >= 1 && HW >= 1 && C >= 1 && C <= TPB, "amsfn_fuse_bwd: need 1 <= C <= %d (C=%d)", TPB, C);
  MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, "amsfn_fuse_bwd: bad dtype %d", dtype);
  Branches br;
  Grads gr;
  for (int k = 0; k < 4; k++) {
    MIS_REQUIRE(z[k] && dact[k], "amsfn_fuse_bwd: null branch %d", k);
    br.z[k] = z[k];
    br.aff[k] = z_affine[k];
    gr.d[k] = dact[k];
  }
  if (dtype == MIS_BF16)
    return fuse_bwd_t<bf16>(br, gr, dout, B, HW, C, W1, W2, s, h, w, dW1, db1, dW2, db2, workspace,
                            (hipStream_t)stream);
  return fuse_bwd_t<float>(br, gr, dout, B, HW, C, W1, W2, s, h, w, dW1, db1, dW2, db2, workspace,
                           (hipStream_t)stream);
}

size_t mis_amsfn_lazy_workspace_bytes(int B, int HW, int C) {
  if (B < 1 || HW < 1 || C < 1) return 0;
  const int rows = stats_rows(B, HW, C, 1) > stats_rows(B, HW, C, 4) ? stats_rows(B, HW, C, 1) : stats_rows(B, HW, C, 4);
  const size_t plane = (size_t)B * 4 * C * sizeof(float);
  return align256((size_t)rows * ST_N_BWD * plane) + align256(ST_N_BWD * plane) + 3 * align256(plane);
}

int mis_amsfn_fuse_bwd_lazy(const void* const* z, const float* const* z_affine4, const float* const* bn_gamma,
                            const float* const* bn_beta, const void* dout, int dtype, int B, int HW, int C,
                            const float* W1, const float* W2, const float* s, const float* h, const float* w,
                            float* bn_sums, float* dact_shift, float* dW1, float* db1, float* dW2, float* db2,
                            void* workspace, mis_stream_t stream) {
  MIS_REQUIRE(z && z_affine4 && bn_gamma && bn_beta && dout && W1 && W2 && s && h && w && bn_sums && dact_shift && dW1 &&
                  db1 && dW2 && db2 && workspace,
              "amsfn_fuse_bwd_lazy: null pointer");
  MIS_REQUIRE(B >= 1 && HW >= 1 && C >= 1 && C <= TPB, "amsfn_fuse_bwd_lazy: need 1 <= C <= %d (C=%d)", TPB, C);
  MIS_REQUIRE((int64_t)B * HW < ((int64_t)1 << 31), "amsfn_fuse_bwd_lazy: too many pixels");
  MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, "amsfn_fuse_bwd_lazy: bad dtype %d", dtype);
  Branches br;
  BranchBn bn;
  for (int k = 0; k < 4; k++) {
    MIS_REQUIRE(z[k] && z_affine4[k] && bn_gamma[k] && bn_beta[k], "amsfn_fuse_bwd_lazy: null branch %d", k);
    br.z[k] = z[k];
    br.aff[k] = z_affine4[k];
    bn.gamma[k] = bn_gamma[k];
    bn.beta[k] = bn_beta[k];
  }
  if (dtype == MIS_BF16)
    return fuse_bwd_lazy_t<bf16>(br, bn, dout, B, HW, C, W1, W2, s, h, w, bn_sums, dact_shift, dW1, db1, dW2, db2,
                                 workspace, (hipStream_t)stream);
  return fuse_bwd_lazy_t<float>(br, bn, dout, B, HW, C, W1, W2, s, h, w, bn_sums, dact_shift, dW1, db1, dW2, db2,
                                workspace, (hipStream_t)stream);
}

size_t mis_sum_gradients_workspace_bytes(int C) {
  return C >= 1 ? align256((size_t)SUM_MAX_ROWS * 2 * C * sizeof(float)) + 256 : 0;
}

int mis_sum_gradients(const void* const* grads, int n_grads, int dtype, int64_t n_pix, int C, const void* z,
                      const float* z_affine4, void* out, float* bn_sums, void* workspace, mis_stream_t stream) {
  MIS_REQUIRE(grads && out, "sum_gradients: null pointer");
  MIS_REQUIRE(n_grads >= 2 && n_grads <= SUM_MAX, "sum_gradients: 2..%d gradients (got %d)", SUM_MAX, n_grads);
  MIS_REQUIRE(n_pix >= 1 && C >= 1 && C <= 4096, "sum_gradients: bad shape");
  MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, "sum_gradients: bad dtype %d", dtype);
  MIS_REQUIRE(!bn_sums || (z && z_affine4 && workspace), "sum_gradients: the BatchNorm sums need z, its [4][C] affine and a workspace");
  MIS_REQUIRE(!bn_sums || (C % (dtype == MIS_BF16 ? 8 : 4) == 0 && C / (dtype == MIS_BF16 ? 8 : 4) <= TPB),
              "sum_gradients: the BatchNorm sums need whole 16-byte channel vectors (C=%d)", C);
  SumPtrs src;
  for (int j = 0; j < SUM_MAX; j++) {
    src.g[j] = j < n_grads ? grads[j] : nullptr;
    MIS_REQUIRE(j >= n_grads || grads[j], "sum_gradients: null gradient %d", j);
  }
  if (dtype == MIS_BF16) sum_gradients_t<bf16>(src, n_grads, out, n_pix, C, z, z_affine4, bn_sums, workspace, (hipStream_t)stream);
  else sum_gradients_t<float>(src, n_grads, out, n_pix, C, z, z_affine4, bn_sums, workspace, (hipStream_t)stream);
  MIS_CHECK_LAUNCH("sum_gradients_kernel");
  return MIS_OK;
}

int mis_dwconv_compose_fwd(const float* P, const float* D, const float* bd, const float* bp, int Cout, int Cin,
                           float* W, float* b, mis_stream_t stream) {
  MIS_REQUIRE(P && D && bd && bp && W && b, "dwconv_compose_fwd: null pointer");
  MIS_REQUIRE(Cout >= 1 && Cin >= 1 && (int64_t)Cout * Cin * 9 < (1 << 30), "dwconv_compose_fwd: bad shape");
  dw_compose_fwd_kernel<<<ew_grid((int64_t)Cout * Cin * 9 + Cout, TPB), TPB, 0, (hipStream_t)stream>>>(P, D, bd, bp, Cout,
                                                                                                     Cin, W, b);
  MIS_CHECK_LAUNCH("dw_compose_fwd_kernel");
  return MIS_OK;
}

int mis_dwconv_compose_multi(const mis_compose_job* jobs, int n_jobs, int total_blocks, mis_stream_t stream) {
  MIS_REQUIRE(jobs && n_jobs >= 1 && total_blocks >= n_jobs, "dwconv_compose_multi: bad arguments");
  dw_compose_multi_kernel<<<total_blocks, TPB, 0, (hipStream_t)stream>>>(jobs, n_jobs);
  MIS_CHECK_LAUNCH("dw_compose_multi_kernel");
  return MIS_OK;
}

int mis_dwconv_compose_bwd_multi(const mis_compose_bwd_job* jobs, int n_jobs, mis_stream_t stream) {
  MIS_REQUIRE(n_jobs >= 0 && (jobs || n_jobs == 0), "dwconv_compose_bwd_multi: null job array");
  for (int j = 0; j < n_jobs; j++) {
    const mis_compose_bwd_job& jb = jobs[j];
    MIS_REQUIRE(jb.dW && jb.dP && jb.Cout >= 1 && jb.Cin >= 1 && (int64_t)jb.Cout * jb.Cin * 9 < (1 << 30),
                "dwconv_compose_bwd_multi: bad job %d", j);
    MIS_REQUIRE(!jb.D || (jb.P && jb.bd && jb.db && jb.dD && jb.dbd), "dwconv_compose_bwd_multi: job %d lacks a DWConv operand", j);
  }
  for (int j0 = 0; j0 < n_jobs; j0 += COMPOSE_BWD_MAX) {
    ComposeBwdArgs a;
    a.n = n_jobs - j0 < COMPOSE_BWD_MAX ? n_jobs - j0 : COMPOSE_BWD_MAX;
    int total = 0;
    for (int j = 0; j < a.n; j++) {
      a.job[j] = jobs[j0 + j];
      a.first[j] = total;
      const int64_t work = a.job[j].D ? (int64_t)a.job[j].Cout * a.job[j].Cin + 8 + a.job[j].Cin * 80
                                      : (int64_t)a.job[j].Cout * a.job[j].Cin;
      int nb = (int)((work + TPB - 1) / TPB);
      total += nb > 64 ? 64 : nb;
    }
    a.first[a.n] = total;
    dw_compose_bwd_multi_kernel<<<total, TPB, 0, (hipStream_t)stream>>>(a);
    MIS_CHECK_LAUNCH("dw_compose_bwd_multi_kernel");
  }
  return MIS_OK;
}

int mis_dwconv_compose_bwd(const float* P, const float* D, const float* bd, const float* dW, const float* db, int Cout,
                           int Cin, float* dP, float* dD, float* dbd, mis_stream_t stream) {
  MIS_REQUIRE(P && D && bd && dW && db && dP && dD && dbd, "dwconv_compose_bwd: null pointer");
  MIS_REQUIRE(Cout >= 1 && Cin >= 1 && (int64_t)Cout * Cin * 9 < (1 << 30), "dwconv_compose_bwd: bad shape");
  dw_compose_bwd_kernel<<<ew_grid((int64_t)Cout * Cin + 8 + Cin * 80, TPB), TPB, 0, (hipStream_t)stream>>>(
      P, D, bd, dW, db, Cout, Cin, dP, dD, dbd);
  MIS_CHECK_LAUNCH("dw_compose_bwd_kernel");
  return MIS_OK;
}

}  // extern "C"
