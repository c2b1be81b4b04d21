// The pieces of RDAM_UNet (model/rdam_unet.py:7-115) that are not convolutions, on channels-last
// tensors, 16 bytes per lane:
//
//   * ResDConv's tail (model/utils/modules.py:447-453): relu(bn(conv3(x)) + bn(conv1x1(x))) from the
//     two conv outputs and their BatchNorm affines in one pass, and the ReLU mask of its backward;
//   * MDAM (model/utils/attention.py:171-190): out = x * sigmoid(w[b][c]).  Of the two terms of w,
//     softmax(avgpool(GroupNorm(x * gates))) does not depend on x at all — every channel is its own
//     group, so the average of the normalised map is the GroupNorm bias — which leaves the
//     squeeze term fc(avgpool(conv3x3(x))).  The average of a zero-padded 3x3 conv needs no conv:
//     per (sample, channel) it is a combination of the nine "shifted-window" sums of x, i.e. of the
//     total, the first/last row and column sums and the four corners.  So MDAM is one reduction
//     pass, a few hundred scalars of host-side arithmetic, and one scaling pass.
//
// Roofline: HBM; algorithmic bytes = elements read + written, times sizeof(T).
#include "common.h"
#include "vec.h"

namespace {

constexpr int RD_THREADS = 256;
constexpr int RD_MAX_BLOCKS = 64;        // workgroups per sample in the per-sample reductions

// relu(za*sa + ha + zb*sb + hb); aff: [2][C] (scale, shift) each.  A thread keeps one channel
// vector (the grid is sized for that), so its eight constants stay in registers.
template <typename T, int V>
__global__ __launch_bounds__(RD_THREADS) void add_bn_relu_kernel(const T* __restrict__ za, const T* __restrict__ zb,
                                                                  T* __restrict__ out, int64_t n_vec, int C,
                                                                  const float* __restrict__ affa,
                                                                  const float* __restrict__ affb) {
  const int CV = C / V;
  const int64_t i0 = (int64_t)blockIdx.x * RD_THREADS + threadIdx.x;
  const int c0 = (int)(i0 % CV) * V;
  float sa[V], ha[V], sb[V], hb[V];
#pragma unroll
  for (int k = 0; k < V; k++) {
    sa[k] = affa[c0 + k];
    ha[k] = affa[C + c0 + k];
    sb[k] = affb[c0 + k];
    hb[k] = affb[C + c0 + k];
  }
  for (int64_t i = i0; i < n_vec; i += (int64_t)gridDim.x * RD_THREADS) {
    float a[V], b[V];
    vload<T, V>(za + i * V, a);
    vload<T, V>(zb + i * V, b);
#pragma unroll
    for (int k = 0; k < V; k++) a[k] = fmaxf((a[k] * sa[k] + ha[k]) + (b[k] * sb[k] + hb[k]), 0.f);
    vstore<T, V>(out + i * V, a);
  }
}

// g = dout where out > 0, else 0 (the ReLU backward, from the stored output).
template <typename T, int V>
__global__ __launch_bounds__(RD_THREADS) void relu_mask_kernel(const T* __restrict__ out, const T* __restrict__ dout,
                                                                T* __restrict__ g, int64_t n_vec) {
  for (int64_t i = (int64_t)blockIdx.x * RD_THREADS + threadIdx.x; i < n_vec; i += (int64_t)gridDim.x * RD_THREADS) {
    float o[V], d[V];
    vload<T, V>(out + i * V, o);
    vload<T, V>(dout + i * V, d);
#pragma unroll
    for (int k = 0; k < V; k++) d[k] = o[k] > 0.f ? d[k] : 0.f;
    vstore<T, V>(g + i * V, d);
  }
}

// Sum over the PR pixel rows of a block of one per-thread channel vector, in a fixed order:
// row[c] = sum over threads holding channel vector c / V.  `red`: [RD_THREADS][V] floats of LDS.
template <int V>
__device__ __forceinline__ void block_channel_sum(const float (&acc)[V], float* red, int CV, int C, float* row) {
  __syncthreads();
#pragma unroll
  for (int k = 0; k < V; k++) red[threadIdx.x * V + k] = acc[k];
  __syncthreads();
  for (int j = threadIdx.x; j < C; j += RD_THREADS) {
    const int cv = j / V, k = j % V;
    float s = 0.f;
    for (int t = cv; t < RD_THREADS; t += CV) s += red[t * V + k];
    row[j] = s;
  }
}

// Per (sample, channel): T = total, R0 / RL = first / last row, C0 / CL = first / last column,
// and the four corners of x — what the nine shifted-window sums are made of.  Grid (nblk, B);
// partial [B][nblk][9][C].  CV must divide RD_THREADS' pixel rows evenly enough: CV <= RD_THREADS.
template <typename T, int V>
__global__ __launch_bounds__(RD_THREADS) void mdam_sums_kernel(const T* __restrict__ x, int H, int W, int C,
                                                                int64_t pix_per_blk, float* __restrict__ partial) {
  extern __shared__ float red[];
  const int CV = C / V, PR = RD_THREADS / CV;
  const int tid = threadIdx.x, cv = tid % CV, pr = tid / CV;
  const int b = blockIdx.y;
  const int64_t HW = (int64_t)H * W;
  const T* xb = x + (int64_t)b * HW * C;
  float acc[9][V];
#pragma unroll
  for (int q = 0; q < 9; q++)
#pragma unroll
    for (int k = 0; k < V; k++) acc[q][k] = 0.f;
  const int64_t p0 = (int64_t)blockIdx.x * pix_per_blk;
  int64_t p1 = p0 + pix_per_blk;
  if (p1 > HW) p1 = HW;
  if (pr < PR) {
    for (int64_t p = p0 + pr; p < p1; p += PR) {
      float v[V];
      vload<T, V>(xb + p * C + cv * V, v);
      const int py = (int)(p / W), px = (int)(p % W);
      const bool r0 = py == 0, rl = py == H - 1, c0 = px == 0, cl = px == W - 1;
#pragma unroll
      for (int k = 0; k < V; k++) acc[0][k] += v[k];
      if (r0 | rl | c0 | cl) {
#pragma unroll
        for (int k = 0; k < V; k++) {
          acc[1][k] += r0 ? v[k] : 0.f;
          acc[2][k] += rl ? v[k] : 0.f;
          acc[3][k] += c0 ? v[k] : 0.f;
          acc[4][k] += cl ? v[k] : 0.f;
          acc[5][k] += (r0 & c0) ? v[k] : 0.f;
          acc[6][k] += (r0 & cl) ? v[k] : 0.f;
          acc[7][k] += (rl & c0) ? v[k] : 0.f;
          acc[8][k] += (rl & cl) ? v[k] : 0.f;
        }
      }
    }
  }
  float* out = partial + ((int64_t)b * gridDim.x + blockIdx.x) * 9 * C;
#pragma unroll
  for (int q = 0; q < 9; q++) block_channel_sum<V>(acc[q], red, CV, C, out + q * C);
}

// S[b][c][ky*3+kx] = sum over the pixels a zero-padded 3x3 tap (ky, kx) reads = rows R(ky) x
// columns C(kx) of x, with R(0) = all but the last row, R(1) = all, R(2) = all but the first
// (an output pixel q reads input q + (ky-1, kx-1)).  Inclusion-exclusion on the nine raw sums.
__global__ void mdam_sums_finalize_kernel(const float* __restrict__ partial, int nblk, int B, int C,
                                          float* __restrict__ S) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i % C;
  // (row k outermost: the nine loads of a row are independent, so they are in flight together; the sums are
  // still added in row order)
  double r[9];
#pragma unroll
  for (int q = 0; q < 9; q++) r[q] = 0.0;
#pragma unroll 4
  for (int k = 0; k < nblk; k++) {
    const float* row = partial + ((int64_t)b * nblk + k) * 9 * C + c;
#pragma unroll
    for (int q = 0; q < 9; q++) r[q] += (double)row[q * C];
  }
  const double T = r[0], R0 = r[1], RL = r[2], C0 = r[3], CL = r[4], k00 = r[5], k0L = r[6], kL0 = r[7], kLL = r[8];
  // excluded row / column per tap index: 0 -> the LAST one, 1 -> none, 2 -> the FIRST one
  const double rowx[3] = {RL, 0.0, R0}, colx[3] = {CL, 0.0, C0};
  const double corner[3][3] = {{kLL, 0.0, kL0}, {0.0, 0.0, 0.0}, {k0L, 0.0, k00}};
  for (int ky = 0; ky < 3; ky++)
    for (int kx = 0; kx < 3; kx++)
      S[(int64_t)i * 9 + ky * 3 + kx] = (float)(T - rowx[ky] - colx[kx] + corner[ky][kx]);
}

// Gradient of the nine sums with respect to x: dx[b][p][c] = sum of dS[b][c][tap] over the taps
// whose window contains p (all nine in the interior).
template <typename T, int V>
__global__ __launch_bounds__(RD_THREADS) void mdam_sums_bwd_kernel(const float* __restrict__ dS, int H, int W, int C,
                                                                    T* __restrict__ dx) {
  const int CV = C / V;
  const int b = blockIdx.y;
  const int64_t n_vec = (int64_t)H * W * CV;
  const int64_t i0 = (int64_t)blockIdx.x * RD_THREADS + threadIdx.x;
  const int c0 = (int)(i0 % CV) * V;
  float g[9][V];
#pragma unroll
  for (int k = 0; k < V; k++)
#pragma unroll
    for (int q = 0; q < 9; q++) g[q][k] = dS[((int64_t)b * C + c0 + k) * 9 + q];
  T* dxb = dx + (int64_t)b * n_vec * V;
  for (int64_t i = i0; i < n_vec; i += (int64_t)gridDim.x * RD_THREADS) {
    const int64_t p = i / CV;
    const int py = (int)(p / W), px = (int)(p % W);
    // tap (ky, kx) reads pixel p for the output pixel p - (ky-1, kx-1), which must lie in the image
    const bool ry[3] = {py + 1 < H, true, py >= 1}, rx[3] = {px + 1 < W, true, px >= 1};
    float o[V];
#pragma unroll
    for (int k = 0; k < V; k++) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 9; q++) s += (ry[q / 3] && rx[q % 3]) ? g[q][k] : 0.f;
      o[k] = s;
    }
    vstore<T, V>(dxb + i * V, o);
  }
}

// out[b][p][c] = x[b][p][c] * s[b][c]
template <typename T, int V>
__global__ __launch_bounds__(RD_THREADS) void channel_scale_kernel(const T* __restrict__ x, const float* __restrict__ s,
                                                                    T* __restrict__ out, int64_t HW, int C) {
  const int CV = C / V;
  const int b = blockIdx.y;
  const int64_t n_vec = HW * CV;
  const int64_t i0 = (int64_t)blockIdx.x * RD_THREADS + threadIdx.x;
  const int c0 = (int)(i0 % CV) * V;
  float sv[V];
#pragma unroll
  for (int k = 0; k < V; k++) sv[k] = s[(int64_t)b * C + c0 + k];
  const T* xb = x + (int64_t)b * n_vec * V;
  T* ob = out + (int64_t)b * n_vec * V;
  for (int64_t i = i0; i < n_vec; i += (int64_t)gridDim.x * RD_THREADS) {
    float v[V];
    vload<T, V>(xb + i * V, v);
#pragma unroll
    for (int k = 0; k < V; k++) v[k] *= sv[k];
    vstore<T, V>(ob + i * V, v);
  }
}

// dot[b][c] = sum_p x[b][p][c] * dout[b][p][c]: grid (nblk, B), partial [B][nblk][C]
template <typename T, int V>
__global__ __launch_bounds__(RD_THREADS) void channel_dot_kernel(const T* __restrict__ x, const T* __restrict__ d,
                                                                  int64_t HW, int C, int64_t pix_per_blk,
                                                                  float* __restrict__ partial) {
  extern __shared__ float red[];
  const int CV = C / V, PR = RD_THREADS / CV;
  const int tid = threadIdx.x, cv = tid % CV, pr = tid / CV;
  const int b = blockIdx.y;
  const T* xb = x + (int64_t)b * HW * C;
  const T* db = d + (int64_t)b * HW * C;
  float acc[V];
#pragma unroll
  for (int k = 0; k < V; k++) acc[k] = 0.f;
  const int64_t p0 = (int64_t)blockIdx.x * pix_per_blk;
  int64_t p1 = p0 + pix_per_blk;
  if (p1 > HW) p1 = HW;
  if (pr < PR) {
#pragma unroll 4
    for (int64_t p = p0 + pr; p < p1; p += PR) {
      float a[V], g[V];
      vload<T, V>(xb + p * C + cv * V, a);
      vload<T, V>(db + p * C + cv * V, g);
#pragma unroll
      for (int k = 0; k < V; k++) acc[k] += a[k] * g[k];
    }
  }
  block_channel_sum<V>(acc, red, CV, C, partial + ((int64_t)b * gridDim.x + blockIdx.x) * C);
}

__global__ void rows_sum_kernel(const float* __restrict__ partial, int nblk, int n_out, int C, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;      // (b, c)
  if (i >= n_out) return;
  const int b = i / C, c = i % C;
  // four interleaved partial sums (rows k = j mod 4), combined in a fixed order: four loads in flight
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int k = 0;
  for (; k + 3 < nblk; k += 4) {
    const float* row = partial + ((int64_t)b * nblk + k) * C + c;
    s0 += (double)row[0];
    s1 += (double)row[C];
    s2 += (double)row[2 * (int64_t)C];
    s3 += (double)row[3 * (int64_t)C];
  }
  for (; k < nblk; k++) s0 += (double)partial[((int64_t)b * nblk + k) * C + c];
  out[i] = (float)((s0 + s1) + (s2 + s3));
}

inline int rd_blocks(int64_t HW, int PR) {
  int64_t want = (HW + (int64_t)PR * 16 - 1) / ((int64_t)PR * 16);          // >= 16 pixels per thread
  if (want > RD_MAX_BLOCKS) want = RD_MAX_BLOCKS;
  return want < 1 ? 1 : (int)want;
}

inline int ew_grid_fixed(int64_t n_vec, int CV, int cap) {
  // grid*256 a multiple of CV, so that a grid-stride thread keeps its channel vector
  int g = 1;
  while (((int64_t)g * RD_THREADS) % CV != 0) g++;
  int64_t want = (n_vec + RD_THREADS - 1) / RD_THREADS;
  if (want > cap) want = cap;
  int grid = (int)(want / g) * g;
  return grid < g ? g : grid;
}

template <typename T>
bool vec_ok(int C, const void* a, const void* b = nullptr, const void* c = nullptr) {
  return pick_vec<T>(C, a, b, c) == VecOf<T>::N && C / VecOf<T>::N <= RD_THREADS;
}

#define RD_DISPATCH(dtype, CALL_F32, CALL_BF16, what)       \
  if (dtype == MIS_F32) { CALL_F32; }                       \
  else if (dtype == MIS_BF16) { CALL_BF16; }                \
  else MIS_REQUIRE(false, what ": bad dtype %d", dtype);

template <typename T>
int add_bn_relu_t(const void* za, const float* affa, const void* zb, const float* affb, int64_t n_pix, int C, void* out,
                  hipStream_t st) {
  constexpr int N = VecOf<T>::N;
  const int64_t n = n_pix * C;
  if (vec_ok<T>(C, za, zb, out)) {
    const int grid = ew_grid_fixed(n / N, C / N, 2048);
    add_bn_relu_kernel<T, N><<<grid, RD_THREADS, 0, st>>>((const T*)za, (const T*)zb, (T*)out, n / N, C, affa, affb);
  } else {
    if (C > RD_THREADS) return -1;
    const int grid = ew_grid_fixed(n, C, 2048);
    add_bn_relu_kernel<T, 1><<<grid, RD_THREADS, 0, st>>>((const T*)za, (const T*)zb, (T*)out, n, C, affa, affb);
  }
  return 0;
}

template <typename T>
int relu_mask_t(const void* out, const void* dout, int64_t n, int C, void* g, hipStream_t st) {
  constexpr int N = VecOf<T>::N;
  if (pick_vec<T>(C, out, dout, g) == N)
    relu_mask_kernel<T, N><<<ew_grid(n / N, RD_THREADS), RD_THREADS, 0, st>>>((const T*)out, (const T*)dout, (T*)g, n / N);
  else
    relu_mask_kernel<T, 1><<<ew_grid(n, RD_THREADS), RD_THREADS, 0, st>>>((const T*)out, (const T*)dout, (T*)g, n);
  return 0;
}

template <typename T>
int mdam_sums_t(const void* x, int B, int H, int W, int C, float* S, void* ws, hipStream_t st) {
  constexpr int N = VecOf<T>::N;
  if (!vec_ok<T>(C, x)) return -1;
  const int CV = C / N, PR = RD_THREADS / CV;
  const int64_t HW = (int64_t)H * W;
  const int nblk = rd_blocks(HW, PR);
  const int64_t ppb = (HW + nblk - 1) / nblk;
  mdam_sums_kernel<T, N><<<dim3(nblk, B), RD_THREADS, RD_THREADS * N * sizeof(float), st>>>((const T*)x, H, W, C, ppb,
                                                                                            (float*)ws);
  mdam_sums_finalize_kernel<<<cdiv((int64_t)B * C, 128), 128, 0, st>>>((const float*)ws, nblk, B, C, S);
  return 0;
}

template <typename T>
int mdam_sums_bwd_t(const float* dS, int B, int H, int W, int C, void* dx, hipStream_t st) {
  constexpr int N = VecOf<T>::N;
  if (!vec_ok<T>(C, dx)) return -1;
  const int grid = ew_grid_fixed((int64_t)H * W * (C / N), C / N, 512);
  mdam_sums_bwd_kernel<T, N><<<dim3(grid, B), RD_THREADS, 0, st>>>(dS, H, W, C, (T*)dx);
  return 0;
}

template <typename T>
int channel_scale_t(const void* x, const float* s, int B, int64_t HW, int C, void* out, hipStream_t st) {
  constexpr int N = VecOf<T>::N;
  if (!vec_ok<T>(C, x, out)) return -1;
  const int grid = ew_grid_fixed(HW * (C / N), C / N, 512);
  channel_scale_kernel<T, N><<<dim3(grid, B), RD_THREADS, 0, st>>>((const T*)x, s, (T*)out, HW, C);
  return 0;
}

template <typename T>
int channel_dot_t(const void* x, const void* d, int B, int64_t HW, int C, float* out, void* ws, hipStream_t st) {
  constexpr int N = VecOf<T>::N;
  if (!vec_ok<T>(C, x, d)) return -1;
  const int CV = C / N, PR = RD_THREADS / CV;
  const int nblk = rd_blocks(HW, PR);
  const int64_t ppb = (HW + nblk - 1) / nblk;
  channel_dot_kernel<T, N><<<dim3(nblk, B), RD_THREADS, RD_THREADS * N * sizeof(float), st>>>(
      (const T*)x, (const T*)d, HW, C, ppb, (float*)ws);
  rows_sum_kernel<<<cdiv((int64_t)B * C, 128), 128, 0, st>>>((const float*)ws, nblk, B * C, C, out);
  return 0;
}

}  // namespace

extern "C" {

int mis_add_bn_relu_fwd(const void* za, const float* za_affine, const void* zb, const float* zb_affine, int dtype,
                        int64_t n_pix, int C, void* out, mis_stream_t stream) {
  MIS_REQUIRE(za && za_affine && zb && zb_affine && out && n_pix >= 1 && C >= 1, "add_bn_relu_fwd: bad arguments");
  int rc = 0;
  RD_DISPATCH(dtype, rc = add_bn_relu_t<float>(za, za_affine, zb, zb_affine, n_pix, C, out, (hipStream_t)stream),
              rc = add_bn_relu_t<bf16>(za, za_affine, zb, zb_affine, n_pix, C, out, (hipStream_t)stream),
              "add_bn_relu_fwd")
  MIS_REQUIRE(rc == 0, "add_bn_relu_fwd: C=%d needs 16-byte channel vectors or C <= 256", C);
  MIS_CHECK_LAUNCH("add_bn_relu_kernel");
  return MIS_OK;
}

int mis_relu_mask_bwd(const void* out, const void* dout, int dtype, int64_t n_pix, int C, void* g, mis_stream_t stream) {
  MIS_REQUIRE(out && dout && g && n_pix >= 1 && C >= 1, "relu_mask_bwd: bad arguments");
  RD_DISPATCH(dtype, relu_mask_t<float>(out, dout, n_pix * C, C, g, (hipStream_t)stream),
              relu_mask_t<bf16>(out, dout, n_pix * C, C, g, (hipStream_t)stream), "relu_mask_bwd")
  MIS_CHECK_LAUNCH("relu_mask_kernel");
  return MIS_OK;
}

size_t mis_mdam_workspace_bytes(int B, int C) {
  if (B < 1 || C < 1) return 0;
  return align256((size_t)B * RD_MAX_BLOCKS * 9 * C * sizeof(float));
}

int mis_mdam_window_sums(const void* x, int dtype, int B, int H, int W, int C, float* S, void* workspace,
                         mis_stream_t stream) {
  MIS_REQUIRE(x && S && workspace && B >= 1 && H >= 1 && W >= 1 && C >= 1, "mdam_window_sums: bad arguments");
  int rc = 0;
  RD_DISPATCH(dtype, rc = mdam_sums_t<float>(x, B, H, W, C, S, workspace, (hipStream_t)stream),
              rc = mdam_sums_t<bf16>(x, B, H, W, C, S, workspace, (hipStream_t)stream), "mdam_window_sums")
  MIS_REQUIRE(rc == 0, "mdam_window_sums: C=%d must be a multiple of the 16-byte channel vector and at most %d vectors",
              C, RD_THREADS);
  MIS_CHECK_LAUNCH("mdam_sums_kernel");
  return MIS_OK;
}

int mis_mdam_window_sums_bwd(const float* dS, int dtype, int B, int H, int W, int C, void* dx, mis_stream_t stream) {
  MIS_REQUIRE(dS && dx && B >= 1 && H >= 1 && W >= 1 && C >= 1, "mdam_window_sums_bwd: bad arguments");
  int rc = 0;
  RD_DISPATCH(dtype, rc = mdam_sums_bwd_t<float>(dS, B, H, W, C, dx, (hipStream_t)stream),
              rc = mdam_sums_bwd_t<bf16>(dS, B, H, W, C, dx, (hipStream_t)stream), "mdam_window_sums_bwd")
  MIS_REQUIRE(rc == 0, "mdam_window_sums_bwd: unsupported channel count %d", C);
  MIS_CHECK_LAUNCH("mdam_sums_bwd_kernel");
  return MIS_OK;
}

int mis_channel_scale(const void* x, const float* s, int dtype, int B, int64_t HW, int C, void* out,
                      mis_stream_t stream) {
  MIS_REQUIRE(x && s && out && B >= 1 && HW >= 1 && C >= 1, "channel_scale: bad arguments");
  int rc = 0;
  RD_DISPATCH(dtype, rc = channel_scale_t<float>(x, s, B, HW, C, out, (hipStream_t)stream),
              rc = channel_scale_t<bf16>(x, s, B, HW, C, out, (hipStream_t)stream), "channel_scale")
  MIS_REQUIRE(rc == 0, "channel_scale: unsupported channel count %d", C);
  MIS_CHECK_LAUNCH("channel_scale_kernel");
  return MIS_OK;
}

int mis_channel_dot(const void* x, const void* d, int dtype, int B, int64_t HW, int C, float* out, void* workspace,
                    mis_stream_t stream) {
  MIS_REQUIRE(x && d && out && workspace && B >= 1 && HW >= 1 && C >= 1, "channel_dot: bad arguments");
  int rc = 0;
  RD_DISPATCH(dtype, rc = channel_dot_t<float>(x, d, B, HW, C, out, workspace, (hipStream_t)stream),
              rc = channel_dot_t<bf16>(x, d, B, HW, C, out, workspace, (hipStream_t)stream), "channel_dot")
  MIS_REQUIRE(rc == 0, "channel_dot: unsupported channel count %d", C);
  MIS_CHECK_LAUNCH("channel_dot_kernel");
  return MIS_OK;
}

}  // extern "C"
