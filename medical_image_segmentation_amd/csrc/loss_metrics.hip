// Fused soft-Dice sums + cross-entropy map + confusion counts: ONE read of the logits.
//
// Roofline: HBM-bound.  Algorithmic bytes per pixel = C*sizeof(T) (logits) + 8 (int64 target)
// [+4 when the CE map is written]; the backward re-reads both and writes C*sizeof(T).
//
// Work split: grid = (blocks per sample, B); a block strides over one sample's pixels so
// the confusion counts of a block all belong to one cm[b].  Float partial sums are
// written per block and reduced in a fixed order by a second tiny kernel (bitwise
// reproducible); integer counts are wave-ballot popcounts (C <= 5) or LDS atomics.
#include "common.h"
#include <atomic>

namespace {

constexpr int LM_THREADS = 256;
constexpr float CE_LABEL_SMOOTHING = 1e-7f;   // utils/loss_fn.py:199

template <typename T, int C, bool NHWC>
__device__ __forceinline__ void load_pixel(const T* __restrict__ base, int64_t pix, int64_t sc, int64_t sp,
                                           float (&x)[C]) {
  if constexpr (NHWC) {
    constexpr int BYTES = C * (int)sizeof(T);
    const T* p = base + pix * C;
    T tmp[C];
    if constexpr (BYTES % 16 == 0) {
#pragma unroll
      for (int i = 0; i < BYTES / 16; i++) ((uint4*)tmp)[i] = ((const uint4*)p)[i];
    } else if constexpr (BYTES % 8 == 0) {
#pragma unroll
      for (int i = 0; i < BYTES / 8; i++) ((uint2*)tmp)[i] = ((const uint2*)p)[i];
    } else if constexpr (BYTES % 4 == 0) {
#pragma unroll
      for (int i = 0; i < BYTES / 4; i++) ((uint32_t*)tmp)[i] = ((const uint32_t*)p)[i];
    } else {
#pragma unroll
      for (int i = 0; i < C; i++) tmp[i] = p[i];
    }
#pragma unroll
    for (int c = 0; c < C; c++) x[c] = to_f32(tmp[c]);
  } else {
#pragma unroll
    for (int c = 0; c < C; c++) x[c] = to_f32(base[c * sc + pix * sp]);
  }
}

template <typename T, int C, bool NHWC>
__device__ __forceinline__ void store_pixel(T* __restrict__ base, int64_t pix, int64_t sc, int64_t sp,
                                            const float (&x)[C]) {
  if constexpr (NHWC) {
    constexpr int BYTES = C * (int)sizeof(T);
    T tmp[C];
#pragma unroll
    for (int c = 0; c < C; c++) tmp[c] = from_f32<T>(x[c]);
    T* p = base + pix * C;
    if constexpr (BYTES % 16 == 0) {
#pragma unroll
      for (int i = 0; i < BYTES / 16; i++) ((uint4*)p)[i] = ((const uint4*)tmp)[i];
    } else if constexpr (BYTES % 8 == 0) {
#pragma unroll
      for (int i = 0; i < BYTES / 8; i++) ((uint2*)p)[i] = ((const uint2*)tmp)[i];
    } else if constexpr (BYTES % 4 == 0) {
#pragma unroll
      for (int i = 0; i < BYTES / 4; i++) ((uint32_t*)p)[i] = ((const uint32_t*)tmp)[i];
    } else {
#pragma unroll
      for (int c = 0; c < C; c++) p[c] = tmp[c];
    }
  } else {
#pragma unroll
    for (int c = 0; c < C; c++) base[c * sc + pix * sp] = from_f32<T>(x[c]);
  }
}

// softmax of one pixel; returns log-sum-exp and the first-max argmax (NaN counts as maximal).
template <int C>
__device__ __forceinline__ void softmax_pixel(const float (&x)[C], float (&p)[C], float& lse, int& amax) {
  float m = x[0];
  amax = 0;
#pragma unroll
  for (int c = 1; c < C; c++) {
    const bool take = (x[c] > m) || (x[c] != x[c] && m == m);
    m = take ? x[c] : m;
    amax = take ? c : amax;
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < C; c++) {
    p[c] = expf(x[c] - m);
    s += p[c];
  }
  const float inv = 1.0f / s;
#pragma unroll
  for (int c = 0; c < C; c++) p[c] *= inv;
  lse = m + logf(s);
}

template <typename T, int C, bool NHWC>
__global__ __launch_bounds__(LM_THREADS) void loss_metrics_kernel(
    const T* __restrict__ logits, int64_t sb, int64_t sc, int64_t sp, const int64_t* __restrict__ target, int HW,
    float* __restrict__ partials,           // [gridDim.y*gridDim.x][2C]
    float* __restrict__ ce_map,             // [B][HW] or null
    unsigned long long* __restrict__ cm,    // [B][C][C]
    int* __restrict__ status) {
  // Confusion counts, C <= 5: every thread keeps its C*C bins as 8-bit fields of NR registers
  // (one shift and NR selects per pixel) and the block adds them up through LDS every <= 252
  // pixels per thread (before a field can wrap) and at the end.
  constexpr bool PACKED = (C <= 5);
  constexpr int NB = C * C, NR = (NB + 3) / 4;
  __shared__ float red[2 * C * (LM_THREADS / 64)];
  __shared__ unsigned int hist[C * C];
  __shared__ unsigned int pkl[PACKED ? LM_THREADS * NR : 1];
  const int b = blockIdx.y, tid = threadIdx.x;
  const T* lg = logits + (int64_t)b * sb;
  const int64_t* tg = target + (int64_t)b * HW;
  for (int i = tid; i < C * C; i += LM_THREADS) hist[i] = 0;
  __syncthreads();

  float acc[2 * C];
#pragma unroll
  for (int i = 0; i < 2 * C; i++) acc[i] = 0.f;
  [[maybe_unused]] unsigned int pk[PACKED ? NR : 1];
#pragma unroll
  for (int k = 0; k < (PACKED ? NR : 1); k++) pk[k] = 0;
  [[maybe_unused]] auto flush_counts = [&]() {
#pragma unroll
    for (int k = 0; k < NR; k++) {
      pkl[tid * NR + k] = pk[k];
      pk[k] = 0;
    }
    __syncthreads();
    // thread -> (bin, one of 8 groups of 32 rows); the 8 partial sums of a bin meet in hist
    const int k = tid & 31, part = tid >> 5;
    if (k < NB) {
      unsigned v = 0;
      for (int r = part * 32; r < part * 32 + 32; r++) v += (pkl[r * NR + (k >> 2)] >> ((k & 3) * 8)) & 0xffu;
      if (v) atomicAdd(&hist[k], v);
    }
    __syncthreads();
  };
  int trips = 0;
  bool bad = false;

  // PPT pixels per thread and trip: all their loads are issued before the first use, so a lane has
  // PPT*(C*sizeof(T) + 8) bytes in flight instead of one pixel's (the pass is pure streaming and
  // latency-bound otherwise).  A thread's pixels are LM_THREADS apart: every load instruction of
  // a wave still covers consecutive pixels.
  constexpr int PPT = 4;
  for (int base = blockIdx.x * (LM_THREADS * PPT); base < HW; base += gridDim.x * (LM_THREADS * PPT)) {
    float x[PPT][C];
    int64_t t64[PPT];
    bool valid[PPT];
#pragma unroll
    for (int u = 0; u < PPT; u++) {
      const int pix = base + u * LM_THREADS + tid;
      valid[u] = pix < HW;
      const int pl = valid[u] ? pix : 0;                 // (a valid address: sample b has HW >= 1 pixels)
      load_pixel<T, C, NHWC>(lg, pl, sc, sp, x[u]);
      t64[u] = tg[pl];
    }
#pragma unroll
    for (int u = 0; u < PPT; u++) {
      const int pix = base + u * LM_THREADS + tid;
      int bin = -1;
      if (valid[u]) {
        float p[C], lse;
        int amax;
        softmax_pixel<C>(x[u], p, lse, amax);
        const bool t_ok = (t64[u] >= 0) && (t64[u] < C);
        bad |= !t_ok;
        const int t = t_ok ? (int)t64[u] : -1;
#pragma unroll
        for (int c = 0; c < C; c++) {
          acc[c] += (t == c) ? p[c] : 0.f;
          acc[C + c] += p[c];
        }
        if (t_ok) bin = t * C + amax;
        if (ce_map) {
          float ce = 0.f;
          if (t > 0) {
            float xt = 0.f, xs = 0.f;
#pragma unroll
            for (int c = 0; c < C; c++) {
              xt = (t == c) ? x[u][c] : xt;
              xs += x[u][c];
            }
            ce = (1.0f - CE_LABEL_SMOOTHING) * (lse - xt) + CE_LABEL_SMOOTHING * (lse - xs * (1.0f / C));
          }
          ce_map[(int64_t)b * HW + pix] = ce;
        }
      }
      if constexpr (PACKED) {
        const unsigned inc = bin >= 0 ? 1u << ((bin & 3) * 8) : 0u;
#pragma unroll
        for (int k = 0; k < NR; k++) pk[k] += ((bin >> 2) == k) ? inc : 0u;
      } else {
        if (bin >= 0) atomicAdd(&hist[bin], 1u);
      }
    }
    if constexpr (PACKED) {
      if (++trips == 63) {          // (uniform over the block: the loop bounds are)
        flush_counts();
        trips = 0;
      }
    }
  }
  if constexpr (PACKED) flush_counts();
  if (bad) atomicOr(status, 1);

  block_sum<2 * C>(acc, red);
  if (tid < 2 * C) {
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * C; i++) v = (tid == i) ? acc[i] : v;
    partials[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (2 * C) + tid] = v;
  }
  __syncthreads();
  for (int i = tid; i < C * C; i += LM_THREADS)
    if (hist[i]) atomicAdd(&cm[(int64_t)b * C * C + i], (unsigned long long)hist[i]);
}

__global__ void clear_counts_kernel(unsigned long long* __restrict__ cm, int n, int* __restrict__ status) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) cm[i] = 0ull;
  if (threadIdx.x == 0) *status = 0;
}

// Fixed-order reduction of the per-block partials; T_c comes from the integer counts.
// 1024 threads: value v = tid % VP (VP = 2C rounded up to a power of two) is summed by
// 1024/VP row lanes (double accumulators), then a tree over the lanes in LDS.
__global__ __launch_bounds__(1024) void loss_sums_finalize_kernel(const float* __restrict__ partials, int nblk, int C,
                                                                  const unsigned long long* __restrict__ cm, int B,
                                                                  float* __restrict__ sums) {
  __shared__ double red[1024];
  const int tid = threadIdx.x;
  int VP = 1;
  while (VP < 2 * C) VP <<= 1;
  const int v = tid % VP, r = tid / VP, RL = 1024 / VP;
  double s = 0.0;
  if (v < 2 * C)
    for (int i = r; i < nblk; i += RL) s += (double)partials[(int64_t)i * 2 * C + v];
  red[tid] = s;
  __syncthreads();
  for (int o = RL / 2; o > 0; o >>= 1) {
    if (r < o) red[tid] += red[tid + o * VP];
    __syncthreads();
  }
  if (tid < 2 * C) sums[tid] = (float)red[tid];
  if (tid >= 64 && tid < 64 + C) {
    const int c = tid - 64;
    unsigned long long t = 0;
    for (int b = 0; b < B; b++)
      for (int p = 0; p < C; p++) t += cm[((int64_t)b * C + c) * C + p];
    sums[2 * C + c] = (float)t;
  }
}

__global__ void dice_from_sums_kernel(const float* __restrict__ sums, int C, float smooth, int kind,
                                      float* __restrict__ losses) {
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int c = 1; c < C; c++) {
      const float I = sums[c], U = sums[C + c] + sums[2 * C + c];
      // Dice: 1 - 2I/(U+s) (loss_fn.py:81-82); IoU: 1 - I/(U-I+s) (loss_fn.py:116-117)
      const float l = (kind == MIS_LOSS_IOU) ? 1.0f - I / (U - I + smooth) : 1.0f - (2.0f * I) / (U + smooth);
      losses[c - 1] = l;
      tot += l;
    }
    losses[C - 1] = tot / (float)C;
  }
}

template <typename T, int C, bool NHWC>
__global__ __launch_bounds__(LM_THREADS) void loss_bwd_kernel(
    const T* __restrict__ logits, int64_t sb, int64_t sc, int64_t sp, const int64_t* __restrict__ target, int HW,
    const float* __restrict__ sums, const float* __restrict__ w, float smooth, int kind,
    const float* __restrict__ dce, T* __restrict__ dlogits) {
  __shared__ float cA[C], cB[C];
  const int b = blockIdx.y, tid = threadIdx.x;
  if (tid < C) {
    float a = 0.f, bb = 0.f;
    if (tid > 0) {
      const float k = w[tid - 1] + w[C - 1] / (float)C;
      // a = k * dloss/dI, bb = k * dloss/dP  (P enters through U = P + T only)
      if (kind == MIS_LOSS_IOU) {
        const float I = sums[tid], U = sums[C + tid] + sums[2 * C + tid], D = U - I + smooth;
        a = -k * (U + smooth) / (D * D);
        bb = k * I / (D * D);
      } else {
        const float I = sums[tid], U = sums[C + tid] + sums[2 * C + tid] + smooth;
        a = -2.0f * k / U;
        bb = 2.0f * k * I / (U * U);
      }
    }
    cA[tid] = a;
    cB[tid] = bb;
  }
  __syncthreads();
  float A[C], Bc[C];
#pragma unroll
  for (int c = 0; c < C; c++) {
    A[c] = cA[c];
    Bc[c] = cB[c];
  }
  const T* lg = logits + (int64_t)b * sb;
  T* dl = dlogits + (int64_t)b * sb;
  const int64_t* tg = target + (int64_t)b * HW;
  constexpr int PPT = 4;          // pixels per thread and trip, all loads issued before the first use
  for (int base = blockIdx.x * (LM_THREADS * PPT); base < HW; base += gridDim.x * (LM_THREADS * PPT)) {
    float xs[PPT][C], ds[PPT];
    int ts[PPT];
#pragma unroll
    for (int u = 0; u < PPT; u++) {
      const int pix = base + u * LM_THREADS + tid;
      const int pl = pix < HW ? pix : 0;
      load_pixel<T, C, NHWC>(lg, pl, sc, sp, xs[u]);
      ts[u] = (int)tg[pl];
      ds[u] = (dce && ts[u] > 0 && ts[u] < C) ? dce[(int64_t)b * HW + pl] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < PPT; u++) {
      const int pix = base + u * LM_THREADS + tid;
      if (pix >= HW) continue;
      float p[C], g[C], lse;
      int amax;
      const int t = ts[u];
      softmax_pixel<C>(xs[u], p, lse, amax);
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < C; c++) {
        g[c] = ((t == c) ? A[c] : 0.f) + Bc[c];
        dot += p[c] * g[c];
      }
      const float d = ds[u];
#pragma unroll
      for (int c = 0; c < C; c++) {
        float v = p[c] * (g[c] - dot);
        const float onehot = (t == c) ? 1.f : 0.f;
        v += d * ((1.0f - CE_LABEL_SMOOTHING) * (p[c] - onehot) + CE_LABEL_SMOOTHING * (p[c] - 1.0f / C));
        g[c] = v;
      }
      store_pixel<T, C, NHWC>(dl, pix, sc, sp, g);
    }
  }
}

inline int lm_blocks_per_sample(int B, int HW) {
  int want = cdiv(HW, LM_THREADS * 16);             // >= 16 pixels (four trips of 4) per thread where possible
  int cap = cdiv(1024, B);                           // ~1024 blocks (four per CU) over the chip
  int n = want < cap ? want : cap;
  return n < 1 ? 1 : n;
}

template <typename T, int C>
int launch_fwd(const void* logits, int64_t sb, int64_t sc, int64_t sp, const int64_t* target, int B, int HW,
               float* partials, float* ce_map, unsigned long long* cm, int* status, hipStream_t st) {
  dim3 grid(lm_blocks_per_sample(B, HW), B);
  const bool nhwc = (sc == 1 && sp == C && sb % C == 0 && ((uintptr_t)logits % 16) == 0);
  if (nhwc)
    loss_metrics_kernel<T, C, true><<<grid, LM_THREADS, 0, st>>>((const T*)logits, sb, sc, sp, target, HW,
                                                                 partials, ce_map, cm, status);
  else
    loss_metrics_kernel<T, C, false><<<grid, LM_THREADS, 0, st>>>((const T*)logits, sb, sc, sp, target, HW,
                                                                  partials, ce_map, cm, status);
  return 0;
}

inline int lm_bwd_blocks_per_sample(int B, int HW) {
  int want = cdiv(HW, LM_THREADS * 4);              // one trip of 4 pixels per thread
  int cap = cdiv(2048, B);
  int n = want < cap ? want : cap;
  return n < 1 ? 1 : n;
}

template <typename T, int C>
int launch_bwd(const void* logits, int64_t sb, int64_t sc, int64_t sp, const int64_t* target, int B, int HW,
               const float* sums, const float* w, float smooth, int kind, const float* dce, void* dlogits,
               hipStream_t st) {
  dim3 grid(lm_bwd_blocks_per_sample(B, HW), B);
  const bool nhwc = (sc == 1 && sp == C && sb % C == 0 && ((uintptr_t)logits % 16) == 0 &&
                     ((uintptr_t)dlogits % 16) == 0);
  if (nhwc)
    loss_bwd_kernel<T, C, true><<<grid, LM_THREADS, 0, st>>>((const T*)logits, sb, sc, sp, target, HW, sums, w,
                                                             smooth, kind, dce, (T*)dlogits);
  else
    loss_bwd_kernel<T, C, false><<<grid, LM_THREADS, 0, st>>>((const T*)logits, sb, sc, sp, target, HW, sums, w,
                                                              smooth, kind, dce, (T*)dlogits);
  return 0;
}

// Scratch of the entry point that allocates nothing (mis_loss_metrics_fwd_static): the per-workgroup
// partial sums and, when the caller wants no counts back, the count matrices.  One copy of this
// array exists on every GPU the library runs on, at its own address, for the life of the process.
// In host code the NAME below is only a handle for hipGetSymbolAddress: kernels and memsets get
// the per-device address kept in static_scratch_ptr.
constexpr size_t STATIC_SCRATCH_BYTES = MIS_LOSS_METRICS_STATIC_SCRATCH_BYTES;
constexpr int STATIC_SCRATCH_MAX_DEVICES = 64;
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];
static std::atomic<void*> static_scratch_ptr[STATIC_SCRATCH_MAX_DEVICES];

// Device address of `scratch` on the CURRENT device, resolved once per device.
static void* static_scratch(int* status_out) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= STATIC_SCRATCH_MAX_DEVICES) {
    *status_out = MIS_ERR_LAUNCH;
    mis_set_error("loss_metrics_fwd_static: cannot identify the current device (hipGetDevice gave %d)", dev);
    return nullptr;
  }
  void* p = static_scratch_ptr[dev].load(std::memory_order_acquire);
  if (p == nullptr) {
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || p == nullptr) {
      *status_out = MIS_ERR_LAUNCH;
      mis_set_error("loss_metrics_fwd_static: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
      return nullptr;
    }
    static_scratch_ptr[dev].store(p, std::memory_order_release);    // (two racing threads store the same address)
  }
  return p;
}

#define DISPATCH_C(FN, T, C, ...)                \
  switch (C) {                                   \
    case 2: FN<T, 2>(__VA_ARGS__); break;        \
    case 3: FN<T, 3>(__VA_ARGS__); break;        \
    case 4: FN<T, 4>(__VA_ARGS__); break;        \
    case 5: FN<T, 5>(__VA_ARGS__); break;        \
    case 6: FN<T, 6>(__VA_ARGS__); break;        \
    case 7: FN<T, 7>(__VA_ARGS__); break;        \
    case 8: FN<T, 8>(__VA_ARGS__); break;        \
    case 9: FN<T, 9>(__VA_ARGS__); break;        \
    case 10: FN<T, 10>(__VA_ARGS__); break;      \
    case 11: FN<T, 11>(__VA_ARGS__); break;      \
    case 12: FN<T, 12>(__VA_ARGS__); break;      \
    case 13: FN<T, 13>(__VA_ARGS__); break;      \
    case 14: FN<T, 14>(__VA_ARGS__); break;      \
    case 15: FN<T, 15>(__VA_ARGS__); break;      \
    case 16: FN<T, 16>(__VA_ARGS__); break;      \
  }

}  // namespace

extern "C" {

size_t mis_loss_metrics_workspace_bytes(int B, int C, int HW) {
  if (B < 1 || C < 2 || HW < 1) return 0;
  const size_t nblk = (size_t)lm_blocks_per_sample(B, HW) * B;
  return align256(nblk * 2 * C * sizeof(float)) + align256((size_t)B * C * C * sizeof(int64_t)) + 256;
}

int mis_loss_metrics_fwd(const void* logits, int dtype, int64_t sb, int64_t sc, int64_t sp, const int64_t* target,
                         int B, int C, int HW, float* sums, float* ce_map, int64_t* cm, int32_t* status,
                         void* workspace, mis_stream_t stream) {
  MIS_REQUIRE(logits && target && workspace && status, "loss_metrics_fwd: null pointer");
  MIS_REQUIRE(B >= 1 && HW >= 1 && C >= 2 && C <= 16, "loss_metrics_fwd: need B,HW >= 1 and 2 <= C <= 16 (C=%d)", C);
  MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, "loss_metrics_fwd: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  const int nblk = lm_blocks_per_sample(B, HW) * B;
  float* partials = (float*)workspace;
  unsigned long long* cmw = cm ? (unsigned long long*)cm
                               : (unsigned long long*)((char*)workspace + align256((size_t)nblk * 2 * C * sizeof(float)));
  // (a kernel, not hipMemsetAsync: captured into a HIP graph, the two memsets left a garbage pattern in the
  // upper halves of the counts on replays after the first, in some processes — and one launch is cheaper)
  clear_counts_kernel<<<1, 256, 0, st>>>(cmw, B * C * C, status);
  MIS_CHECK_LAUNCH("clear_counts_kernel");
  if (dtype == MIS_F32) {
    DISPATCH_C(launch_fwd, float, C, logits, sb, sc, sp, target, B, HW, partials, ce_map, cmw, status, st)
  } else {
    DISPATCH_C(launch_fwd, bf16, C, logits, sb, sc, sp, target, B, HW, partials, ce_map, cmw, status, st)
  }
  MIS_CHECK_LAUNCH("loss_metrics_kernel");
  if (sums) {
    loss_sums_finalize_kernel<<<1, 1024, 0, st>>>(partials, nblk, C, cmw, B, sums);
    MIS_CHECK_LAUNCH("loss_sums_finalize_kernel");
  }
  return MIS_OK;
}

int mis_loss_metrics_fwd_static(const void* logits, int dtype, int64_t sb, int64_t sc, int64_t sp,
                                const int64_t* target, int B, int C, int HW, float* sums, float* ce_map, int64_t* cm,
                                int32_t* status, mis_stream_t stream) {
  MIS_REQUIRE(B >= 1 && HW >= 1 && C >= 2 && C <= 16, "loss_metrics_fwd_static: need B,HW >= 1 and 2 <= C <= 16 (C=%d)", C);
  const size_t need = mis_loss_metrics_workspace_bytes(B, C, HW);
  MIS_REQUIRE(need <= STATIC_SCRATCH_BYTES,
              "loss_metrics_fwd_static: B=%d C=%d HW=%d needs %zu bytes of scratch, the built-in buffer holds %zu "
              "(use mis_loss_metrics_fwd with a caller-provided workspace)", B, C, HW, need, STATIC_SCRATCH_BYTES);
  int rc = MIS_OK;
  void* ws = static_scratch(&rc);
  if (ws == nullptr) return rc;
  // everything the kernels read from the scratch is written earlier in the same call (partial rows by
  // every workgroup; the count area is cleared at the start of mis_loss_metrics_fwd), so what the
  // previous call left behind is never seen
  return mis_loss_metrics_fwd(logits, dtype, sb, sc, sp, target, B, C, HW, sums, ce_map, cm, status, ws, stream);
}

int mis_dice_from_sums(const float* sums, int C, float smooth, int kind, float* losses, mis_stream_t stream) {
  MIS_REQUIRE(sums && losses && C >= 2 && C <= 16, "dice_from_sums: bad arguments");
  MIS_REQUIRE(kind == MIS_LOSS_DICE || kind == MIS_LOSS_IOU, "dice_from_sums: bad loss kind %d", kind);
  dice_from_sums_kernel<<<1, 64, 0, (hipStream_t)stream>>>(sums, C, smooth, kind, losses);
  MIS_CHECK_LAUNCH("dice_from_sums_kernel");
  return MIS_OK;
}

int mis_loss_bwd(const void* logits, int dtype, int64_t sb, int64_t sc, int64_t sp, const int64_t* target, int B,
                 int C, int HW, const float* sums, const float* w, float smooth, int kind, const float* dce,
                 void* dlogits, mis_stream_t stream) {
  MIS_REQUIRE(logits && target && sums && w && dlogits, "loss_bwd: null pointer");
  MIS_REQUIRE(kind == MIS_LOSS_DICE || kind == MIS_LOSS_IOU, "loss_bwd: bad loss kind %d", kind);
  MIS_REQUIRE(B >= 1 && HW >= 1 && C >= 2 && C <= 16, "loss_bwd: need B,HW >= 1 and 2 <= C <= 16 (C=%d)", C);
  MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, "loss_bwd: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIS_F32) {
    DISPATCH_C(launch_bwd, float, C, logits, sb, sc, sp, target, B, HW, sums, w, smooth, kind, dce, dlogits, st)
  } else {
    DISPATCH_C(launch_bwd, bf16, C, logits, sb, sc, sp, target, B, HW, sums, w, smooth, kind, dce, dlogits, st)
  }
  MIS_CHECK_LAUNCH("loss_bwd_kernel");
  return MIS_OK;
}

}  // extern "C"
