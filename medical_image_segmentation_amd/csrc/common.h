// Shared device/host helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/misamd.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int WAVE = 64;

void mis_set_error(const char* fmt, ...);

#define MIS_REQUIRE(cond, ...)                       \
  do {                                               \
    if (!(cond)) {                                   \
      mis_set_error(__VA_ARGS__);                    \
      return MIS_ERR_ARG;                            \
    }                                                \
  } while (0)

#define MIS_CHECK_LAUNCH(what)                                                   \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      mis_set_error("%s: %s", what, hipGetErrorString(e__));                     \
      return MIS_ERR_LAUNCH;                                                     \
    }                                                                            \
  } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

template <typename T> struct DT;
template <> struct DT<float> { static constexpr int id = MIS_F32; };
template <> struct DT<bf16> { static constexpr int id = MIS_BF16; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }

// 64-lane butterfly sum: every lane ends with the total.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}

// Block sum of NV values per thread through LDS; result valid in thread 0..NV-1 style:
// returns in `out[NV]` on every thread of wave 0.  `red` needs NV * (blockDim/64) floats.
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; i++) {
    float s = wave_sum(v[i]);
    if (lane == 0) red[wid * NV + i] = s;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; i++) {
    float s = 0.f;
    for (int w = 0; w < nw; w++) s += red[w * NV + i];
    v[i] = s;
  }
  __syncthreads();
}
