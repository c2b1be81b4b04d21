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

// Cross-lane sums on the DPP path (no LDS round trips: a `__shfl_xor` is a ds_bpermute plus a wait
// for it, ~100 cycles each, which dominates epilogues that reduce dozens of values).
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over each row of 16 lanes; every lane of the row ends with the row's total
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_move<0xB1>(v);      // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);      // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);     // row_half_mirror: quad q with quad 1-q of its 8 lanes
  v += dpp_move<0x140>(v);     // row_mirror: the two halves of the row
  return v;
}
// 64-lane sum: every lane ends with the total (the four row totals are read as scalars).
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return (r0 + r1) + (r2 + r3);
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
