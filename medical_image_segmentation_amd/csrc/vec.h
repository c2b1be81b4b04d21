// 16-byte channel vectors for channels-last tensors: 8 x bf16 or 4 x float per lane.
#pragma once
#include "common.h"

template <typename T> struct VecOf;
template <> struct VecOf<float> { static constexpr int N = 4; };
template <> struct VecOf<bf16> { static constexpr int N = 8; };

// Loads N consecutive elements (N*sizeof(T) == 16 -> one dwordx4) into floats.
template <typename T, int N>
__device__ __forceinline__ void vload(const T* __restrict__ p, float (&v)[N]) {
  if constexpr (N * sizeof(T) == 16) {
    union { uint4 u; T t[N]; } r;
    r.u = *reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = to_f32(r.t[i]);
  } else if constexpr (N * sizeof(T) == 8) {
    union { uint2 u; T t[N]; } r;
    r.u = *reinterpret_cast<const uint2*>(p);
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = to_f32(r.t[i]);
  } else {
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = to_f32(p[i]);
  }
}

template <typename T, int N>
__device__ __forceinline__ void vstore(T* __restrict__ p, const float (&v)[N]) {
  if constexpr (N * sizeof(T) == 16) {
    union { uint4 u; T t[N]; } r;
#pragma unroll
    for (int i = 0; i < N; i++) r.t[i] = from_f32<T>(v[i]);
    *reinterpret_cast<uint4*>(p) = r.u;
  } else if constexpr (N * sizeof(T) == 8) {
    union { uint2 u; T t[N]; } r;
#pragma unroll
    for (int i = 0; i < N; i++) r.t[i] = from_f32<T>(v[i]);
    *reinterpret_cast<uint2*>(p) = r.u;
  } else {
#pragma unroll
    for (int i = 0; i < N; i++) p[i] = from_f32<T>(v[i]);
  }
}

// Lazy activations: a tensor may be stored as the pre-BatchNorm conv output z together with a
// per-channel affine (scale, shift); its value is relu(z*scale + shift), applied by whoever
// loads it.  `aff` is a [2][C] table (scale then shift), normally staged in LDS; null = plain.
template <int V>
__device__ __forceinline__ void affine_relu(float (&v)[V], const float* aff, int C, int c0) {
  if (aff) {
#pragma unroll
    for (int k = 0; k < V; k++) v[k] = fmaxf(v[k] * aff[c0 + k] + aff[C + c0 + k], 0.f);
  }
}

// Same, rounded to the storage type T: the convolutions use the activation as a T operand on
// every path (matrix-core and vector-ALU), so they all see identical values.
template <typename T, int V>
__device__ __forceinline__ void affine_relu_rounded(float (&v)[V], const float* aff, int C, int c0) {
  if (aff) {
#pragma unroll
    for (int k = 0; k < V; k++) v[k] = to_f32(from_f32<T>(fmaxf(v[k] * aff[c0 + k] + aff[C + c0 + k], 0.f)));
  }
}

// Picks the widest channel vector (full 16 B, else scalar) that divides C and keeps alignment.
template <typename T>
static inline int pick_vec(int C, const void* a, const void* b = nullptr, const void* c = nullptr,
                           const void* d = nullptr) {
  const int n = VecOf<T>::N;
  const uintptr_t m = (uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d;
  return (C % n == 0 && (m % 16) == 0) ? n : 1;
}

static inline int ew_grid(int64_t work_items, int threads) {
  int64_t g = (work_items + threads - 1) / threads;
  if (g > 256 * 16) g = 256 * 16;   // grid-stride beyond ~16 blocks per CU (1024..8192 measure the same)
  if (g < 1) g = 1;
  return (int)g;
}
