// Streaming-rate sweep on one GPU: what launch shape and cache policy gives the highest rate for
// the access patterns of the elementwise kernels here (one read + one write; two reads + one write).
// Build: hipcc -O3 --offload-arch=gfx950 stream_sweep.hip -o stream_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float v4 __attribute__((ext_vector_type(4)));

template <int NT>
__device__ inline v4 ld(const v4* p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}
template <int NT>
__device__ inline void st(v4* p, v4 v) {
  if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

// MODE 0: grid-stride, consecutive workgroups on consecutive 4 KiB*U pieces.
// MODE 1: every workgroup owns one contiguous range.
template <int U, int NTL, int NTS, int MODE, int READS>
__global__ __launch_bounds__(256) void stream_kernel(const v4* __restrict__ a, const v4* __restrict__ b,
                                                     v4* __restrict__ c, long n) {
  const long T = blockDim.x;
  if (MODE == 0) {
    const long stride = (long)gridDim.x * T * U;
    for (long base = (long)blockIdx.x * T * U + threadIdx.x; base < n; base += stride) {
      v4 va[U], vb[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        long i = base + u * T;
        if (i < n) { va[u] = ld<NTL>(a + i); if (READS == 2) vb[u] = ld<NTL>(b + i); }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        long i = base + u * T;
        if (i < n) st<NTS>(c + i, READS == 2 ? va[u] * vb[u] + va[u] : va[u]);
      }
    }
  } else {
    const long per = ((n + gridDim.x - 1) / gridDim.x + T * U - 1) / (T * U) * (T * U);
    const long lo = per * blockIdx.x, hi = (lo + per < n) ? lo + per : n;
    for (long base = lo + threadIdx.x; base < hi; base += T * U) {
      v4 va[U], vb[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        long i = base + u * T;
        if (i < hi) { va[u] = ld<NTL>(a + i); if (READS == 2) vb[u] = ld<NTL>(b + i); }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        long i = base + u * T;
        if (i < hi) st<NTS>(c + i, READS == 2 ? va[u] * vb[u] + va[u] : va[u]);
      }
    }
  }
}

struct Cfg { const char* name; void (*fn)(const v4*, const v4*, v4*, long); int reads; };

template <int U, int NTL, int NTS, int MODE, int READS>
static void run(const char* tag, const v4* a, const v4* b, v4* c, long n, const std::vector<int>& grids) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int g : grids) {
    for (int w = 0; w < 3; w++) stream_kernel<U, NTL, NTS, MODE, READS><<<g, 256>>>(a, b, c, n);
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int r = 0; r < reps; r++) stream_kernel<U, NTL, NTS, MODE, READS><<<g, 256>>>(a, b, c, n);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1000.0 / reps;
    double bytes = (double)n * 16 * (READS + 1);
    printf("%-28s U=%d ntl=%d nts=%d mode=%d grid=%5d  %8.1f us  %6.2f TB/s\n", tag, U, NTL, NTS, MODE, g, us, bytes / us * 1e-6);
  }
  fflush(stdout);
}

int main(int argc, char** argv) {
  const long bytes = argc > 1 ? atol(argv[1]) : 268435456L;   // one 16 x 512^2 x 32 bf16 tensor
  const long n = bytes / 16;
  v4 *a, *b, *c;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 1, bytes)); CK(hipMemset(c, 0, bytes));
  std::vector<int> grids = {256, 512, 1024, 2048, 4096, 8192, 16384, 65536};
  printf("tensor bytes %ld\n", bytes);
  // hipMemcpyAsync D2D for reference
  {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; w++) CK(hipMemcpyAsync(c, a, bytes, hipMemcpyDeviceToDevice, 0));
    CK(hipEventRecord(e0));
    for (int r = 0; r < 20; r++) CK(hipMemcpyAsync(c, a, bytes, hipMemcpyDeviceToDevice, 0));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("hipMemcpyAsync D2D: %.1f us  %.2f TB/s\n", ms * 50.0, 2.0 * bytes / (ms * 50.0) * 1e-6);
  }
  run<1, 0, 0, 0, 1>("copy", a, b, c, n, grids);
  run<4, 0, 0, 0, 1>("copy", a, b, c, n, grids);
  run<8, 0, 0, 0, 1>("copy", a, b, c, n, grids);
  run<4, 1, 1, 0, 1>("copy", a, b, c, n, grids);
  run<4, 0, 1, 0, 1>("copy", a, b, c, n, grids);
  run<4, 1, 0, 0, 1>("copy", a, b, c, n, grids);
  run<4, 0, 0, 1, 1>("copy", a, b, c, n, grids);
  run<4, 1, 1, 1, 1>("copy", a, b, c, n, grids);
  run<2, 0, 0, 0, 2>("2read+1write", a, b, c, n, grids);
  run<4, 0, 0, 0, 2>("2read+1write", a, b, c, n, grids);
  run<4, 1, 1, 0, 2>("2read+1write", a, b, c, n, grids);
  run<4, 0, 1, 0, 2>("2read+1write", a, b, c, n, grids);
  run<4, 0, 0, 1, 2>("2read+1write", a, b, c, n, grids);
  run<4, 1, 1, 1, 2>("2read+1write", a, b, c, n, grids);
  return 0;
}
