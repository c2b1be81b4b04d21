// Does the memory-side cache reward a consumer that starts where its producer stopped?
// Kernel A writes Y (streaming copy of X), kernel B reads Y (sum) either in the same order or reversed
// (each workgroup walks its contiguous range from the end).  Prints B's time for both orders and sizes.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/zigzag.hip -o tools/micro/zigzag && tools/micro/zigzag
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void copy_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int64_t n_vec) {
  const int64_t per = (n_vec + gridDim.x - 1) / gridDim.x;
  const int64_t b0 = (int64_t)blockIdx.x * per, b1 = b0 + per < n_vec ? b0 + per : n_vec;
  for (int64_t i = b0 + threadIdx.x; i < b1; i += 256) y[i] = x[i];
}

template <bool REV>
__global__ __launch_bounds__(256) void read_kernel(const uint4* __restrict__ y, unsigned* __restrict__ out, int64_t n_vec) {
  const int64_t per = (n_vec + gridDim.x - 1) / gridDim.x;
  const int64_t b0 = (int64_t)blockIdx.x * per, b1 = b0 + per < n_vec ? b0 + per : n_vec;
  unsigned acc = 0;
  if (!REV) {
    for (int64_t i = b0 + threadIdx.x; i < b1; i += 256) { const uint4 v = y[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  } else {
    for (int64_t i = b1 - 256 + threadIdx.x; i >= b0; i -= 256) { const uint4 v = y[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// producer that also reads: y = a ^ b (two streams in, one out), as a BatchNorm-backward pass does
__global__ __launch_bounds__(256) void xor_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ y,
                                                  int64_t n_vec) {
  const int64_t per = (n_vec + gridDim.x - 1) / gridDim.x;
  const int64_t b0 = (int64_t)blockIdx.x * per, b1 = b0 + per < n_vec ? b0 + per : n_vec;
  for (int64_t i = b0 + threadIdx.x; i < b1; i += 256) {
    const uint4 u = a[i], v = b[i];
    y[i] = uint4{u.x ^ v.x, u.y ^ v.y, u.z ^ v.z, u.w ^ v.w};
  }
}

// grid-stride forms: the whole grid sweeps the tensor front to back (or back to front)
__global__ __launch_bounds__(256) void xor_sweep_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ y,
                                                        int64_t n_vec) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (int64_t)gridDim.x * 256) {
    const uint4 u = a[i], v = b[i];
    y[i] = uint4{u.x ^ v.x, u.y ^ v.y, u.z ^ v.z, u.w ^ v.w};
  }
}
template <bool REV>
__global__ __launch_bounds__(256) void read_sweep_kernel(const uint4* __restrict__ y, unsigned* __restrict__ out, int64_t n_vec) {
  const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
  const int64_t trips = i0 < n_vec ? (n_vec - 1 - i0) / stride + 1 : 0;
  unsigned acc = 0;
  int64_t i = REV ? i0 + (trips - 1) * stride : i0;
  for (int64_t t = 0; t < trips; t++, i += REV ? -stride : stride) { const uint4 v = y[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}

int main3() {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  unsigned* out;
  hipMalloc(&out, 64);
  const int64_t bytes = (int64_t)268 << 20, n_vec = bytes / 16;
  uint4 *a, *b, *y;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&y, bytes);
  hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
  for (int prod = 0; prod < 2; prod++)
    for (int cons = 0; cons < 2; cons++) {
      float t[2] = {0, 0};
      for (int rev = 0; rev < 2; rev++)
        for (int it = 0; it < 6; it++) {
          if (prod) xor_sweep_kernel<<<512, 256>>>(a, b, y, n_vec); else xor_kernel<<<512, 256>>>(a, b, y, n_vec);
          hipEventRecord(e0);
          if (cons) { if (rev) read_sweep_kernel<true><<<1024, 256>>>(y, out, n_vec); else read_sweep_kernel<false><<<1024, 256>>>(y, out, n_vec); }
          else { if (rev) read_kernel<true><<<1024, 256>>>(y, out, n_vec); else read_kernel<false><<<1024, 256>>>(y, out, n_vec); }
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          if (it >= 2) t[rev] += ms / 4;
        }
      printf("268 MB: producer %s, consumer %s: same order %6.1f us, reversed %6.1f us\n", prod ? "grid sweep" : "ranges   ",
             cons ? "grid sweep" : "ranges   ", t[0] * 1e3, t[1] * 1e3);
    }
  return 0;
}

int main2() {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  unsigned* out;
  hipMalloc(&out, 64);
  for (int64_t mb : {134, 268}) {
    const int64_t bytes = mb << 20, n_vec = bytes / 16;
    uint4 *a, *b, *y;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&y, bytes);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    float t[2] = {0, 0};
    for (int rev = 0; rev < 2; rev++)
      for (int it = 0; it < 6; it++) {
        xor_kernel<<<512, 256>>>(a, b, y, n_vec);
        hipEventRecord(e0);
        if (rev) read_kernel<true><<<1024, 256>>>(y, out, n_vec);
        else read_kernel<false><<<1024, 256>>>(y, out, n_vec);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (it >= 2) t[rev] += ms / 4;
      }
    printf("two reads + one write of %lld MB each, then the written tensor is read: same order %7.1f us, reversed %7.1f us\n",
           (long long)mb, t[0] * 1e3, t[1] * 1e3);
    hipFree(a); hipFree(b); hipFree(y);
  }
  return 0;
}

int main() {
  main3();
  main2();
  const int64_t sizes_mb[] = {64, 128, 192, 256, 320, 512, 1024};
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  unsigned* out;
  CHECK(hipMalloc(&out, 64));
  for (int64_t mb : sizes_mb) {
    const int64_t bytes = mb << 20, n_vec = bytes / 16;
    uint4 *x, *y;
    CHECK(hipMalloc(&x, bytes));
    CHECK(hipMalloc(&y, bytes));
    CHECK(hipMemset(x, 1, bytes));
    float t[2] = {0, 0};
    for (int rev = 0; rev < 2; rev++) {
      for (int it = 0; it < 6; it++) {
        copy_kernel<<<1024, 256>>>(x, y, n_vec);
        CHECK(hipEventRecord(e0));
        if (rev) read_kernel<true><<<1024, 256>>>(y, out, n_vec);
        else read_kernel<false><<<1024, 256>>>(y, out, n_vec);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 2) t[rev] += ms / 4;
      }
    }
    printf("%5lld MB written then read: same order %7.1f us (%5.2f TB/s)   reversed %7.1f us (%5.2f TB/s)\n", (long long)mb,
           t[0] * 1e3, bytes / (t[0] * 1e-3) / 1e12, t[1] * 1e3, bytes / (t[1] * 1e-3) / 1e12);
    CHECK(hipFree(x));
    CHECK(hipFree(y));
  }
  return 0;
}
