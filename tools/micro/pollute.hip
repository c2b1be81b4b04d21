// Debugging aid: fills the whole LDS of every CU with a bit pattern (and leaves it behind), so that a
// kernel launched next that reads LDS it never wrote sees that pattern instead of its own leftovers.
//   hipcc --offload-arch=gfx950 -shared -fPIC -O2 tools/micro/pollute.hip -o tools/micro/libpollute.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void pollute_lds_kernel(uint32_t pattern, int words, uint32_t* sink) {
  extern __shared__ uint32_t lds[];
  for (int i = threadIdx.x; i < words; i += 256) lds[i] = pattern;
  __syncthreads();
  // keep the stores alive
  if (sink && lds[(threadIdx.x * 97) % words] != pattern) sink[0] = 1;
}

extern "C" int pollute_lds(uint32_t pattern, void* stream) {
  static bool init = false;
  const int bytes = 160 * 1024;
  if (!init) {
    if (hipFuncSetAttribute((const void*)pollute_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
      return 1;
    init = true;
  }
  pollute_lds_kernel<<<512, 256, bytes, (hipStream_t)stream>>>(pattern, bytes / 4, nullptr);
  return (int)hipGetLastError();
}
