// microbenchmark: random 4-byte scatter / gather rates as a function of the address window
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
// element a writes to window (a / per_window) at a random offset inside it
__global__ void scatter_k(uint32_t *dst, uint64_t m, uint64_t window, uint64_t nwin) {
  uint64_t a = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= m) return;
  uint64_t w = a / ((m + nwin - 1) / nwin);
  uint64_t off = mix(a) % window;
  dst[w * window + off] = (uint32_t)a;
}
__global__ void gather_k(const uint32_t *src, uint32_t *out, uint64_t m, uint64_t window, uint64_t nwin) {
  uint64_t a = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= m) return;
  uint64_t w = a / ((m + nwin - 1) / nwin);
  uint64_t off = mix(a) % window;
  out[a] = src[w * window + off];
}
int main() {
  const uint64_t total = 1ull << 30;      // 4 GB of uint32
  const uint64_t m = 1ull << 28;          // 268 M accesses
  uint32_t *buf, *out;
  CK(hipMalloc(&buf, total * 4)); CK(hipMalloc(&out, m * 4));
  CK(hipMemset(buf, 0, total * 4));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (uint64_t wbytes : {1ull << 20, 4ull << 20, 16ull << 20, 64ull << 20, 128ull << 20, 256ull << 20, 1ull << 30, 4ull << 30}) {
    uint64_t window = wbytes / 4, nwin = total / window;
    float ms[2];
    for (int mode = 0; mode < 2; mode++) {
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(scatter_k, dim3((m + 255) / 256), dim3(256), 0, 0, buf, m, window, nwin);
        else hipLaunchKernelGGL(gather_k, dim3((m + 255) / 256), dim3(256), 0, 0, buf, out, m, window, nwin);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[mode], e0, e1);
      }
    }
    printf("window %6llu MB: scatter %7.2f ms (%6.1f G/s)   gather %7.2f ms (%6.1f G/s)\n", (unsigned long long)(wbytes >> 20), ms[0],
           m / ms[0] / 1e6, ms[1], m / ms[1] / 1e6);
  }
  CK(hipDeviceSynchronize());
  return 0;
}
