// gather-rate microbenchmark (gfx950): independent random reads of ELEM bytes from a table of a given size,
// one per lane, index stream read coalesced.  Prints G gathers/s per (table bytes, element bytes, dependent depth).
// Build: hipcc -O3 --offload-arch=gfx950 gather.hip -o gather ; run: ./gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <class T, int DEPTH, int PER>
__global__ __launch_bounds__(256) void gather_kernel(const T *__restrict__ tab, uint64_t mask, const uint32_t *__restrict__ idx, uint64_t n,
                                                     uint32_t *__restrict__ out) {
  const uint64_t t0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * PER;
  if (t0 >= n) return;
  uint32_t acc = 0;
  uint32_t ix[PER];
#pragma unroll
  for (int k = 0; k < PER; k++) ix[k] = idx[t0 + k];
#pragma unroll
  for (int k = 0; k < PER; k++) {
    uint64_t j = ix[k] & mask;
    uint32_t v = (uint32_t)tab[j];
    if (DEPTH >= 2) { j = (mix(v + ix[k]) ) & mask; v = (uint32_t)tab[j]; }
    if (DEPTH >= 3) { j = (mix(v ^ ix[k]) ) & mask; v = (uint32_t)tab[j]; }
    acc += v;
  }
  out[t0 / PER] = acc;
}
__global__ void fill_idx(uint32_t *idx, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) idx[i] = mix((uint32_t)i * 2654435761u + 12345u);
}
// sorted-ish indices: locality like "members of one family": windows of 64 consecutive lanes read within a 4 KB region
__global__ void fill_idx_local(uint32_t *idx, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) idx[i] = (mix((uint32_t)(i >> 6) * 2654435761u) & ~1023u) | (mix((uint32_t)i) & 1023u);
}

template <class T, int DEPTH, int PER>
static double run(const void *tab, uint64_t tab_elems, const uint32_t *idx, uint64_t n, uint32_t *out) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const uint64_t mask = tab_elems - 1;
  dim3 grid((unsigned)((n / PER + 255) / 256));
  hipLaunchKernelGGL((gather_kernel<T, DEPTH, PER>), grid, dim3(256), 0, 0, (const T *)tab, mask, idx, n, out);
  hipEventRecord(a);
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL((gather_kernel<T, DEPTH, PER>), grid, dim3(256), 0, 0, (const T *)tab, mask, idx, n, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return (double)n * DEPTH * 3 / (ms * 1e-3) / 1e9;
}

int main() {
  const uint64_t n = 1ull << 27;          // 128 M gathers per launch
  uint32_t *idx, *idxl, *out; void *tab;
  const uint64_t tab_max = 8ull << 30;
  CHK(hipMalloc(&idx, n * 4)); CHK(hipMalloc(&idxl, n * 4)); CHK(hipMalloc(&out, n * 4)); CHK(hipMalloc(&tab, tab_max));
  CHK(hipMemset(tab, 1, tab_max));
  hipLaunchKernelGGL(fill_idx, dim3((unsigned)(n / 256)), dim3(256), 0, 0, idx, n);
  hipLaunchKernelGGL(fill_idx_local, dim3((unsigned)(n / 256)), dim3(256), 0, 0, idxl, n);
  CHK(hipDeviceSynchronize());
  printf("%-10s %-5s %-6s %-4s %10s\n", "table", "elem", "depth", "per", "Ggather/s");
  for (uint64_t tb = 1ull << 20; tb <= tab_max; tb <<= 2) {
    printf("%7.1f MB  u8    1      8  %10.1f\n", tb / 1048576.0, run<uint8_t, 1, 8>(tab, tb, idx, n, out));
    printf("%7.1f MB  u16   1      8  %10.1f\n", tb / 1048576.0, run<uint16_t, 1, 8>(tab, tb / 2, idx, n, out));
    printf("%7.1f MB  u32   1      8  %10.1f\n", tb / 1048576.0, run<uint32_t, 1, 8>(tab, tb / 4, idx, n, out));
    printf("%7.1f MB  u32   1      1  %10.1f\n", tb / 1048576.0, run<uint32_t, 1, 1>(tab, tb / 4, idx, n, out));
    printf("%7.1f MB  u64   1      8  %10.1f\n", tb / 1048576.0, run<uint64_t, 1, 8>(tab, tb / 8, idx, n, out));
    printf("%7.1f MB  u32   2      8  %10.1f\n", tb / 1048576.0, run<uint32_t, 2, 8>(tab, tb / 4, idx, n, out));
    printf("%7.1f MB  u32   3      8  %10.1f\n", tb / 1048576.0, run<uint32_t, 3, 8>(tab, tb / 4, idx, n, out));
    printf("%7.1f MB  u32   1      8  %10.1f  (64 lanes inside 4 KB)\n", tb / 1048576.0, run<uint32_t, 1, 8>(tab, tb / 4, idxl, n, out));
    fflush(stdout);
  }
  return 0;
}
