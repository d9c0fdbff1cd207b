// does a wider onesweep digit pay?  rocPRIM's default is 8 bits per pass; radix_sort_onesweep_config takes up to log2(block size).
// The chain's first-round sorts: 80 M u64 keys on bits [28, 64) (configs[2] dictionary, keys only: 5 passes of 8 bits, 4 of 9),
// (u64, u32) pairs on 40 bits (12.6 GB dictionary: 5 passes, 4 of 10) and on 64 bits (phrase hashes: 8 passes, 7 of 10).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o sortcfg sortcfg.hip && ./sortcfg [millions of elements] [other]   (built on demand: 40 MB, not kept in the tree)
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <class K, class V>
__global__ void fill(K *k, V *v, size_t n, int skew) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint64_t x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    if (skew) x &= 0x7777777777777777ull;      // (the alphabetic code's skewed bits, roughly)
    k[i] = (K)x; if (v) v[i] = (V)i;
  }
}
template <class K>
__global__ void check_sorted(const K *k, size_t n, int bb, int eb, unsigned long long *bad) {
  const uint64_t mask = (eb >= 64 ? ~0ull : ((1ull << eb) - 1)) & ~((1ull << bb) - 1);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + 1 < n; i += (size_t)gridDim.x * blockDim.x)
    if (((uint64_t)k[i] & mask) > ((uint64_t)k[i + 1] & mask)) atomicAdd(bad, 1ull);
}
template <class Config, bool PAIRS, class K = uint64_t, class V = uint32_t>
static float run(const char *what, K *k, K *k2, V *v, V *v2, size_t n, int bb, int eb, int skew) {
  size_t tb = 0;
  void *tmp = nullptr;
  rocprim::double_buffer<K> dk(k, k2);
  rocprim::double_buffer<V> dv(v, v2);
  if (PAIRS) rocprim::radix_sort_pairs<Config>(nullptr, tb, dk, dv, n, bb, eb, 0);
  else rocprim::radix_sort_keys<Config>(nullptr, tb, dk, n, bb, eb, 0);
  hipMalloc(&tmp, tb);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  unsigned long long *bad; hipMalloc(&bad, 8); hipMemset(bad, 0, 8);
  for (int it = 0; it < 4; it++) {
    hipLaunchKernelGGL((fill<K, V>), dim3(4096), dim3(256), 0, 0, k, PAIRS ? v : (V *)nullptr, n, skew);
    rocprim::double_buffer<K> a(k, k2);
    rocprim::double_buffer<V> b(v, v2);
    hipEventRecord(e0, 0);
    hipError_t e = PAIRS ? rocprim::radix_sort_pairs<Config>(tmp, tb, a, b, n, bb, eb, 0) : rocprim::radix_sort_keys<Config>(tmp, tb, a, n, bb, eb, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    if (e != hipSuccess) { printf("%s: %s\n", what, hipGetErrorString(e)); return -1; }
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    if (it == 3) hipLaunchKernelGGL((check_sorted<K>), dim3(4096), dim3(256), 0, 0, (const K *)a.current(), n, bb, eb, bad);
  }
  unsigned long long hb = 0; hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
  printf("  %-34s %8.3f ms   %s\n", what, best, hb ? "NOT SORTED" : "sorted");
  hipFree(tmp); hipFree(bad);
  return best;
}
template <unsigned BS, unsigned IPT, unsigned BITS>
using Cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                       rocprim::radix_sort_onesweep_config<rocprim::kernel_config<BS, IPT>, rocprim::kernel_config<BS, IPT>, BITS,
                                                                           rocprim::block_radix_rank_algorithm::match>, 1024 * 1024>;
template <bool PAIRS>
static void series(const char *title, uint64_t *k, uint64_t *k2, uint32_t *v, uint32_t *v2, size_t n, int bb, int eb, int skew) {
  printf("%s: %zu M elements, bits [%d, %d)%s\n", title, n >> 20, bb, eb, skew ? ", skewed" : "");
  run<rocprim::default_config, PAIRS>("default", k, k2, v, v2, n, bb, eb, skew);
  if (!PAIRS) {
    run<Cfg<1024, 8, 9>, PAIRS>("1024 x 8, 9 bits", k, k2, v, v2, n, bb, eb, skew);
    run<Cfg<1024, 10, 9>, PAIRS>("1024 x 10, 9 bits", k, k2, v, v2, n, bb, eb, skew);
    run<Cfg<1024, 12, 9>, PAIRS>("1024 x 12, 9 bits", k, k2, v, v2, n, bb, eb, skew);
    run<Cfg<1024, 16, 9>, PAIRS>("1024 x 16, 9 bits", k, k2, v, v2, n, bb, eb, skew);
    run<Cfg<512, 16, 9>, PAIRS>("512 x 16, 9 bits", k, k2, v, v2, n, bb, eb, skew);
    run<Cfg<1024, 8, 8>, PAIRS>("1024 x 8, 8 bits", k, k2, v, v2, n, bb, eb, skew);
    run<Cfg<1024, 12, 8>, PAIRS>("1024 x 12, 8 bits", k, k2, v, v2, n, bb, eb, skew);
    run<Cfg<1024, 12, 10>, PAIRS>("1024 x 12, 10 bits", k, k2, v, v2, n, bb, eb, skew);
  } else {
    run<Cfg<1024, 4, 8>, PAIRS>("1024 x 4, 8 bits", k, k2, v, v2, n, bb, eb, skew);
    run<Cfg<1024, 5, 8>, PAIRS>("1024 x 5, 8 bits", k, k2, v, v2, n, bb, eb, skew);
    run<Cfg<1024, 6, 8>, PAIRS>("1024 x 6, 8 bits", k, k2, v, v2, n, bb, eb, skew);
    run<Cfg<1024, 7, 8>, PAIRS>("1024 x 7, 8 bits", k, k2, v, v2, n, bb, eb, skew);
    run<Cfg<1024, 8, 8>, PAIRS>("1024 x 8, 8 bits", k, k2, v, v2, n, bb, eb, skew);
    run<Cfg<1024, 10, 8>, PAIRS>("1024 x 10, 8 bits", k, k2, v, v2, n, bb, eb, skew);
    run<Cfg<1024, 8, 9>, PAIRS>("1024 x 8, 9 bits", k, k2, v, v2, n, bb, eb, skew);
    run<Cfg<1024, 10, 9>, PAIRS>("1024 x 10, 9 bits", k, k2, v, v2, n, bb, eb, skew);
    run<Cfg<1024, 5, 10>, PAIRS>("1024 x 5, 10 bits", k, k2, v, v2, n, bb, eb, skew);
  }
}
template <class K, class V>
static void series_kv(const char *title, size_t n, int bb, int eb) {
  K *k, *k2; V *v, *v2;
  hipMalloc(&k, n * sizeof(K)); hipMalloc(&k2, n * sizeof(K)); hipMalloc(&v, n * sizeof(V)); hipMalloc(&v2, n * sizeof(V));
  printf("%s: %zu M elements, bits [%d, %d)\n", title, n >> 20, bb, eb);
  run<rocprim::default_config, true, K, V>("default", k, k2, v, v2, n, bb, eb, 0);
  run<Cfg<1024, 4, 8>, true, K, V>("1024 x 4, 8 bits", k, k2, v, v2, n, bb, eb, 0);
  run<Cfg<1024, 5, 8>, true, K, V>("1024 x 5, 8 bits", k, k2, v, v2, n, bb, eb, 0);
  run<Cfg<1024, 6, 8>, true, K, V>("1024 x 6, 8 bits", k, k2, v, v2, n, bb, eb, 0);
  run<Cfg<1024, 7, 8>, true, K, V>("1024 x 7, 8 bits", k, k2, v, v2, n, bb, eb, 0);
  run<Cfg<1024, 8, 8>, true, K, V>("1024 x 8, 8 bits", k, k2, v, v2, n, bb, eb, 0);
  run<Cfg<1024, 10, 8>, true, K, V>("1024 x 10, 8 bits", k, k2, v, v2, n, bb, eb, 0);
  run<Cfg<1024, 12, 8>, true, K, V>("1024 x 12, 8 bits", k, k2, v, v2, n, bb, eb, 0);
  run<Cfg<1024, 8, 9>, true, K, V>("1024 x 8, 9 bits", k, k2, v, v2, n, bb, eb, 0);
  run<Cfg<1024, 6, 9>, true, K, V>("1024 x 6, 9 bits", k, k2, v, v2, n, bb, eb, 0);
  hipFree(k); hipFree(k2); hipFree(v); hipFree(v2);
}
int main(int argc, char **argv) {
  const size_t n = (size_t)(argc > 1 ? atoll(argv[1]) : 80) << 20;
  if (argc > 2) {      // the other pair types of the chain
    series_kv<uint32_t, uint32_t>("pairs (u32, u32)", n, 0, 24);
    series_kv<uint32_t, uint32_t>("pairs (u32, u32)", n, 0, 32);
    series_kv<uint64_t, uint64_t>("pairs (u64, u64)", n, 0, 48);
    return 0;
  }
  uint64_t *k, *k2; uint32_t *v, *v2;
  hipMalloc(&k, n * 8); hipMalloc(&k2, n * 8); hipMalloc(&v, n * 4); hipMalloc(&v2, n * 4);
  series<false>("keys only", k, k2, v, v2, n, 28, 64, 1);
  series<true>("pairs (u64, u32)", k, k2, v, v2, n, 0, 40, 1);
  series<true>("pairs (u64, u32)", k, k2, v, v2, n, 0, 64, 0);
  series<true>("pairs (u64, u32)", k, k2, v, v2, n, 0, 48, 1);
  return 0;
}
