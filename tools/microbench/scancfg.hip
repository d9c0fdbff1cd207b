// rocPRIM's device scan with other tiles than its gfx942 default (256 x 21 for 4-byte values): the chain makes 32 exclusive sums of
// 5-80 M u32 counts per step on configs[2] (1.2 ms), 49 of up to 260 M on 12.6 GB (7 ms).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o scancfg scancfg.hip && ./scancfg
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdlib>
template <class Config>
static void run(const char *what, const uint32_t *in, uint32_t *out, size_t n) {
  size_t tb = 0; void *tmp = nullptr;
  rocprim::exclusive_scan<Config>(nullptr, tb, in, out, 0u, n, rocprim::plus<uint32_t>(), 0);
  hipMalloc(&tmp, tb);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int it = 0; it < 6; it++) {
    hipEventRecord(e0, 0);
    rocprim::exclusive_scan<Config>(tmp, tb, in, out, 0u, n, rocprim::plus<uint32_t>(), 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  printf("  %-22s %8.1f us  (%.2f TB/s)\n", what, best * 1e3, n * 8.0 / best / 1e9);
  hipFree(tmp);
}
template <unsigned BS, unsigned IPT>
using Cfg = rocprim::scan_config<BS, IPT, rocprim::block_load_method::block_load_transpose, rocprim::block_store_method::block_store_transpose,
                                 rocprim::block_scan_algorithm::using_warp_scan>;
int main() {
  for (size_t n : {size_t(5) << 20, size_t(16) << 20, size_t(80) << 20, size_t(260) << 20}) {
    uint32_t *in, *out; hipMalloc(&in, n * 4); hipMalloc(&out, n * 4); hipMemset(in, 1, n * 4);
    printf("%zu M u32\n", n >> 20);
    run<rocprim::default_config>("default", in, out, n);
    run<Cfg<256, 16>>("256 x 16", in, out, n);
    run<Cfg<256, 24>>("256 x 24", in, out, n);
    run<Cfg<256, 32>>("256 x 32", in, out, n);
    run<Cfg<256, 12>>("256 x 12", in, out, n);
    run<Cfg<256, 8>>("256 x 8", in, out, n);
    run<Cfg<128, 16>>("128 x 16", in, out, n);
    run<Cfg<128, 32>>("128 x 32", in, out, n);
    hipFree(in); hipFree(out);
  }
  return 0;
}
