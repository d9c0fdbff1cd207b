// what does fresh device memory cost?  (the one-shot CLI and the first call of a context pay it for every byte of the pool)
// hipcc --offload-arch=gfx950 -O2 -o alloc alloc.hip && ./alloc
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(unsigned char *p, size_t n, size_t stride) { size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * stride; if (i < n) p[i] = 1; }
int main(int argc, char **argv) {
  const size_t only = argc > 1 ? (size_t)atoll(argv[1]) : 0;
  double t0 = now();
  hipFree(0);
  printf("runtime + context: %.3f s\n", now() - t0);
  const size_t GB = 1ull << 30;
  for (size_t piece : {size_t(1) * GB, size_t(16) * GB, size_t(128) * GB}) {
    if (only && piece != only * GB) continue;
    const size_t total = 128 * GB;
    std::vector<void *> ps;
    t0 = now();
    for (size_t got = 0; got < total; got += piece) { void *p = nullptr; if (hipMalloc(&p, piece) != hipSuccess) { printf("hipMalloc failed\n"); return 1; } ps.push_back(p); }
    double t_alloc = now() - t0;
    t0 = now();
    for (void *p : ps) hipLaunchKernelGGL(touch, dim3((unsigned)((piece / 4096 + 255) / 256)), dim3(256), 0, 0, (unsigned char *)p, piece, (size_t)4096);
    hipDeviceSynchronize();
    double t_touch = now() - t0;
    t0 = now();
    for (void *p : ps) hipFree(p);
    double t_free = now() - t0;
    printf("128 GB in pieces of %3zu GB: hipMalloc %.3f s (%.1f ms/GB), first touch of every page %.3f s, hipFree %.3f s\n", piece / GB, t_alloc, 1e3 * t_alloc / 128, t_touch, t_free);
  }
  // the stream-ordered pool
  if (only == 999) {
    hipMemPool_t pool; hipDeviceGetDefaultMemPool(&pool, 0);
    uint64_t thr = ~0ull; hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
    void *p = nullptr;
    t0 = now();
    hipError_t e = hipMallocAsync(&p, 128 * GB, 0); hipStreamSynchronize(0);
    printf("hipMallocAsync 128 GB: %s %.3f s\n", hipGetErrorString(e), now() - t0);
    if (e == hipSuccess) { t0 = now(); hipFreeAsync(p, 0); hipStreamSynchronize(0); e = hipMallocAsync(&p, 64 * GB, 0); hipStreamSynchronize(0); printf("free + hipMallocAsync 64 GB from the pool: %.3f s\n", now() - t0); }
  }
  return 0;
}
