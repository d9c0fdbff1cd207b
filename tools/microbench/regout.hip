// can the output files' own pages be the D2H target?  (the one-shot CLI spends 2.2 of its 3.5 s copying 13.7 GB from pinned
// buffers into /dev/shm files with pwrite(): ~6 GB/s, one writer per inode.)  Map the file, register the mapping with the
// runtime, copy HBM -> mapping in one go; time every step, in pieces too (registration can run beside the chain).
// hipcc --offload-arch=gfx950 -O2 -o regout regout.hip -lpthread && ./regout [GB] [piece_MB]
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void fill(unsigned char *p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned char)(i * 2654435761u >> 13);
}
static unsigned char expect(size_t i) { return (unsigned char)(i * 2654435761u >> 13); }
int main(int argc, char **argv) {
  const double gb = argc > 1 ? atof(argv[1]) : 2.0;
  const size_t piece = (argc > 2 ? (size_t)atoll(argv[2]) : 1024) << 20;
  const int mode = argc > 3 ? atoi(argv[3]) : 0;      // 0: ftruncate only; 1: fallocate first; 2: MAP_POPULATE
  const size_t n = ((size_t)(gb * 1e9) + 4095) & ~size_t(4095);
  hipFree(0);
  unsigned char *d = nullptr;
  if (hipMalloc(&d, n) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, d, n);
  hipDeviceSynchronize();
  const char *path = "/dev/shm/regout.bin";
  unlink(path);
  double t0 = now(), T0 = t0;
  int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
  if (fd < 0 || ftruncate(fd, (off_t)n) != 0) { printf("cannot create %s\n", path); return 1; }
  if (mode == 1) { if (posix_fallocate(fd, 0, (off_t)n) != 0) printf("fallocate failed\n"); printf("fallocate %.3f s\n", now() - t0); t0 = now(); }
  unsigned char *m = (unsigned char *)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED | (mode == 2 ? MAP_POPULATE : 0), fd, 0);
  if (m == MAP_FAILED) { printf("mmap failed\n"); return 1; }
  printf("create + map %.2f GB: %.3f s\n", n / 1e9, now() - t0);
  if (mode == 3) {      // fault the pages in from T threads (MADV_POPULATE_WRITE per piece), then register from T threads
    const int T = argc > 4 ? atoi(argv[4]) : 8;
    const size_t np = (n + piece - 1) / piece;
    for (int phase = 0; phase < 2; phase++) {
      t0 = now();
      std::vector<std::thread> th;
      std::vector<double> busy(T, 0.0);
      for (int t = 0; t < T; t++) th.emplace_back([&, t]() {
        hipSetDevice(0);
        for (size_t i = t; i < np; i += T) {
          const size_t off = i * piece, len = n - off < piece ? n - off : piece;
          double a = now();
          if (phase == 0) { if (madvise(m + off, len, 23 /* MADV_POPULATE_WRITE */) != 0) for (size_t q = 0; q < len; q += 4096) ((volatile unsigned char *)m)[off + q] = 0; }
          else if (hipHostRegister(m + off, len, hipHostRegisterDefault) != hipSuccess) printf("register failed\n");
          busy[t] += now() - a;
        }
      });
      for (auto &x : th) x.join();
      double sum = 0; for (double b : busy) sum += b;
      printf("%s from %d threads: %.3f s wall (%.2f GB/s), %.3f thread seconds\n", phase == 0 ? "populate" : "register", T, now() - t0, n / 1e9 / (now() - t0), sum);
    }
    t0 = now();
    hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    for (size_t off = 0; off < n; off += piece) hipMemcpyAsync(m + off, d + off, n - off < piece ? n - off : piece, hipMemcpyDeviceToHost, s2);
    hipStreamSynchronize(s2);
    printf("copy into the registered mapping: %.3f s = %.2f GB/s\n", now() - t0, n / 1e9 / (now() - t0));
    t0 = now();
    for (size_t off = 0; off < n; off += piece) hipHostUnregister(m + off);
    printf("unregister %.3f s\n", now() - t0);
    t0 = now();
    munmap(m, n); close(fd); unlink(path);
    printf("unmap + close + unlink %.3f s\n", now() - t0);
    return 0;
  }
  if (mode == 6 || mode == 7) {      // the whole file allocated first (one fallocate), THEN T threads map (6: read faults, 7: write faults) and register
    const int T = argc > 4 ? atoi(argv[4]) : 4;
    const size_t np = (n + piece - 1) / piece;
    t0 = now();
    if (fallocate(fd, 0, 0, (off_t)n) != 0) printf("fallocate failed\n");
    const double t_alloc = now() - t0;
    double t1 = now();
    std::vector<std::thread> th;
    std::vector<double> bp(T, 0.0), br(T, 0.0);
    for (int t = 0; t < T; t++) th.emplace_back([&, t]() {
      hipSetDevice(0);
      for (size_t i = t; i < np; i += T) {
        const size_t off = i * piece, len = n - off < piece ? n - off : piece;
        double a = now();
        if (madvise(m + off, len, mode == 6 ? 22 : 23) != 0) printf("madvise failed\n");
        double b = now();
        if (hipHostRegister(m + off, len, hipHostRegisterDefault) != hipSuccess) printf("register failed\n");
        bp[t] += b - a; br[t] += now() - b;
      }
    });
    for (auto &x : th) x.join();
    double sp = 0, sr = 0; for (int t = 0; t < T; t++) { sp += bp[t]; sr += br[t]; }
    printf("fallocate %.3f s; then %d threads mapping (%s faults) %.3f and registering %.3f thread seconds in %.3f s: all ready after %.3f s (%.2f GB/s)\n", t_alloc, T, mode == 6 ? "read" : "write", sp, sr, now() - t1, now() - t0, n / 1e9 / (now() - t0));
    t0 = now();
    hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    for (size_t off = 0; off < n; off += piece) hipMemcpyAsync(m + off, d + off, n - off < piece ? n - off : piece, hipMemcpyDeviceToHost, s2);
    hipStreamSynchronize(s2);
    printf("copy into the registered mapping: %.3f s = %.2f GB/s\n", now() - t0, n / 1e9 / (now() - t0));
    for (size_t off = 0; off < n; off += piece) hipHostUnregister(m + off);
    munmap(m, n); close(fd);
    fd = open(path, O_RDONLY);
    std::vector<unsigned char> buf(1 << 20);
    size_t bad = 0;
    for (size_t off : {size_t(0), n / 2 & ~size_t(4095), n - buf.size()}) {
      ssize_t r = pread(fd, buf.data(), buf.size(), (off_t)off);
      for (ssize_t i = 0; i < r; i++) bad += buf[i] != expect(off + i);
    }
    close(fd); unlink(path);
    printf("file content %s\n", bad ? "WRONG" : "ok");
    return 0;
  }
  if (mode == 4 || mode == 5) {      // one thread allocates the pages piece by piece (fallocate), T threads map (4: read faults, 5: write faults) and register behind it
    const int T = argc > 4 ? atoi(argv[4]) : 4;
    const size_t np = (n + piece - 1) / piece;
    std::vector<int> ready(np, 0);
    volatile int *rd = ready.data();
    t0 = now();
    double t_alloc = 0;
    std::thread A([&]() {
      for (size_t i = 0; i < np; i++) {
        const size_t off = i * piece, len = n - off < piece ? n - off : piece;
        if (fallocate(fd, 0, (off_t)off, (off_t)len) != 0) printf("fallocate failed\n");
        __atomic_store_n(&rd[i], 1, __ATOMIC_RELEASE);
      }
      t_alloc = now() - t0;
    });
    std::vector<std::thread> th;
    std::vector<double> bp(T, 0.0), br(T, 0.0);
    for (int t = 0; t < T; t++) th.emplace_back([&, t]() {
      hipSetDevice(0);
      for (size_t i = t; i < np; i += T) {
        const size_t off = i * piece, len = n - off < piece ? n - off : piece;
        while (!__atomic_load_n(&rd[i], __ATOMIC_ACQUIRE)) usleep(200);
        double a = now();
        if (madvise(m + off, len, mode == 4 ? 22 : 23) != 0) printf("madvise failed\n");
        double b = now();
        if (hipHostRegister(m + off, len, hipHostRegisterDefault) != hipSuccess) printf("register failed\n");
        bp[t] += b - a; br[t] += now() - b;
      }
    });
    A.join();
    for (auto &x : th) x.join();
    double sp = 0, sr = 0; for (int t = 0; t < T; t++) { sp += bp[t]; sr += br[t]; }
    printf("fallocate thread %.3f s; + %d threads mapping (%s faults) %.3f and registering %.3f thread seconds: all ready after %.3f s (%.2f GB/s)\n", t_alloc, T, mode == 4 ? "read" : "write", sp, sr, now() - t0, n / 1e9 / (now() - t0));
    t0 = now();
    hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    for (size_t off = 0; off < n; off += piece) hipMemcpyAsync(m + off, d + off, n - off < piece ? n - off : piece, hipMemcpyDeviceToHost, s2);
    hipStreamSynchronize(s2);
    printf("copy into the registered mapping: %.3f s = %.2f GB/s\n", now() - t0, n / 1e9 / (now() - t0));
    for (size_t off = 0; off < n; off += piece) hipHostUnregister(m + off);
    t0 = now();
    munmap(m, n);
    printf("unmap %.3f s\n", now() - t0);
    close(fd);
    fd = open(path, O_RDONLY);
    std::vector<unsigned char> buf(1 << 20);
    size_t bad = 0;
    for (size_t off : {size_t(0), n / 2 & ~size_t(4095), n - buf.size()}) {
      ssize_t r = pread(fd, buf.data(), buf.size(), (off_t)off);
      for (ssize_t i = 0; i < r; i++) bad += buf[i] != expect(off + i);
    }
    close(fd); unlink(path);
    printf("file content %s\n", bad ? "WRONG" : "ok");
    return 0;
  }
  // register in pieces on a helper thread, copy each piece as soon as it is registered
  t0 = now();
  std::vector<double> treg;
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  double t_reg = 0, t_first = 0;
  for (size_t off = 0; off < n; off += piece) {
    const size_t len = n - off < piece ? n - off : piece;
    double a = now();
    hipError_t e = hipHostRegister(m + off, len, hipHostRegisterDefault);
    if (e != hipSuccess) { printf("hipHostRegister failed at %zu: %s\n", off, hipGetErrorString(e)); return 2; }
    t_reg += now() - a;
    if (off == 0) t_first = now() - a;
    e = hipMemcpyAsync(m + off, d + off, len, hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) { printf("hipMemcpyAsync failed: %s\n", hipGetErrorString(e)); return 2; }
  }
  double t_issue = now() - t0;
  hipStreamSynchronize(s);
  double t_all = now() - t0;
  printf("register (pieces of %zu MB) %.3f s total (first %.3f s), register+copy issue %.3f s, all copies done %.3f s = %.2f GB/s\n", piece >> 20, t_reg, t_first, t_issue, t_all, n / 1e9 / t_all);
  t0 = now();
  for (size_t off = 0; off < n; off += piece) hipHostUnregister(m + off);
  double t_unreg = now() - t0;
  t0 = now();
  munmap(m, n); close(fd);
  printf("unregister %.3f s, unmap + close %.3f s, whole %.3f s = %.2f GB/s\n", t_unreg, now() - t0, now() - T0, n / 1e9 / (now() - T0));
  // check through the file
  fd = open(path, O_RDONLY);
  std::vector<unsigned char> buf(1 << 20);
  size_t bad = 0;
  for (size_t off : {size_t(0), n / 2 & ~size_t(4095), n - buf.size()}) {
    ssize_t r = pread(fd, buf.data(), buf.size(), (off_t)off);
    for (ssize_t i = 0; i < r; i++) bad += buf[i] != expect(off + i);
  }
  close(fd);
  printf("file content %s\n", bad ? "WRONG" : "ok");
  // the present way for comparison: D2H into two pinned buffers, pwrite
  t0 = now();
  fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
  (void)!ftruncate(fd, (off_t)n);
  unsigned char *pin[2]; hipHostMalloc((void **)&pin[0], 32 << 20, hipHostMallocNonCoherent); hipHostMalloc((void **)&pin[1], 32 << 20, hipHostMallocNonCoherent);
  hipEvent_t ev[2]; hipEventCreate(&ev[0]); hipEventCreate(&ev[1]);
  const size_t P = 32 << 20;
  size_t np = (n + P - 1) / P;
  for (size_t i = 0; i < np + 1; i++) {
    if (i < np) { size_t len = n - i * P < P ? n - i * P : P; hipMemcpyAsync(pin[i & 1], d + i * P, len, hipMemcpyDeviceToHost, s); hipEventRecord(ev[i & 1], s); }
    if (i > 0) { size_t j = i - 1; size_t len = n - j * P < P ? n - j * P : P; hipEventSynchronize(ev[j & 1]); if (pwrite(fd, pin[j & 1], len, (off_t)(j * P)) != (ssize_t)len) printf("short write\n"); }
  }
  close(fd);
  printf("pinned buffers + pwrite: %.3f s = %.2f GB/s\n", now() - t0, n / 1e9 / (now() - t0));
  unlink(path);
  return 0;
}
