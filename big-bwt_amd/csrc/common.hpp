// common.hpp -- shared host-side plumbing for libpfpgpu (context, errors, device buffers).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include <chrono>
#include <map>
#include <thread>
#include "../../include/pfpgpu.h"

namespace pfp {

// special bytes of the parse (utils.h:6-8)
constexpr uint8_t kDollar = 2, kEndOfWord = 1, kEndOfDict = 0;
// Karp-Rabin window modulus (newscan.cpp:172)
constexpr uint32_t kPrime = 1999999973u;

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

#define PFP_HIP(expr)                                                                     \
  do {                                                                                    \
    hipError_t e__ = (expr);                                                              \
    if (e__ != hipSuccess)                                                                \
      throw ::pfp::Error(PFP_EHIP, std::string(#expr) + ": " + hipGetErrorString(e__) +   \
                                       " (" + __FILE__ + ":" + std::to_string(__LINE__) + ")"); \
  } while (0)

#define PFP_REQUIRE(cond, code, msg)                       \
  do {                                                     \
    if (!(cond)) throw ::pfp::Error((code), (msg));        \
  } while (0)

inline uint64_t cdiv(uint64_t a, uint64_t b) { return (a + b - 1) / b; }
// Launch grid for nblocks workgroups.  HIP keeps the global work size of a dimension in 32 bits (blocks x threads
// per block must stay below 2^32: a 1-D launch of 2^24 or more workgroups of 256 threads wraps around - met at
// |D| > 2^32 with one thread per suffix), so beyond 2^22 workgroups the grid becomes 2-D and kernels number their
// workgroup with BID; the last row may hold workgroups past nblocks: every kernel bounds-checks what it derives
// from BID.
inline dim3 gdim(uint64_t nblocks) {
  const unsigned gx = 1u << 22;
  if (nblocks <= gx) return dim3((unsigned)(nblocks ? nblocks : 1));
  return dim3(gx, (unsigned)((nblocks + gx - 1) / gx));
}
#define BID ((uint64_t)blockIdx.y * gridDim.x + blockIdx.x)
#define GDIM ((uint64_t)gridDim.x * gridDim.y)
inline uint64_t cdiv64(uint64_t a, uint64_t b) { return (a + b - 1) / b; }
inline int bits_for(uint64_t maxval) { int b = 1; while (b < 64 && (maxval >> b)) b++; return b; }

}  // namespace pfp

// Device memory pool of one context.  All work of a context is issued on ONE stream, so a block
// released by the host can be handed out again at once: every later use is ordered behind every
// earlier use by the stream itself.  Blocks are cached until pfp_ctx_destroy (steady-state calls
// do not touch hipMalloc); best-fit reuse within 25 % slack.
//
// PFP_POOL_DEBUG=1 (read at pfp_ctx_create) turns the pool into a checker: every request gets its own
// hipMalloc of exactly the bytes asked for plus a canary band before and after, the body is filled with a
// poison pattern (a kernel that reads memory nobody wrote sees 0xCD garbage, not the zeros of a fresh
// mapping or the leftovers of the block's previous life), the bands are verified when the block is
// released (after a stream sync) and the block goes back to the driver.  A damaged band is reported on
// stderr with the allocation site and fails every later API call on the context (pfp_debug_check).
struct pfp_pool {
  struct Block { void *p; size_t bytes; void *base; const char *file; int line; size_t req; };      // req: bytes asked for by the current holder
  static constexpr size_t kBand = 1024;            // bytes of canary on either side (debug mode)
  static constexpr unsigned char kCanary = 0xA5, kPoison = 0xCD;
  std::vector<Block> free_list;
  std::vector<Block> all;
  // total: held from the driver; live / peak: bytes the holders ASKED for (a recycled block may be larger than the request)
  size_t total_bytes = 0, peak_bytes = 0, live_bytes = 0;
  bool debug = false;
  hipStream_t stream = nullptr;
  std::string corrupt;                             // first canary damage seen (debug mode)
  uint64_t debug_blocks = 0;
  size_t soft_limit = 0;                           // bytes held from the driver beyond which any cached block that fits is reused
  size_t test_limit = 0;                           // PFP_TEST_POOL_LIMIT (read at pfp_ctx_create): requests beyond this many live bytes fail like a full device
  uint64_t driver_allocs = 0, trims = 0;           // hipMalloc calls; times a failed one made the pool give its cache back
  bool trace = false;                              // PFP_TRACE_POOL=1: remember what was live at the peak (printed by pfp_ctx_destroy)
  std::vector<Block> peak_blocks;
  void note_peak() {
    if (live_bytes <= peak_bytes) return;
    peak_bytes = live_bytes;
    if (!trace) return;
    peak_blocks.clear();
    for (const auto &b : all) {
      bool is_free = false;
      for (const auto &f : free_list) if (f.p == b.p) { is_free = true; break; }
      if (!is_free) peak_blocks.push_back(b);
    }
  }
  void print_peak() const {
    if (!trace) return;
    fprintf(stderr, "[pfp] pool peak %.2f GB in %zu blocks:\n", peak_bytes / 1e9, peak_blocks.size());
    for (const auto &b : peak_blocks)
      if (b.bytes >= (64u << 20)) fprintf(stderr, "[pfp]   %10.3f GB asked (block of %.3f)  %s:%d\n", b.req / 1e9, b.bytes / 1e9, b.file, b.line);
  }
  void *get(size_t bytes, hipError_t *err, const char *file = "", int line = 0) {
    *err = hipSuccess;
    if (test_limit && live_bytes + bytes > test_limit) { *err = hipErrorOutOfMemory; return nullptr; }
    if (debug) return get_debug(bytes, err, file, line);
    size_t best = (size_t)-1, bi = (size_t)-1;
    for (size_t i = 0; i < free_list.size(); i++) {
      size_t b = free_list[i].bytes;
      if (b >= bytes && b <= bytes + bytes / 4 + 4096 && b < best) { best = b; bi = i; }
    }
    if (bi == (size_t)-1 && soft_limit && total_bytes + bytes > soft_limit) {
      // the pool already holds most of the device: a cached block that is merely too large is better than a
      // driver call that may fail and cost every cached block (smallest block that fits, whatever the slack)
      for (size_t i = 0; i < free_list.size(); i++) {
        size_t b = free_list[i].bytes;
        if (b >= bytes && b < best) { best = b; bi = i; }
      }
    }
    if (bi != (size_t)-1) {
      void *p = free_list[bi].p;
      live_bytes += bytes;
      free_list[bi] = free_list.back(); free_list.pop_back();
      for (auto &b : all) if (b.p == p) { b.file = file; b.line = line; b.req = bytes; break; }
      note_peak();
      return p;
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    driver_allocs++;
    if (e != hipSuccess) {   // give cached blocks back to the driver and retry once
      (void)hipGetLastError();
      trim();
      trims++;
      e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) { *err = e; (void)hipGetLastError(); return nullptr; }
    all.push_back({p, bytes, p, file, line, bytes}); total_bytes += bytes;
    live_bytes += bytes; note_peak();
    return p;
  }
  void put(void *p) {
    for (size_t i = 0; i < all.size(); i++)
      if (all[i].p == p) {
        live_bytes -= all[i].req;
        if (debug) { put_debug(i); return; }
        free_list.push_back(all[i]);
        return;
      }
  }
  void *get_debug(size_t bytes, hipError_t *err, const char *file, int line) {
    void *base = nullptr;
    hipError_t e = hipMalloc(&base, bytes + 2 * kBand);
    if (e != hipSuccess) { *err = e; (void)hipGetLastError(); return nullptr; }
    unsigned char *b = (unsigned char *)base;
    (void)hipMemsetAsync(b, kCanary, kBand, stream);
    (void)hipMemsetAsync(b + kBand, kPoison, bytes, stream);
    (void)hipMemsetAsync(b + kBand + bytes, kCanary, kBand, stream);
    all.push_back({b + kBand, bytes, base, file, line, bytes});
    total_bytes += bytes; live_bytes += bytes; if (live_bytes > peak_bytes) peak_bytes = live_bytes;
    debug_blocks++;
    return b + kBand;
  }
  void put_debug(size_t i) {
    const Block blk = all[i];
    all[i] = all.back(); all.pop_back();
    total_bytes -= blk.bytes;
    (void)hipStreamSynchronize(stream);
    std::vector<unsigned char> h(2 * kBand);
    unsigned char *b = (unsigned char *)blk.base;
    (void)hipMemcpy(h.data(), b, kBand, hipMemcpyDeviceToHost);
    (void)hipMemcpy(h.data() + kBand, b + kBand + blk.bytes, kBand, hipMemcpyDeviceToHost);
    for (size_t k = 0; k < 2 * kBand; k++)
      if (h[k] != kCanary) {
        char msg[512];
        const long off = k < kBand ? (long)k - (long)kBand : (long)(k - kBand);
        snprintf(msg, sizeof msg, "pool debug: canary damaged %s a block of %zu bytes allocated at %s:%d (byte offset %ld %s, value 0x%02x)",
                 k < kBand ? "BEFORE" : "AFTER", blk.bytes, blk.file, blk.line, off, k < kBand ? "relative to its start" : "past its end", h[k]);
        fprintf(stderr, "[pfp] %s\n", msg);
        if (corrupt.empty()) corrupt = msg;
        break;
      }
    (void)hipMemset(blk.base, 0xDD, blk.bytes + 2 * kBand);      // use after release reads 0xDD
    (void)hipFree(blk.base);
  }
  void trim() {   // caller guarantees the stream is idle
    (void)hipDeviceSynchronize();
    for (auto &f : free_list) {
      (void)hipFree(f.base);
      for (size_t i = 0; i < all.size(); i++) if (all[i].p == f.p) { total_bytes -= all[i].bytes; all[i] = all.back(); all.pop_back(); break; }
    }
    free_list.clear();
  }
  void destroy() {
    for (auto &b : all) (void)hipFree(b.base);
    all.clear(); free_list.clear(); total_bytes = 0; live_bytes = 0;
  }
};

// Per-kernel timing with HIP events on the context's own stream (bench.py's roofline leg):
// every instrumented launch is bracketed by two events; pfp_get_kernel_trace resolves them.
struct pfp_ktrace {
  struct Pending { const char *name; uint64_t bytes; hipEvent_t a, b; };
  struct Agg { uint64_t launches = 0; double ms = 0; uint64_t bytes = 0; };
  bool on = false;
  std::vector<Pending> pending;
  std::vector<hipEvent_t> free_events;
  std::map<std::string, Agg> agg;
  hipEvent_t get() {
    if (!free_events.empty()) { hipEvent_t e = free_events.back(); free_events.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }
  void resolve() {   // caller has synchronised the stream
    for (auto &p : pending) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
        Agg &g = agg[p.name];
        g.launches++; g.ms += ms; g.bytes += p.bytes;
      }
      free_events.push_back(p.a); free_events.push_back(p.b);
    }
    pending.clear();
  }
  void destroy() {
    for (auto &p : pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto e : free_events) (void)hipEventDestroy(e);
    pending.clear(); free_events.clear(); agg.clear();
  }
};

struct pfp_ctx {
  int device = 0;
  pfp_pool pool;
  pfp_ktrace kt;
  bool debug = false;             // PFP_DEBUG=1: validate every intermediate on the host
  bool force_wide = false;        // PFP_FORCE_IDX64=1 / pfp_set_index_bits(ctx, 64): 64-bit dictionary positions whatever the size
  bool force_narrow = false;      // pfp_set_index_bits(ctx, 32): 32-bit positions wherever they fit (below 2^32 - 16 bytes of dictionary)
  uint64_t max_phrase = 1u << 15; // fused chain: split phrases longer than this with extra triggers (0 = off)
  double parse_density = 0.0;     // pfp_set_parse_density: the window hash cuts with probability density / p (fused chain only); 0 = chosen by repetitiveness
  bool fast_triggers = true;      // fused chain: cut the text by the cheap window hash of scan.hip (false / PFP_WINDOW_HASH=kr: Karp-Rabin)
  hipStream_t stream = nullptr;
  std::string err;
  bool profiling = false;
  pfp_stats stats{};
  int n_cu = 256;
  // pinned staging scalars for D2H counters
  uint64_t *h_scalars = nullptr;  // 16 x u64, pinned
  void *staged = nullptr;         // pfp::StagedText kept by pfp_stage_text_dev
  void *k1scratch = nullptr;      // scratch of pfp_scan_k1_enqueue
  void *dist = nullptr;           // DistState of the multi-GPU entry points
  // two pinned staging buffers for chunked, double-buffered host <-> device streams (allocated on first use)
  static constexpr size_t kPinBytes = 32u << 20;
  void *pin[2] = {nullptr, nullptr};
  hipEvent_t pin_ev[2] = {nullptr, nullptr};
  uint64_t n_syncs = 0;           // host waits on the stream (PFP_TRACE_HOST prints the count when the context goes)
  std::vector<std::thread> background;      // host work that may outlive a call (unmapping an output file): joined by the next file call and by destroy
};

namespace pfp {

// Device buffer drawn from the context's pool (see pfp_pool).
template <class T>
struct DBuf {
  T *p = nullptr;
  size_t n = 0;
  pfp_ctx *ctx = nullptr;
  DBuf() = default;
  DBuf(pfp_ctx *c, size_t count, const char *file = __builtin_FILE(), int line = __builtin_LINE()) { alloc(c, count, file, line); }
  DBuf(const DBuf &) = delete;
  DBuf &operator=(const DBuf &) = delete;
  DBuf(DBuf &&o) noexcept { *this = std::move(o); }
  DBuf &operator=(DBuf &&o) noexcept {
    if (this != &o) { release(); p = o.p; n = o.n; ctx = o.ctx; o.p = nullptr; o.n = 0; }
    return *this;
  }
  ~DBuf() { release(); }
  void alloc(pfp_ctx *c, size_t count, const char *file = __builtin_FILE(), int line = __builtin_LINE()) {
    release();
    ctx = c; n = count;
    size_t bytes = (count ? count : 1) * sizeof(T);
    if (!c->pool.debug) bytes = (bytes + 511) & ~size_t(511);      // debug: exactly what was asked for, canaries right behind
    hipError_t e;
    p = (T *)c->pool.get(bytes, &e, file, line);
    if (!p)
      throw Error(e == hipErrorOutOfMemory ? PFP_ENOMEM : PFP_EHIP,
                  std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e));
  }
  void release() {
    if (p) { ctx->pool.put(p); p = nullptr; n = 0; }
  }
  void zero() { PFP_HIP(hipMemsetAsync(p, 0, n * sizeof(T), ctx->stream)); }
  size_t bytes() const { return n * sizeof(T); }
};

template <class T>
inline void d2h(pfp_ctx *c, T *dst, const T *src, size_t count) {
  PFP_HIP(hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyDeviceToHost, c->stream));
}
template <class T>
inline void h2d(pfp_ctx *c, T *dst, const T *src, size_t count) {
  PFP_HIP(hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyHostToDevice, c->stream));
}
inline void sync(pfp_ctx *c) { c->n_syncs++; PFP_HIP(hipStreamSynchronize(c->stream)); }

// read one device scalar (syncs the stream)
template <class T>
inline T read_scalar(pfp_ctx *c, const T *dptr) {
  static_assert(sizeof(T) <= 8, "scalar");
  PFP_HIP(hipMemcpyAsync(c->h_scalars, dptr, sizeof(T), hipMemcpyDeviceToHost, c->stream));
  sync(c);
  T v;
  memcpy(&v, c->h_scalars, sizeof(T));
  return v;
}

// brackets the launches issued in its scope with two events (no-op unless tracing is on);
// algo_bytes = algorithmic bytes of those launches (DESIGN.md lists the formula per kernel)
struct KScope {
  pfp_ctx *c; size_t idx = (size_t)-1;
  KScope(pfp_ctx *ctx, const char *name, uint64_t algo_bytes) : c(ctx) {
    if (!c->kt.on) return;
    pfp_ktrace::Pending p{name, algo_bytes, c->kt.get(), c->kt.get()};
    (void)hipEventRecord(p.a, c->stream);
    idx = c->kt.pending.size();
    c->kt.pending.push_back(p);
  }
  ~KScope() { if (idx != (size_t)-1) (void)hipEventRecord(c->kt.pending[idx].b, c->stream); }
};

struct PhaseTimer {
  pfp_ctx *c; double *slot; std::chrono::steady_clock::time_point t0;
  PhaseTimer(pfp_ctx *ctx, double *s) : c(ctx), slot(s) {
    if (c->profiling) { (void)hipStreamSynchronize(c->stream); t0 = std::chrono::steady_clock::now(); }
  }
  ~PhaseTimer() {
    if (c->profiling) {
      (void)hipStreamSynchronize(c->stream);
      *slot += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
  }
};

}  // namespace pfp
