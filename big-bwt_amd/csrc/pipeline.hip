// pipeline.hip -- orchestration of the parse -> SA -> BWT chain and the extern "C" boundary
// (include/pfpgpu.h).  The chain mirrors bigbwt:69-156 (newscan -> bwtparse -> pfbwt) but every
// intermediate stays in HBM; the staged entry points ingest/emit the reference's file formats.
#include <atomic>
#include "kernels.hpp"
#include "prims.hpp"
#include "devutil.hpp"
#include <cstdlib>
#include <new>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/vfs.h>
#include <condition_variable>
#include <cerrno>
#include <thread>
#include <mutex>
#include <algorithm>

using namespace pfp;

namespace pfp {

static constexpr int TB = 256;

// host-side copy with a few threads: one core moves ~10 GB/s, PCIe Gen5 x16 takes 50
static void par_memcpy(void *dst, const void *src, size_t len) {
  const unsigned hw = std::thread::hardware_concurrency();
  const size_t T = std::min<size_t>(std::min<size_t>(8, hw ? hw : 1), len >> 22);
  if (T <= 1) { memcpy(dst, src, len); return; }
  std::vector<std::thread> th;
  const size_t part = (len / T + 4095) & ~size_t(4095);
  for (size_t k = 0; k < T; k++) {
    const size_t off = k * part;
    if (off >= len) break;
    const size_t l = std::min(part, len - off);
    th.emplace_back([=]() { memcpy((uint8_t *)dst + off, (const uint8_t *)src + off, l); });
  }
  for (auto &t : th) t.join();
}

// the same from a file: a few threads pread their parts straight into the (pinned) destination - one copy out of the page
// cache and no page-table work, where an mmap'ed source costs a fault per 4 KB page (12.6 GB in /dev/shm: 2 GB/s through the
// mapping, an order of magnitude more through pread)
static void par_pread(int fd, uint64_t file_off, void *dst, size_t len) {
  const unsigned hw = std::thread::hardware_concurrency();
  static const size_t tmax = []() { const char *e = getenv("PFP_READ_THREADS"); return e ? (size_t)atoi(e) : (size_t)8; }();
  const size_t T = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(tmax, hw ? hw : 1), len >> 21));
  std::atomic<int> bad{0};
  auto work = [&](size_t off, size_t l) {
    size_t done = 0;
    while (done < l) {
      const ssize_t r = pread(fd, (uint8_t *)dst + off + done, l - done, (off_t)(file_off + off + done));
      if (r <= 0) { bad.store(r == 0 ? -1 : errno ? errno : -1); return; }
      done += (size_t)r;
    }
  };
  if (T <= 1) work(0, len);
  else {
    std::vector<std::thread> th;
    const size_t part = (len / T + 4095) & ~size_t(4095);
    for (size_t k = 0; k < T && k * part < len; k++) th.emplace_back(work, k * part, std::min(part, len - k * part));
    for (auto &t : th) t.join();
  }
  PFP_REQUIRE(bad.load() == 0, PFP_EINVAL, bad.load() == -1 ? std::string("input file is shorter than announced") : std::string("reading the input: ") + strerror(bad.load()));
}

static void ensure_pinned(pfp_ctx *c) {
  for (int k = 0; k < 2; k++) {
    // (PFP_PIN_NONCOHERENT=1: host-cacheable staging buffers - the CPU fills them, the copy engine reads them)
    static const unsigned pin_flags = getenv("PFP_PIN_NONCOHERENT") ? hipHostMallocNonCoherent : hipHostMallocDefault;
    if (!c->pin[k]) PFP_HIP(hipHostMalloc(&c->pin[k], pfp_ctx::kPinBytes, pin_flags));
    if (!c->pin_ev[k]) PFP_HIP(hipEventCreateWithFlags(&c->pin_ev[k], hipEventDisableTiming));
  }
}
// Device -> host stream in chunks through the two pinned buffers: while chunk i crosses PCIe, `sink`
// consumes chunk i-1 on the host (file write, copy into the caller's buffer).
template <class Sink>
static void stream_d2h(pfp_ctx *c, const uint8_t *d_src, uint64_t nbytes, Sink &&sink) {
  ensure_pinned(c);
  const uint64_t CH = pfp_ctx::kPinBytes;
  uint64_t prev_off = 0, prev_len = 0;
  int k = 0;
  for (uint64_t off = 0; off < nbytes || prev_len; off += CH) {
    const uint64_t len = off < nbytes ? std::min<uint64_t>(CH, nbytes - off) : 0;
    if (len) {
      PFP_HIP(hipMemcpyAsync(c->pin[k], d_src + off, len, hipMemcpyDeviceToHost, c->stream));
      PFP_HIP(hipEventRecord(c->pin_ev[k], c->stream));
    }
    if (prev_len) {
      PFP_HIP(hipEventSynchronize(c->pin_ev[k ^ 1]));
      sink((const uint8_t *)c->pin[k ^ 1], prev_off, prev_len);
    }
    prev_off = off; prev_len = len;
    k ^= 1;
  }
}
// Host -> device the same way: `fill` writes chunk i into a pinned buffer while chunk i-1 crosses PCIe.
template <class Fill>
static void stream_h2d(pfp_ctx *c, uint8_t *d_dst, uint64_t nbytes, Fill &&fill) {
  ensure_pinned(c);
  const uint64_t CH = pfp_ctx::kPinBytes;
  static const bool trace_host = getenv("PFP_TRACE_HOST") != nullptr;
  auto now = []() { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  double t_wait = 0, t_fill = 0, t_issue = 0;
  int k = 0;
  for (uint64_t off = 0; off < nbytes; off += CH, k ^= 1) {
    const uint64_t len = std::min<uint64_t>(CH, nbytes - off);
    const auto a0 = now();
    if (off >= 2 * CH) PFP_HIP(hipEventSynchronize(c->pin_ev[k]));      // the copy that last used this buffer is done
    const auto a1 = now();
    fill((uint8_t *)c->pin[k], off, len);
    const auto a2 = now();
    PFP_HIP(hipMemcpyAsync(d_dst + off, c->pin[k], len, hipMemcpyHostToDevice, c->stream));
    PFP_HIP(hipEventRecord(c->pin_ev[k], c->stream));
    if (trace_host) { t_wait += secs(a0, a1); t_fill += secs(a1, a2); t_issue += secs(a2, now()); }
  }
  if (trace_host && nbytes >= (64u << 20))
    fprintf(stderr, "[pfp] host -> device, %.2f GB in %llu-MB pieces: filling the pinned buffers %.3f s, waiting for the copy engine %.3f s, issuing %.3f s\n",
            nbytes / 1e9, (unsigned long long)(CH >> 20), t_fill, t_wait, t_issue);
}

void StagedText::stage(pfp_ctx *c, const void *src, bool src_on_device, uint64_t n_, int w_) {
  n = n_; w = w_;
  size_t total = kFront + n + (size_t)w + kBack;
  buf.alloc(c, total);
  PFP_HIP(hipMemsetAsync(buf.p, 0, kFront - 1, c->stream));
  PFP_HIP(hipMemsetAsync(buf.p + kFront - 1, kDollar, 1, c->stream));
  if (n && src_on_device) PFP_HIP(hipMemcpyAsync(buf.p + kFront, src, n, hipMemcpyDeviceToDevice, c->stream));
  if (n && !src_on_device)      // pageable host text (a caller's buffer, an mmap of the input file): chunks through pinned buffers
    stream_h2d(c, buf.p + kFront, n, [&](uint8_t *pin, uint64_t off, uint64_t len) { par_memcpy(pin, (const uint8_t *)src + off, len); });
  restage_tail(c, n, w);
}
void StagedText::stage_fd(pfp_ctx *c, int fd, uint64_t file_off, uint64_t n_, int w_) {
  n = n_; w = w_;
  size_t total = kFront + n + (size_t)w + kBack;
  buf.alloc(c, total);
  PFP_HIP(hipMemsetAsync(buf.p, 0, kFront - 1, c->stream));
  PFP_HIP(hipMemsetAsync(buf.p + kFront - 1, kDollar, 1, c->stream));
  if (n) stream_h2d(c, buf.p + kFront, n, [&](uint8_t *pin, uint64_t off, uint64_t len) { par_pread(fd, file_off + off, pin, len); });
  restage_tail(c, n, w);
}
void StagedText::restage_tail(pfp_ctx *c, uint64_t new_n, int w_) const {
  PFP_HIP(hipMemsetAsync(buf.p + kFront + new_n, kDollar, (size_t)w_, c->stream));
  PFP_HIP(hipMemsetAsync(buf.p + kFront + new_n + w_, 0, kBack, c->stream));
}

// occ in lexicographic order and the parse as 1-based lexicographic ranks (newscan.cpp:436,456)
__global__ void occ_lex_kernel(uint32_t d, const uint32_t *__restrict__ lexrank, const uint32_t *__restrict__ wocc,
                               uint32_t *__restrict__ occ_lex, uint32_t *__restrict__ word_at_rank) {
  uint32_t j = BID * blockDim.x + threadIdx.x;
  if (j >= d) return;
  uint32_t r = lexrank[j];
  occ_lex[r] = wocc[j];
  if (word_at_rank) word_at_rank[r] = j;
}
__global__ void parse_sym_kernel(uint64_t P, const uint32_t *__restrict__ pid, const uint32_t *__restrict__ lexrank,
                                 uint32_t *__restrict__ sym) {
  uint64_t k = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (k < P) sym[k] = lexrank[pid[k]] + 1;
}
__global__ void sorted_len1_kernel(uint32_t d, const uint32_t *__restrict__ word_at_rank,
                                   const uint32_t *__restrict__ wlen, uint32_t *__restrict__ len1) {
  uint32_t r = BID * blockDim.x + threadIdx.x;
  if (r == 0) len1[d] = 0;
  if (r < d) len1[r] = wlen[word_at_rank[r]] + 1;
}
// .dict in lexicographic order (newscan.cpp:406-438): 8 lanes per word, 16-byte pieces
__global__ __launch_bounds__(256) void dict_permute_kernel(uint32_t d, const uint32_t *__restrict__ word_at_rank,
                                                           const uint64_t *__restrict__ woff,
                                                           const uint32_t *__restrict__ wlen,
                                                           const uint8_t *__restrict__ src,
                                                           const uint64_t *__restrict__ doff, uint8_t *__restrict__ dst) {
  uint64_t t = (uint64_t)BID * 256 + threadIdx.x;
  uint64_t r = t >> 3;
  int l8 = (int)(t & 7);
  if (r >= d) return;
  uint32_t j = word_at_rank[r];
  uint64_t len = (uint64_t)wlen[j] + 1;   // with terminator
  const uint8_t *s = src + woff[j];
  uint8_t *o = dst + doff[r];
  for (uint64_t off = (uint64_t)l8 * 16; off < len; off += 128) {
    if (off + 16 <= len) st16u(o + off, ld16u(s + off));
    else for (uint64_t b = off; b < len; b++) o[b] = s[b];
  }
}

// suffix order of the dictionary in the index width the dictionary's size asks for (use_wide_index)
struct DictOrder {
  bool wide = false;
  SuffixOrderT<uint32_t> so32;
  SuffixOrderT<uint64_t> so64;
  template <class I> SuffixOrderT<I> &get() {
    if constexpr (sizeof(I) == 8) return so64; else return so32;
  }
  uint64_t rounds() const { return wide ? so64.rounds : so32.rounds; }
};
// f(I{}) with I = uint64_t (wide) or uint32_t
template <class F> static void with_width(bool wide, F &&f) {
  if (wide) f(uint64_t{}); else f(uint32_t{});
}

struct Chain {
  StagedText tx;
  DBuf<uint64_t> ends;
  uint64_t n_ends = 0, n_used = 0;
  Dictionary D;
  DictIndex ix;
  DictOrder ord;
  DBuf<uint32_t> occ_lex, word_at_rank, sym;
  ParseBWT pb;
  DBuf<uint64_t> sa_own;      // every SA value (-S) when the caller keeps none (host / file entry points)
  BwtOutputs out;             // outputs of the merge; -s / -e with no caller array: SA values at the run boundaries only (out.sa_c)
};

// narrowing / widening copy of an index array to the host (the staged gsacak.h entry points fix their SA width)
template <class A, class B>
__global__ void convert_kernel(const A *__restrict__ in, uint64_t n, B *__restrict__ out) {
  uint64_t i = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (B)in[i];
}
template <class A, class B>
static void fetch_converted(pfp_ctx *c, const A *d_in, uint64_t n, B *h_out) {
  if constexpr (sizeof(A) == sizeof(B)) {
    d2h(c, (A *)h_out, d_in, n);
    sync(c);
  } else {
    DBuf<B> tmp(c, n);
    hipLaunchKernelGGL((convert_kernel<A, B>), gdim(cdiv(n, TB)), gdim(TB), 0, c->stream, d_in, n, tmp.p);
    PFP_HIP(hipGetLastError());
    d2h(c, h_out, tmp.p, n);
    sync(c);
  }
}

static void check_args(int w, uint64_t p, int flags) {
  PFP_REQUIRE(w >= 4, PFP_EINVAL, "Windows size must be at least 4 (newscan.cpp:537)");
  PFP_REQUIRE(w <= 4096, PFP_EINVAL, "window size above 4096 is not supported");
  PFP_REQUIRE(p >= 10, PFP_EINVAL, "Modulus must be at leas 10 (newscan.cpp:541)");
  PFP_REQUIRE(!((flags & PFP_FLAG_SA) && (flags & (PFP_FLAG_SSA | PFP_FLAG_ESA))), PFP_EINVAL,
              "You can either compute the full SA or a sample of it, not both (bigbwt:59-61)");
  PFP_REQUIRE((flags & ~7) == 0, PFP_EINVAL, "unknown flag bits");
}

// stage 1 on a staged text: scan, dictionary, dictionary suffix order, lexicographic ranks
static void run_parse(pfp_ctx *c, Chain &ch, uint64_t n, int w, uint64_t p, bool want_sai, bool exact_reference_parse,
                      bool dense_sa = false) {
  pfp_stats &st = c->stats;
  {
    PhaseTimer t(c, &st.ms_scan);
    uint32_t n_extra = 0;
    // the staged entry points parse exactly as the reference does (newscan.cpp:168-202, 363-377); the fused chain cuts by
    // its own window hash (pfp_set_window_hash) and splits giant phrases with extra triggers (pfp_set_max_phrase)
    st.parse_density = 1.0;
    if (exact_reference_parse || (!c->max_phrase && !c->fast_triggers)) ch.n_ends = scan_text(c, ch.tx, n, w, p, ch.ends, &ch.n_used);
    else ch.n_ends = scan_text_adaptive(c, ch.tx, n, w, p, c->max_phrase, ch.ends, &ch.n_used, &n_extra);
    st.extra_triggers = n_extra;
    if (c->debug) validate_scan(c, ch.ends, ch.n_ends, ch.n_used, w);
  }
  {
    PhaseTimer t(c, &st.ms_phrases);
    build_dictionary(c, ch.tx, ch.n_used, w, ch.ends, ch.n_ends, want_sai, ch.D);
    if (c->debug) validate_dictionary(c, ch.D, w);
    build_dict_index(c, ch.D, ch.ix);
    if (c->debug) validate_index(c, ch.D, ch.ix);
    // the text and its phrase ends have done their part: dictionary, parse, last and sai are all there is from here on
    ch.tx.buf.release();
    ch.ends.release();
  }
  {
    PhaseTimer t(c, &st.ms_sa_dict);
    // BWT only: the merge records ride in the spare bits of the first-round keys (SuffixOrder::paybits)
    const WordView wv = word_view(ch.D, ch.ix);
    const SlotPayloadSrc pay{wv, ch.D.wocc.p, w};
    ch.ord.wide = prefer_wide_index(c, ch.D.dsize);      // 32- or 64-bit dictionary positions (bigbwt:130-151)
    with_width(ch.ord.wide, [&](auto tag) {
      using I = decltype(tag);
      auto &so = ch.ord.get<I>();
      so.rep_hint = (double)ch.n_used / (double)std::max<uint64_t>(ch.D.dsize, 1);
      sort_dict_suffixes<I>(c, ch.D.bytes.p, ch.D.dsize, wv, so, dense_sa ? nullptr : &pay);
      if (c->debug) validate_suffix_order<I>(c, ch.D.bytes.p, so, true, "dict SA");
      compute_lexrank<I>(c, ch.D, so, ch.ix);
      if (!c->debug) { so.rank.release(); so.tab.release(); }      // the merge reads sa / grp / skeys only
    });
    if (c->debug) validate_lexrank(c, ch.D, ch.ix);
    const uint32_t d = (uint32_t)ch.D.d;
    ch.occ_lex.alloc(c, d); ch.word_at_rank.alloc(c, d);
    hipLaunchKernelGGL(occ_lex_kernel, gdim(cdiv(d, TB)), gdim(TB), 0, c->stream, d, ch.ix.lexrank.p, ch.D.wocc.p,
                       ch.occ_lex.p, ch.word_at_rank.p);
    ch.sym.alloc(c, ch.D.P);
    hipLaunchKernelGGL(parse_sym_kernel, gdim(cdiv(ch.D.P, TB)), gdim(TB), 0, c->stream, ch.D.P, ch.D.pid.p,
                       ch.ix.lexrank.p, ch.sym.p);
    PFP_HIP(hipGetLastError());
  }
  st.n = ch.n_used; st.n_phrases = ch.D.P; st.n_words = ch.D.d; st.dict_size = ch.D.dsize;
  st.sa_rounds_dict = ch.ord.rounds(); st.hash_reseeds = ch.D.reseeds; st.index_bits = ch.ord.wide ? 64 : 32;
}

// d_sa == nullptr with SA flags: the SA values live in buffers of the chain, allocated when the suffix sorter has
// given its scratch back - all of them for -S (ch.sa_own), those at the run boundaries of the BWT for -s / -e
// (ch.out.sa_c); the caller never sees them, only what is sampled / packed from them
static void run_chain_dev(pfp_ctx *c, Chain &ch, uint64_t n, int w, uint64_t p, int flags, uint8_t *d_bwt,
                          uint64_t *d_sa, uint64_t *n_used) {
  pfp_stats &st = c->stats;
  st = pfp_stats{};
  auto t0 = std::chrono::steady_clock::now();
  run_parse(c, ch, n, w, p, flags != 0, false, (flags & PFP_FLAG_SA) != 0);
  {
    PhaseTimer t(c, &st.ms_sa_parse);
    parse_bwt(c, ch.sym.p, ch.D.P, ch.D.last.p, flags ? ch.D.sai.p : nullptr, ch.occ_lex.p, ch.D.d, ch.pb);
    st.sa_rounds_parse = ch.pb.rounds;
    if (c->debug) validate_parse_bwt(c, ch.pb);
  }
  {
    PhaseTimer t(c, &st.ms_merge);
    BwtOutputs &bo = ch.out;
    if ((flags & PFP_FLAG_SA) && !d_sa) { ch.sa_own.alloc(c, ch.n_used + 1); d_sa = ch.sa_own.p; }
    bo.d_bwt = d_bwt; bo.d_sa = d_sa;
    with_width(ch.ord.wide, [&](auto tag) {
      using I = decltype(tag);
      merge_bwt<I>(c, ch.D, ch.ix, ch.ord.get<I>(), ch.pb, ch.occ_lex.p, w, flags, ch.n_used + 1, bo);
    });
    st.hard_groups = bo.hard_groups; st.hard_chars = bo.hard_chars;
    st.hard_big_groups = bo.hard_big_groups; st.hard_max_chars = bo.hard_max_chars; st.hard_max_members = bo.hard_max_members;
    st.hard_minor_groups = bo.hard_minor_groups; st.hard_minor_chars = bo.hard_minor_chars;
  }
  sync(c);
  st.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  *n_used = ch.n_used;
}

template <class T>
static T *host_alloc(size_t count) {
  T *p = (T *)malloc((count ? count : 1) * sizeof(T));
  if (!p) throw Error(PFP_ENOMEM, "host malloc failed");
  return p;
}

// device bytes -> a fresh host buffer, streamed through the pinned buffers (the first touch of the fresh pages
// and the copy are spread over a few threads)
static uint8_t *fetch_bytes(pfp_ctx *c, const uint8_t *d_src, uint64_t nbytes) {
  uint8_t *h = nullptr;
  if (nbytes >= (64u << 20)) {      // large result: 2 MiB pages where the kernel offers them (hundreds of first-touch faults, not hundreds of thousands)
    void *q = nullptr;
    if (posix_memalign(&q, 2u << 20, nbytes) == 0 && q) { (void)madvise(q, nbytes, MADV_HUGEPAGE); h = (uint8_t *)q; }
  }
  if (!h) h = host_alloc<uint8_t>(nbytes);
  stream_d2h(c, d_src, nbytes, [&](const uint8_t *pin, uint64_t off, uint64_t len) { par_memcpy(h + off, pin, len); });
  return h;
}
// the reference's output files from the device results of a finished chain, produced one after the other into
// `sink(name, device pointer, bytes)` (host buffers of a pfp_bwt_result, or files)
template <class Sink>
static void emit_outputs(pfp_ctx *c, const uint8_t *d_bwt, const SaView &d_sa, uint64_t n_out, int flags, Sink &&sink) {
  sink("bwt", d_bwt, n_out);
  if (flags & PFP_FLAG_SA) {                       // .sa: n entries, SA[0]=n omitted (pfbwt.cpp:158-162)
    uint64_t cnt = n_out - 1;
    DBuf<uint8_t> packed(c, cnt * 5 + 16);
    PFP_REQUIRE(d_sa.dense, PFP_EINVAL, "full SA output without the SA values");
    pack5_dev(c, d_sa.dense + 1, cnt, packed.p);
    sink("sa", packed.p, cnt * 5);
    sync(c);
  }
  if (flags & PFP_FLAG_SSA) {
    DBuf<uint8_t> pairs;
    uint64_t k = sample_runs_dev(c, d_bwt, d_sa, n_out, false, pairs);
    sink("ssa", pairs.p, k * 10);
    sync(c);
  }
  if (flags & PFP_FLAG_ESA) {
    DBuf<uint8_t> pairs;
    uint64_t k = sample_runs_dev(c, d_bwt, d_sa, n_out, true, pairs);
    sink("esa", pairs.p, k * 10);
    sync(c);
  }
  sync(c);
}
static void fetch_outputs(pfp_ctx *c, const uint8_t *d_bwt, const SaView &d_sa, uint64_t n_out, int flags,
                          pfp_bwt_result *out) {
  emit_outputs(c, d_bwt, d_sa, n_out, flags, [&](const char *name, const uint8_t *d, uint64_t bytes) {
    uint8_t *h = fetch_bytes(c, d, bytes);
    if (name[0] == 'b') { out->bwt = h; out->bwt_size = bytes; }
    else if (name[0] == 's' && name[1] == 'a') { out->sa = h; out->sa_bytes = bytes; }
    else if (name[0] == 's') { out->ssa = h; out->ssa_bytes = bytes; }
    else { out->esa = h; out->esa_bytes = bytes; }
  });
}
// device bytes -> file (created / truncated), streamed through the pinned buffers
static void write_dev_file(pfp_ctx *c, const std::string &path, uint64_t file_offset, const uint8_t *d_src, uint64_t nbytes, bool trunc) {
  const int fd = open(path.c_str(), O_WRONLY | O_CREAT | (trunc ? O_TRUNC : 0), 0644);
  PFP_REQUIRE(fd >= 0, PFP_EINVAL, "cannot open " + path + ": " + strerror(errno));
  if (trunc && nbytes) (void)!ftruncate(fd, (off_t)(file_offset + nbytes));      // the final size at once: the writers only fill pages
  bool ok = true;
  std::string werr;
  // (round 4, measured and dropped: a shared mapping of the output file filled by eight threads - buffered pwrite()s to one file
  //  serialise on the inode - was SLOWER into /dev/shm, 2.9-3.3 s against 1.95 s for 13.7 GB: faulting fresh pages in through a
  //  mapping costs more than the write path's own allocation.  profiles/r04_cli_probe_mmap_output.txt)
  try {
    // a chunk can be written by several threads, each its own range at its own offset (PFP_PWRITE_THREADS; pfthreads.hpp:369-376
    // has every worker pwrite its range).  Default one: on tmpfs more writers only contend (1.1 GB: 172 ms with one
    // thread, 220-290 ms with 2-8, MI355X box).
    std::mutex mu;
    stream_d2h(c, d_src, nbytes, [&](const uint8_t *h, uint64_t off, uint64_t len) {
      static const unsigned wthreads = []() { const char *e = getenv("PFP_PWRITE_THREADS"); return e ? (unsigned)atoi(e) : 1u; }();
      const unsigned hw = std::thread::hardware_concurrency();
      const uint64_t T = std::min<uint64_t>(std::min<uint64_t>(wthreads ? wthreads : 1, hw ? hw : 1), std::max<uint64_t>(len >> 22, 1));
      const uint64_t part = ((len + T - 1) / T + 4095) & ~uint64_t(4095);
      auto work = [&](uint64_t lo, uint64_t hi) {
        uint64_t done = lo;
        while (done < hi) {
          const ssize_t w = pwrite(fd, h + done, hi - done, (off_t)(file_offset + off + done));
          if (w <= 0) { std::lock_guard<std::mutex> g(mu); ok = false; werr = strerror(errno); return; }
          done += (uint64_t)w;
        }
      };
      if (T <= 1) { work(0, len); return; }
      std::vector<std::thread> th;
      for (uint64_t k = 0; k < T && k * part < len; k++) th.emplace_back(work, k * part, std::min(len, (k + 1) * part));
      for (auto &t : th) t.join();
    });
  } catch (...) { close(fd); throw; }
  sync(c);
  PFP_REQUIRE(close(fd) == 0 && ok, PFP_EINVAL, "error writing " + path + ": " + werr);
}


// An output file whose OWN pages are the target of the device -> host copy (round 4).  write_dev_file() above moves every byte
// twice on the host side of PCIe - copy engine -> pinned buffer, pwrite() -> the file's pages - and a file takes one writer at a
// time: ~6 GB/s into /dev/shm, 2.2 of the 3.5 s of the 12.6 GB command-line run.  The .bwt's size is known before the text is
// read, so: create the file at its final size, map it, and - on helper threads, beside the text input and the chain - fault
// its pages in and register the mapping with the runtime piece by piece; when the BWT exists it crosses PCIe once, at the copy
// engine's rate, straight into the file (tools/microbench/regout.hip: 31 GB/s into a populated, registered mapping against
// 4.1-4.7 for the pinned-buffer path).  Files in a memory file system only (anything else: the pwrite path); every failure on
// the way - no mapping, a piece the runtime will not register - falls back to the pwrite path for the whole file.
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
struct MappedOut {
  uint64_t kPiece = 128ull << 20;      // (a small file in smaller pieces: its first copy starts sooner)
  std::string path;
  int fd = -1, device = 0;
  uint8_t *m = nullptr;
  uint64_t bytes = 0, npieces = 0;
  std::vector<int> state;                 // per piece: 0 pending, 1 registered, -1 failed (under mu)
  std::mutex mu;
  std::condition_variable cv;
  std::atomic<bool> cancel{false};
  std::vector<std::thread> workers;
  hipStream_t cs = nullptr;               // the copies' own stream: the run sampling that follows the BWT overlaps them
  hipEvent_t ev = nullptr;
  bool copying = false, done = false;
  std::chrono::steady_clock::time_point t_start;
  double s_populate = 0, s_register = 0, s_ready = 0, s_waited = 0;      // PFP_TRACE_HOST (under mu)

  static bool wanted(uint64_t nbytes) {
    static const int mode = []() { const char *e = getenv("PFP_MAP_OUTPUT"); return e ? atoi(e) : -1; }();      // 0: never
    static const uint64_t min_bytes = []() { const char *e = getenv("PFP_MAP_MIN_BYTES"); return e ? (uint64_t)atoll(e) : (64ull << 20); }();      // (tests: small files too)
    return mode != 0 && nbytes >= std::max<uint64_t>(min_bytes, 1);
  }
  // false: this file is written the ordinary way
  bool start(pfp_ctx *c, const std::string &path_, uint64_t nbytes) {
    if (!wanted(nbytes)) return false;
    path = path_; bytes = nbytes; device = c->device;
    // (a disk file system tracks dirty pages through write faults, which a copy engine does not take: pwrite there)
    struct statfs sf;
    const size_t slash = path.rfind('/');
    const std::string dir = slash == std::string::npos ? std::string(".") : (slash == 0 ? std::string("/") : path.substr(0, slash));
    if (statfs(dir.c_str(), &sf) != 0 || (unsigned long)sf.f_type != 0x01021994ul /* tmpfs */) return false;
    fd = open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return false;      // (the ordinary path reports it)
    if (fstatfs(fd, &sf) != 0 || (unsigned long)sf.f_type != 0x01021994ul || ftruncate(fd, (off_t)bytes) != 0) { close(fd); fd = -1; unlink(path.c_str()); return false; }
    t_start = std::chrono::steady_clock::now();
    void *q = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (q == MAP_FAILED) { close(fd); fd = -1; unlink(path.c_str()); return false; }
    m = (uint8_t *)q;
    static const uint64_t piece_mb = []() { const char *e = getenv("PFP_MAP_PIECE_MB"); return e ? (uint64_t)atoll(e) : (uint64_t)32; }();
    kPiece = std::min<uint64_t>(std::max<uint64_t>(piece_mb, 2) << 20, std::max<uint64_t>(2ull << 20, (bytes / 8 + (2u << 20) - 1) & ~uint64_t((2u << 20) - 1)));
    npieces = (bytes + kPiece - 1) / kPiece;
    state.assign(npieces, 0);
    if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError(); abandon(true); return false;
    }
    workers.emplace_back([this]() { work(); });
    return true;
  }
  // ONE helper: page allocation in one file does not scale over threads - 1.96 s for 12.6 GB from one thread, 3.0-3.2 s from
  // 4-16, and an allocating thread beside a mapping one slows both (tools/microbench/regout.hip, profiles/r04_regout_*.txt).
  // Per piece: its pages allocated by fallocate (19 GB/s; 6 when the registration's faults have to allocate them), then
  // mapped and pinned by the registration.  Pieces of 32 MB: a registration in flight holds up the calling thread's own
  // allocations and copies (text in 0.35 -> 0.9 s, cold chain 0.6 -> 0.9 s with 128 MB pieces), smaller ones cost the helper more.
  void work() {
    (void)hipSetDevice(device);
    for (uint64_t i = 0; i < npieces && !cancel.load(); i++) {
      const uint64_t off = i * kPiece, len = std::min(kPiece, bytes - off);
      const auto a0 = std::chrono::steady_clock::now();
      // (a file system without room: no page of the piece may be touched through the mapping - that would be a SIGBUS where
      //  write() says ENOSPC; the piece counts as refused and the pwrite path reports the error)
      const bool have_pages = fallocate(fd, 0, (off_t)off, (off_t)len) == 0;
      const auto a1 = std::chrono::steady_clock::now();
      static const long fail_at = []() { const char *t = getenv("PFP_TEST_MAP_FAIL"); return t ? atol(t) : -1L; }();      // (test hook: piece k is refused)
      const hipError_t e = (long)i == fail_at || !have_pages ? hipErrorOutOfMemory : hipHostRegister(m + off, len, hipHostRegisterDefault);
      if (e != hipSuccess) (void)hipGetLastError();
      const auto a2 = std::chrono::steady_clock::now();
      {
        std::lock_guard<std::mutex> g(mu);
        state[i] = e == hipSuccess ? 1 : -1;
        s_populate += std::chrono::duration<double>(a1 - a0).count(); s_register += std::chrono::duration<double>(a2 - a1).count();
        s_ready = std::chrono::duration<double>(a2 - t_start).count();
      }
      cv.notify_all();
      if (e != hipSuccess) return;      // (the copy gives up at this piece)
    }
  }
  // device bytes [0, nbytes) -> the file, behind what the context's stream holds; returns at once (finish() waits).
  // false: a piece could not be registered - nothing usable was written, the caller takes the ordinary path.
  bool write(pfp_ctx *c, const uint8_t *d_src, uint64_t nbytes) {
    PFP_REQUIRE(nbytes <= bytes, PFP_EINVAL, "mapped output smaller than the result");
    PFP_HIP(hipEventRecord(ev, c->stream));
    PFP_HIP(hipStreamWaitEvent(cs, ev, 0));
    copying = true;
    for (uint64_t i = 0; i < npieces && i * kPiece < nbytes; i++) {
      {
        const auto a0 = std::chrono::steady_clock::now();
        std::unique_lock<std::mutex> g(mu); cv.wait(g, [&]() { return state[i] != 0; });
        s_waited += std::chrono::duration<double>(std::chrono::steady_clock::now() - a0).count();
        if (state[i] < 0) return false;
      }
      const uint64_t off = i * kPiece, len = std::min(kPiece, nbytes - off);
      PFP_HIP(hipMemcpyAsync(m + off, d_src + off, len, hipMemcpyDeviceToHost, cs));
    }
    return true;
  }
  void join_workers() { cancel.store(true); for (auto &t : workers) if (t.joinable()) t.join(); workers.clear(); }
  void unregister_all() {
    join_workers();
    if (m) for (uint64_t i = 0; i < npieces; i++) if (state[i] == 1) { (void)hipHostUnregister(m + i * kPiece); state[i] = 0; }
    if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
    if (cs) { (void)hipStreamDestroy(cs); cs = nullptr; }
  }
  // waits for the copies, gives the file its final length and lets go of the mapping.  Taking 12.6 GB out of the page table
  // costs a quarter of a second: on a thread the context joins later - the file is complete before that.  (Measured and
  // dropped: unmapping every piece as soon as its copy has landed - the address space's lock, taken for every piece, held up
  // this thread's own mappings and stream calls: files out 45 -> 125 ms at 0.79 GB.)
  void finish(pfp_ctx *c, uint64_t final_bytes) {
    const auto a0 = std::chrono::steady_clock::now();
    if (copying) PFP_HIP(hipStreamSynchronize(cs));
    copying = false;
    const auto a1 = std::chrono::steady_clock::now();
    unregister_all();
    bool ok = final_bytes == bytes || ftruncate(fd, (off_t)final_bytes) == 0;
    ok = (close(fd) == 0) && ok; fd = -1;
    done = true;
    if (getenv("PFP_TRACE_HOST"))
      fprintf(stderr, "[pfp] %s, %.2f GB through its mapping: pages allocated %.3f s, mapped and registered %.3f s (%zu pieces), all ready %.3f s after the start; the copy waited %.3f s for pieces, %.3f s for the copy engine, %.3f s to unregister and close\n",
              path.c_str(), bytes / 1e9, s_populate, s_register, (size_t)npieces, s_ready, s_waited, std::chrono::duration<double>(a1 - a0).count(),
              std::chrono::duration<double>(std::chrono::steady_clock::now() - a1).count());
    uint8_t *mm = m; const uint64_t len = bytes; m = nullptr;
    // (last: an unmapping in flight holds the address space's lock, and creating a thread or destroying a stream would wait for it)
    c->background.emplace_back([mm, len]() { munmap(mm, len); });
    PFP_REQUIRE(ok, PFP_EINVAL, "error writing " + path + ": " + strerror(errno));
  }
  // drop everything; the (incomplete) file goes too unless the ordinary path is about to rewrite it
  void abandon(bool remove) {
    if (copying && cs) (void)hipStreamSynchronize(cs);
    copying = false;
    unregister_all();
    if (m) { munmap(m, bytes); m = nullptr; }
    if (fd >= 0) { close(fd); fd = -1; if (remove) unlink(path.c_str()); }
    (void)hipGetLastError();
    done = true;
  }
  bool active() const { return m != nullptr && !done; }
  ~MappedOut() { if (!done && (m || fd >= 0)) abandon(true); }
};
static void join_background(pfp_ctx *c) {
  for (auto &t : c->background) if (t.joinable()) t.join();
  c->background.clear();
}

}  // namespace pfp

// ======================================================================== extern "C"

#define PFP_TRY(ctx) try {                                                                \
  if ((ctx) && !(ctx)->pool.corrupt.empty()) throw ::pfp::Error(PFP_EHIP, (ctx)->pool.corrupt);
#define PFP_CATCH(ctx)                                                                    \
  }                                                                                       \
  catch (const pfp::Error &e) { if (ctx) (ctx)->err = e.what(); (void)hipGetLastError(); return e.code; } \
  catch (const std::bad_alloc &) { if (ctx) (ctx)->err = "host out of memory"; return PFP_ENOMEM; }   \
  catch (const std::exception &e) { if (ctx) (ctx)->err = e.what(); return PFP_EHIP; }

extern "C" {

const char *pfp_version(void) { return "pfpgpu 0.1 (gfx950, wave64; prefix-free parsing BWT)"; }

const char *pfp_strerror(int code) {
  switch (code) {
    case PFP_OK: return "ok";
    case PFP_EINVAL: return "invalid argument";
    case PFP_ENODEV: return "no usable HIP device";
    case PFP_EHIP: return "HIP runtime error";
    case PFP_ECOLLISION: return "phrase hash collision";
    case PFP_ELIMIT: return "size limit exceeded";
    case PFP_EFORMAT: return "inconsistent input";
    case PFP_ENOMEM: return "out of memory";
    case PFP_ESHORT: return "input too short";
    default: return "unknown error";
  }
}

int pfp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}
int pfp_ctx_create(pfp_ctx **out, int device) {
  if (!out) return PFP_EINVAL;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    (void)hipGetLastError();
    return PFP_ENODEV;
  }
  pfp_ctx *c = new (std::nothrow) pfp_ctx();
  if (!c) return PFP_ENOMEM;
  try {
    c->device = device;
    PFP_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    PFP_HIP(hipGetDeviceProperties(&prop, device));
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    PFP_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->pool.stream = c->stream;
    { const char *pd = getenv("PFP_POOL_DEBUG"); c->pool.debug = pd && pd[0] && pd[0] != '0'; }
    { const char *tl = getenv("PFP_TEST_POOL_LIMIT"); if (tl) c->pool.test_limit = (size_t)strtoull(tl, nullptr, 10); }
    c->pool.trace = getenv("PFP_TRACE_POOL") != nullptr;
    { size_t fr = 0, tot = 0; if (hipMemGetInfo(&fr, &tot) == hipSuccess) c->pool.soft_limit = tot / 10 * 7; else (void)hipGetLastError(); }
    PFP_HIP(hipHostMalloc((void **)&c->h_scalars, 16 * sizeof(uint64_t), hipHostMallocDefault));
    const char *dbg = getenv("PFP_DEBUG");
    c->debug = dbg && dbg[0] && dbg[0] != '0';
    { const char *fw = getenv("PFP_FORCE_IDX64"); c->force_wide = fw && fw[0] && fw[0] != '0'; }
    { const char *wh = getenv("PFP_WINDOW_HASH"); if (wh && !strcmp(wh, "kr")) c->fast_triggers = false; }
    { const char *pd = getenv("PFP_PARSE_DENSITY"); if (pd && atof(pd) >= 0.01 && atof(pd) <= 64.0) c->parse_density = atof(pd); }
    (void)hipGetLastError();
  } catch (const pfp::Error &e) {
    (void)hipGetLastError();
    delete c;
    return e.code == PFP_EHIP ? PFP_ENODEV : e.code;
  }
  *out = c;
  return PFP_OK;
}

struct K1Scratch { pfp::DBuf<uint16_t> flags; pfp::DBuf<uint32_t> bcnt; pfp::DBuf<unsigned long long> fbad; uint64_t n = 0; };
static pfp::StagedText *&staged_of(pfp_ctx *c) { return *reinterpret_cast<pfp::StagedText **>(&c->staged); }

void pfp_dist_release(pfp_ctx *c);
void pfp_ctx_destroy(pfp_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  join_background(c);
  pfp_dist_release(c);
  delete staged_of(c);
  delete reinterpret_cast<K1Scratch *>(c->k1scratch);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  c->kt.destroy();
  if (getenv("PFP_TRACE_HOST")) fprintf(stderr, "[pfp] host waits on the stream over the context's life: %llu\n", (unsigned long long)c->n_syncs);
  c->pool.print_peak();
  c->pool.destroy();
  if (c->h_scalars) (void)hipHostFree(c->h_scalars);
  for (int k = 0; k < 2; k++) {
    if (c->pin[k]) (void)hipHostFree(c->pin[k]);
    if (c->pin_ev[k]) (void)hipEventDestroy(c->pin_ev[k]);
  }
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char *pfp_last_error(const pfp_ctx *c) { return c ? c->err.c_str() : "null context"; }
int pfp_debug_check(pfp_ctx *c) {
  if (!c) return PFP_EINVAL;
  if (c->pool.corrupt.empty()) return PFP_OK;
  c->err = c->pool.corrupt;
  return PFP_EHIP;
}
void pfp_pool_trim(pfp_ctx *c) {      // give the cached device blocks back to the driver (several contexts sharing one GPU)
  if (!c) return;
  (void)hipSetDevice(c->device);
  c->pool.trim();
}
int pfp_get_mem_stats(const pfp_ctx *c, uint64_t out[4]) {
  if (!c || !out) return PFP_EINVAL;
  out[0] = c->pool.total_bytes; out[1] = c->pool.peak_bytes; out[2] = c->pool.live_bytes; out[3] = c->pool.debug ? c->pool.debug_blocks : 0;
  return PFP_OK;
}
int pfp_get_pool_counters(const pfp_ctx *c, uint64_t out[2]) {
  if (!c || !out) return PFP_EINVAL;
  out[0] = c->pool.driver_allocs; out[1] = c->pool.trims;
  return PFP_OK;
}
void *pfp_ctx_stream(pfp_ctx *c) { return c ? (void *)c->stream : nullptr; }
void pfp_free(void *p) { free(p); }
void pfp_set_profiling(pfp_ctx *c, int on) { if (c) c->profiling = on != 0; }
void pfp_set_kernel_trace(pfp_ctx *c, int on) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  c->kt.resolve();
  c->kt.agg.clear();
  c->kt.on = on != 0;
}
int pfp_get_kernel_trace(pfp_ctx *c, pfp_kernel_stat *out, int cap) {
  if (!c) return PFP_EINVAL;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  c->kt.resolve();
  int k = 0;
  for (auto &kv : c->kt.agg) {
    if (out && k < cap) {
      memset(&out[k], 0, sizeof out[k]);
      strncpy(out[k].name, kv.first.c_str(), sizeof out[k].name - 1);
      out[k].launches = kv.second.launches; out[k].total_ms = kv.second.ms; out[k].algo_bytes = kv.second.bytes;
    }
    k++;
  }
  return k;
}
void pfp_set_max_phrase(pfp_ctx *c, uint64_t max_phrase) { if (c) c->max_phrase = max_phrase; }
void pfp_set_window_hash(pfp_ctx *c, int fast) { if (c) c->fast_triggers = fast != 0; }
int pfp_set_parse_density(pfp_ctx *c, double density) {
  if (!c || !(density == 0.0 || (density >= 0.01 && density <= 64.0))) return PFP_EINVAL;
  c->parse_density = density;
  return PFP_OK;
}
int pfp_set_index_bits(pfp_ctx *c, int bits) {
  if (!c || (bits != 0 && bits != 32 && bits != 64)) return PFP_EINVAL;
  c->force_wide = bits == 64;
  c->force_narrow = bits == 32;
  return PFP_OK;
}
int pfp_get_stats(const pfp_ctx *c, pfp_stats *st) {
  if (!c || !st) return PFP_EINVAL;
  *st = c->stats;
  return PFP_OK;
}

void pfp_parse_result_free(pfp_parse_result *r) {
  if (!r) return;
  free(r->dict); free(r->occ); free(r->parse); free(r->last); free(r->sai);
  memset(r, 0, sizeof *r);
}
void pfp_bwt_result_free(pfp_bwt_result *r) {
  if (!r) return;
  free(r->bwt); free(r->sa); free(r->ssa); free(r->esa);
  memset(r, 0, sizeof *r);
}

// ---------------------------------------------------------------- stage 1a
int pfp_scan(pfp_ctx *c, const uint8_t *text, uint64_t n, int w, uint64_t p, uint64_t **ends, uint64_t *n_ends,
             uint64_t *n_used) {
  if (!c || (!text && n) || !ends || !n_ends) return PFP_EINVAL;
  *ends = nullptr; *n_ends = 0;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  PFP_REQUIRE(w >= 1 && w <= 4096 && p >= 1, PFP_EINVAL, "bad window or modulus");
  StagedText tx;
  tx.stage(c, text, false, n, w);
  DBuf<uint64_t> d_ends;
  uint64_t used = n;
  uint64_t k = scan_text(c, tx, n, w, p, d_ends, &used);
  uint64_t *h = host_alloc<uint64_t>(k);
  if (k) d2h(c, h, d_ends.p, k);
  sync(c);
  *ends = h; *n_ends = k;
  if (n_used) *n_used = used;
  return PFP_OK;
  PFP_CATCH(c)
}

// ---------------------------------------------------------------- stage 1
int pfp_parse(pfp_ctx *c, const uint8_t *text, uint64_t n, int w, uint64_t p, int want_sai, pfp_parse_result *out) {
  if (!c || (!text && n) || !out) return PFP_EINVAL;
  memset(out, 0, sizeof *out);
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  check_args(w, p, 0);
  c->stats = pfp_stats{};
  Chain ch;
  ch.tx.stage(c, text, false, n, w);
  run_parse(c, ch, n, w, p, want_sai != 0, true, true);      // the staged parser's outputs do not depend on the key payload
  const uint32_t d = (uint32_t)ch.D.d;
  const uint64_t P = ch.D.P;
  // .dict in lexicographic order
  DBuf<uint32_t> len1(c, (size_t)d + 1);
  DBuf<uint64_t> doff(c, (size_t)d + 1);
  hipLaunchKernelGGL(sorted_len1_kernel, gdim(cdiv(d, TB)), gdim(TB), 0, c->stream, d, ch.word_at_rank.p, ch.D.wlen.p,
                     len1.p);
  exclusive_sum_u32_u64(c, len1.p, doff.p, (size_t)d + 1);
  DBuf<uint8_t> sdict(c, ch.D.dsize + 64);
  PFP_HIP(hipMemsetAsync(sdict.p + ch.D.dsize - 1, 0, 1, c->stream));
  hipLaunchKernelGGL(dict_permute_kernel, gdim(cdiv((uint64_t)d * 8, TB)), gdim(TB), 0, c->stream, d, ch.word_at_rank.p,
                     ch.D.woff.p, ch.D.wlen.p, ch.D.bytes.p, doff.p, sdict.p);
  PFP_HIP(hipGetLastError());
  out->n_used = ch.n_used;
  out->dict_size = ch.D.dsize; out->n_words = d; out->n_phrases = P;
  out->dict = host_alloc<uint8_t>(ch.D.dsize);
  out->occ = host_alloc<uint32_t>(d);
  out->parse = host_alloc<uint32_t>(P);
  out->last = host_alloc<uint8_t>(P);
  d2h(c, out->dict, sdict.p, ch.D.dsize);
  d2h(c, out->occ, ch.occ_lex.p, d);
  d2h(c, out->parse, ch.sym.p, P);
  d2h(c, out->last, ch.D.last.p, P);
  if (want_sai) {
    DBuf<uint8_t> packed(c, P * 5);
    pack5_dev(c, ch.D.sai.p, P, packed.p);
    out->sai = host_alloc<uint8_t>(P * 5);
    d2h(c, out->sai, packed.p, P * 5);
    sync(c);
  }
  sync(c);
  return PFP_OK;
  PFP_CATCH(c)
}

// ---------------------------------------------------------------- suffix sorting
int pfp_sacak_int(pfp_ctx *c, const uint32_t *s, uint32_t *SA, uint64_t n, uint64_t k) {
  (void)k;
  if (!c || !s || !SA) return PFP_EINVAL;   // gsacak.c:2498 returns -1 on NULL
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  PFP_REQUIRE(n >= 1, PFP_EINVAL, "empty string");
  PFP_REQUIRE(s[n - 1] == 0, PFP_EFORMAT, "sacak_int: last symbol must be 0");
  DBuf<uint32_t> ds(c, n);
  h2d(c, ds.p, s, n);
  SuffixOrder so;
  sort_int_suffixes(c, ds.p, n, so, k ? (uint32_t)std::min<uint64_t>(k - 1, 0xFFFFFFFFull) : 0xFFFFFFFFu);      // symbols < k (k = 0: not stated)
  d2h(c, SA, so.sa.p, n);
  sync(c);
  return PFP_OK;
  PFP_CATCH(c)
}

}  // extern "C"
// suffix array of a byte string ending in a unique 0 into a host array of OUT-wide entries
template <class OUT>
static void sacak_any(pfp_ctx *c, const uint8_t *s, OUT *SA, uint64_t n) {
  PFP_REQUIRE(n >= 1, PFP_EINVAL, "empty string");
  PFP_REQUIRE(s[n - 1] == 0, PFP_EFORMAT, "sacak: last symbol must be 0");
  PFP_REQUIRE(sizeof(OUT) == 8 || n < 0xFFFFFFF0ull, PFP_ELIMIT, "text of 4 GiB or more needs the 64-bit entry point (simplebwt64)");
  DBuf<uint8_t> ds(c, n + 64);
  h2d(c, ds.p, s, n);
  PFP_HIP(hipMemsetAsync(ds.p + n, 0, 64, c->stream));
  with_width(use_wide_index(c, n), [&](auto tag) {
    using I = decltype(tag);
    SuffixOrderT<I> so;
    sort_byte_suffixes<I>(c, ds.p, n, so);
    fetch_converted<I, OUT>(c, so.sa.p, n, SA);
  });
}
extern "C" {
int pfp_sacak(pfp_ctx *c, const uint8_t *s, uint32_t *SA, uint64_t n) {
  if (!c || !s || !SA) return PFP_EINVAL;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  sacak_any<uint32_t>(c, s, SA, n);
  return PFP_OK;
  PFP_CATCH(c)
}
int pfp_sacak64(pfp_ctx *c, const uint8_t *s, uint64_t *SA, uint64_t n) {
  if (!c || !s || !SA) return PFP_EINVAL;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  sacak_any<uint64_t>(c, s, SA, n);
  return PFP_OK;
  PFP_CATCH(c)
}
int pfp_sacak_int64(pfp_ctx *c, const uint32_t *s, uint64_t *SA, uint64_t n, uint64_t k) {
  if (!c || !s || !SA) return PFP_EINVAL;   // -DM64: uint_t SA entries, int_text stays 32 bits (gsacak.h:42-60)
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  PFP_REQUIRE(n >= 1, PFP_EINVAL, "empty string");
  PFP_REQUIRE(s[n - 1] == 0, PFP_EFORMAT, "sacak_int: last symbol must be 0");
  DBuf<uint32_t> ds(c, n);
  h2d(c, ds.p, s, n);
  SuffixOrder so;
  sort_int_suffixes(c, ds.p, n, so, k ? (uint32_t)std::min<uint64_t>(k - 1, 0xFFFFFFFFull) : 0xFFFFFFFFu);
  fetch_converted<uint32_t, uint64_t>(c, so.sa.p, n, SA);
  return PFP_OK;
  PFP_CATCH(c)
}

}  // extern "C"
// dictionary given as bytes: fill D.{bytes,dsize,d,woff,wlen} and the index
static void dictionary_from_host(pfp_ctx *c, const uint8_t *s, uint64_t n, Dictionary &D, DictIndex &ix) {
  PFP_REQUIRE(n >= 2 && s[n - 1] == kEndOfDict && s[n - 2] == kEndOfWord, PFP_EFORMAT,
              "dictionary must end with 0x01 0x00 (pfbwt.cpp:498-503)");
  D.dsize = n;
  D.bytes.alloc(c, n + 64);
  h2d(c, D.bytes.p, s, n);
  PFP_HIP(hipMemsetAsync(D.bytes.p + n, 0, 64, c->stream));
  word_table_from_bytes(c, D, n);
  build_dict_index(c, D, ix);
}

// gsacak's optional outputs (gsa/gsacak.h:78-105): LCP[i] = length of the common prefix of the suffixes SA[i-1] and
// SA[i], where a separator (1) or the final 0 ends the count (gsa/README.md:76-104); DA[i] = index of the string
// the suffix SA[i] starts in.
// Pass 1: one thread per slot compares its two suffixes 8 bytes at a time, for at most kLcpCap bytes; a pair still equal
// there is flagged.  Pass 2 (only if something was flagged - long exact repeats: an 18 Mb run of N inside one string has
// 18 M pairs with a mean common prefix of 9 MB, which one thread per pair would never finish): the flagged slots in TEXT
// order, 1024 of them per wave.  Neighbours in text order inherit (Kasai et al.: lcp(phi(b+1), b+1) >= lcp(phi(b), b) - 1;
// separators rank as distinct smallest symbols, gsacak.c:2493, so the lemma holds for a collection), the compare itself
// is the whole wave's: 64 lanes x 16 bytes per step, a ballot finds the first difference or end.
constexpr uint32_t kLcpCap = 2048;
constexpr uint32_t kLcpChunk = 1024;
template <class I, class L>
__global__ void lcp_da_kernel(const uint8_t *__restrict__ s, uint64_t n, const I *__restrict__ sa, WordView wv,
                              L *__restrict__ lcp, L *__restrict__ da, uint8_t *__restrict__ longf) {
  const uint64_t t = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const uint64_t b = sa[t];
  if (da) da[t] = (L)word_of(wv, b);
  if (!lcp) return;
  longf[t] = 0;
  if (t == 0) { lcp[0] = 0; return; }
  const uint64_t a = sa[t - 1];
  uint64_t l = 0;
  for (; l < kLcpCap; l += 8) {
    const uint64_t x = ld8u(s + a + l), y = ld8u(s + b + l);
    const uint64_t end = (x - 0x0202020202020202ull) & ~x & 0x8080808080808080ull;      // bytes < 2 of x (lowest flag exact)
    const uint64_t diff = x ^ y;
    if (diff | end) {
      const int fd = diff ? (__builtin_ctzll(diff) >> 3) : 8, fe = end ? (__builtin_ctzll(end) >> 3) : 8;
      lcp[t] = (L)(l + (fd < fe ? fd : fe));
      return;
    }
  }
  longf[t] = 1;      // equal for kLcpCap bytes: pass 2
}
template <class I>
__global__ void lcp_long_pos_kernel(uint64_t m, const uint64_t *__restrict__ slot, const I *__restrict__ sa, uint64_t *__restrict__ pos) {
  const uint64_t j = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (j < m) pos[j] = sa[slot[j]];
}
// one wave per kLcpChunk flagged pairs (sorted by text position b; slot[j] = their suffix-array slot)
template <class I, class L>
__global__ __launch_bounds__(64) void lcp_long_kernel(const uint8_t *__restrict__ s, uint64_t n, const I *__restrict__ sa, uint64_t m,
                                                      const uint64_t *__restrict__ pos, const uint64_t *__restrict__ slot, L *__restrict__ lcp) {
  const uint64_t j0 = (uint64_t)BID * kLcpChunk;
  if (j0 >= m) return;
  const uint64_t j1 = j0 + kLcpChunk < m ? j0 + kLcpChunk : m;
  const int lane = threadIdx.x;
  uint64_t prev_b = ~0ull, prev_l = 0;
  for (uint64_t j = j0; j < j1; j++) {
    const uint64_t b = pos[j], t = slot[j], a = sa[t - 1];      // (slot 0 is never flagged)
    uint64_t l = kLcpCap;
    if (prev_b + 1 == b && prev_l > (uint64_t)kLcpCap + 1) l = prev_l - 1;
    for (;;) {
      const uint64_t off = l + (uint64_t)lane * 16;
      // (the buffer is padded with 64 zero bytes past n: a lane that would read beyond them sees an end instead)
      const bool in = a + off + 16 <= n + 64 && b + off + 16 <= n + 64;
      uint4 x = make_uint4(0u, 0u, 0u, 0u), y = x;
      if (in) { x = ld16u(s + a + off); y = ld16u(s + b + off); }
      const uint32_t xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
      int ev = 16;
#pragma unroll
      for (int q = 3; q >= 0; q--) {
        const uint32_t end = (xs[q] - 0x02020202u) & ~xs[q] & 0x80808080u, diff = xs[q] ^ ys[q];
        if (diff | end) {
          const int fd = diff ? (__builtin_ctz(diff) >> 3) : 4, fe = end ? (__builtin_ctz(end) >> 3) : 4;
          ev = 4 * q + (fd < fe ? fd : fe);
        }
      }
      const unsigned long long hit = __ballot(ev < 16);
      if (hit) {
        const int first = __ffsll((long long)hit) - 1;
        l += (uint64_t)first * 16 + (uint64_t)__shfl(ev, first, 64);
        break;
      }
      l += 1024;
    }
    if (lane == 0) lcp[t] = (L)l;
    prev_b = b; prev_l = l;
  }
}
template <class I, class L>
static void lcp_da_device(pfp_ctx *c, const uint8_t *bytes, uint64_t n, const I *sa, const WordView &wv, L *d_lcp, L *d_da) {
  DBuf<uint8_t> longf(c, d_lcp ? n + 16 : 16);
  if (d_lcp) PFP_HIP(hipMemsetAsync(longf.p + n, 0, 16, c->stream));
  hipLaunchKernelGGL((lcp_da_kernel<I, L>), gdim(cdiv(n, TB)), gdim(TB), 0, c->stream, bytes, n, sa, wv, d_lcp, d_da, longf.p);
  PFP_HIP(hipGetLastError());
  if (!d_lcp) return;
  const uint64_t m = count_flags(c, longf.p, n);
  if (!m) return;
  DBuf<uint64_t> slot(c, m), slot2(c, m), pos(c, m), pos2(c, m), cnt(c, 1);
  select_index<uint64_t>(c, longf.p, slot.p, cnt.p, n);
  hipLaunchKernelGGL(lcp_long_pos_kernel<I>, gdim(cdiv(m, TB)), gdim(TB), 0, c->stream, m, slot.p, sa, pos.p);
  sort_pairs_db(c, pos, pos2, slot, slot2, m, 0, bits_for(n));
  hipLaunchKernelGGL((lcp_long_kernel<I, L>), gdim((unsigned)cdiv64(m, kLcpChunk)), gdim(64), 0, c->stream, bytes, n, sa, m, pos.p, slot.p, d_lcp);
  PFP_HIP(hipGetLastError());
}
template <class OUT, class L>
static void gsacak_any(pfp_ctx *c, const uint8_t *s, OUT *SA, uint64_t n, L *LCP = nullptr, L *DA = nullptr) {
  PFP_REQUIRE(sizeof(OUT) == 8 || n < 0xFFFFFFF0ull, PFP_ELIMIT, "collection of 4 GiB or more needs the 64-bit entry point (gsacak.h -DM64)");
  Dictionary D; DictIndex ix;
  dictionary_from_host(c, s, n, D, ix);
  with_width(use_wide_index(c, n), [&](auto tag) {
    using I = decltype(tag);
    SuffixOrderT<I> so;
    sort_dict_suffixes<I>(c, D.bytes.p, n, word_view(D, ix), so);
    fetch_converted<I, OUT>(c, so.sa.p, n, SA);
    if (LCP || DA) {
      DBuf<L> dl(c, LCP ? n : 1), dd(c, DA ? n : 1);
      lcp_da_device<I, L>(c, D.bytes.p, n, so.sa.p, word_view(D, ix), LCP ? dl.p : (L *)nullptr, DA ? dd.p : (L *)nullptr);
      if (LCP) d2h(c, LCP, dl.p, n);
      if (DA) d2h(c, DA, dd.p, n);
      sync(c);
    }
  });
}
extern "C" {
int pfp_gsacak(pfp_ctx *c, const uint8_t *s, uint32_t *SA, uint64_t n) {
  if (!c || !s || !SA) return PFP_EINVAL;   // gsacak.c:2503
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  gsacak_any<uint32_t, int32_t>(c, s, SA, n);
  return PFP_OK;
  PFP_CATCH(c)
}
int pfp_gsacak64(pfp_ctx *c, const uint8_t *s, uint64_t *SA, uint64_t n) {
  if (!c || !s || !SA) return PFP_EINVAL;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  gsacak_any<uint64_t, int64_t>(c, s, SA, n);
  return PFP_OK;
  PFP_CATCH(c)
}
int pfp_gsacak_lcp_da(pfp_ctx *c, const uint8_t *s, uint32_t *SA, int32_t *LCP, int32_t *DA, uint64_t n) {
  if (!c || !s || !SA) return PFP_EINVAL;   // gsacak.c:2503; LCP and DA are optional like there
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  gsacak_any<uint32_t, int32_t>(c, s, SA, n, LCP, DA);
  return PFP_OK;
  PFP_CATCH(c)
}
int pfp_gsacak_lcp_da64(pfp_ctx *c, const uint8_t *s, uint64_t *SA, int64_t *LCP, int64_t *DA, uint64_t n) {
  if (!c || !s || !SA) return PFP_EINVAL;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  gsacak_any<uint64_t, int64_t>(c, s, SA, n, LCP, DA);
  return PFP_OK;
  PFP_CATCH(c)
}

// ---------------------------------------------------------------- stage 2
int pfp_bwtparse(pfp_ctx *c, const uint32_t *parse, uint64_t P, const uint8_t *last, const uint8_t *sai,
                 const uint32_t *occ, uint64_t n_words, uint32_t *ilist, uint8_t *bwlast, uint8_t *bwsai) {
  if (!c || !parse || !last || !occ || !ilist || !bwlast || (sai && !bwsai)) return PFP_EINVAL;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  PFP_REQUIRE(P >= 2, PFP_ESHORT, "parse has fewer than 2 phrases (bwtparse.c:244)");
  PFP_REQUIRE(P <= 0xFFFFFFFEull, PFP_ELIMIT, "Input containing more than 2^32-2 phrases (bwtparse.c:93)");
  DBuf<uint32_t> dparse(c, P), docc(c, n_words);
  DBuf<uint8_t> dlast(c, P);
  DBuf<uint64_t> dsai;
  h2d(c, dparse.p, parse, P); h2d(c, dlast.p, last, P); h2d(c, docc.p, occ, n_words);
  if (sai) {
    DBuf<uint8_t> packed(c, P * 5);
    h2d(c, packed.p, sai, P * 5);
    dsai.alloc(c, P);
    unpack5_dev(c, packed.p, P, dsai.p);
    sync(c);
  }
  ParseBWT pb;
  parse_bwt(c, dparse.p, P, dlast.p, sai ? dsai.p : nullptr, docc.p, n_words, pb);
  d2h(c, ilist, pb.ilist.p, P + 1);
  d2h(c, bwlast, pb.bwlast.p, P + 1);
  if (sai) {
    DBuf<uint8_t> packed(c, (P + 1) * 5);
    pack5_dev(c, pb.bwsai.p, P + 1, packed.p);
    d2h(c, bwsai, packed.p, (P + 1) * 5);
    sync(c);
  }
  sync(c);
  PFP_REQUIRE(ilist[0] == 1, PFP_EFORMAT, "ilist[0] != 1 (bwtparse.c:305): parse does not start with the smallest word");
  return PFP_OK;
  PFP_CATCH(c)
}

// ---------------------------------------------------------------- stage 3
int pfp_merge(pfp_ctx *c, const uint8_t *dict, uint64_t dict_size, const uint32_t *occ, uint64_t n_words,
              const uint32_t *ilist, const uint8_t *bwlast, const uint8_t *bwsai, uint64_t n_plus_1, int w, int flags,
              pfp_bwt_result *out) {
  if (!c || !dict || !occ || !ilist || !bwlast || !out) return PFP_EINVAL;
  memset(out, 0, sizeof *out);
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  check_args(w, 10, flags);
  PFP_REQUIRE(!flags || bwsai, PFP_EINVAL, "SA output requested but no bwsai given");
  PFP_REQUIRE(dict_size > 1 + (uint64_t)w, PFP_EFORMAT, "invalid dictionary file (pfbwt.cpp:332)");
  PFP_REQUIRE(ilist[0] == 1, PFP_EFORMAT, "ilist[0] != 1 (pfbwt.cpp:377)");
  PFP_REQUIRE(dict[0] == kDollar, PFP_EFORMAT, "dictionary must start with Dollar (pfbwt.cpp:125)");
  Dictionary D; DictIndex ix; DictOrder ord;
  dictionary_from_host(c, dict, dict_size, D, ix);
  PFP_REQUIRE(D.d == n_words, PFP_EFORMAT, "occ entries != dictionary words (pfbwt.cpp:357)");
  // expected output size: every suffix longer than w of every word, once per occurrence
  uint64_t expect = 0, tot_occ = 0;
  {
    uint64_t s = 0, j = 0;
    for (uint64_t i = 0; i < dict_size; i++)
      if (dict[i] == kEndOfWord) {
        uint64_t len = i - s;
        if (len > (uint64_t)w) expect += (len - (uint64_t)w) * occ[j];
        tot_occ += occ[j];
        j++; s = i + 1;
      }
  }
  PFP_REQUIRE(tot_occ + 1 == n_plus_1, PFP_EFORMAT, "sum(occ)+1 != parse size (pfbwt.cpp:397)");
  D.wocc.alloc(c, D.d);
  h2d(c, D.wocc.p, occ, D.d);
  if (c->debug) validate_index(c, D, ix);
  const WordView wv = word_view(D, ix);
  const SlotPayloadSrc pay{wv, D.wocc.p, w};
  ord.wide = use_wide_index(c, D.dsize);      // pfbwt[NT].x or pfbwt[NT]64.x (bigbwt:130-151)
  with_width(ord.wide, [&](auto tag) {
    using I = decltype(tag);
    auto &so = ord.get<I>();
    sort_dict_suffixes<I>(c, D.bytes.p, D.dsize, wv, so, (flags & PFP_FLAG_SA) ? nullptr : &pay);
    if (c->debug) validate_suffix_order<I>(c, D.bytes.p, so, true, "dict SA");
    compute_lexrank<I>(c, D, so, ix);
  });
  if (c->debug) validate_lexrank(c, D, ix);
  DBuf<uint32_t> occ_lex(c, D.d);
  hipLaunchKernelGGL(occ_lex_kernel, gdim(cdiv(D.d, TB)), gdim(TB), 0, c->stream, (uint32_t)D.d, ix.lexrank.p, D.wocc.p,
                     occ_lex.p, (uint32_t *)nullptr);
  ParseBWT pb;
  pb.P = n_plus_1 - 1;
  pb.ilist.alloc(c, n_plus_1); pb.bwlast.alloc(c, n_plus_1);
  h2d(c, pb.ilist.p, ilist, n_plus_1); h2d(c, pb.bwlast.p, bwlast, n_plus_1);
  if (flags) {
    DBuf<uint8_t> packed(c, n_plus_1 * 5);
    h2d(c, packed.p, bwsai, n_plus_1 * 5);
    pb.bwsai.alloc(c, n_plus_1);
    unpack5_dev(c, packed.p, n_plus_1, pb.bwsai.p);
    sync(c);
  }
  DBuf<uint8_t> d_bwt(c, expect + 16);
  DBuf<uint64_t> d_sa;
  if (flags & PFP_FLAG_SA) d_sa.alloc(c, expect + 1);      // -s / -e: the merge keeps the values at the run boundaries (bo.sa_c)
  BwtOutputs bo;
  bo.d_bwt = d_bwt.p; bo.d_sa = d_sa.p;
  with_width(ord.wide, [&](auto tag) {
    using I = decltype(tag);
    merge_bwt<I>(c, D, ix, ord.get<I>(), pb, occ_lex.p, w, flags, expect, bo);
  });
  c->stats.hard_groups = bo.hard_groups; c->stats.hard_chars = bo.hard_chars;
  fetch_outputs(c, d_bwt.p, sa_view(bo), expect, flags, out);
  return PFP_OK;
  PFP_CATCH(c)
}

// ---------------------------------------------------------------- output formats of device-resident results
int pfp_pack5_dev(pfp_ctx *c, const void *d_vals, uint64_t count, void *d_out5) {
  if (!c || ((!d_vals || !d_out5) && count)) return PFP_EINVAL;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  pack5_dev(c, (const uint64_t *)d_vals, count, (uint8_t *)d_out5);
  sync(c);
  return PFP_OK;
  PFP_CATCH(c)
}

int pfp_sample_runs_dev(pfp_ctx *c, const void *d_bwt, const void *d_sa, uint64_t count, uint64_t pos_base, int left_byte,
                        int right_byte, int run_end, void *d_out10, uint64_t cap_pairs, uint64_t *n_pairs) {
  if (!c || !n_pairs || (count && !d_bwt) || (d_out10 && count && !d_sa)) return PFP_EINVAL;
  *n_pairs = 0;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  PFP_REQUIRE(left_byte >= -1 && left_byte <= 255 && right_byte >= -1 && right_byte <= 255, PFP_EINVAL, "neighbour bytes are -1 or 0..255");
  PFP_REQUIRE(pos_base + count <= (1ull << 40), PFP_ELIMIT, "positions do not fit 5 bytes");
  RunSampler rs(c, (const uint8_t *)d_bwt, count, left_byte, right_byte, run_end != 0);
  *n_pairs = rs.pairs;
  if (!d_out10) return PFP_OK;                  // count only
  PFP_REQUIRE(rs.pairs <= cap_pairs, PFP_ELIMIT, "output buffer holds " + std::to_string(cap_pairs) + " pairs, the slice has " +
                                                     std::to_string(rs.pairs) + " run boundaries");
  SaView sv;
  sv.dense = (const uint64_t *)d_sa;
  rs.place(sv, pos_base, (uint8_t *)d_out10);
  sync(c);
  return PFP_OK;
  PFP_CATCH(c)
}

// pfthreads.hpp:369-376: every worker pwrite()s its range of the output file at its offset
int pfp_pwrite_dev(pfp_ctx *c, const char *path, uint64_t file_offset, const void *d_src, uint64_t nbytes) {
  if (!c || !path || (!d_src && nbytes)) return PFP_EINVAL;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  write_dev_file(c, path, file_offset, (const uint8_t *)d_src, nbytes, false);
  return PFP_OK;
  PFP_CATCH(c)
}

// ---------------------------------------------------------------- whole chain
int pfp_bigbwt_dev(pfp_ctx *c, const void *d_text, uint64_t n, int w, uint64_t p, int flags, void *d_bwt, void *d_sa,
                   uint64_t *n_used) {
  if (!c || (!d_text && n) || !d_bwt || (flags && !d_sa)) return PFP_EINVAL;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  check_args(w, p, flags);
  PFP_REQUIRE(((uintptr_t)d_bwt & 15) == 0, PFP_EINVAL, "d_bwt must be 16-byte aligned");
  Chain ch;
  ch.tx.stage(c, d_text, true, n, w);
  uint64_t used = 0;
  run_chain_dev(c, ch, n, w, p, flags, (uint8_t *)d_bwt, (uint64_t *)d_sa, &used);
  if (n_used) *n_used = used;
  return PFP_OK;
  PFP_CATCH(c)
}

int pfp_bigbwt(pfp_ctx *c, const uint8_t *text, uint64_t n, int w, uint64_t p, int flags, pfp_bwt_result *out) {
  if (!c || (!text && n) || !out) return PFP_EINVAL;
  memset(out, 0, sizeof *out);
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  check_args(w, p, flags);
  static const bool trace_host = getenv("PFP_TRACE_HOST") != nullptr;      // where the time of the host boundary goes
  auto now = []() { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto t0 = now();
  Chain ch;
  ch.tx.stage(c, text, false, n, w);
  if (trace_host) sync(c);
  const auto t1 = now();
  DBuf<uint8_t> d_bwt(c, n + 1 + 16);
  uint64_t used = 0;
  run_chain_dev(c, ch, n, w, p, flags, d_bwt.p, nullptr, &used);
  const auto t2 = now();
  fetch_outputs(c, d_bwt.p, sa_view(ch.out), used + 1, flags, out);
  if (trace_host)
    fprintf(stderr, "[pfp] host boundary: text in %.1f ms, chain %.1f ms, outputs out %.1f ms (%.2f GB in, %.2f GB out)\n", ms(t0, t1), ms(t1, t2),
            ms(t2, now()), n / 1e9, (out->bwt_size + out->sa_bytes + out->ssa_bytes + out->esa_bytes) / 1e9);
  return PFP_OK;
  PFP_CATCH(c)
}

// Device-resident chain whose SA-derived outputs are the reference's files, not SA values: .sa (PFP_FLAG_SA, 5-byte
// ints), .ssa / .esa (10-byte pairs) as device buffers of the library (pfp_dev_free).  The SA values themselves stay
// inside (allocated after the suffix sorter has returned its scratch): the 8 bytes per text byte a d_sa array takes
// are what keeps a 12.6 GB input with -s from fitting one GPU next to the sorter.
int pfp_bigbwt_formats_dev(pfp_ctx *c, const void *d_text, uint64_t n, int w, uint64_t p, int flags, void *d_bwt,
                           void *d_out[3], uint64_t out_bytes[3], uint64_t *n_used) {
  if (!c || (!d_text && n) || !d_bwt || !d_out || !out_bytes) return PFP_EINVAL;
  for (int k = 0; k < 3; k++) { d_out[k] = nullptr; out_bytes[k] = 0; }
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  check_args(w, p, flags);
  PFP_REQUIRE(((uintptr_t)d_bwt & 15) == 0, PFP_EINVAL, "d_bwt must be 16-byte aligned");
  Chain ch;
  ch.tx.stage(c, d_text, true, n, w);
  uint64_t used = 0;
  run_chain_dev(c, ch, n, w, p, flags, (uint8_t *)d_bwt, nullptr, &used);
  if (n_used) *n_used = used;
  emit_outputs(c, (const uint8_t *)d_bwt, sa_view(ch.out), used + 1, flags, [&](const char *name, const uint8_t *d, uint64_t bytes) {
    if (name[0] == 'b') return;
    const int k = name[1] == 'a' ? 0 : (name[0] == 's' ? 1 : 2);
    // a block of the context's pool (handed back by pfp_dev_free): steady-state calls do not reach the driver
    hipError_t e = hipSuccess;
    void *q = c->pool.get(bytes ? bytes : 1, &e, __FILE__, __LINE__);
    if (!q) throw Error(PFP_ENOMEM, std::string("device allocation of an output buffer failed: ") + hipGetErrorString(e));
    PFP_HIP(hipMemcpyAsync(q, d, bytes, hipMemcpyDeviceToDevice, c->stream));
    d_out[k] = q; out_bytes[k] = bytes;
  });
  sync(c);
  return PFP_OK;
  PFP_CATCH(c)
}
int pfp_memcpy_d2h(pfp_ctx *c, void *host_dst, const void *d_src, uint64_t nbytes) {
  if (!c || ((!host_dst || !d_src) && nbytes)) return PFP_EINVAL;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  stream_d2h(c, (const uint8_t *)d_src, nbytes, [&](const uint8_t *pin, uint64_t off, uint64_t len) { par_memcpy((uint8_t *)host_dst + off, pin, len); });
  sync(c);
  return PFP_OK;
  PFP_CATCH(c)
}
void pfp_dev_free(pfp_ctx *c, void *d_ptr) {
  if (!c || !d_ptr) return;
  (void)hipSetDevice(c->device);
  for (const auto &b : c->pool.all)
    if (b.p == d_ptr) { c->pool.put(d_ptr); return; }      // (later work of the context is ordered behind the caller's reads only if
  (void)hipFree(d_ptr);                                    //  those were on the context's stream or have completed: pfpgpu.h)
}

// file to files: the host text (an mmap of the input works) is streamed in, the outputs are streamed from HBM
// straight into <base>.bwt / .sa / .ssa / .esa - no host copy of any output is held
}  // extern "C"
template <class Stage>
static int bigbwt_to_files(pfp_ctx *c, uint64_t n, int w, uint64_t p, int flags, const char *base, uint64_t out_bytes[4], Stage &&stage) {
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  check_args(w, p, flags);
  static const bool trace_host = getenv("PFP_TRACE_HOST") != nullptr;
  auto now = []() { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto t0 = now();
  join_background(c);
  // the two outputs whose sizes the text's length fixes get their files now: their pages are made ready beside the input and the chain
  MappedOut m_bwt, m_sa;
  // (measured: holding the registrations back until the text is in, or starting only then, moves the helper's 1.6-1.9 s for
  //  12.6 GB behind the chain instead of beside the input - same total; profiles/r04_cli_probe_hold.txt)
  const bool map_bwt = m_bwt.start(c, std::string(base) + ".bwt", n + 1);
  const bool map_sa = (flags & PFP_FLAG_SA) && n && m_sa.start(c, std::string(base) + ".sa", n * 5);
  Chain ch;
  stage(ch.tx);
  if (trace_host) sync(c);
  const auto t1 = now();
  DBuf<uint8_t> d_bwt(c, n + 1 + 16);
  uint64_t used = 0;
  run_chain_dev(c, ch, n, w, p, flags, d_bwt.p, nullptr, &used);
  const auto t2 = now();
  uint64_t sizes[4] = {0, 0, 0, 0};
  int n_mapped = 0;
  emit_outputs(c, d_bwt.p, sa_view(ch.out), used + 1, flags, [&](const char *name, const uint8_t *d, uint64_t bytes) {
    sizes[name[0] == 'b' ? 0 : (name[1] == 'a' ? 1 : (name[0] == 's' ? 2 : 3))] = bytes;
    if (trace_host) fprintf(stderr, "[pfp]   %.1f ms after the chain: .%s (%.2f GB) is on the device\n", ms(t2, now()), name, bytes / 1e9);
    MappedOut *mo = name[0] == 'b' ? (map_bwt ? &m_bwt : nullptr) : (name[0] == 's' && name[1] == 'a' && name[2] == 0 ? (map_sa ? &m_sa : nullptr) : nullptr);
    MappedOut late;      // .ssa / .esa (and an .sa that could not start early): their sizes are only known now - the helper works while the copies follow it
    if (!mo && late.start(c, std::string(base) + "." + name, bytes)) mo = &late;
    if (mo && mo->active() && bytes <= mo->bytes) {
      if (mo->write(c, d, bytes)) {
        n_mapped++;
        if (mo != &m_bwt) mo->finish(c, bytes);      // (its device buffer goes back to the pool when this returns; the BWT's lives to the end)
        return;
      }
      mo->abandon(false);
    } else if (mo && mo->active()) mo->abandon(false);
    write_dev_file(c, std::string(base) + "." + name, 0, d, bytes, true);
  });
  if (trace_host) fprintf(stderr, "[pfp]   %.1f ms after the chain: the other files are written\n", ms(t2, now()));
  if (m_bwt.active()) m_bwt.finish(c, sizes[0]);
  if (m_sa.active()) m_sa.abandon(true);
  if (trace_host)
    fprintf(stderr, "[pfp] file to files: text in %.1f ms, chain %.1f ms (first call: the pool is cold), files out %.1f ms%s\n", ms(t0, t1), ms(t1, t2),
            ms(t2, now()), n_mapped ? (std::string(" (") + std::to_string(n_mapped) + " of them straight into the files' mapped pages)").c_str() : "");
  if (out_bytes) memcpy(out_bytes, sizes, sizeof sizes);
  return PFP_OK;
  PFP_CATCH(c)
}
extern "C" {
int pfp_bigbwt_files(pfp_ctx *c, const uint8_t *text, uint64_t n, int w, uint64_t p, int flags, const char *base,
                     uint64_t out_bytes[4]) {
  if (!c || (!text && n) || !base) return PFP_EINVAL;
  return bigbwt_to_files(c, n, w, p, flags, base, out_bytes, [&](StagedText &tx) { tx.stage(c, text, false, n, w); });
}
// the same with the text read from bytes [file_offset, file_offset + n) of an open file (parallel pread into the pinned
// staging buffers: what the C driver uses for a plain input file)
int pfp_bigbwt_fd(pfp_ctx *c, int fd, uint64_t file_offset, uint64_t n, int w, uint64_t p, int flags, const char *base,
                  uint64_t out_bytes[4]) {
  if (!c || fd < 0 || !base) return PFP_EINVAL;
  return bigbwt_to_files(c, n, w, p, flags, base, out_bytes, [&](StagedText &tx) { tx.stage_fd(c, fd, file_offset, n, w); });
}

}  // extern "C"

// ---------------------------------------------------------------- multi-GPU chain (one rank's share)
//
// the text is sharded over the ranks, phrases are independent units.  Every rank
//   1. pfp_dist_local_parse : scans (halo + its shard), owns the phrases that END inside the shard,
//                             deduplicates them locally                       [n/R bytes of work]
//   -- allgather of the local dictionaries (RCCL, done by the caller) --
//   2. pfp_dist_global      : deduplicates the union into the global dictionary, suffix-sorts it
//                             (replicated: O(|D|) << n for repetitive input), translates its own
//                             parse to global lexicographic ranks
//   -- allgather of parse symbols / last / sai --
//   3. pfp_dist_merge       : BWT of the parse (replicated) and the slice [lo,hi) of the final BWT/SA
// The collectives live in big-bwt_amd/dist.py (torch.distributed over RCCL/xGMI).
struct DistState {
  StagedText tx;
  DBuf<uint64_t> ends;
  uint64_t n_ends = 0, n_local = 0, k0 = 0, P_local = 0;
  int w = 0;
  Dictionary L;                 // local dictionary + local parse
  Dictionary G;                 // global dictionary (identical on every rank)
  DictIndex ix;
  DictOrder ord;
  DBuf<uint32_t> occ_lex;
  uint64_t local_total = 0;     // BWT positions the held slots emit
  BwtOutputs out;               // -s / -e without an SA slice: run maps and per-boundary SA values of the emitted slice
  uint64_t out_lo = 0;
  bool slice_empty = true;      // the last pfp_dist_merge emitted no position
  bool want_sai = false;
  int flags = 0;                // output flags announced at pfp_dist_local_parse (0: BWT only)
  // hash-partitioned dedup (pfp_dist_partition_words ...): local words in owner order, the words this rank owns,
  // and the global id of every local word once the owners have answered
  DBuf<uint32_t> part_order;    // [L.d] local word ids, grouped by owner
  Dictionary Own;               // distinct words of this rank's hash class (first-arrival order) with summed occ
  DBuf<uint32_t> gid_local;     // [L.d] global word id of local word j
  bool have_gid = false;
  DBuf<uint32_t> parse_sa;      // [P + 1] the parse's suffix array gathered from the ranks' shares (pfp_dist_set_parse_sa), used by the next merge
  uint64_t parse_sa_n = 0;
  uint64_t parse_share_rounds = 0;      // rounds of this rank's share sort (reported when the gathered array is used)
};
static DistState *dist_of(pfp_ctx *c) {
  if (!c->dist) c->dist = new DistState();
  return reinterpret_cast<DistState *>(c->dist);
}

__global__ void count_below_kernel(const uint64_t *__restrict__ ends, uint64_t ne, uint64_t bound, uint64_t *out) {
  uint64_t lo = 0, hi = ne;                   // first index with ends[idx] >= bound
  while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if (ends[mid] < bound) lo = mid + 1; else hi = mid; }
  out[0] = lo;
  out[1] = ne ? ends[ne - 1] : ~0ull;
}
__global__ void dist_sym_kernel(uint64_t P, const uint32_t *__restrict__ lpid, uint64_t word_base,
                                const uint32_t *__restrict__ gid_of_union, const uint32_t *__restrict__ lexrank,
                                uint32_t *__restrict__ sym) {
  uint64_t k = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (k < P) sym[k] = lexrank[gid_of_union[word_base + lpid[k]]] + 1;
}
// hash class of every local word: owner = (hash >> 20) % parts; per-owner word and byte counts
// (counts are summed per workgroup in LDS first: one atomic per workgroup and owner, not one per word - with few
// owners every word would hit the same two addresses)
__global__ __launch_bounds__(256) void word_owner_kernel(uint32_t d, const uint64_t *__restrict__ hash, const uint32_t *__restrict__ wlen,
                                                         uint32_t parts, uint32_t *__restrict__ owner, uint32_t *__restrict__ ids,
                                                         unsigned long long *__restrict__ counts) {
  __shared__ unsigned long long lc[2 * 64];
  const uint32_t np = parts < 64 ? parts : 64;      // owners beyond 64 (not a single-node case) go straight to global memory
  for (uint32_t q = threadIdx.x; q < 2 * np; q += 256) lc[q] = 0;
  __syncthreads();
  uint32_t j = BID * blockDim.x + threadIdx.x;
  if (j < d) {
    const uint32_t o = (uint32_t)((hash[j] >> 20) % parts);
    owner[j] = o; ids[j] = j;
    unsigned long long *dst = o < np ? lc : counts;
    atomicAdd(&dst[2 * o], 1ull);
    atomicAdd(&dst[2 * o + 1], (unsigned long long)wlen[j] + 1);
  }
  __syncthreads();
  for (uint32_t q = threadIdx.x; q < 2 * np; q += 256) if (lc[q]) atomicAdd(&counts[q], lc[q]);
}
__global__ void gather_u32_kernel(uint32_t n, const uint32_t *__restrict__ idx, const uint32_t *__restrict__ src, uint32_t *__restrict__ dst) {
  uint32_t i = BID * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}
__global__ void scatter_u32_kernel(uint32_t n, const uint32_t *__restrict__ idx, const uint32_t *__restrict__ src, uint32_t *__restrict__ dst) {
  uint32_t i = BID * blockDim.x + threadIdx.x;
  if (i < n) dst[idx[i]] = src[i];
}
__global__ void len1_of_kernel(uint32_t n, const uint32_t *__restrict__ order, const uint32_t *__restrict__ wlen, uint32_t *__restrict__ len1) {
  uint32_t i = BID * blockDim.x + threadIdx.x;
  if (i == 0) len1[n] = 0;
  if (i < n) len1[i] = wlen[order[i]] + 1;
}
__global__ void dist_sym_gid_kernel(uint64_t P, const uint32_t *__restrict__ lpid, const uint32_t *__restrict__ gid_local,
                                    const uint32_t *__restrict__ lexrank, uint32_t *__restrict__ sym) {
  uint64_t k = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (k < P) sym[k] = lexrank[gid_local[lpid[k]]] + 1;
}

__global__ __launch_bounds__(256) void sum_u32_kernel(const uint32_t *__restrict__ v, uint64_t n, unsigned long long *__restrict__ total) {
  unsigned long long s = 0;
  for (uint64_t i = (uint64_t)BID * 256 + threadIdx.x; i < n; i += (uint64_t)GDIM * 256) s += v[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0 && s) atomicAdd(total, s);
}
template <class I>
__global__ void add_one_kernel(uint32_t n, const I *__restrict__ in, uint64_t *__restrict__ out) {
  uint32_t j = BID * blockDim.x + threadIdx.x;
  if (j < n) out[j] = (uint64_t)in[j] + 1;
}

extern "C" {

// Extra trigger hashes this rank would add to split its giant phrases (<= 8, see scan.hip).  The
// caller allgathers the proposals and hands the union to every rank's pfp_dist_local_parse, so that
// all ranks scan with ONE trigger set.
int pfp_dist_propose_triggers(pfp_ctx *c, const void *d_text, uint64_t n, int w, uint64_t p, uint32_t out_hashes[8],
                              uint32_t *n_hashes) {
  if (!c || (!d_text && n) || !out_hashes || !n_hashes) return PFP_EINVAL;
  *n_hashes = 0;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  check_args(w, p, 0);
  if (!c->max_phrase) return PFP_OK;
  StagedText tx;
  tx.stage(c, d_text, true, n, w);
  DBuf<uint64_t> ends;
  uint64_t used = 0;
  KRParams kp = make_kr_params(w, p);
  uint64_t ne = scan_text(c, tx, n, w, p, ends, &used, &kp);
  propose_extra_triggers(c, tx, used, w, c->max_phrase, ends, ne, kp);
  for (uint32_t q = 0; q < kp.nextra && q < 8; q++) out_hashes[(*n_hashes)++] = kp.extra[q];
  return PFP_OK;
  PFP_CATCH(c)
}

// ---- the collection's parse plan (round 4): which hash cuts the text, with which seed, how densely
// plan[0] 0 = the reference's Karp-Rabin hash, 1 = the window hash of scan.hip; plan[1] its seed; plan[2] the density (a double's
// bits: the text is cut with probability density / p); plan[3] 1 while the density is a candidate the ranks still have to decide on
static KRParams params_from_plan(int w, uint64_t p, const uint64_t plan[4]) {
  KRParams kp = make_kr_params(w, p);
  if (!plan || plan[0] == 0) return kp;
  double dens;
  memcpy(&dens, &plan[2], 8);
  PFP_REQUIRE(w >= 4 && w <= 17 && dens >= 0.01 && dens <= 64.0, PFP_EINVAL, "bad parse plan");
  const double thr_nom = 4294967296.0 / (double)p, thr = thr_nom * dens;
  kp.fast = 1; kp.fseed = (uint32_t)plan[1]; kp.fdens = (float)dens; kp.fauto = 0;
  kp.fthr = thr >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)thr;
  kp.fthr_nom = thr_nom >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)thr_nom;
  return kp;
}
// Rank 0 (which holds the text's first bytes) makes the plan; the hosts hand it to every rank.  first_hash: the hash of the text's
// first window under the plan (it must not become an extra trigger: SURVEY.md 2.2-Q1), ~0 when the text is shorter than a window.
int pfp_dist_parse_plan(pfp_ctx *c, const uint8_t *first_bytes, uint64_t n_bytes, int w, uint64_t p, uint32_t ranks, uint64_t plan[4],
                        uint64_t *first_hash) {
  if (!c || !plan || !first_hash || (!first_bytes && n_bytes)) return PFP_EINVAL;
  PFP_TRY(c)
  check_args(w, p, 0);
  const bool have = n_bytes >= (uint64_t)w;
  const double one = 1.0;
  *first_hash = ~0ull;
  if (!c->fast_triggers || w < 4 || w > 17) {
    plan[0] = 0; plan[1] = 0; memcpy(&plan[2], &one, 8); plan[3] = 0;
    if (have) { uint64_t h = 0; for (int k = 0; k < w; k++) h = (h * 256 + first_bytes[k]) % kPrime; *first_hash = h; }      // newscan.cpp:168-202
    return PFP_OK;
  }
  // (the dictionary's work is shared between the ranks, the parse's is not - every rank's merge reads the whole parse: shorter
  //  phrases pay on one or two ranks - 42.3 -> 35.8 and 51.1 -> 42.2 ms per rank - and no longer on eight, 71.9 -> 72.3; so the
  //  density is a candidate only there, and nominal beyond)
  const double setting = c->parse_density > 0 ? c->parse_density : (ranks > 2 ? 1.0 : 0.0);
  const KRParams kp = make_fast_params_host(have ? first_bytes : nullptr, w, p, setting);
  const double dens = (double)kp.fdens;
  plan[0] = 1; plan[1] = kp.fseed; memcpy(&plan[2], &dens, 8); plan[3] = kp.fauto;
  if (have) *first_hash = window_hash_host(first_bytes, w, kp.fseed);
  return PFP_OK;
  PFP_CATCH(c)
}
// pfp_dist_propose_triggers under a plan; while the plan's density is a candidate (plan[3]) also this rank's sample of the cuts
// at or after halo_len: at most sample_cap sorted context hashes in d_sample, *n_sample of them (the hosts all-gather the samples
// and every rank decides alike: pfp_dist_decide_density)
int pfp_dist_propose_triggers2(pfp_ctx *c, const void *d_text, uint64_t n, uint64_t halo_len, int w, uint64_t p, const uint64_t plan[4],
                               uint32_t out_hashes[8], uint32_t *n_hashes, void *d_sample, uint64_t sample_cap, uint64_t *n_sample) {
  if (!c || (!d_text && n) || !out_hashes || !n_hashes || !plan || !n_sample || (sample_cap && !d_sample)) return PFP_EINVAL;
  *n_hashes = 0; *n_sample = 0;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  check_args(w, p, 0);
  if (!c->max_phrase && !plan[3]) return PFP_OK;
  StagedText tx;
  tx.stage(c, d_text, true, n, w);
  DBuf<uint64_t> ends;
  uint64_t used = 0;
  KRParams kp = params_from_plan(w, p, plan);
  const uint64_t ne = scan_text(c, tx, n, w, p, ends, &used, &kp);
  // (proposals only under a SETTLED plan: a window of a periodic stretch may cut at the candidate density and not at the nominal
  //  one - the stretch then looks harmless here and is one giant phrase in the parse; the hosts call this twice while plan[3] is set:
  //  for the sample first, for the proposals after pfp_dist_decide_density)
  if (c->max_phrase && !plan[3]) {
    propose_extra_triggers(c, tx, used, w, c->max_phrase, ends, ne, kp);
    for (uint32_t q = 0; q < kp.nextra && q < 8; q++) out_hashes[(*n_hashes)++] = kp.extra[q];
  }
  if (plan[3] && kp.fast && ne) {
    CutSample cs;
    classify_cuts(c, tx, w, ends, ne, kp, halo_len, cs);
    const uint64_t take = std::min<uint64_t>(cs.ns, sample_cap);
    if (take) PFP_HIP(hipMemcpyAsync(d_sample, cs.hashes.p, take * 8, hipMemcpyDeviceToDevice, c->stream));
    sync(c);
    *n_sample = take;
  }
  return PFP_OK;
  PFP_CATCH(c)
}
// the ranks' samples, gathered: every rank sorts them and settles the plan's density the same way (candidate kept or 1)
int pfp_dist_decide_density(pfp_ctx *c, const void *d_samples, uint64_t count, uint64_t p, uint64_t plan[4]) {
  if (!c || !plan || (count && !d_samples)) return PFP_EINVAL;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  if (!plan[3]) return PFP_OK;
  bool dense = false;
  if (count >= 1024) {
    DBuf<uint64_t> sorted(c, count);
    sort_keys_raw(c, (const uint64_t *)d_samples, sorted.p, count, 0, 64);
    dense = sample_says_dense(c, sorted.p, count, p);
  }
  const double one = 1.0;
  if (!dense) memcpy(&plan[2], &one, 8);
  plan[3] = 0;
  return PFP_OK;
  PFP_CATCH(c)
}

static int dist_local_parse_impl(pfp_ctx *c, const void *d_text, uint64_t n, uint64_t halo_len, int w, uint64_t p, int is_first,
                                 int is_last, uint64_t global_offset, int want_sai, const uint64_t *plan, const uint32_t *extra_hashes,
                                 uint32_t n_extra, uint64_t out_sizes[4]);
int pfp_dist_local_parse(pfp_ctx *c, const void *d_text, uint64_t n, uint64_t halo_len, int w, uint64_t p, int is_first,
                         int is_last, uint64_t global_offset, int want_sai, const uint32_t *extra_hashes,
                         uint32_t n_extra, uint64_t out_sizes[4]) {
  return dist_local_parse_impl(c, d_text, n, halo_len, w, p, is_first, is_last, global_offset, want_sai, nullptr, extra_hashes, n_extra, out_sizes);
}
// the same under a (decided) parse plan
int pfp_dist_local_parse2(pfp_ctx *c, const void *d_text, uint64_t n, uint64_t halo_len, int w, uint64_t p, int is_first,
                          int is_last, uint64_t global_offset, int want_sai, const uint64_t plan[4], const uint32_t *extra_hashes,
                          uint32_t n_extra, uint64_t out_sizes[4]) {
  if (!plan || plan[3]) return PFP_EINVAL;
  return dist_local_parse_impl(c, d_text, n, halo_len, w, p, is_first, is_last, global_offset, want_sai, plan, extra_hashes, n_extra, out_sizes);
}
static int dist_local_parse_impl(pfp_ctx *c, const void *d_text, uint64_t n, uint64_t halo_len, int w, uint64_t p, int is_first,
                                 int is_last, uint64_t global_offset, int want_sai, const uint64_t *plan, const uint32_t *extra_hashes,
                                 uint32_t n_extra, uint64_t out_sizes[4]) {
  if (!c || (!d_text && n) || !out_sizes || (n_extra && !extra_hashes)) return PFP_EINVAL;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  check_args(w, p, want_sai & 7);      // (callers pass the output flags here: -S with -s/-e is refused like bigbwt:59-61)
  PFP_REQUIRE(n_extra <= KRParams::kMaxExtra, PFP_EINVAL, "too many extra trigger hashes");
  PFP_REQUIRE(is_first ? halo_len == 0 : halo_len >= (uint64_t)w, PFP_EINVAL, "halo must hold at least one window");
  PFP_REQUIRE(halo_len <= n, PFP_EINVAL, "halo longer than the local text");
  DistState *ds = dist_of(c);
  *ds = DistState();
  ds->w = w; ds->n_local = n;
  pfp_stats &st = c->stats;
  st = pfp_stats{};
  ds->tx.stage(c, d_text, true, n, w);
  uint64_t used = 0;
  KRParams kp = params_from_plan(w, p, plan);                      // one trigger set on all ranks
  st.parse_density = kp.fast ? (double)kp.fdens : 1.0;
  for (uint32_t q = 0; q < n_extra; q++) { kp.extra[kp.nextra++] = extra_hashes[q]; kp.bloom |= 1ull << (extra_hashes[q] & 63); }
  { PhaseTimer t(c, &st.ms_scan);
    ds->n_ends = scan_text(c, ds->tx, n, w, p, ds->ends, &used, &kp); }
  PFP_REQUIRE(used == n, PFP_EFORMAT, "bytes <= 2 inside a text shard are not supported in the multi-GPU chain");
  DBuf<uint64_t> tmp(c, 2);
  hipLaunchKernelGGL(count_below_kernel, gdim(1), gdim(1), 0, c->stream, ds->ends.p, ds->n_ends, halo_len, tmp.p);
  PFP_HIP(hipMemcpyAsync(c->h_scalars, tmp.p, 16, hipMemcpyDeviceToHost, c->stream));
  sync(c);
  ds->k0 = is_first ? 0 : c->h_scalars[0];
  const uint64_t last_end = c->h_scalars[1];
  PFP_REQUIRE(is_first || ds->k0 >= 1, PFP_ELIMIT, "no phrase boundary inside the halo: a phrase is longer than the halo");
  const uint64_t kend = is_last ? ds->n_ends + 1 : ds->n_ends;      // the partial tail phrase belongs to the next rank
  PFP_REQUIRE(kend > ds->k0, PFP_ESHORT, "text shard holds no complete phrase");
  ds->P_local = kend - ds->k0;
  // T' index x of the local text is global text position global_offset - halo_len + x - 1; sai = end position + 1
  const uint64_t sai_base = global_offset - halo_len;
  ds->want_sai = want_sai != 0;
  ds->flags = want_sai & 7;     // callers pass the output flags here (any non-zero value asks for sa info)
  { PhaseTimer t(c, &st.ms_phrases);
    build_dictionary_shard(c, ds->tx, n, w, ds->ends, ds->n_ends, ds->k0, ds->P_local, want_sai != 0, sai_base, ds->L); }
  st.n = n - halo_len; st.n_phrases = ds->P_local; st.extra_triggers = n_extra;
  out_sizes[0] = ds->L.dsize - 1;      // local dictionary bytes without the final 0x00
  out_sizes[1] = ds->L.d;
  out_sizes[2] = ds->P_local;
  out_sizes[3] = last_end;             // local position of the last trigger (the next rank's halo must reach it)
  return PFP_OK;
  PFP_CATCH(c)
}

int pfp_dist_export_local(pfp_ctx *c, void *d_dict, void *d_occ, void *d_last, void *d_sai) {
  if (!c || !c->dist) return PFP_EINVAL;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  DistState *ds = dist_of(c);
  if (d_dict) PFP_HIP(hipMemcpyAsync(d_dict, ds->L.bytes.p, ds->L.dsize - 1, hipMemcpyDeviceToDevice, c->stream));
  if (d_occ) PFP_HIP(hipMemcpyAsync(d_occ, ds->L.wocc.p, ds->L.d * 4, hipMemcpyDeviceToDevice, c->stream));
  if (d_last) PFP_HIP(hipMemcpyAsync(d_last, ds->L.last.p, ds->P_local, hipMemcpyDeviceToDevice, c->stream));
  if (d_sai) {
    PFP_REQUIRE(ds->L.sai.p, PFP_EINVAL, "sa info was not requested in pfp_dist_local_parse");
    PFP_HIP(hipMemcpyAsync(d_sai, ds->L.sai.p, ds->P_local * 8, hipMemcpyDeviceToDevice, c->stream));
  }
  sync(c);
  return PFP_OK;
  PFP_CATCH(c)
}

}  // extern "C"
// the global dictionary ds->G is in place: index it and sort its suffixes (replicated, or this rank's key range)
static void dist_sort_global(pfp_ctx *c, DistState *ds, uint32_t part, uint32_t parts, void *d_wslot_out, uint64_t out_info[8]) {
  PhaseTimer t_sa(c, &c->stats.ms_sa_dict);
  ds->ix = DictIndex();
  ds->ord = DictOrder();
  build_dict_index(c, ds->G, ds->ix);
  const uint32_t d = (uint32_t)ds->G.d;
  const WordView wv = word_view(ds->G, ds->ix);
  const SlotPayloadSrc pay{wv, ds->G.wocc.p, ds->w};
  const SlotPayloadSrc *payp = (ds->flags & PFP_FLAG_SA) ? nullptr : &pay;      // full SA: the merge gathers wider records itself
  // a share of a dictionary of 2^31 bytes or more takes the wide build: the 32-bit one has no spare bit for the settled flag
  // there, so no pivot rounds - and a share cannot run doubling rounds instead (they read other shares' ranks).  The share is
  // 1 / parts of the slots, so 8-byte indices are affordable where they would not be for the whole array.
  ds->ord.wide = use_wide_index(c, ds->G.dsize) || (parts > 1 && ds->G.dsize >= (1ull << 31));
  // phrases per distinct word = text bytes per dictionary byte, near enough (the same on every rank: the keys-only
  // first round of the sorter is chosen from it)
  double rep_hint = 0;
  {
    DBuf<unsigned long long> tot(c, 1);
    tot.zero();
    hipLaunchKernelGGL(sum_u32_kernel, gdim((int)std::min<uint64_t>(cdiv64(d, 256), 1024)), gdim(256), 0, c->stream, ds->G.wocc.p, (uint64_t)d, tot.p);
    rep_hint = (double)read_scalar(c, (const uint64_t *)tot.p) / (double)std::max<uint32_t>(d, 1);
  }
  uint64_t info_rounds = 0, info_complete = 1, info_N = 0, info_base = 0;
  with_width(ds->ord.wide, [&](auto tag) {
    using I = decltype(tag);
    auto &so = ds->ord.get<I>();
    so.rep_hint = rep_hint;
    if (parts == 1) {
      sort_dict_suffixes<I>(c, ds->G.bytes.p, ds->G.dsize, wv, so, payp);
      if (c->debug) validate_suffix_order<I>(c, ds->G.bytes.p, so, true, "global dict SA");
      DBuf<I> slots(c, d);
      gather_ranks<I>(c, so, ds->G.woff.p, d, slots.p);
      hipLaunchKernelGGL(add_one_kernel<I>, gdim(cdiv(d, TB)), gdim(TB), 0, c->stream, d, slots.p, (uint64_t *)d_wslot_out);
      ds->local_total = 0;
    } else {
      sort_dict_suffixes_range<I>(c, ds->G.bytes.p, ds->G.dsize, wv, part, parts, so, payp, &pay);
      gather_slots_range<I>(c, so, wv, d, (uint64_t *)d_wslot_out);
      ds->local_total = so.complete ? so.range_emits : 0;
      if (c->debug && so.complete)
        PFP_REQUIRE(count_slot_outputs<I>(c, ds->G, ds->ix, so, ds->w) == ds->local_total, PFP_EHIP,
                    "emit count by position differs from the count by slot");
    }
    info_rounds = so.rounds; info_complete = so.complete ? 1 : 0; info_N = so.N; info_base = so.slot_base;
  });
  PFP_HIP(hipGetLastError());
  sync(c);
  out_info[0] = ds->G.d; out_info[1] = ds->G.dsize; out_info[2] = info_rounds; out_info[3] = info_complete;
  out_info[4] = info_N; out_info[5] = info_base; out_info[6] = ds->local_total; out_info[7] = ds->ord.wide ? 64 : 32;
  c->stats.n_words = ds->G.d; c->stats.dict_size = ds->G.dsize; c->stats.sa_rounds_dict = info_rounds;
  c->stats.index_bits = ds->ord.wide ? 64 : 32;
}
extern "C" {

int pfp_dist_global_sort(pfp_ctx *c, const void *d_union, uint64_t union_bytes, const void *d_union_occ, uint64_t n_union,
                         uint32_t part, uint32_t parts, void *d_wslot_out, uint64_t out_info[8]) {
  if (!c || !c->dist || !d_union || !d_union_occ || !d_wslot_out || !out_info || parts < 1 || part >= parts) return PFP_EINVAL;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  DistState *ds = dist_of(c);
  PFP_REQUIRE(n_union >= 1, PFP_EINVAL, "empty union");
  // the union is itself a (not sorted, not duplicate-free) dictionary: words + 0x01, closed by one 0x00
  Dictionary U;
  U.dsize = union_bytes + 1;
  U.bytes.alloc(c, U.dsize + 64);
  PFP_HIP(hipMemcpyAsync(U.bytes.p, d_union, union_bytes, hipMemcpyDeviceToDevice, c->stream));
  PFP_HIP(hipMemsetAsync(U.bytes.p + union_bytes, 0, 65, c->stream));
  word_table_from_bytes(c, U, n_union);           // only the word boundaries of the union are needed
  PFP_REQUIRE(U.d == n_union, PFP_EFORMAT, "the union holds a different number of words than occ entries");
  ds->G = Dictionary();
  ds->have_gid = false;
  build_dictionary_words(c, U.bytes.p, U.woff.p, U.wlen.p, n_union, (const uint32_t *)d_union_occ, union_bytes, ds->G);
  dist_sort_global(c, ds, part, parts, d_wslot_out, out_info);
  return PFP_OK;
  PFP_CATCH(c)
}

// ---- hash-partitioned dedup: every distinct word is owned by the rank its hash points at (SURVEY 8e exchange A; the
//      reference's threaded parser shards its maps by hash % (3 N) the same way, pscan.cpp:137-205)
int pfp_dist_partition_words(pfp_ctx *c, uint32_t parts, uint64_t *counts /* [2 * parts]: words, bytes (+1 per word) per owner */) {
  if (!c || !c->dist || !counts || parts < 1) return PFP_EINVAL;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  DistState *ds = dist_of(c);
  const uint32_t d = (uint32_t)ds->L.d;
  PFP_REQUIRE(d >= 1, PFP_EINVAL, "pfp_dist_local_parse has not run");
  DBuf<uint64_t> hash(c, d);
  hash_word_list(c, ds->L.bytes.p, ds->L.woff.p, ds->L.wlen.p, d, 0x6A09E667F3BCC909ULL, hash.p);
  DBuf<uint32_t> owner(c, d), ownero(c, d), ids(c, d);
  DBuf<unsigned long long> cnt(c, 2 * (size_t)parts);
  cnt.zero();
  hipLaunchKernelGGL(word_owner_kernel, gdim(cdiv(d, TB)), gdim(TB), 0, c->stream, d, hash.p, ds->L.wlen.p, parts, owner.p, ids.p, cnt.p);
  ds->part_order.alloc(c, d);
  sort_pairs_u32_u32(c, owner.p, ownero.p, ids.p, ds->part_order.p, d, 0, bits_for(parts));      // stable: local order inside an owner
  PFP_HIP(hipGetLastError());
  PFP_HIP(hipMemcpyAsync(counts, cnt.p, 2 * (size_t)parts * 8, hipMemcpyDeviceToHost, c->stream));
  sync(c);
  return PFP_OK;
  PFP_CATCH(c)
}

int pfp_dist_export_partition(pfp_ctx *c, void *d_bytes, void *d_occ) {
  if (!c || !c->dist || !d_bytes || !d_occ) return PFP_EINVAL;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  DistState *ds = dist_of(c);
  const uint32_t d = (uint32_t)ds->L.d;
  PFP_REQUIRE(ds->part_order.p, PFP_EINVAL, "pfp_dist_partition_words has not run");
  DBuf<uint32_t> len1(c, (size_t)d + 1);
  DBuf<uint64_t> doff(c, (size_t)d + 1);
  hipLaunchKernelGGL(len1_of_kernel, gdim(cdiv(d, TB)), gdim(TB), 0, c->stream, d, ds->part_order.p, ds->L.wlen.p, len1.p);
  exclusive_sum_u32_u64(c, len1.p, doff.p, (size_t)d + 1);
  hipLaunchKernelGGL(dict_permute_kernel, gdim(cdiv((uint64_t)d * 8, TB)), gdim(TB), 0, c->stream, d, ds->part_order.p, ds->L.woff.p,
                     ds->L.wlen.p, ds->L.bytes.p, doff.p, (uint8_t *)d_bytes);
  hipLaunchKernelGGL(gather_u32_kernel, gdim(cdiv(d, TB)), gdim(TB), 0, c->stream, d, ds->part_order.p, ds->L.wocc.p, (uint32_t *)d_occ);
  PFP_HIP(hipGetLastError());
  sync(c);
  return PFP_OK;
  PFP_CATCH(c)
}

int pfp_dist_owner_dedup(pfp_ctx *c, const void *d_bytes, uint64_t nbytes, const void *d_occ, uint64_t n_words, void *d_pid_out,
                         uint64_t out[2]) {
  if (!c || !c->dist || !out || (n_words && (!d_bytes || !d_occ || !d_pid_out))) return PFP_EINVAL;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  DistState *ds = dist_of(c);
  ds->Own = Dictionary();
  out[0] = out[1] = 0;
  if (!n_words) return PFP_OK;                 // nobody sent a word of this hash class
  Dictionary U;
  U.dsize = nbytes + 1;
  U.bytes.alloc(c, U.dsize + 64);
  PFP_HIP(hipMemcpyAsync(U.bytes.p, d_bytes, nbytes, hipMemcpyDeviceToDevice, c->stream));
  PFP_HIP(hipMemsetAsync(U.bytes.p + nbytes, 0, 65, c->stream));
  word_table_from_bytes(c, U, n_words);
  PFP_REQUIRE(U.d == n_words, PFP_EFORMAT, "the received words do not match their occ entries");
  build_dictionary_words(c, U.bytes.p, U.woff.p, U.wlen.p, n_words, (const uint32_t *)d_occ, nbytes, ds->Own);
  PFP_HIP(hipMemcpyAsync(d_pid_out, ds->Own.pid.p, n_words * 4, hipMemcpyDeviceToDevice, c->stream));
  sync(c);
  out[0] = ds->Own.d; out[1] = ds->Own.dsize - 1;
  return PFP_OK;
  PFP_CATCH(c)
}

int pfp_dist_export_owned(pfp_ctx *c, void *d_bytes, void *d_occ) {
  if (!c || !c->dist) return PFP_EINVAL;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  DistState *ds = dist_of(c);
  if (!ds->Own.d) return PFP_OK;
  PFP_REQUIRE(d_bytes && d_occ, PFP_EINVAL, "null output");
  PFP_HIP(hipMemcpyAsync(d_bytes, ds->Own.bytes.p, ds->Own.dsize - 1, hipMemcpyDeviceToDevice, c->stream));
  PFP_HIP(hipMemcpyAsync(d_occ, ds->Own.wocc.p, ds->Own.d * 4, hipMemcpyDeviceToDevice, c->stream));
  sync(c);
  return PFP_OK;
  PFP_CATCH(c)
}

// d_dict / d_occ: the owners' distinct words back to back (owner 0 first) = the global dictionary, duplicate free by
// construction; d_gid_sent: the global id of every local word in the order pfp_dist_export_partition sent them
int pfp_dist_global_sort_distinct(pfp_ctx *c, const void *d_dict, uint64_t dict_bytes, const void *d_occ, uint64_t n_words,
                                  const void *d_gid_sent, uint32_t part, uint32_t parts, void *d_wslot_out, uint64_t out_info[8]) {
  if (!c || !c->dist || !d_dict || !d_occ || !d_gid_sent || !d_wslot_out || !out_info || parts < 1 || part >= parts) return PFP_EINVAL;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  DistState *ds = dist_of(c);
  PFP_REQUIRE(n_words >= 1 && n_words < 0xFFFFFFFFull, PFP_EINVAL, "bad word count");
  PFP_REQUIRE(ds->part_order.p, PFP_EINVAL, "pfp_dist_partition_words has not run");
  ds->G = Dictionary();
  ds->G.dsize = dict_bytes + 1;
  ds->G.bytes.alloc(c, ds->G.dsize + 64);
  PFP_HIP(hipMemcpyAsync(ds->G.bytes.p, d_dict, dict_bytes, hipMemcpyDeviceToDevice, c->stream));
  PFP_HIP(hipMemsetAsync(ds->G.bytes.p + dict_bytes, 0, 65, c->stream));
  word_table_from_bytes(c, ds->G, n_words);
  PFP_REQUIRE(ds->G.d == n_words, PFP_EFORMAT, "the global dictionary holds a different number of words than occ entries");
  ds->G.wocc.alloc(c, n_words);
  PFP_HIP(hipMemcpyAsync(ds->G.wocc.p, d_occ, n_words * 4, hipMemcpyDeviceToDevice, c->stream));
  const uint32_t dl = (uint32_t)ds->L.d;
  ds->gid_local.alloc(c, dl);
  static const bool by_occ = getenv("PFP_DIST_NO_OCC_ORDER") == nullptr;      // (diagnostic: the owners' order as it arrived)
  if (by_occ) {
    // most frequent words first: the pivots of the suffix sorter's pivot rounds become the words the variants deviate from
    DBuf<uint32_t> perm, gid(c, std::max<uint32_t>(dl, 1));
    { PhaseTimer t(c, &c->stats.ms_phrases); reorder_dictionary_by_occ(c, ds->G, perm); }
    hipLaunchKernelGGL(gather_u32_kernel, gdim(cdiv(dl, TB)), gdim(TB), 0, c->stream, dl, (const uint32_t *)d_gid_sent, perm.p, gid.p);
    hipLaunchKernelGGL(scatter_u32_kernel, gdim(cdiv(dl, TB)), gdim(TB), 0, c->stream, dl, ds->part_order.p, gid.p, ds->gid_local.p);
    sync(c);      // perm and gid are released on return
  } else
    hipLaunchKernelGGL(scatter_u32_kernel, gdim(cdiv(dl, TB)), gdim(TB), 0, c->stream, dl, ds->part_order.p, (const uint32_t *)d_gid_sent,
                       ds->gid_local.p);
  ds->have_gid = true;
  dist_sort_global(c, ds, part, parts, d_wslot_out, out_info);
  return PFP_OK;
  PFP_CATCH(c)
}

int pfp_dist_global_finish(pfp_ctx *c, const void *d_wslot_all, uint32_t parts, uint64_t my_word_base, void *d_sym_out) {
  if (!c || !c->dist || !d_wslot_all || !d_sym_out || parts < 1) return PFP_EINVAL;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  DistState *ds = dist_of(c);
  const uint32_t d = (uint32_t)ds->G.d;
  PFP_REQUIRE(d >= 1, PFP_EINVAL, "pfp_dist_global_sort has not run");
  compute_lexrank_from_slots(c, ds->G, (const uint64_t *)d_wslot_all, parts, ds->ix);
  if (c->debug) validate_lexrank(c, ds->G, ds->ix);
  ds->occ_lex.alloc(c, d);
  hipLaunchKernelGGL(occ_lex_kernel, gdim(cdiv(d, TB)), gdim(TB), 0, c->stream, d, ds->ix.lexrank.p, ds->G.wocc.p,
                     ds->occ_lex.p, (uint32_t *)nullptr);
  if (ds->have_gid)
    hipLaunchKernelGGL(dist_sym_gid_kernel, gdim(cdiv(ds->P_local, TB)), gdim(TB), 0, c->stream, ds->P_local, ds->L.pid.p,
                       ds->gid_local.p, ds->ix.lexrank.p, (uint32_t *)d_sym_out);
  else
    hipLaunchKernelGGL(dist_sym_kernel, gdim(cdiv(ds->P_local, TB)), gdim(TB), 0, c->stream, ds->P_local, ds->L.pid.p,
                       my_word_base, ds->G.pid.p, ds->ix.lexrank.p, (uint32_t *)d_sym_out);
  PFP_HIP(hipGetLastError());
  sync(c);
  return PFP_OK;
  PFP_CATCH(c)
}

// the two steps above on one rank holding the whole suffix array (no exchange in between)
int pfp_dist_global(pfp_ctx *c, const void *d_union, uint64_t union_bytes, const void *d_union_occ, uint64_t n_union,
                    uint64_t my_word_base, void *d_sym_out, uint64_t out_info[3]) {
  if (!c || !c->dist || !d_union || !d_union_occ || !d_sym_out || !out_info) return PFP_EINVAL;
  uint64_t info[8];
  uint64_t *wslot = nullptr;
  if (hipSetDevice(c->device) != hipSuccess || hipMalloc((void **)&wslot, (n_union ? n_union : 1) * 8) != hipSuccess) return PFP_ENOMEM;
  int rc = pfp_dist_global_sort(c, d_union, union_bytes, d_union_occ, n_union, 0, 1, wslot, info);
  if (rc == PFP_OK) rc = pfp_dist_global_finish(c, wslot, 1, my_word_base, d_sym_out);
  (void)hipFree(wslot);
  if (rc == PFP_OK) { out_info[0] = info[0]; out_info[1] = info[1]; out_info[2] = info[2]; }
  return rc;
}

int pfp_dist_parse_sort(pfp_ctx *c, const void *d_sym, uint64_t P, uint32_t part, uint32_t parts, void *d_sa_out, uint64_t out_info[4]) {
  if (!c || !c->dist || !d_sym || !d_sa_out || !out_info || parts < 1 || part >= parts) return PFP_EINVAL;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  DistState *ds = dist_of(c);
  PFP_REQUIRE(ds->occ_lex.p && ds->G.d, PFP_EINVAL, "pfp_dist_global_finish has not run");
  PFP_REQUIRE(P >= 2, PFP_ESHORT, "parse has fewer than 2 phrases (bwtparse.c:244)");
  PhaseTimer t(c, &c->stats.ms_sa_parse);
  DBuf<uint32_t> sym(c, P + 1);      // the parse and its end symbol (bwtparse.c:212-230)
  PFP_HIP(hipMemcpyAsync(sym.p, d_sym, P * 4, hipMemcpyDeviceToDevice, c->stream));
  PFP_HIP(hipMemsetAsync(sym.p + P, 0, 4, c->stream));
  SuffixOrder so;
  sort_int_suffixes_range(c, sym.p, P + 1, (uint32_t)ds->G.d, ds->occ_lex.p, (uint32_t)ds->G.d, part, parts, so);
  if (so.complete && so.N) PFP_HIP(hipMemcpyAsync(d_sa_out, so.sa.p, so.N * 4, hipMemcpyDeviceToDevice, c->stream));
  sync(c);
  out_info[0] = so.N; out_info[1] = so.slot_base; out_info[2] = so.complete ? 1 : 0; out_info[3] = so.rounds;
  ds->parse_share_rounds = so.rounds;
  return PFP_OK;
  PFP_CATCH(c)
}

int pfp_dist_set_parse_sa(pfp_ctx *c, const void *d_sa, uint64_t count) {
  if (!c || !c->dist || (!d_sa && count)) return PFP_EINVAL;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  DistState *ds = dist_of(c);
  ds->parse_sa.release(); ds->parse_sa_n = 0;
  if (!count) return PFP_OK;
  ds->parse_sa.alloc(c, count);
  PFP_HIP(hipMemcpyAsync(ds->parse_sa.p, d_sa, count * 4, hipMemcpyDeviceToDevice, c->stream));
  sync(c);
  ds->parse_sa_n = count;
  return PFP_OK;
  PFP_CATCH(c)
}

int pfp_dist_merge(pfp_ctx *c, const void *d_sym, uint64_t P, const void *d_last, const void *d_sai, int flags,
                   uint64_t n_total, uint64_t out_lo, uint64_t out_hi, void *d_bwt_slice, void *d_sa_slice) {
  if (!c || !c->dist || !d_sym || !d_last || !d_bwt_slice || (flags && !d_sai) || ((flags & PFP_FLAG_SA) && !d_sa_slice)) return PFP_EINVAL;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  DistState *ds = dist_of(c);
  check_args(ds->w, 10, flags);
  PFP_REQUIRE(flags == ds->flags, PFP_EINVAL, "output flags differ from those announced at pfp_dist_local_parse");
  PFP_REQUIRE(out_lo <= out_hi && out_hi <= n_total + 1, PFP_EINVAL, "bad output slice");
  ParseBWT pb;
  { PhaseTimer t(c, &c->stats.ms_sa_parse);
    PFP_REQUIRE(!ds->parse_sa_n || ds->parse_sa_n == P + 1, PFP_EINVAL, "the gathered suffix array of the parse has " +
                std::to_string(ds->parse_sa_n) + " entries, the parse " + std::to_string(P) + " phrases");
    parse_bwt(c, (const uint32_t *)d_sym, P, (const uint8_t *)d_last, flags ? (const uint64_t *)d_sai : nullptr,
              ds->occ_lex.p, ds->G.d, pb, ds->parse_sa_n ? ds->parse_sa.p : nullptr);
    c->stats.sa_rounds_parse = ds->parse_sa_n ? ds->parse_share_rounds : pb.rounds;
    ds->parse_sa.release(); ds->parse_sa_n = 0; }
  if (c->debug) validate_parse_bwt(c, pb);
  ds->out = BwtOutputs();       // (-s / -e with d_sa_slice == NULL: what pfp_dist_sample_runs reads afterwards)
  ds->out_lo = out_lo;
  ds->slice_empty = out_hi == out_lo;
  BwtOutputs &bo = ds->out;
  bo.d_bwt = (uint8_t *)d_bwt_slice; bo.d_sa = (uint64_t *)d_sa_slice;
  bool empty_share = false;
  PhaseTimer t_merge(c, &c->stats.ms_merge);
  with_width(ds->ord.wide, [&](auto tag) {
    using I = decltype(tag);
    auto &so = ds->ord.get<I>();
    if (so.range) {
      // the held slots are one contiguous range of SA(D): they emit exactly [out_lo, out_hi)
      PFP_REQUIRE(so.complete, PFP_EINVAL, "this share of the suffix array is incomplete: redo pfp_dist_global_sort with parts = 1");
      PFP_REQUIRE(out_hi - out_lo == ds->local_total, PFP_EINVAL, "output range does not match this share's occurrence count");
      if (so.N == 0) { empty_share = true; return; }      // an empty share of the key space emits nothing
      merge_bwt<I>(c, ds->G, ds->ix, so, pb, ds->occ_lex.p, ds->w, flags, ds->local_total, bo, 0, ~0ull, out_lo, n_total + 1);
    } else {
      merge_bwt<I>(c, ds->G, ds->ix, so, pb, ds->occ_lex.p, ds->w, flags, n_total + 1, bo, out_lo, out_hi);
    }
  });
  (void)empty_share;
  { pfp_stats &st = c->stats;
    st.hard_groups = bo.hard_groups; st.hard_chars = bo.hard_chars; st.hard_big_groups = bo.hard_big_groups;
    st.hard_max_members = bo.hard_max_members; st.hard_minor_groups = bo.hard_minor_groups; st.hard_minor_chars = bo.hard_minor_chars; }
  sync(c);
  return PFP_OK;
  PFP_CATCH(c)
}

int pfp_dist_sample_runs(pfp_ctx *c, int run_end, int drop_edge, void *d_out10, uint64_t cap_pairs, uint64_t *n_pairs) {
  if (!c || !c->dist || !n_pairs) return PFP_EINVAL;
  *n_pairs = 0;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  DistState *ds = dist_of(c);
  PFP_REQUIRE((ds->flags & (PFP_FLAG_SSA | PFP_FLAG_ESA)) && !(ds->flags & PFP_FLAG_SA), PFP_EINVAL,
              "pfp_dist_sample_runs: the chain was not run with -s / -e");
  PFP_REQUIRE((run_end ? PFP_FLAG_ESA : PFP_FLAG_SSA) & ds->flags, PFP_EINVAL, "this sampled file was not asked for");
  if (ds->slice_empty) return PFP_OK;      // this rank's slice of the BWT holds no position
  PFP_REQUIRE(ds->out.slice_n && ds->out.sa_c.p, PFP_EINVAL, "the last pfp_dist_merge left no run maps (it must run before, with an SA-less slice)");
  const SaView sv = sa_view(ds->out);
  const uint64_t k = sample_runs_maps(c, sv, ds->out.slice_n, run_end != 0, drop_edge != 0, ds->out_lo, nullptr);
  *n_pairs = k;
  if (!d_out10) return PFP_OK;
  PFP_REQUIRE(k <= cap_pairs, PFP_ELIMIT, "output buffer holds " + std::to_string(cap_pairs) + " pairs, the slice has " + std::to_string(k));
  sample_runs_maps(c, sv, ds->out.slice_n, run_end != 0, drop_edge != 0, ds->out_lo, (uint8_t *)d_out10);
  sync(c);
  return PFP_OK;
  PFP_CATCH(c)
}

void pfp_dist_release(pfp_ctx *c) {
  if (!c || !c->dist) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  delete reinterpret_cast<DistState *>(c->dist);
  c->dist = nullptr;
}

}  // extern "C"

extern "C" {

// ---------------------------------------------------------------- micro entry points
int pfp_stage_text_dev(pfp_ctx *c, const void *d_text, uint64_t n, int w) {
  if (!c || (!d_text && n)) return PFP_EINVAL;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  PFP_REQUIRE(w >= 1 && w <= 4096, PFP_EINVAL, "bad window");
  StagedText *&s = staged_of(c);
  if (!s) s = new StagedText();
  s->stage(c, d_text, true, n, w);
  sync(c);
  return PFP_OK;
  PFP_CATCH(c)
}

int pfp_scan_staged(pfp_ctx *c, uint64_t p, uint64_t *n_ends) {
  if (!c) return PFP_EINVAL;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  StagedText *s = staged_of(c);
  PFP_REQUIRE(s && s->buf.p, PFP_EINVAL, "no staged text");
  DBuf<uint64_t> d_ends;
  uint64_t used = 0;
  uint64_t k = scan_text(c, *s, s->n, s->w, p, d_ends, &used);
  sync(c);
  if (n_ends) *n_ends = k;
  return PFP_OK;
  PFP_CATCH(c)
}

// diagnostic: the first-round sort of radix.hip on caller data - keys (and 32-bit values, may be null) sorted in place, stable
// on key bits [lo, hi)
int pfp_debug_msd_sort(pfp_ctx *c, uint64_t *keys, uint32_t *vals, uint64_t n, int lo, int hi) {
  if (!c || (!keys && n)) return PFP_EINVAL;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  if (!n) return PFP_OK;
  DBuf<uint64_t> k(c, n), ka(c, n);
  h2d(c, k.p, keys, n);
  if (vals) {
    DBuf<uint32_t> v(c, n), va(c, n);
    h2d(c, v.p, vals, n);
    msd_sort_pairs_db<uint32_t>(c, k, ka, v, va, n, lo, hi);
    d2h(c, vals, v.p, n);
    d2h(c, keys, k.p, n);
    sync(c);
  } else {
    msd_sort_keys_db(c, k, ka, n, lo, hi);
    d2h(c, keys, k.p, n);
    sync(c);
  }
  return PFP_OK;
  PFP_CATCH(c)
}

int pfp_scan_k1_enqueue(pfp_ctx *c, uint64_t p) {
  if (!c) return PFP_EINVAL;
  PFP_TRY(c)
  PFP_HIP(hipSetDevice(c->device));
  StagedText *s = staged_of(c);
  PFP_REQUIRE(s && s->buf.p, PFP_EINVAL, "no staged text");
  // scratch kept across calls so that the enqueue itself does no allocation
  if (!c->k1scratch) c->k1scratch = new K1Scratch();
  K1Scratch &sc = *reinterpret_cast<K1Scratch *>(c->k1scratch);
  uint64_t nchunks = cdiv64(s->n, 16);
  if (sc.n != s->n || !sc.flags.p) {
    sc.flags.alloc(c, nchunks + 1); sc.bcnt.alloc(c, cdiv64(nchunks, 256) + 1); sc.fbad.alloc(c, 1); sc.n = s->n;
  }
  scan_flags(c, s->tbase(), s->n, s->w, p, sc.flags.p, sc.bcnt.p, sc.fbad.p);
  return PFP_OK;
  PFP_CATCH(c)
}

}  // extern "C"
