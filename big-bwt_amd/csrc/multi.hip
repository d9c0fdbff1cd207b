// multi.hip -- pfp_bigbwt_files_multi: one BWT on N GPUs of one node from ONE process.
//
// One host thread and one pfp_ctx per GPU; the threads run the multi-GPU chain (the
// pfp_dist_* calls of pipeline.hip) and meet in collectives: RCCL over xGMI (grouped ncclSend / ncclRecv of the ranks' pieces,
// one communicator per thread from ncclCommInitAll; librccl is loaded with dlopen when the first multi-GPU call comes, so a
// single-GPU user of the library never maps it), or - PFP_MULTI_LOOPBACK=1, for tests on a one-GPU box: every rank on the same
// device - plain device copies between the ranks' buffers.  This is the host side of `bigbwt -G N` (host/bigbwt.c); the
// Python driver (dist.py over torch.distributed, used by bench.py) is the same sequence of calls and collectives.
//
// The reference's analogue is the threaded parser / merger of one process (pscan.hpp:114-165 splits the input by byte range,
// pscan.cpp:137-205 shards the dictionary by hash, pfthreads.hpp:171-176, 369-376, 456-493 shard the suffix array by range and
// pwrite the output ranges).
//
// Error protocol (as dist._Step): after every rank-local step the ranks exchange their status before any of them enters the
// next data collective; a rank whose step failed makes all ranks return that error instead of leaving the others waiting.
#include <dlfcn.h>
#include <fcntl.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "common.hpp"

namespace pfp {
namespace {

// ---------------------------------------------------------------- RCCL, resolved at run time
typedef struct ncclComm *ncclComm_t;
enum { kNcclUint8 = 1 };                      // ncclDataType_t: ncclUint8 == ncclChar + 1 (rccl.h)
struct Rccl {
  void *h = nullptr;
  int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*CommAbort)(ncclComm_t) = nullptr;
  int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  bool complete() const { return h && CommInitAll && CommDestroy && CommAbort && Send && Recv && AllGather && GroupStart && GroupEnd && GetErrorString; }
  // (idempotent: a call after a partial failure starts over instead of returning a table with holes)
  bool load(std::string &err) {
    if (complete()) return true;
    if (h) { dlclose(h); h = nullptr; }
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) { h = dlopen(nm, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    if (!h) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
    auto sym = [&](const char *n) { void *p = dlsym(h, n); if (!p) err = std::string("librccl lacks ") + n; return p; };
    CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
    CommAbort = (decltype(CommAbort))sym("ncclCommAbort");
    Send = (decltype(Send))sym("ncclSend");
    Recv = (decltype(Recv))sym("ncclRecv");
    AllGather = (decltype(AllGather))sym("ncclAllGather");
    GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
    GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
    if (complete()) return true;
    dlclose(h); h = nullptr;
    return false;
  }
};

struct Failure { int code; std::string msg; };
[[noreturn]] void fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  throw Failure{code, buf};
}
#define MG_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) fail(e_ == hipErrorOutOfMemory ? PFP_ENOMEM : PFP_EHIP, "%s: %s", #x, hipGetErrorString(e_)); } while (0)

// device buffer of one rank (plain hipMalloc: these are the exchange buffers between the library calls)
struct Dev {
  void *p = nullptr; uint64_t bytes = 0;
  Dev() = default;
  explicit Dev(uint64_t n) { alloc(n); }
  Dev(const Dev &) = delete; Dev &operator=(const Dev &) = delete;
  Dev(Dev &&o) noexcept { p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
  Dev &operator=(Dev &&o) noexcept { if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; } return *this; }
  ~Dev() { release(); }
  void alloc(uint64_t n) { release(); MG_HIP(hipMalloc(&p, n ? n : 16)); bytes = n; }
  void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
  uint8_t *u8() const { return (uint8_t *)p; }
};

// the ranks are threads of one process: small host values are exchanged through shared memory at a barrier (RCCL only
// carries the bulk device data).  A rank that leaves unexpectedly aborts the barrier, so nobody waits for it forever.
class Barrier {
  std::mutex m; std::condition_variable cv; int n, waiting = 0; uint64_t gen = 0; bool aborted = false;
 public:
  explicit Barrier(int n_) : n(n_) {}
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    if (aborted) fail(PFP_EHIP, "another rank left the chain");
    const uint64_t g = gen;
    if (++waiting == n) { waiting = 0; gen++; cv.notify_all(); return; }
    cv.wait(lk, [&] { return gen != g || aborted; });
    if (gen == g) fail(PFP_EHIP, "another rank left the chain");
  }
  void abort() { std::lock_guard<std::mutex> lk(m); aborted = true; cv.notify_all(); }
};

// what the ranks (threads) share
struct Shared {
  int size;
  bool loopback;
  Rccl *rccl = nullptr;
  std::vector<ncclComm_t> comms;
  Barrier bar;
  std::vector<const void *> ptr;            // loopback: every rank's current send buffer
  std::vector<std::vector<uint64_t>> vals;  // small host values of the current exchange
  explicit Shared(int n, bool lb) : size(n), loopback(lb), bar(n), ptr(n), vals(n) {}
  // a rank that fails takes everybody out: the host barrier is aborted and every communicator too, so that send / recv kernels
  // already waiting for the failed rank end (ncclCommAbort may be called from any thread) instead of hanging their streams
  std::atomic<bool> aborted{false};
  std::mutex abort_mu;
  bool comms_gone = false;
  void abort_all() {
    aborted.store(true);
    bar.abort();
    std::lock_guard<std::mutex> lk(abort_mu);
    if (!loopback && rccl && !comms_gone) {
      for (auto &cm : comms) if (cm) { (void)rccl->CommAbort(cm); cm = nullptr; }
      comms_gone = true;
    }
  }
};

__global__ void add_u32_kernel(uint32_t *v, uint64_t n, uint32_t add) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] += add;
}

// one rank's view of the collectives; every call is made by all ranks in the same order
struct Coll {
  Shared &S; int rank, size; hipStream_t stream;
  Coll(Shared &s, int r, hipStream_t st) : S(s), rank(r), size(s.size), stream(st) {}

  // k 64-bit host values from every rank -> out[r * k + j]
  void allgather_u64(const uint64_t *in, int k, std::vector<uint64_t> &out) {
    out.assign((size_t)size * k, 0);
    S.vals[rank].assign(in, in + k);
    S.bar.wait();
    for (int r = 0; r < size; r++) std::copy(S.vals[r].begin(), S.vals[r].end(), out.begin() + (size_t)r * k);
    S.bar.wait();
  }
  // the receive buffer of a data collective: allocated before anything moves, and the ranks agree that everybody got theirs
  Dev recv_buffer(uint64_t bytes, const char *what) {
    Dev out;
    uint64_t ok = 1;
    try { out.alloc(bytes + 16); } catch (const Failure &) { ok = 0; }
    std::vector<uint64_t> all;
    allgather_u64(&ok, 1, all);
    for (int r = 0; r < size; r++) if (!all[r]) fail(PFP_ENOMEM, "rank %d: out of device memory for the %s buffer", r, what);
    return out;
  }
  // every rank's `bytes` from d_send, back to back in rank order; counts[r] = rank r's bytes
  Dev allgatherv(const void *d_send, uint64_t bytes, std::vector<uint64_t> &counts) {
    allgather_u64(&bytes, 1, counts);
    uint64_t total = 0;
    for (uint64_t c : counts) total += c;
    Dev out = recv_buffer(total, "allgather");
    if (S.loopback) {
      S.ptr[rank] = d_send;
      S.bar.wait();
      uint64_t off = 0;
      for (int r = 0; r < size; r++) {
        if (counts[r]) MG_HIP(hipMemcpyAsync(out.u8() + off, S.ptr[r], counts[r], hipMemcpyDeviceToDevice, stream));
        off += counts[r];
      }
      MG_HIP(hipStreamSynchronize(stream));
      S.bar.wait();
      return out;
    }
    exchange_same(d_send, bytes, out, counts);
    wait_stream();
    return out;
  }
  // d_send = the pieces for ranks 0 .. size-1 back to back (send[r] bytes each) -> what everybody sent here, in rank order
  Dev alltoallv(const void *d_send, const std::vector<uint64_t> &send, std::vector<uint64_t> &recv) {
    std::vector<uint64_t> table;
    allgather_u64(send.data(), size, table);          // table[s * size + r] = bytes s sends to r
    recv.assign(size, 0);
    uint64_t total = 0;
    for (int s = 0; s < size; s++) { recv[s] = table[(size_t)s * size + rank]; total += recv[s]; }
    Dev out = recv_buffer(total, "all-to-all");
    if (S.loopback) {
      S.ptr[rank] = d_send;
      S.bar.wait();
      uint64_t off = 0;
      for (int s = 0; s < size; s++) {
        uint64_t soff = 0;
        for (int r = 0; r < rank; r++) soff += table[(size_t)s * size + r];
        if (recv[s]) MG_HIP(hipMemcpyAsync(out.u8() + off, (const uint8_t *)S.ptr[s] + soff, recv[s], hipMemcpyDeviceToDevice, stream));
        off += recv[s];
      }
      MG_HIP(hipStreamSynchronize(stream));
      S.bar.wait();
      return out;
    }
    exchange(d_send, send, out.p, recv);
    wait_stream();
    return out;
  }

 public:
  bool force_grouped = false;      // (self test: the send / recv form of the all-gather whatever the counts)
  bool fail_in_group = false;      // (self test: an error between ncclGroupStart and ncclGroupEnd, after the first send was queued)
 private:
  void nccl_check(int rc, const char *what) { if (rc != 0) fail(PFP_EHIP, "rank %d: RCCL %s: %s", rank, what, S.rccl->GetErrorString(rc)); }
  // the stream's RCCL work is done - or some rank has given up (its abort_all ends the kernels this stream waits in)
  void wait_stream() {
    for (;;) {
      const hipError_t e = hipStreamQuery(stream);
      if (e == hipSuccess) return;
      if (e != hipErrorNotReady) { (void)hipGetLastError(); fail(PFP_EHIP, "rank %d: stream: %s", rank, hipGetErrorString(e)); }
      if (S.aborted.load()) fail(PFP_EHIP, "another rank left the chain");
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
  }
  // a group that was opened is closed on the way out of an error, or the communicator stays in group mode
  struct Group {
    Rccl &R; bool open = false;
    explicit Group(Rccl &r) : R(r) {}
    ~Group() { if (open) (void)R.GroupEnd(); }
  };
  // RCCL: piece r of d_send (send[r] bytes, back to back) goes to rank r; recv[s] bytes arrive from rank s, back to back
  void exchange(const void *d_send, const std::vector<uint64_t> &send, void *d_recv, const std::vector<uint64_t> &recv) {
    Rccl &R = *S.rccl;
    Group g(R);
    nccl_check(R.GroupStart(), "group start"); g.open = true;
    uint64_t so = 0, ro = 0;
    for (int r = 0; r < size; r++) {
      if (send[r]) nccl_check(R.Send((const uint8_t *)d_send + so, send[r], kNcclUint8, r, S.comms[rank], stream), "send");
      if (fail_in_group) fail(PFP_EHIP, "rank %d: injected failure inside a group", rank);
      if (recv[r]) nccl_check(R.Recv((uint8_t *)d_recv + ro, recv[r], kNcclUint8, r, S.comms[rank], stream), "recv");
      so += send[r]; ro += recv[r];
    }
    g.open = false;
    nccl_check(R.GroupEnd(), "group end");
  }
  // RCCL: the same `bytes` of d_send to every rank.  ONE collective where that costs little: ncclAllGather when every rank
  // brings the same number of bytes, and - counts within a quarter of each other - ncclAllGather of pieces padded to the
  // largest, closed up by device copies; grouped send / recv for the rest (a rank that brings nothing next to one that brings much)
  void exchange_same(const void *d_send, uint64_t bytes, Dev &out, const std::vector<uint64_t> &recv) {
    Rccl &R = *S.rccl;
    uint64_t mx = 0, total = 0;
    for (uint64_t c : recv) { mx = std::max(mx, c); total += c; }
    if (!total) return;
    bool equal = true;
    for (uint64_t c : recv) equal = equal && c == mx;
    if (equal && !force_grouped) {
      nccl_check(R.AllGather(d_send, out.p, mx, kNcclUint8, S.comms[rank], stream), "all-gather");
      return;
    }
    if (!force_grouped && (uint64_t)size * mx <= total + total / 4) {
      // (every rank takes this branch or none: the counts are the same everywhere)
      Dev pad(mx + 16), all((uint64_t)size * mx + 16);
      if (bytes) MG_HIP(hipMemcpyAsync(pad.p, d_send, bytes, hipMemcpyDeviceToDevice, stream));
      nccl_check(R.AllGather(pad.p, all.p, mx, kNcclUint8, S.comms[rank], stream), "all-gather");
      uint64_t ro = 0;
      for (int r = 0; r < size; r++) {
        if (recv[r]) MG_HIP(hipMemcpyAsync(out.u8() + ro, all.u8() + (uint64_t)r * mx, recv[r], hipMemcpyDeviceToDevice, stream));
        ro += recv[r];
      }
      wait_stream();      // (pad / all go out of scope)
      return;
    }
    Group g(R);
    nccl_check(R.GroupStart(), "group start"); g.open = true;
    uint64_t ro = 0;
    for (int r = 0; r < size; r++) {
      if (bytes) nccl_check(R.Send(d_send, bytes, kNcclUint8, r, S.comms[rank], stream), "send");
      if (recv[r]) nccl_check(R.Recv((uint8_t *)out.p + ro, recv[r], kNcclUint8, r, S.comms[rank], stream), "recv");
      ro += recv[r];
    }
    g.open = false;
    nccl_check(R.GroupEnd(), "group end");
  }
};

struct RankResult { int code = PFP_OK; std::string msg; pfp_multi_stats st{}; };

struct Job {
  const uint8_t *text; uint64_t n; int w; uint64_t p; int flags; uint64_t halo; const char *base;
};

// A rank-local step: library calls and the allocations around them.  Whatever fails - a PFP_E* code, an allocation - is kept
// as this rank's status until the ranks have compared theirs (agree): nobody enters a data collective alone.
struct Step {
  int rc = PFP_OK; std::string msg;
  template <class F> void run(F &&f) {
    if (rc != PFP_OK) return;
    try { rc = f(); } catch (const Failure &e) { rc = e.code; msg = e.msg; }
  }
};
// all continue, or all leave with the code of the first failing rank
void agree(Coll &C, Step &st, const char *what, pfp_ctx *ctx, uint64_t *extra = nullptr, int n_extra = 0, std::vector<uint64_t> *gathered = nullptr) {
  std::vector<uint64_t> v(1 + n_extra), all;
  v[0] = (uint64_t)(int64_t)st.rc;
  for (int k = 0; k < n_extra; k++) v[1 + k] = extra[k];
  C.allgather_u64(v.data(), 1 + n_extra, all);
  for (int r = 0; r < C.size; r++) {
    const int code = (int)(int64_t)all[(size_t)r * (1 + n_extra)];
    if (code == 0) continue;
    if (r == C.rank) fail(code, "rank %d: %s: %s (%s)", r, what, pfp_strerror(code), !st.msg.empty() ? st.msg.c_str() : (ctx ? pfp_last_error(ctx) : ""));
    fail(code, "%s failed on rank %d (%s); every rank stops here", what, r, pfp_strerror(code));
  }
  if (gathered) *gathered = std::move(all);
}

void create_sized(const std::string &path, uint64_t size) {      // pfbwt.cpp:132-141 opens its outputs "wb"
  const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (fd < 0 || ftruncate(fd, (off_t)size) != 0) { if (fd >= 0) close(fd); fail(PFP_EINVAL, "cannot create %s", path.c_str()); }
  fsync(fd);
  close(fd);
}

void run_rank(Shared &S, int rank, int device, const Job &J, RankResult &res) {
  const int size = S.size;
  pfp_ctx *ctx = nullptr;
  try {
    Step st;
    st.run([&] { MG_HIP(hipSetDevice(device)); return pfp_ctx_create(&ctx, device); });
    Coll C(S, rank, st.rc == PFP_OK ? (hipStream_t)pfp_ctx_stream(ctx) : nullptr);
    agree(C, st, "context", ctx);
    hipStream_t stream = C.stream;
    const auto t_start = std::chrono::steady_clock::now();
    const int w = J.w; const uint64_t p = J.p; const int flags = J.flags;
    const bool want_sai = flags != 0;
    // ---- this rank's byte range (pscan.hpp:114-165) and the tail of its left neighbour's
    const uint64_t lo_t = J.n * (uint64_t)rank / size, hi_t = J.n * (uint64_t)(rank + 1) / size, n_shard = hi_t - lo_t;
    const uint64_t tail_len = std::min<uint64_t>(J.halo, n_shard);
    Dev d_shard, d_local;
    st.run([&] {
      d_shard.alloc(n_shard + 16);
      if (n_shard) MG_HIP(hipMemcpyAsync(d_shard.p, J.text + lo_t, n_shard, hipMemcpyHostToDevice, stream));
      MG_HIP(hipStreamSynchronize(stream));
      return PFP_OK;
    });
    std::vector<uint64_t> lens;
    agree(C, st, "text upload", ctx, const_cast<uint64_t *>(&n_shard), 1, &lens);
    uint64_t n_total = 0, goff = 0;
    for (int r = 0; r < size; r++) { n_total += lens[(size_t)r * 2 + 1]; if (r < rank) goff += lens[(size_t)r * 2 + 1]; }
    uint64_t left_len = 0;
    {
      std::vector<uint64_t> tails_cnt;
      Dev tails = C.allgatherv(d_shard.u8() + (n_shard - tail_len), tail_len, tails_cnt);
      uint64_t toff = 0;
      for (int r = 0; r + 1 < rank; r++) toff += tails_cnt[r];
      left_len = rank > 0 ? tails_cnt[rank - 1] : 0;
      st.run([&] {
        d_local.alloc(left_len + n_shard + 64);
        if (left_len) MG_HIP(hipMemcpyAsync(d_local.p, tails.u8() + toff, left_len, hipMemcpyDeviceToDevice, stream));
        if (n_shard) MG_HIP(hipMemcpyAsync(d_local.u8() + left_len, d_shard.p, n_shard, hipMemcpyDeviceToDevice, stream));
        MG_HIP(hipStreamSynchronize(stream));
        return PFP_OK;
      });
      d_shard.release();
    }
    const uint64_t n_local = left_len + n_shard;
    // ---- one trigger set for all ranks (proposals for splitting giant phrases; rank 0 bans its first window's hash: it must
    //      not become an extra trigger)
    std::vector<uint32_t> extra;
    // ---- the collection's parse plan (round 4): rank 0 holds the text's first window and makes the plan - which hash cuts the
    //      text, with which seed, how densely - every rank receives it; a candidate density is settled from the ranks' samples
    uint64_t plan[4] = {0, 0, 0, 0};
    {
      uint64_t pv[5] = {0, 0, 0, 0, ~0ull};
      if (rank == 0) st.run([&] { return pfp_dist_parse_plan(ctx, J.text, std::min<uint64_t>(n_shard, 32), w, p, (uint32_t)size, pv, &pv[4]); });
      std::vector<uint64_t> plans;
      agree(C, st, "parse plan", ctx, pv, 5, &plans);
      for (int k = 0; k < 4; k++) plan[k] = plans[1 + k];      // (rank 0's row)
      const uint64_t first_hash = plans[5];
      if (plan[3]) {          // a candidate density: the ranks' samples of their cuts, gathered, settle it
        const uint64_t scap = std::max<uint64_t>(4096, n_local / std::max<uint64_t>(p, 1) / 2 + 4096);
        Dev smp;
        uint64_t n_sample = 0;
        st.run([&] {
          smp.alloc(scap * 8 + 16);
          uint32_t hashes[8]; uint32_t nh = 0;
          return pfp_dist_propose_triggers2(ctx, d_local.p, n_local, left_len, w, p, plan, hashes, &nh, smp.p, scap, &n_sample);
        });
        agree(C, st, "sampling the cuts", ctx);
        std::vector<uint64_t> sc;
        Dev all = C.allgatherv(smp.p, n_sample * 8, sc);
        uint64_t tot = 0;
        for (uint64_t x : sc) tot += x;
        st.run([&] { return pfp_dist_decide_density(ctx, all.p, tot / 8, p, plan); });
        agree(C, st, "parse density", ctx);
      }
      uint64_t prop[9];
      for (int k = 0; k < 9; k++) prop[k] = ~0ull;
      st.run([&] {
        uint32_t hashes[8]; uint32_t nh = 0;
        uint64_t none = 0;
        const int rc = pfp_dist_propose_triggers2(ctx, d_local.p, n_local, left_len, w, p, plan, hashes, &nh, nullptr, 0, &none);
        if (rc == PFP_OK) for (uint32_t k = 0; k < nh && k < 8; k++) prop[k] = hashes[k];
        return rc;
      });
      if (rank == 0) prop[8] = first_hash;
      std::vector<uint64_t> props;
      agree(C, st, "trigger proposal", ctx, prop, 9, &props);
      auto at = [&](int r, int q) { return props[(size_t)r * 10 + 1 + q]; };
      std::vector<uint64_t> banned;
      for (int r = 0; r < size; r++) if (at(r, 8) != ~0ull) banned.push_back(at(r, 8));
      for (int k = 0; k < 8; k++)
        for (int r = 0; r < size; r++) {
          // the k-th surviving proposal of rank r: at most 32 in all, round-robin over the ranks
          int seen = -1; uint64_t v = ~0ull;
          for (int q = 0; q < 8; q++) {
            const uint64_t x = at(r, q);
            if (x == ~0ull || std::find(banned.begin(), banned.end(), x) != banned.end()) continue;
            if (++seen == k) { v = x; break; }
          }
          if (v != ~0ull && extra.size() < 32 && std::find(extra.begin(), extra.end(), (uint32_t)v) == extra.end()) extra.push_back((uint32_t)v);
        }
      std::sort(extra.begin(), extra.end());
    }
    // ---- local parse
    uint64_t sizes[4] = {0, 0, 0, 0};
    st.run([&] {
      const int rc = pfp_dist_local_parse2(ctx, d_local.p, n_local, left_len, w, p, rank == 0, rank == size - 1, goff, flags, plan,
                                           extra.empty() ? nullptr : extra.data(), (uint32_t)extra.size(), sizes);
      // the next rank re-derives this rank's last phrase boundary from the last tail_len bytes of the shard
      if (rc == PFP_OK && rank < size - 1 && (int64_t)sizes[3] - (int64_t)(w - 1) < (int64_t)(n_local - tail_len))
        fail(PFP_ELIMIT, "last phrase boundary lies outside the %llu-byte halo; raise --halo", (unsigned long long)tail_len);
      return rc;
    });
    agree(C, st, "local parse", ctx);
    const uint64_t P_local = sizes[2];
    Dev d_last, d_sai, d_sym, sendb, sendo;
    std::vector<uint64_t> wsplit(size, 0), bsplit(size, 0);
    st.run([&] {
      d_last.alloc(P_local + 16); d_sym.alloc(P_local * 4 + 16);
      if (want_sai) d_sai.alloc(P_local * 8 + 16);
      int rc = pfp_dist_export_local(ctx, nullptr, nullptr, d_last.p, want_sai ? d_sai.p : nullptr);
      d_local.release();
      // ---- exchange A: words travel to the owner of their hash class and come back as global ids (pscan.cpp:137-205)
      std::vector<uint64_t> counts((size_t)2 * size, 0);
      if (rc == PFP_OK) rc = pfp_dist_partition_words(ctx, (uint32_t)size, counts.data());
      if (rc != PFP_OK) return rc;
      uint64_t sw = 0, sb = 0;
      for (int o = 0; o < size; o++) { wsplit[o] = counts[2 * o] * 4; bsplit[o] = counts[2 * o + 1]; sw += counts[2 * o]; sb += counts[2 * o + 1]; }
      sendb.alloc(sb + 16); sendo.alloc(sw * 4 + 16);
      return pfp_dist_export_partition(ctx, sendb.p, sendo.p);
    });
    agree(C, st, "word partition", ctx);
    std::vector<uint64_t> rbs, ros;
    Dev rb = C.alltoallv(sendb.p, bsplit, rbs);
    Dev ro = C.alltoallv(sendo.p, wsplit, ros);
    sendb.release(); sendo.release();
    uint64_t rb_bytes = 0, ro_words = 0;
    for (int s = 0; s < size; s++) { rb_bytes += rbs[s]; ro_words += ros[s] / 4; }
    Dev pid, ownb, owno;
    uint64_t own[2] = {0, 0};
    st.run([&] {
      pid.alloc(ro_words * 4 + 16);
      const int rc = pfp_dist_owner_dedup(ctx, rb.p, rb_bytes, ro.p, ro_words, pid.p, own);
      if (rc != PFP_OK) return rc;
      ownb.alloc(own[1] + 16); owno.alloc(own[0] * 4 + 16);
      return pfp_dist_export_owned(ctx, ownb.p, owno.p);
    });
    rb.release(); ro.release();
    std::vector<uint64_t> owned;
    agree(C, st, "owner dedup", ctx, own, 1, &owned);
    uint64_t base = 0, n_union = 0;
    for (int r = 0; r < size; r++) { if (r < rank) base += owned[(size_t)r * 2 + 1]; n_union += owned[(size_t)r * 2 + 1]; }
    st.run([&] {
      if (ro_words) {
        hipLaunchKernelGGL(add_u32_kernel, dim3((unsigned)((ro_words + 255) / 256)), dim3(256), 0, stream, (uint32_t *)pid.p, ro_words, (uint32_t)base);
        MG_HIP(hipGetLastError());
        MG_HIP(hipStreamSynchronize(stream));
      }
      return PFP_OK;
    });
    agree(C, st, "global word ids", ctx);
    std::vector<uint64_t> gs, dcnt, ocnt;
    Dev gid_sent = C.alltoallv(pid.p, ros, gs);       // answers go back in the order the words came; owner order on arrival
    pid.release();
    Dev uni = C.allgatherv(ownb.p, own[1], dcnt);
    Dev uocc = C.allgatherv(owno.p, own[0] * 4, ocnt);
    ownb.release(); owno.release();
    uint64_t union_bytes = 0;
    for (uint64_t c : dcnt) union_bytes += c;
    // ---- suffix array of the global dictionary, one key range per rank (pfthreads.hpp:171-176 shards it by index range)
    Dev wslot;
    uint32_t parts = (uint32_t)size;
    uint64_t info[8] = {0};
    st.run([&] {
      wslot.alloc(n_union * 8 + 16);
      MG_HIP(hipMemsetAsync(wslot.p, 0, n_union * 8 + 16, stream));
      MG_HIP(hipStreamSynchronize(stream));
      return pfp_dist_global_sort_distinct(ctx, uni.p, union_bytes, uocc.p, n_union, gid_sent.p, parts > 1 ? (uint32_t)rank : 0u, parts, wslot.p, info);
    });
    std::vector<uint64_t> status;
    { uint64_t v[2] = {(st.rc == PFP_OK && info[3]) ? 1ull : 0ull, st.rc == PFP_OK ? info[6] : 0ull}; agree(C, st, "dictionary suffix sort", ctx, v, 2, &status); }
    bool all_complete = true;
    for (int r = 0; r < size; r++) if (!status[(size_t)r * 3 + 1]) all_complete = false;
    if (parts > 1 && !all_complete) {
      // some range needs ranks it does not hold: every rank sorts everything (equal output slices)
      parts = 1;
      st.run([&] {
        MG_HIP(hipMemsetAsync(wslot.p, 0, n_union * 8 + 16, stream));
        MG_HIP(hipStreamSynchronize(stream));
        return pfp_dist_global_sort_distinct(ctx, uni.p, union_bytes, uocc.p, n_union, gid_sent.p, 0u, 1u, wslot.p, info);
      });
      agree(C, st, "replicated dictionary suffix sort", ctx);
    }
    const uint64_t d_words = info[0];
    if (parts > 1) {
      std::vector<uint64_t> wc;
      Dev wall = C.allgatherv(wslot.p, d_words * 8, wc);
      st.run([&] { return pfp_dist_global_finish(ctx, wall.p, parts, 0, d_sym.p); });
    } else {
      st.run([&] { return pfp_dist_global_finish(ctx, wslot.p, 1, 0, d_sym.p); });
    }
    agree(C, st, "word ranking", ctx);
    uni.release(); uocc.release(); wslot.release(); gid_sent.release();
    // ---- the whole parse everywhere (bwtparse.c works on the whole parse)
    std::vector<uint64_t> pc;
    Dev sym_all = C.allgatherv(d_sym.p, P_local * 4, pc);
    Dev last_all = C.allgatherv(d_last.p, P_local, pc);
    uint64_t P_total = 0;
    for (uint64_t c : pc) P_total += c;
    Dev sai_all;
    if (want_sai) sai_all = C.allgatherv(d_sai.p, P_local * 8, pc);
    d_sym.release(); d_last.release(); d_sai.release();
    // ---- the suffix array of the parse in key ranges, one per rank (first sort, pivot round, comparison finisher: no ranks of
    //      other ranges); if any range would need a doubling round nobody sets anything and the merge sorts the whole parse itself
    uint64_t parse_shares = 1;
    if (size > 1 && P_total >= 2) {
      Dev share;
      uint64_t pinfo[4] = {0, 0, 0, 0};
      st.run([&] {
        share.alloc((P_total + 1) * 4 + 16);
        return pfp_dist_parse_sort(ctx, sym_all.p, P_total, (uint32_t)rank, (uint32_t)size, share.p, pinfo);
      });
      std::vector<uint64_t> pst;
      { uint64_t v[3] = {st.rc == PFP_OK && pinfo[2] ? pinfo[0] : 0ull, st.rc == PFP_OK && pinfo[2] ? 1ull : 0ull, pinfo[1]};
        agree(C, st, "parse suffix sort", ctx, v, 3, &pst); }
      bool all_ok = true;
      uint64_t entries = 0;
      for (int r = 0; r < size; r++) {
        if (!pst[(size_t)r * 4 + 2]) all_ok = false;
        // every share must start where the shares before it end (a gap in one and an overlap in another could cancel in the sum)
        if (pst[(size_t)r * 4 + 3] != entries) all_ok = false;
        entries += pst[(size_t)r * 4 + 1];
      }
      if (all_ok && entries == P_total + 1) {
        std::vector<uint64_t> sc;
        Dev sa_parse = C.allgatherv(share.p, pinfo[0] * 4, sc);
        st.run([&] { return pfp_dist_set_parse_sa(ctx, sa_parse.p, P_total + 1); });
        parse_shares = (uint64_t)size;
      }
    }
    const uint64_t n_out = n_total + 1;
    uint64_t lo = 0, hi = 0;
    if (parts > 1) {
      uint64_t sum = 0;
      for (int r = 0; r < size; r++) { if (r < rank) lo += status[(size_t)r * 3 + 2]; sum += status[(size_t)r * 3 + 2]; }
      if (sum != n_out) fail(PFP_EFORMAT, "the ranges of SA(D) emit %llu positions, text length + 1 is %llu", (unsigned long long)sum, (unsigned long long)n_out);
      hi = lo + status[(size_t)rank * 3 + 2];
    } else {
      lo = n_out * (uint64_t)rank / size; hi = n_out * (uint64_t)(rank + 1) / size;
    }
    const uint64_t cnt = hi - lo;
    Dev bwt, sa;
    uint8_t e0 = 0, e1 = 0;
    st.run([&] {
      bwt.alloc(cnt + 16);
      if (flags & PFP_FLAG_SA) sa.alloc((cnt + 1) * 8);
      int rc = pfp_dist_merge(ctx, sym_all.p, P_total, last_all.p, want_sai ? sai_all.p : nullptr, flags, n_total, lo, hi, bwt.p,
                              (flags & PFP_FLAG_SA) ? sa.p : nullptr);
      if (rc == PFP_OK && cnt) {
        rc = pfp_memcpy_d2h(ctx, &e0, bwt.p, 1);
        if (rc == PFP_OK) rc = pfp_memcpy_d2h(ctx, &e1, bwt.u8() + (cnt - 1), 1);
      }
      return rc;
    });
    std::vector<uint64_t> edges;      // per rank: status, positions emitted, first and last BWT byte of the slice
    { uint64_t v[3] = {cnt, e0, e1}; agree(C, st, "merge", ctx, v, 3, &edges); }
    sym_all.release(); last_all.release(); sai_all.release();
    // ---- the reference's output formats
    Dev sa5, ssa, esa;
    uint64_t sa5_bytes = 0, sa5_off = 0, k_ssa = 0, k_esa = 0;
    if (flags & PFP_FLAG_SA) {
      // .sa holds SA[1..n] (SA[0] = n is not written: pfbwt.cpp:158-162)
      const uint64_t first = lo == 0 ? 1 : 0, k = cnt > first ? cnt - first : 0;
      st.run([&] {
        sa5.alloc(5 * k + 16);
        return k ? pfp_pack5_dev(ctx, (const uint8_t *)sa.p + 8 * first, k, sa5.p) : PFP_OK;
      });
      agree(C, st, "5-byte packing", ctx);
      sa5_bytes = 5 * k; sa5_off = 5 * (lo + first - 1);
      sa.release();
    }
    std::vector<uint64_t> ks;
    if (flags & (PFP_FLAG_SSA | PFP_FLAG_ESA)) {
      int lb = -1, rbyte = -1;      // the BWT byte just before / after this rank's slice (slices may be empty)
      for (int r = 0; r < rank; r++) if (edges[(size_t)r * 4 + 1]) lb = (int)edges[(size_t)r * 4 + 3];
      for (int r = size - 1; r > rank; r--) if (edges[(size_t)r * 4 + 1]) rbyte = (int)edges[(size_t)r * 4 + 2];
      auto sample = [&](int run_end, Dev &buf, uint64_t &k) {
        k = 0;
        if (!cnt) return (int)PFP_OK;
        // the slice's own edge is a run start (end) unless the neighbour's adjacent byte is the same
        const int drop = run_end ? (rbyte >= 0 && rbyte == (int)e1) : (lb >= 0 && lb == (int)e0);
        int rc = pfp_dist_sample_runs(ctx, run_end, drop, nullptr, 0, &k);
        if (rc != PFP_OK) return rc;
        buf.alloc(10 * k + 16);
        if (k) rc = pfp_dist_sample_runs(ctx, run_end, drop, buf.p, k, &k);
        return rc;
      };
      if (flags & PFP_FLAG_SSA) st.run([&] { return sample(0, ssa, k_ssa); });
      if (flags & PFP_FLAG_ESA) st.run([&] { return sample(1, esa, k_esa); });
      uint64_t v[2] = {k_ssa, k_esa};
      agree(C, st, "run sampling", ctx, v, 2, &ks);
    }
    const double ms_chain = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    // ---- files: created at their final sizes by rank 0, then every rank pwrites its ranges (pfthreads.hpp:369-376)
    const std::string basep(J.base);
    uint64_t ssa_off = 0, esa_off = 0, ssa_tot = 0, esa_tot = 0;
    for (int r = 0; r < size && !ks.empty(); r++) {
      if (r < rank) { ssa_off += ks[(size_t)r * 3 + 1]; esa_off += ks[(size_t)r * 3 + 2]; }
      ssa_tot += ks[(size_t)r * 3 + 1]; esa_tot += ks[(size_t)r * 3 + 2];
    }
    if (rank == 0)
      st.run([&] {
        create_sized(basep + ".bwt", n_out);
        if (flags & PFP_FLAG_SA) create_sized(basep + ".sa", 5 * n_total);
        if (flags & PFP_FLAG_SSA) create_sized(basep + ".ssa", 10 * ssa_tot);
        if (flags & PFP_FLAG_ESA) create_sized(basep + ".esa", 10 * esa_tot);
        return PFP_OK;
      });
    agree(C, st, "creating the output files", ctx);
    st.run([&] {
      int rc = cnt ? pfp_pwrite_dev(ctx, (basep + ".bwt").c_str(), lo, bwt.p, cnt) : PFP_OK;
      if (rc == PFP_OK && sa5_bytes) rc = pfp_pwrite_dev(ctx, (basep + ".sa").c_str(), sa5_off, sa5.p, sa5_bytes);
      if (rc == PFP_OK && k_ssa) rc = pfp_pwrite_dev(ctx, (basep + ".ssa").c_str(), 10 * ssa_off, ssa.p, 10 * k_ssa);
      if (rc == PFP_OK && k_esa) rc = pfp_pwrite_dev(ctx, (basep + ".esa").c_str(), 10 * esa_off, esa.p, 10 * k_esa);
      return rc;
    });
    agree(C, st, "writing the output files", ctx);
    res.st.n = n_total; res.st.n_words = d_words; res.st.n_phrases = P_total; res.st.dict_size = info[1];
    res.st.index_bits = info[7]; res.st.sa_shares = parts; res.st.parse_shares = parse_shares; res.st.ranks = (uint64_t)size;
    res.st.ms_chain = ms_chain;
    { double dens = 1.0; if (plan[0]) memcpy(&dens, &plan[2], 8); res.st.parse_density = dens; }
    res.st.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
  } catch (const Failure &f) {
    res.code = f.code; res.msg = f.msg;
    S.abort_all();
  } catch (const std::exception &e) {
    res.code = PFP_EHIP; res.msg = e.what();
    S.abort_all();
  }
  if (ctx) pfp_ctx_destroy(ctx);
}

}  // namespace
}  // namespace pfp

extern "C" int pfp_bigbwt_files_multi(int n_dev, const int *devices, const uint8_t *text, uint64_t n, int w, uint64_t p, int flags,
                                      uint64_t halo, const char *out_base, pfp_multi_stats *stats, char *errbuf, uint64_t errbuf_len) {
  using namespace pfp;
  auto say = [&](const std::string &m) { if (errbuf && errbuf_len) { snprintf(errbuf, errbuf_len, "%s", m.c_str()); } };
  if (n_dev < 1 || n_dev > 64 || !devices || (!text && n) || !out_base) { say("bad argument"); return PFP_EINVAL; }
  if ((flags & PFP_FLAG_SA) && (flags & (PFP_FLAG_SSA | PFP_FLAG_ESA))) { say("either the full SA or a sample of it, not both (bigbwt:59-61)"); return PFP_EINVAL; }
  const char *lbe = getenv("PFP_MULTI_LOOPBACK");
  const bool loopback = lbe && atoi(lbe) != 0;
  int have = 0;
  if (hipGetDeviceCount(&have) != hipSuccess || have < 1) { say("no HIP device"); return PFP_ENODEV; }
  std::vector<int> dev(devices, devices + n_dev);
  for (int r = 0; r < n_dev; r++) {
    if (loopback) dev[r] = dev[r] % have;
    if (dev[r] < 0 || dev[r] >= have) { say("device " + std::to_string(dev[r]) + " does not exist (" + std::to_string(have) + " visible)"); return PFP_ENODEV; }
    if (!loopback) for (int q = 0; q < r; q++) if (dev[q] == dev[r]) { say("the same device twice (RCCL wants one rank per GPU; PFP_MULTI_LOOPBACK=1 runs the ranks on one device without it)"); return PFP_EINVAL; }
  }
  static Rccl rccl;
  static std::mutex rccl_mu;
  Shared S(n_dev, loopback);
  if (!loopback) {
    std::lock_guard<std::mutex> lk(rccl_mu);
    std::string err;
    if (!rccl.load(err)) { say(err); return PFP_ENODEV; }
    S.rccl = &rccl;
    S.comms.assign(n_dev, nullptr);
    const int rc = rccl.CommInitAll(S.comms.data(), n_dev, dev.data());
    if (rc != 0) { say(std::string("ncclCommInitAll: ") + rccl.GetErrorString(rc)); return PFP_EHIP; }
  }
  const Job J{text, n, w, p, flags, halo ? halo : (1ull << 20), out_base};
  std::vector<RankResult> res(n_dev);
  std::vector<std::thread> th;
  for (int r = 0; r < n_dev; r++) th.emplace_back([&, r] { run_rank(S, r, dev[r], J, res[r]); });
  for (auto &t : th) t.join();
  if (!loopback && !S.comms_gone) for (auto c : S.comms) if (c) rccl.CommDestroy(c);
  int code = PFP_OK;
  for (int r = 0; r < n_dev; r++)
    if (res[r].code != PFP_OK) {
      // prefer the message of the rank whose own step failed ("rank r: ...") over the ones that only heard of it
      if (code == PFP_OK || res[r].msg.rfind("rank ", 0) == 0) { code = res[r].code; say(res[r].msg); }
    }
  if (code == PFP_OK && stats) *stats = res[0].st;
  return code;
}

// One-GPU check of the RCCL transport (tests): the dlopen'ed table, the enum values and the group semantics, with a communicator
// over ONE device - every exchange shape of the chain as a self send / recv, and the one-collective all-gather.
// inject_failure != 0: after the exchanges, one more all-to-all that fails between ncclGroupStart and ncclGroupEnd with a send
// already queued - the error path of a rank (group closed on the way out, abort_all: ncclCommAbort on every communicator) must come
// back instead of hanging; PFP_OK then means "failed as intended and cleaned up".
extern "C" int pfp_multi_rccl_selftest2(int device, int inject_failure, char *errbuf, uint64_t errbuf_len);
extern "C" int pfp_multi_rccl_selftest(int device, char *errbuf, uint64_t errbuf_len) { return pfp_multi_rccl_selftest2(device, 0, errbuf, errbuf_len); }
extern "C" int pfp_multi_rccl_selftest2(int device, int inject_failure, char *errbuf, uint64_t errbuf_len) {
  using namespace pfp;
  auto say = [&](const std::string &m) { if (errbuf && errbuf_len) snprintf(errbuf, errbuf_len, "%s", m.c_str()); };
  int have = 0;
  if (hipGetDeviceCount(&have) != hipSuccess || device < 0 || device >= have) { say("no such HIP device"); return PFP_ENODEV; }
  static Rccl rccl;
  std::string err;
  if (!rccl.load(err)) { say(err); return PFP_ENODEV; }
  Shared S(1, false);
  S.rccl = &rccl;
  S.comms.assign(1, nullptr);
  if (hipSetDevice(device) != hipSuccess) { say("hipSetDevice"); return PFP_EHIP; }
  int rc = rccl.CommInitAll(S.comms.data(), 1, &device);
  if (rc != 0) { say(std::string("ncclCommInitAll: ") + rccl.GetErrorString(rc)); return PFP_EHIP; }
  int code = PFP_OK;
  hipStream_t stream = nullptr;
  try {
    MG_HIP(hipStreamCreate(&stream));
    Coll C(S, 0, stream);
    const uint64_t n = (1u << 20) + 123;
    std::vector<uint8_t> h(n), back(n);
    for (uint64_t i = 0; i < n; i++) h[i] = (uint8_t)(i * 2654435761u >> 13);
    Dev src(n + 16);
    MG_HIP(hipMemcpy(src.p, h.data(), n, hipMemcpyHostToDevice));
    auto same = [&](const Dev &d, uint64_t bytes, const char *what) {
      MG_HIP(hipMemcpy(back.data(), d.p, bytes, hipMemcpyDeviceToHost));
      if (memcmp(back.data(), h.data(), bytes) != 0) fail(PFP_EHIP, "RCCL self test: %s returned other bytes", what);
    };
    std::vector<uint64_t> counts;
    for (int grouped = 0; grouped < 2; grouped++) {
      C.force_grouped = grouped != 0;
      { Dev out = C.allgatherv(src.p, n, counts); if (counts.size() != 1 || counts[0] != n) fail(PFP_EHIP, "RCCL self test: counts"); same(out, n, grouped ? "grouped all-gather" : "ncclAllGather"); }
      { Dev out = C.allgatherv(src.p, 0, counts); (void)out; }      // an empty contribution
    }
    C.force_grouped = false;
    { std::vector<uint64_t> send(1, n), recv; Dev out = C.alltoallv(src.p, send, recv); if (recv.size() != 1 || recv[0] != n) fail(PFP_EHIP, "RCCL self test: all-to-all counts"); same(out, n, "all-to-all"); }
    { std::vector<uint64_t> send(1, 0), recv; Dev out = C.alltoallv(src.p, send, recv); (void)out; }
    if (inject_failure) {
      bool failed = false;
      C.fail_in_group = true;
      try { std::vector<uint64_t> send(1, n), recv; Dev out = C.alltoallv(src.p, send, recv); (void)out; }
      catch (const Failure &f) { failed = std::string(f.msg).find("injected") != std::string::npos; }
      C.fail_in_group = false;
      if (!failed) fail(PFP_EHIP, "RCCL self test: the injected failure did not surface");
      S.abort_all();          // what run_rank's catch block does: the queued send must not keep anything waiting
      if (!S.comms_gone) fail(PFP_EHIP, "RCCL self test: the communicators were not aborted");
    }
  } catch (const Failure &f) {
    code = f.code; say(f.msg);
    S.abort_all();
  }
  if (stream) (void)hipStreamDestroy(stream);
  if (!S.comms_gone) for (auto cm : S.comms) if (cm) rccl.CommDestroy(cm);
  return code;
}
