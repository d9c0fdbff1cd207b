// validate.hip -- PFP_DEBUG=1 host-side validation of every intermediate array of the chain.
// Debug aid only (copies whole arrays to the host); it never computes a result, it only stops
// the chain with PFP_EFORMAT before a later kernel would use an inconsistent index array.
#include "kernels.hpp"
#include <algorithm>

namespace pfp {

template <class T>
static std::vector<T> fetch(pfp_ctx *c, const T *d, size_t n) {
  std::vector<T> h(n);
  if (n) PFP_HIP(hipMemcpyAsync(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost, c->stream));
  sync(c);
  return h;
}
#define VFAIL(msg) throw Error(PFP_EFORMAT, std::string("PFP_DEBUG validation failed: ") + msg)

void validate_scan(pfp_ctx *c, const DBuf<uint64_t> &ends, uint64_t n_ends, uint64_t n, int w) {
  auto e = fetch(c, ends.p, n_ends);
  for (uint64_t k = 0; k < n_ends; k++) {
    if (e[k] >= n || e[k] + 1 < (uint64_t)w) VFAIL("scan: end out of range at " + std::to_string(k));
    if (k && e[k] <= e[k - 1]) VFAIL("scan: ends not increasing at " + std::to_string(k));
  }
}

void validate_dictionary(pfp_ctx *c, const Dictionary &D, int w) {
  auto pid = fetch(c, D.pid.p, D.P);
  auto wlen = fetch(c, D.wlen.p, D.d);
  auto wocc = fetch(c, D.wocc.p, D.d);
  auto woff = fetch(c, D.woff.p, D.d + 1);
  auto b = fetch(c, D.bytes.p, D.dsize + 64);
  uint64_t tot = 0;
  std::vector<uint32_t> cnt(D.d, 0);
  for (uint64_t k = 0; k < D.P; k++) { if (pid[k] >= D.d) VFAIL("dict: pid out of range"); cnt[pid[k]]++; }
  for (uint64_t j = 0; j < D.d; j++) {
    if (wlen[j] <= (uint32_t)w) VFAIL("dict: word not longer than w");
    if (woff[j + 1] != woff[j] + wlen[j] + 1) VFAIL("dict: woff inconsistent at word " + std::to_string(j));
    if (cnt[j] != wocc[j]) VFAIL("dict: occ mismatch at word " + std::to_string(j));
    if (b[woff[j] + wlen[j]] != kEndOfWord) VFAIL("dict: missing terminator of word " + std::to_string(j));
    for (uint64_t i = 0; i < wlen[j]; i++) if (b[woff[j] + i] < kDollar) VFAIL("dict: special byte inside word " + std::to_string(j));
    tot += wocc[j];
  }
  if (tot != D.P) VFAIL("dict: sum(occ) != P");
  if (woff[D.d] != D.dsize - 1 || b[D.dsize - 1] != kEndOfDict) VFAIL("dict: bad end");
  for (int i = 0; i < 64; i++) if (b[D.dsize + i]) VFAIL("dict: pad not zero");
}

// every position asks the word lookup (wordview.hpp) for its word and its distance to the terminator
__global__ void word_lookup_kernel(WordView wv, uint32_t *__restrict__ pw, uint64_t *__restrict__ sl) {
  const uint64_t i = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (i >= wv.NP) return;
  pw[i] = word_of(wv, i);
  sl[i] = slen_of(wv, i);
}
void validate_index(pfp_ctx *c, const Dictionary &D, const DictIndex &ix) {
  auto b = fetch(c, D.bytes.p, D.dsize);
  auto we = fetch(c, ix.wend.p, D.d + 1);
  auto bw = fetch(c, ix.blk_word.p, (D.dsize + 63) / 64 + 1);
  DBuf<uint32_t> d_pw(c, D.dsize);
  DBuf<uint64_t> d_sl(c, D.dsize);
  hipLaunchKernelGGL(word_lookup_kernel, gdim(cdiv(D.dsize, 256)), gdim(256), 0, c->stream, word_view(D, ix), d_pw.p, d_sl.p);
  auto pw = fetch(c, d_pw.p, D.dsize);
  auto sl = fetch(c, d_sl.p, D.dsize);
  uint32_t wd = 0;
  for (uint64_t i = 0; i < D.dsize; i++) {
    if ((i & 63) == 0 && bw[i >> 6] != wd) VFAIL("index: blk_word wrong at line " + std::to_string(i >> 6));
    if (pw[i] != wd) VFAIL("index: word lookup wrong at " + std::to_string(i));
    if (pw[i] > D.d || i + sl[i] != we[pw[i]]) VFAIL("index: suffix length wrong at " + std::to_string(i));
    if (b[i] == kEndOfWord) { if (we[wd] != i) VFAIL("index: wend wrong"); wd++; }
  }
  if (wd != D.d || we[D.d] != D.dsize - 1) VFAIL("index: word count");
}

// compare two dictionary suffixes as 0x01-terminated strings
static int cmp_suffix(const std::vector<uint8_t> &b, uint64_t x, uint64_t y) {
  for (;;) {
    uint8_t cx = b[x], cy = b[y];
    if (cx != cy) return cx < cy ? -1 : 1;
    if (cx <= kEndOfWord) return 0;
    x++; y++;
  }
}

template <class I>
void validate_suffix_order(pfp_ctx *c, const uint8_t *bytes, SuffixOrderT<I> &so, bool dict_mode, const char *what) {
  const uint64_t N = so.N;
  auto sa = fetch(c, so.sa.p, N);
  // ranks through the sorter's own lookup (rank[] is sparse in dictionary mode), checked against grp[]
  std::vector<I> rk(N);
  {
    std::vector<uint64_t> pos(N);
    for (uint64_t i = 0; i < N; i++) pos[i] = i;
    DBuf<uint64_t> dpos(c, N);
    DBuf<I> dr(c, N);
    h2d(c, dpos.p, pos.data(), N);
    gather_ranks<I>(c, so, dpos.p, N, dr.p);
    rk = fetch(c, dr.p, N);
  }
  auto gr = fetch(c, so.grp.p, N);
  std::vector<uint8_t> seen(N, 0);
  for (uint64_t t = 0; t < N; t++) {
    if (sa[t] >= N || seen[sa[t]]) VFAIL(std::string(what) + ": sa is not a permutation at slot " + std::to_string(t));
    seen[sa[t]] = 1;
    if (rk[sa[t]] > t) VFAIL(std::string(what) + ": rank above slot");
    if (rk[sa[t]] != gr[t]) VFAIL(std::string(what) + ": rank lookup differs from the slot's group head at slot " + std::to_string(t));
  }
  if (dict_mode && N < (64u << 20)) {
    auto b = fetch(c, bytes, N);
    for (uint64_t t = 1; t < N; t++) {
      int cmp = cmp_suffix(b, sa[t - 1], sa[t]);
      bool same = rk[sa[t - 1]] == rk[sa[t]];
      if (cmp > 0) VFAIL(std::string(what) + ": suffixes out of order at slot " + std::to_string(t));
      if ((cmp == 0) != same) VFAIL(std::string(what) + ": rank equality != string equality at slot " + std::to_string(t));
      if (cmp == 0 && sa[t - 1] > sa[t]) VFAIL(std::string(what) + ": equal suffixes not in position order");
    }
  }
}

template void validate_suffix_order<uint32_t>(pfp_ctx *, const uint8_t *, SuffixOrderT<uint32_t> &, bool, const char *);
template void validate_suffix_order<uint64_t>(pfp_ctx *, const uint8_t *, SuffixOrderT<uint64_t> &, bool, const char *);

void validate_int_sa(pfp_ctx *c, const uint32_t *sym, const SuffixOrder &so) {
  const uint64_t N = so.N;
  auto sa = fetch(c, so.sa.p, N);
  auto s = fetch(c, sym, N);
  std::vector<uint8_t> seen(N, 0);
  for (uint64_t t = 0; t < N; t++) {
    if (sa[t] >= N || seen[sa[t]]) VFAIL("parse SA is not a permutation");
    seen[sa[t]] = 1;
  }
  for (uint64_t t = 1; t < N; t++) {
    uint64_t x = sa[t - 1], y = sa[t];
    while (x < N && y < N && s[x] == s[y]) { x++; y++; }
    if (x < N && y < N && s[x] > s[y]) VFAIL("parse SA out of order at " + std::to_string(t));
  }
}

void validate_lexrank(pfp_ctx *c, const Dictionary &D, const DictIndex &ix) {
  auto lr = fetch(c, ix.lexrank.p, D.d);
  std::vector<uint8_t> seen(D.d, 0);
  for (uint64_t j = 0; j < D.d; j++) {
    if (lr[j] >= D.d || seen[lr[j]]) VFAIL("lexrank is not a permutation");
    seen[lr[j]] = 1;
  }
}

void validate_parse_bwt(pfp_ctx *c, const ParseBWT &pb) {
  auto il = fetch(c, pb.ilist.p, pb.P + 1);
  std::vector<uint8_t> seen(pb.P + 1, 0);
  for (uint64_t j = 0; j <= pb.P; j++) {
    if (il[j] > pb.P || seen[il[j]]) VFAIL("ilist is not a permutation");
    seen[il[j]] = 1;
  }
  if (il[0] != 1) VFAIL("ilist[0] != 1");
}

}  // namespace pfp
