// phrase.hip -- stage 1b: phrase identity, deduplication, dictionary build.
//
// Replaces save_update_word + the std::map<hash,word_stats> of the reference parser
// (newscan.cpp:229-304; sharded twin pscan.cpp:137-205) and the dictionary/occ writer
// (newscan.cpp:394-441).  Design, MI355X-first:
//   * phrase k is T'[start_k .. end_k] with end_k from the scan's ends[]; nothing is copied.
//   * identity hash: order-sensitive *commutative* sum  sum_j mix(chunk_j, j)  over 8-byte
//     chunks, so an 8-lane group hashes a typical phrase with one 16-byte unaligned load per
//     lane and a giant phrase (an 18 Mb run of N) is split over the whole chip and combined
//     with 64-bit atomic adds - no sequential Horner chain as in kr_hash (newscan.cpp:229-239).
//   * dedup = radix sort of (hash, phrase#) + adjacent compare.  Every occurrence is verified
//     byte-for-byte against its predecessor (the reference compares strings on every hit,
//     newscan.cpp:282); a mismatch reseeds the hash and retries, so the result is exact.
//   * words are numbered by descending occurrence count (ties: first occurrence); the lexicographic ranks the reference
//     assigns with std::sort (newscan.cpp:622-636) fall out of the dictionary suffix sort.
#include "kernels.hpp"
#include "prims.hpp"
#include "devutil.hpp"

namespace pfp {

constexpr uint32_t kLongPhrase = 8192;  // phrases longer than this are split across blocks

// Where phrase k lives.  Implicit mode: phrase k0+k of the scan (ends[], overlap w) - k0 > 0 lets a
// text shard own only the phrases that END inside it (multi-GPU).  Explicit mode (xstart != null):
// an arbitrary word list over a byte buffer (the union of the ranks' local dictionaries).
struct PhraseGeom {
  const uint64_t *ends; uint64_t ne; uint64_t n; int w;
  uint64_t k0;
  const uint64_t *xstart; const uint32_t *xlen;
};
__device__ __forceinline__ uint64_t ph_end(const PhraseGeom &g, uint64_t k) {   // index of last byte
  if (g.xstart) return g.xstart[k] + g.xlen[k] - 1;
  k += g.k0;
  return k < g.ne ? g.ends[k] + 1 : g.n + (uint64_t)g.w;
}
__device__ __forceinline__ uint64_t ph_start(const PhraseGeom &g, uint64_t k) { // index of first byte
  if (g.xstart) return g.xstart[k];
  k += g.k0;
  return k == 0 ? 0 : g.ends[k - 1] + 2 - (uint64_t)g.w;
}

// One term of the commutative phrase hash.  The chunk is XORed with a position key and pushed
// through a full-avalanche finaliser (two multiplies, three xor-shifts): with a single multiply the
// top byte of a chunk only reaches the top byte of the term, and two SNPs that both fall in the
// last byte of their 8-byte chunks then cancel with probability ~2^-8 (seen on 16 mutated copies).
__device__ __forceinline__ uint64_t chunk_term(uint64_t x, uint64_t idx, uint64_t seed) {
  return fmix64(x ^ (seed + idx * 0x9E3779B97F4A7C15ULL));
}
// contribution of the 16-byte piece at byte offset off of a phrase of length len
__device__ __forceinline__ uint64_t piece_terms(const uint8_t *src, uint64_t off, uint64_t len, uint64_t seed) {
  uint4 v = ld16u(src + off);
  int keep = (len - off) >= 16 ? 16 : (int)(len - off);
  v = keep_bytes16(v, keep);
  uint64_t x0 = (uint64_t)v.x | ((uint64_t)v.y << 32), x1 = (uint64_t)v.z | ((uint64_t)v.w << 32);
  uint64_t s = chunk_term(x0, off >> 3, seed);
  if (keep > 8) s += chunk_term(x1, (off >> 3) + 1, seed);
  return s;
}
__device__ __forceinline__ uint64_t finish_hash(uint64_t sum, uint64_t len) {
  return fmix64(sum ^ (len * 0xA24BAED4963EE407ULL));
}

// 2^LG lanes per phrase, 16 bytes a lane and step: 8 lanes for phrases of ~100 bytes (the nominal -p 100), 4 where the chain chose
// phrases of ~48 bytes (with 8 lanes five of them had nothing to hash).  The sum of the terms does not depend on who adds them.
template <int LG>
__global__ __launch_bounds__(256) void phrase_hash_kernel(const uint8_t *__restrict__ tp, PhraseGeom g, uint64_t P,
                                                          uint64_t seed, uint64_t *__restrict__ hash,
                                                          uint32_t *__restrict__ long_list,
                                                          uint32_t *__restrict__ long_count) {
  uint64_t t = (uint64_t)BID * 256 + threadIdx.x;
  uint64_t k = t >> LG;
  int l8 = (int)(t & ((1u << LG) - 1u));
  if (k >= P) return;
  uint64_t s = ph_start(g, k), len = ph_end(g, k) - s + 1;
  if (len > kLongPhrase) {
    if (l8 == 0) { uint32_t i = atomicAdd(long_count, 1u); long_list[i] = (uint32_t)k; }
    return;
  }
  uint64_t sum = 0;
  for (uint64_t off = (uint64_t)l8 * 16; off < len; off += (16u << LG)) sum += piece_terms(tp + s, off, len, seed);
  sum = group_sum<LG>(sum);
  if (l8 == 0) hash[k] = finish_hash(sum, len);
}
// lanes per phrase from the average phrase length
static inline int lanes_log2_for(uint64_t bytes, uint64_t phrases) { return phrases && bytes / phrases <= 72 ? 2 : 3; }

__global__ __launch_bounds__(256) void phrase_hash_long_kernel(const uint8_t *__restrict__ tp, PhraseGeom g,
                                                               uint64_t seed, const uint32_t *__restrict__ long_list,
                                                               uint32_t nlong, unsigned long long *__restrict__ hsum) {
  __shared__ uint64_t wsum[4];
  for (uint32_t li = 0; li < nlong; li++) {
    uint64_t k = long_list[li];
    uint64_t s = ph_start(g, k), len = ph_end(g, k) - s + 1;
    uint64_t npieces = (len + 15) >> 4;
    uint64_t sum = 0;
    for (uint64_t c = (uint64_t)BID * 256 + threadIdx.x; c < npieces; c += (uint64_t)GDIM * 256)
      sum += piece_terms(tp + s, c * 16, len, seed);
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint64_t b = wsum[0] + wsum[1] + wsum[2] + wsum[3];
      if (b) atomicAdd(&hsum[li], (unsigned long long)b);
    }
    __syncthreads();
  }
}

__global__ void phrase_hash_long_finish_kernel(PhraseGeom g, const uint32_t *__restrict__ long_list, uint32_t nlong,
                                               const unsigned long long *__restrict__ hsum,
                                               uint64_t *__restrict__ hash) {
  uint32_t li = BID * blockDim.x + threadIdx.x;
  if (li >= nlong) return;
  uint64_t k = long_list[li];
  uint64_t len = ph_end(g, k) - ph_start(g, k) + 1;
  hash[k] = finish_hash(hsum[li], len);
}

__global__ void iota_u32_kernel(uint32_t *p, uint64_t n) {
  uint64_t i = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (uint32_t)i;
}

// sorted by hash: head flags + byte verification against the predecessor
template <int LG>
__global__ __launch_bounds__(256) void dedup_verify_kernel(const uint8_t *__restrict__ tp, PhraseGeom g, uint64_t P,
                                                           const uint64_t *__restrict__ ks,
                                                           const uint32_t *__restrict__ vs,
                                                           uint32_t *__restrict__ head, uint32_t *__restrict__ collision) {
  uint64_t t = (uint64_t)BID * 256 + threadIdx.x;
  uint64_t i = t >> LG;
  int l8 = (int)(t & ((1u << LG) - 1u));
  if (i >= P) return;
  bool hd = (i == 0) || ks[i] != ks[i - 1];
  if (l8 == 0) head[i] = hd ? 1u : 0u;
  if (hd) return;
  uint64_t a = vs[i], b = vs[i - 1];
  uint64_t sa = ph_start(g, a), la = ph_end(g, a) - sa + 1;
  uint64_t sb = ph_start(g, b), lb = ph_end(g, b) - sb + 1;
  uint32_t diff = (la != lb) ? 1u : 0u;
  if (!diff && sa != sb) {
    for (uint64_t off = (uint64_t)l8 * 16; off < la; off += (16u << LG)) {
      int keep = (la - off) >= 16 ? 16 : (int)(la - off);
      uint4 va = keep_bytes16(ld16u(tp + sa + off), keep), vb = keep_bytes16(ld16u(tp + sb + off), keep);
      diff |= (va.x ^ vb.x) | (va.y ^ vb.y) | (va.z ^ vb.z) | (va.w ^ vb.w);
    }
  }
  diff = group_or<LG>(diff);
  if (diff && l8 == 0) atomicOr(collision, 1u);
}

__global__ void collect_words_kernel(uint64_t P, const uint32_t *__restrict__ head, const uint32_t *__restrict__ hscan,
                                     const uint32_t *__restrict__ vs, uint32_t *__restrict__ hrep,
                                     uint32_t *__restrict__ hpos, uint32_t d) {
  uint64_t i = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (i == 0) hpos[d] = (uint32_t)P;
  if (i >= P || !head[i]) return;
  uint32_t hw = hscan[i] - 1;
  hrep[hw] = vs[i];   // stable sort => smallest phrase index of the run
  hpos[hw] = (uint32_t)i;
}

// weighted occurrence counts (explicit mode): occ of a distinct word = sum of the weights of its copies
__global__ void weighted_occ_kernel(uint64_t P, const uint32_t *__restrict__ hscan, const uint32_t *__restrict__ vs,
                                    const uint32_t *__restrict__ weight, uint32_t *__restrict__ hocc) {
  uint64_t i = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (i < P) atomicAdd(&hocc[hscan[i] - 1], weight[vs[i]]);
}

// order of the words in the device dictionary: most frequent first, ties by first occurrence (key = ~occ : first phrase)
__global__ void word_order_keys_kernel(uint32_t d, const uint32_t *__restrict__ hrep, const uint32_t *__restrict__ hpos,
                                       const uint32_t *__restrict__ hocc, uint64_t *__restrict__ key) {
  uint32_t hw = BID * blockDim.x + threadIdx.x;
  if (hw >= d) return;
  const uint32_t occ = hocc ? hocc[hw] : hpos[hw + 1] - hpos[hw];
  key[hw] = ((uint64_t)(~occ) << 32) | hrep[hw];
}

__global__ void finish_words_kernel(PhraseGeom g, uint32_t d, const uint64_t *__restrict__ rep_sorted,
                                    const uint32_t *__restrict__ hw_sorted, const uint32_t *__restrict__ hpos,
                                    const uint32_t *__restrict__ hocc,
                                    uint32_t *__restrict__ fo_of_hw, uint32_t *__restrict__ wlen,
                                    uint32_t *__restrict__ wlen1, uint64_t *__restrict__ wsrc,
                                    uint32_t *__restrict__ wocc, uint32_t *__restrict__ toolong) {
  uint32_t j = BID * blockDim.x + threadIdx.x;
  if (j == 0) wlen1[d] = 0;
  if (j >= d) return;
  uint32_t hw = hw_sorted[j];
  fo_of_hw[hw] = j;
  uint64_t k = rep_sorted[j] & 0xFFFFFFFFull;
  uint64_t s = ph_start(g, k), len = ph_end(g, k) - s + 1;
  if (len >= 0xFFFFFFF0ull) { atomicOr(toolong, 1u); len = 0xFFFFFFF0ull; }
  wlen[j] = (uint32_t)len;
  wlen1[j] = (uint32_t)len + 1;
  wsrc[j] = s;
  wocc[j] = hocc ? hocc[hw] : hpos[hw + 1] - hpos[hw];
}

__global__ void assign_pid_kernel(uint64_t P, const uint32_t *__restrict__ hscan, const uint32_t *__restrict__ vs,
                                  const uint32_t *__restrict__ fo_of_hw, uint32_t *__restrict__ pid) {
  uint64_t i = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (i < P) pid[vs[i]] = fo_of_hw[hscan[i] - 1];
}

// copy one 16-byte piece (or its partial tail) of a word
__device__ __forceinline__ void copy_piece(const uint8_t *src, uint8_t *dst, uint64_t off, uint64_t len) {
  if (off + 16 <= len) {
    st16u(dst + off, ld16u(src + off));
  } else {
    for (uint64_t b = off; b < len; b++) dst[b] = src[b];
  }
}

__global__ __launch_bounds__(256) void dict_copy_kernel(const uint8_t *__restrict__ tp, uint32_t d,
                                                        const uint64_t *__restrict__ wsrc,
                                                        const uint32_t *__restrict__ wlen,
                                                        const uint64_t *__restrict__ woff, uint8_t *__restrict__ D,
                                                        uint32_t *__restrict__ long_list,
                                                        uint32_t *__restrict__ long_count) {
  uint64_t t = (uint64_t)BID * 256 + threadIdx.x;
  uint64_t j = t >> 3;
  int l8 = (int)(t & 7);
  if (j >= d) return;
  uint64_t len = wlen[j];
  uint8_t *dst = D + woff[j];
  if (l8 == 0) dst[len] = kEndOfWord;
  if (len > kLongPhrase) {
    if (l8 == 0) { uint32_t i = atomicAdd(long_count, 1u); long_list[i] = (uint32_t)j; }
    return;
  }
  const uint8_t *src = tp + wsrc[j];
  for (uint64_t off = (uint64_t)l8 * 16; off < len; off += 128) copy_piece(src, dst, off, len);
}

__global__ __launch_bounds__(256) void dict_copy_long_kernel(const uint8_t *__restrict__ tp,
                                                             const uint64_t *__restrict__ wsrc,
                                                             const uint32_t *__restrict__ wlen,
                                                             const uint64_t *__restrict__ woff, uint8_t *__restrict__ D,
                                                             const uint32_t *__restrict__ long_list, uint32_t nlong) {
  for (uint32_t li = 0; li < nlong; li++) {
    uint32_t j = long_list[li];
    uint64_t len = wlen[j];
    const uint8_t *src = tp + wsrc[j];
    uint8_t *dst = D + woff[j];
    uint64_t npieces = (len + 15) >> 4;
    for (uint64_t c = (uint64_t)BID * 256 + threadIdx.x; c < npieces; c += (uint64_t)GDIM * 256)
      copy_piece(src, dst, c * 16, len);
  }
}

// .last (newscan.cpp:296) and .sai (newscan.cpp:298-301)
__global__ void last_sai_kernel(const uint8_t *__restrict__ tp, PhraseGeom g, uint64_t P, uint64_t sai_base,
                                uint8_t *__restrict__ last, uint64_t *__restrict__ sai) {
  uint64_t k = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (k >= P) return;
  uint64_t e = ph_end(g, k);
  last[k] = tp[e - (uint64_t)g.w];
  if (sai) sai[k] = e + sai_base;
}

// identity hashes of a word list over a byte buffer (the multi-GPU chain routes every word to the rank that owns
// its hash class)
void hash_word_list(pfp_ctx *c, const uint8_t *bytes, const uint64_t *wstart, const uint32_t *wlen, uint64_t U, uint64_t seed,
                    uint64_t *d_hash) {
  if (!U) return;
  const int TB = 256;
  PhraseGeom g{nullptr, 0, 0, 0, 0, wstart, wlen};
  DBuf<uint32_t> long_list(c, U), counters(c, 1);
  counters.zero();
  hipLaunchKernelGGL(phrase_hash_kernel<3>, gdim(cdiv(U * 8, TB)), gdim(TB), 0, c->stream, bytes, g, U, seed, d_hash, long_list.p, counters.p);
  const uint32_t nlong = read_scalar(c, counters.p);
  if (nlong) {
    DBuf<unsigned long long> hsum(c, nlong);
    hsum.zero();
    hipLaunchKernelGGL(phrase_hash_long_kernel, gdim(c->n_cu * 4), gdim(TB), 0, c->stream, bytes, g, seed, long_list.p, nlong, hsum.p);
    hipLaunchKernelGGL(phrase_hash_long_finish_kernel, gdim(cdiv(nlong, 64)), gdim(64), 0, c->stream, g, long_list.p, nlong, hsum.p, d_hash);
  }
  PFP_HIP(hipGetLastError());
  sync(c);
}

// Core of the dictionary build over any phrase geometry.  weight == nullptr: every phrase counts
// once; otherwise occ of a distinct word is the sum of its copies' weights.  hbytes = total bytes
// the phrases cover (for the kernel-trace byte accounting only).
__global__ void mask_u64_kernel(uint64_t *v, uint64_t n, uint64_t mask) {
  const uint64_t i = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (i < n) v[i] &= mask;
}
static void build_dictionary_core(pfp_ctx *c, const uint8_t *tp, PhraseGeom g, uint64_t P, const uint32_t *weight,
                                  uint64_t hbytes, bool want_last, bool want_sai, uint64_t sai_base, Dictionary &D) {
  PFP_REQUIRE(P >= 1 && P <= 0xFFFFFFFEull, PFP_ELIMIT,
              "parse has more than 2^32-2 phrases (bwtparse.c:93); use a larger -p");
  const uint64_t n = hbytes;
  D.P = P;
  const int TB = 256;
  DBuf<uint64_t> hash(c, P), ks(c, P);
  DBuf<uint32_t> iota(c, P), vs(c, P), head(c, P), hscan(c, P);
  DBuf<uint32_t> long_list(c, P), counters(c, 4);  // [0] long count, [1] collision, [2] too long, [3] long count (copy)
  hipLaunchKernelGGL(iota_u32_kernel, gdim(cdiv(P, TB)), gdim(TB), 0, c->stream, iota.p, P);
  uint64_t seed = 0x243F6A8885A308D3ULL;
  uint32_t d = 0;
  const int lg = lanes_log2_for(n, P);
  for (int attempt = 0;; attempt++) {
    counters.zero();
    { KScope kscope(c, "pfp::phrase_hash_kernel", n + 16 * P);
    if (lg == 2) hipLaunchKernelGGL(phrase_hash_kernel<2>, gdim(cdiv(P * 4, TB)), gdim(TB), 0, c->stream, tp, g, P, seed, hash.p, long_list.p, counters.p);
    else hipLaunchKernelGGL(phrase_hash_kernel<3>, gdim(cdiv(P * 8, TB)), gdim(TB), 0, c->stream, tp, g, P, seed, hash.p, long_list.p, counters.p); }
    uint32_t nlong = read_scalar(c, counters.p);
    if (nlong) {
      DBuf<unsigned long long> hsum(c, nlong);
      hsum.zero();
      hipLaunchKernelGGL(phrase_hash_long_kernel, gdim(c->n_cu * 4), gdim(TB), 0, c->stream, tp, g, seed, long_list.p,
                         nlong, hsum.p);
      hipLaunchKernelGGL(phrase_hash_long_finish_kernel, gdim(cdiv(nlong, 64)), gdim(64), 0, c->stream, g, long_list.p,
                         nlong, hsum.p, hash.p);
    }
    // test hook (PFP_TEST_HASH_BITS=n: the first attempt keeps only n bits of every hash; -n: every attempt): different phrases then
    // share a hash, the byte verification below finds it and the pass is repeated with another seed / gives up after four
    const char *tb_env = getenv("PFP_TEST_HASH_BITS");      // (per call: tests switch it)
    const int test_bits = tb_env ? atoi(tb_env) : 0;
    if (test_bits && (attempt == 0 || test_bits < 0)) {
      const int b = test_bits < 0 ? -test_bits : test_bits;
      hipLaunchKernelGGL(mask_u64_kernel, gdim(cdiv(P, TB)), gdim(TB), 0, c->stream, hash.p, P, b >= 64 ? ~0ull : ((1ull << b) - 1ull));
    }
    { SortTag tag("phrase hashes"); sort_pairs_u64_u32(c, hash.p, ks.p, iota.p, vs.p, P, 0, 64); }
    { KScope kscope(c, "pfp::dedup_verify_kernel", 2 * n + 16 * P);
    if (lg == 2) hipLaunchKernelGGL(dedup_verify_kernel<2>, gdim(cdiv(P * 4, TB)), gdim(TB), 0, c->stream, tp, g, P, ks.p, vs.p, head.p, counters.p + 1);
    else hipLaunchKernelGGL(dedup_verify_kernel<3>, gdim(cdiv(P * 8, TB)), gdim(TB), 0, c->stream, tp, g, P, ks.p, vs.p, head.p, counters.p + 1); }
    inclusive_sum_u32(c, head.p, hscan.p, P);
    PFP_HIP(hipGetLastError());
    PFP_HIP(hipMemcpyAsync(c->h_scalars, counters.p + 1, 4, hipMemcpyDeviceToHost, c->stream));
    PFP_HIP(hipMemcpyAsync(c->h_scalars + 1, hscan.p + (P - 1), 4, hipMemcpyDeviceToHost, c->stream));
    sync(c);
    uint32_t coll; memcpy(&coll, c->h_scalars, 4);
    memcpy(&d, c->h_scalars + 1, 4);
    if (!coll) break;
    PFP_REQUIRE(attempt < 3, PFP_ECOLLISION, "phrase hash collision survived 4 seeds (newscan.cpp:282-286)");
    seed = seed * 0x9E3779B97F4A7C15ULL + 0x7F4A7C15ULL;
    D.reseeds++;
  }
  D.d = d;
  // words by descending occurrence count, ties in first-occurrence order.  (The order inside the device dictionary is free -
  // the outputs depend on the words' lexicographic ranks only - and this one makes the lowest position of a family of
  // near-identical words its most frequent member: the pivot of the suffix sorter's pivot rounds is then the word the variants
  // deviate from, and every variant is placed by its own difference in one round.)
  DBuf<uint32_t> hrep(c, d), hpos(c, (size_t)d + 1), hwi(c, d), hw_sorted(c, d), fo_of_hw(c, d);
  DBuf<uint64_t> okey(c, d), rep_sorted(c, d);
  DBuf<uint32_t> hocc;
  if (weight) {
    hocc.alloc(c, d);
    hocc.zero();
    hipLaunchKernelGGL(weighted_occ_kernel, gdim(cdiv(P, TB)), gdim(TB), 0, c->stream, P, hscan.p, vs.p, weight, hocc.p);
  }
  hipLaunchKernelGGL(collect_words_kernel, gdim(cdiv(P, TB)), gdim(TB), 0, c->stream, P, head.p, hscan.p, vs.p, hrep.p,
                     hpos.p, d);
  hipLaunchKernelGGL(iota_u32_kernel, gdim(cdiv(d, TB)), gdim(TB), 0, c->stream, hwi.p, (uint64_t)d);
  hipLaunchKernelGGL(word_order_keys_kernel, gdim(cdiv(d, TB)), gdim(TB), 0, c->stream, d, hrep.p, hpos.p,
                     weight ? hocc.p : (const uint32_t *)nullptr, okey.p);
  sort_pairs_u64_u32(c, okey.p, rep_sorted.p, hwi.p, hw_sorted.p, d, 0, 64);
  D.wlen.alloc(c, d); D.wocc.alloc(c, d); D.woff.alloc(c, (size_t)d + 1); D.pid.alloc(c, P);
  DBuf<uint32_t> wlen1(c, (size_t)d + 1);
  DBuf<uint64_t> wsrc(c, d);
  hipLaunchKernelGGL(finish_words_kernel, gdim(cdiv(d, TB)), gdim(TB), 0, c->stream, g, d, rep_sorted.p, hw_sorted.p,
                     hpos.p, weight ? hocc.p : (const uint32_t *)nullptr, fo_of_hw.p, D.wlen.p, wlen1.p, wsrc.p, D.wocc.p,
                     counters.p + 2);
  hipLaunchKernelGGL(assign_pid_kernel, gdim(cdiv(P, TB)), gdim(TB), 0, c->stream, P, hscan.p, vs.p, fo_of_hw.p, D.pid.p);
  exclusive_sum_u32_u64(c, wlen1.p, D.woff.p, (size_t)d + 1);
  PFP_HIP(hipMemcpyAsync(c->h_scalars, D.woff.p + d, 8, hipMemcpyDeviceToHost, c->stream));
  PFP_HIP(hipMemcpyAsync(c->h_scalars + 1, counters.p + 2, 4, hipMemcpyDeviceToHost, c->stream));
  sync(c);
  uint32_t toolong; memcpy(&toolong, c->h_scalars + 1, 4);
  D.dsize = c->h_scalars[0] + 1;
  PFP_REQUIRE(!toolong, PFP_ELIMIT, "a phrase of 4 GiB or more");
  PFP_REQUIRE(D.dsize < (1ull << 40), PFP_ELIMIT, "dictionary of 2^40 bytes or more");
  D.bytes.alloc(c, D.dsize + 64);
  PFP_HIP(hipMemsetAsync(D.bytes.p + (D.dsize - 1), 0, 65, c->stream));
  PFP_HIP(hipMemsetAsync(counters.p + 3, 0, 4, c->stream));
  hipLaunchKernelGGL(dict_copy_kernel, gdim(cdiv((uint64_t)d * 8, TB)), gdim(TB), 0, c->stream, tp, d, wsrc.p, D.wlen.p,
                     D.woff.p, D.bytes.p, long_list.p, counters.p + 3);
  uint32_t nlongw = read_scalar(c, counters.p + 3);
  if (nlongw)
    hipLaunchKernelGGL(dict_copy_long_kernel, gdim(c->n_cu * 4), gdim(TB), 0, c->stream, tp, wsrc.p, D.wlen.p, D.woff.p,
                       D.bytes.p, long_list.p, nlongw);
  if (want_last) {
    D.last.alloc(c, P);
    if (want_sai) D.sai.alloc(c, P);
    hipLaunchKernelGGL(last_sai_kernel, gdim(cdiv(P, TB)), gdim(TB), 0, c->stream, tp, g, P, sai_base, D.last.p,
                       want_sai ? D.sai.p : (uint64_t *)nullptr);
  }
  PFP_HIP(hipGetLastError());
}

void build_dictionary(pfp_ctx *c, const StagedText &tx, uint64_t n, int w, const DBuf<uint64_t> &ends, uint64_t n_ends,
                      bool want_sai, Dictionary &D) {
  PhraseGeom g{ends.p, n_ends, n, w, 0, nullptr, nullptr};
  build_dictionary_core(c, tx.tprime(), g, n_ends + 1, nullptr, n + (uint64_t)w * (n_ends + 1), true, want_sai, 0, D);
}

// a text shard owning phrases [k0, k0+P) of its local scan (multi-GPU); sai values shifted to global positions
void build_dictionary_shard(pfp_ctx *c, const StagedText &tx, uint64_t n, int w, const DBuf<uint64_t> &ends,
                            uint64_t n_ends, uint64_t k0, uint64_t P, bool want_sai, uint64_t sai_base, Dictionary &D) {
  PhraseGeom g{ends.p, n_ends, n, w, k0, nullptr, nullptr};
  build_dictionary_core(c, tx.tprime(), g, P, nullptr, n + (uint64_t)w * P, true, want_sai, sai_base, D);
}

// union of word lists (each word `weight` times): the global dictionary of a multi-GPU run
void build_dictionary_words(pfp_ctx *c, const uint8_t *bytes, const uint64_t *wstart, const uint32_t *wlen, uint64_t U,
                            const uint32_t *weight, uint64_t total_bytes, Dictionary &D) {
  PhraseGeom g{nullptr, 0, 0, 0, 0, wstart, wlen};
  build_dictionary_core(c, bytes, g, U, weight, total_bytes, false, false, 0, D);
}

// Words of a (duplicate-free) dictionary re-ordered by descending occurrence count, ties in their old order.
// The multi-GPU chain assembles its global dictionary owner by owner (hash classes), so the lowest position of a
// family of near-identical words - the pivot of the suffix sorter's pivot rounds - is a random variant; a variant's
// own difference splits the family in two each round (15 rounds on 8 x 64 copies).  After this pass the lowest
// position is the family's most frequent word, in a collection of variants the one the others deviate from: every
// member then differs from the pivot at its OWN difference and is placed in one round, as the single-GPU chain's
// first-occurrence order has it by construction.  perm[old index] = new index.
__global__ void occ_sort_keys_kernel(uint32_t d, const uint32_t *__restrict__ occ, uint32_t *__restrict__ key, uint32_t *__restrict__ val) {
  uint32_t j = BID * blockDim.x + threadIdx.x;
  if (j < d) { key[j] = ~occ[j]; val[j] = j; }
}
__global__ void occ_order_apply_kernel(uint32_t d, const uint32_t *__restrict__ order, const uint64_t *__restrict__ woff,
                                       const uint32_t *__restrict__ wlen, const uint32_t *__restrict__ occ, uint64_t *__restrict__ wsrc,
                                       uint32_t *__restrict__ nlen, uint32_t *__restrict__ len1, uint32_t *__restrict__ nocc,
                                       uint32_t *__restrict__ perm) {
  uint32_t r = BID * blockDim.x + threadIdx.x;
  if (r == 0) len1[d] = 0;
  if (r >= d) return;
  const uint32_t j = order[r];
  wsrc[r] = woff[j]; nlen[r] = wlen[j]; len1[r] = wlen[j] + 1; nocc[r] = occ[j]; perm[j] = r;
}
void reorder_dictionary_by_occ(pfp_ctx *c, Dictionary &D, DBuf<uint32_t> &perm) {
  const uint32_t d = (uint32_t)D.d;
  const int TB = 256;
  perm.alloc(c, std::max<uint32_t>(d, 1));
  if (d < 2) { if (d) perm.zero(); return; }
  KScope ks(c, "pfp::occ_order_apply_kernel", 2 * D.dsize + 40ull * d);
  DBuf<uint32_t> key(c, d), keyo(c, d), val(c, d), order(c, d), nlen(c, d), len1(c, (size_t)d + 1), nocc(c, d), long_list(c, d), cnt(c, 1);
  DBuf<uint64_t> wsrc(c, d), nwoff(c, (size_t)d + 1);
  hipLaunchKernelGGL(occ_sort_keys_kernel, gdim(cdiv(d, TB)), gdim(TB), 0, c->stream, d, D.wocc.p, key.p, val.p);
  sort_pairs_u32_u32(c, key.p, keyo.p, val.p, order.p, d, 0, 32);      // stable: equal counts keep their order
  hipLaunchKernelGGL(occ_order_apply_kernel, gdim(cdiv(d, TB)), gdim(TB), 0, c->stream, d, order.p, D.woff.p, D.wlen.p, D.wocc.p, wsrc.p,
                     nlen.p, len1.p, nocc.p, perm.p);
  exclusive_sum_u32_u64(c, len1.p, nwoff.p, (size_t)d + 1);
  DBuf<uint8_t> nb(c, D.dsize + 64);
  PFP_HIP(hipMemsetAsync(nb.p + (D.dsize - 1), 0, 65, c->stream));
  cnt.zero();
  hipLaunchKernelGGL(dict_copy_kernel, gdim(cdiv((uint64_t)d * 8, TB)), gdim(TB), 0, c->stream, D.bytes.p, d, wsrc.p, nlen.p, nwoff.p, nb.p,
                     long_list.p, cnt.p);
  const uint32_t nlong = read_scalar(c, cnt.p);
  if (nlong)
    hipLaunchKernelGGL(dict_copy_long_kernel, gdim(c->n_cu * 4), gdim(TB), 0, c->stream, D.bytes.p, wsrc.p, nlen.p, nwoff.p, nb.p, long_list.p,
                       nlong);
  PFP_HIP(hipGetLastError());
  D.bytes = std::move(nb); D.woff = std::move(nwoff); D.wlen = std::move(nlen); D.wocc = std::move(nocc);
}

}  // namespace pfp
