// merge.hip -- stage 2 (BWT of the parse, inverted lists) and stage 3 (final BWT / SA).
//
// Stage 2 replaces bwtparse.c:212-322; stage 3 replaces bwt() (pfbwt.cpp:109-242), its
// writers fwrite_chars_same_suffix{,_sa,_ssa} (pfbwt.cpp:520-676) and the threaded variant
// (pfthreads.hpp:83-518).  The reference walks SA(D) serially and fputc()s each char; here
// the walk is a data-parallel decomposition (the same one pfthreads.hpp:456-493 uses for its
// thread ranges): every dictionary suffix longer than w contributes occ(word) output chars,
// an exclusive scan of those counts fixes every output offset, and then
//   * fill / full-word entries are expanded output-centrically (each thread owns 16
//     consecutive BWT bytes, finds its SA(D) slot by binary search and walks forward),
//   * groups of equal suffixes whose members disagree (or any multi-word group when SA values
//     are requested) are merged by rank computation over the members' inverted lists - the
//     data-parallel form of the reference's heap merge (pfbwt.cpp:537-556).
#include "kernels.hpp"
#include "prims.hpp"
#include "devutil.hpp"
#include <algorithm>
#include <cstdlib>

namespace pfp {

static constexpr int TB = 256;
constexpr int kOffTileLog = 11;      // slots per offset tile (= one expand workgroup): 2048

// ------------------------------------------------------------------ dictionary index

// Word lookup over the dictionary (wordview.hpp): terminators per 64-byte line, scanned, and the word ends from the
// word table.  |D| / 16 + 8 d bytes instead of the 8 bytes per dictionary byte of pos_word[] / slen[] (rounds 1-2).
__global__ __launch_bounds__(256) void line_terms_kernel(const uint8_t *__restrict__ b, uint64_t N, uint64_t nlines, uint32_t *__restrict__ cnt) {
  const uint64_t ln = (uint64_t)BID * 256 + threadIdx.x;
  if (ln > nlines) return;
  if (ln == nlines) { cnt[ln] = 0; return; }
  const uint64_t b0 = ln * 64;
  const uint32_t nb = N - b0 >= 64 ? 64u : (uint32_t)(N - b0);      // (the padding behind the dictionary is zero, but keep the count exact)
  const uint4 *line = reinterpret_cast<const uint4 *>(b + b0);
  uint32_t n = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) if (nb > 16u * q) n += count_term_bytes16(line[q], nb - 16u * q < 16u ? nb - 16u * q : 16u);
  cnt[ln] = n;
}
__global__ void word_ends_kernel(uint32_t d, const uint64_t *__restrict__ woff, const uint32_t *__restrict__ wlen, uint64_t dsize,
                                 uint64_t *__restrict__ wend) {
  const uint32_t j = BID * blockDim.x + threadIdx.x;
  if (j > d) return;
  wend[j] = j == d ? dsize - 1 : woff[j] + wlen[j];      // the word's 0x01; the final 0x00 is its own word
}

void build_dict_index(pfp_ctx *c, const Dictionary &D, DictIndex &ix) {
  const uint64_t N = D.dsize;
  PFP_REQUIRE(D.woff.p && D.wlen.p, PFP_EINVAL, "dictionary without a word table");
  PFP_REQUIRE(((uintptr_t)D.bytes.p & 63) == 0, PFP_EINVAL, "dictionary bytes must be 64-byte aligned");
  const uint64_t nlines = cdiv64(N, 64);
  DBuf<uint32_t> cnt(c, nlines + 1);
  ix.blk_word.alloc(c, nlines + 1);
  ix.wend.alloc(c, D.d + 1);
  KScope ks(c, "pfp::line_terms_kernel", N + nlines * 12 + D.d * 20);
  hipLaunchKernelGGL(line_terms_kernel, gdim(cdiv(nlines + 1, TB)), gdim(TB), 0, c->stream, D.bytes.p, N, nlines, cnt.p);
  exclusive_sum_u32(c, cnt.p, ix.blk_word.p, nlines + 1);
  hipLaunchKernelGGL(word_ends_kernel, gdim(cdiv((uint64_t)D.d + 1, TB)), gdim(TB), 0, c->stream, (uint32_t)D.d, D.woff.p, D.wlen.p, N, ix.wend.p);
  PFP_HIP(hipGetLastError());
}

// word table of a dictionary given as bytes (words + 0x01, closed by 0x00): terminator positions by
// compaction of the 0x01 bytes, then starts and lengths
__global__ void words_from_ends_kernel(uint32_t d, const uint64_t *__restrict__ ends, uint64_t dsize, uint64_t *__restrict__ woff,
                                       uint32_t *__restrict__ wlen) {
  uint32_t j = BID * blockDim.x + threadIdx.x;
  if (j == 0) woff[d] = dsize - 1;
  if (j >= d) return;
  const uint64_t s0 = j ? (uint64_t)ends[j - 1] + 1 : 0;
  woff[j] = s0;
  wlen[j] = (uint32_t)(ends[j] - s0);
}
void word_table_from_bytes(pfp_ctx *c, Dictionary &D, uint64_t max_words) {
  DBuf<uint64_t> ends(c, max_words + 1), cnt(c, 1);
  select_byte_index<uint64_t>(c, D.bytes.p, kEndOfWord, ends.p, cnt.p, D.dsize);
  D.d = read_scalar(c, cnt.p);
  PFP_REQUIRE(D.d <= max_words, PFP_EFORMAT, "more words in the dictionary bytes than announced");
  D.woff.alloc(c, D.d + 1); D.wlen.alloc(c, std::max<uint64_t>(D.d, 1));
  if (D.d)
    hipLaunchKernelGGL(words_from_ends_kernel, gdim(cdiv(D.d, TB)), gdim(TB), 0, c->stream, (uint32_t)D.d, ends.p, D.dsize, D.woff.p,
                       D.wlen.p);
  PFP_HIP(hipGetLastError());
}

// Lexicographic rank of every word.  A whole word is a singleton group in SA(D) (the parse is
// prefix free), so rank[start of word] is its slot: sorting the d words by that slot gives the
// order std::sort produces in the reference (newscan.cpp:622-636) without touching all N slots.
__global__ void iota_u32_kernel(uint32_t d, uint32_t *__restrict__ val) {
  uint32_t j = BID * blockDim.x + threadIdx.x;
  if (j < d) val[j] = j;
}
__global__ void lexrank_from_order_kernel(uint32_t d, const uint32_t *__restrict__ word_sorted, uint32_t *__restrict__ lexrank) {
  uint32_t r = BID * blockDim.x + threadIdx.x;
  if (r < d) lexrank[word_sorted[r]] = r;
}

// multi-GPU: every share of the suffix array reported 1 + slot for the words it holds, 0 for the others
__global__ void combine_word_slots_kernel(uint32_t d, uint32_t parts, const uint64_t *__restrict__ wslot_all,
                                          uint64_t *__restrict__ key, uint32_t *__restrict__ missing) {
  uint32_t j = BID * blockDim.x + threadIdx.x;
  if (j >= d) return;
  uint64_t v = 0;
  for (uint32_t r = 0; r < parts; r++) { const uint64_t x = wslot_all[(uint64_t)r * d + j]; v = x > v ? x : v; }
  if (v == 0) atomicAdd(missing, 1u);
  key[j] = v - 1;
}
void compute_lexrank_from_slots(pfp_ctx *c, const Dictionary &D, const uint64_t *d_wslot_all, uint32_t parts, DictIndex &ix) {
  const uint32_t d = (uint32_t)D.d;
  ix.lexrank.alloc(c, d);
  DBuf<uint64_t> key(c, d), keyo(c, d);
  DBuf<uint32_t> val(c, d), valo(c, d), missing(c, 1);
  missing.zero();
  hipLaunchKernelGGL(combine_word_slots_kernel, gdim(cdiv(d, TB)), gdim(TB), 0, c->stream, d, parts, d_wslot_all, key.p, missing.p);
  PFP_REQUIRE(read_scalar(c, missing.p) == 0, PFP_EFORMAT, "a dictionary word was claimed by no share of the suffix array");
  hipLaunchKernelGGL(iota_u32_kernel, gdim(cdiv(d, TB)), gdim(TB), 0, c->stream, d, val.p);
  sort_pairs_u64_u32(c, key.p, keyo.p, val.p, valo.p, d, 0, bits_for(D.dsize));
  hipLaunchKernelGGL(lexrank_from_order_kernel, gdim(cdiv(d, TB)), gdim(TB), 0, c->stream, d, valo.p, ix.lexrank.p);
  PFP_HIP(hipGetLastError());
  ix.wslot_lex = std::move(keyo);      // the slots in ascending order = in the words' lexicographic order
}

// number of BWT positions the slots of `so` emit (sum of the occurrence counts of their words)
template <class I>
__global__ __launch_bounds__(256) void slot_output_count_kernel(uint64_t n, const I *__restrict__ sa, WordView wv,
                                                                const uint32_t *__restrict__ wocc, uint32_t d, int w,
                                                                unsigned long long *__restrict__ total) {
  __shared__ unsigned long long ws[4];
  unsigned long long cnt = 0;
  for (uint64_t t = (uint64_t)BID * 256 + threadIdx.x; t < n; t += (uint64_t)GDIM * 256) {
    const I i = sa[t];
    const uint32_t wd = word_of(wv, i);
    if (wd < d && wv.wend[wd] - i > (uint64_t)w) cnt += wocc[wd];
  }
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) { const unsigned long long t2 = ws[0] + ws[1] + ws[2] + ws[3]; if (t2) atomicAdd(total, t2); }
}
template <class I>
uint64_t count_slot_outputs(pfp_ctx *c, const Dictionary &D, const DictIndex &ix, const SuffixOrderT<I> &so, int w) {
  DBuf<unsigned long long> total(c, 1);
  total.zero();
  if (so.N)
    hipLaunchKernelGGL(slot_output_count_kernel<I>, gdim((int)std::min<uint64_t>(cdiv64(so.N, 256), (uint64_t)c->n_cu * 16)), gdim(256),
                       0, c->stream, so.N, so.sa.p, word_view(D, ix), D.wocc.p, (uint32_t)D.d, w, total.p);
  PFP_HIP(hipGetLastError());
  PFP_HIP(hipMemcpyAsync(c->h_scalars, total.p, 8, hipMemcpyDeviceToHost, c->stream));
  sync(c);
  return c->h_scalars[0];
}
template uint64_t count_slot_outputs<uint32_t>(pfp_ctx *, const Dictionary &, const DictIndex &, const SuffixOrderT<uint32_t> &, int);
template uint64_t count_slot_outputs<uint64_t>(pfp_ctx *, const Dictionary &, const DictIndex &, const SuffixOrderT<uint64_t> &, int);

template <class I>
__global__ void widen_kernel(uint32_t n, const I *__restrict__ in, uint64_t *__restrict__ out) {
  uint32_t j = BID * blockDim.x + threadIdx.x;
  if (j < n) out[j] = (uint64_t)in[j];
}
template <class I>
void compute_lexrank(pfp_ctx *c, const Dictionary &D, SuffixOrderT<I> &so, DictIndex &ix) {
  const uint32_t d = (uint32_t)D.d;
  ix.lexrank.alloc(c, d);
  DBuf<I> key(c, d), keyo(c, d);
  DBuf<uint32_t> val(c, d), valo(c, d);
  gather_ranks<I>(c, so, D.woff.p, d, key.p);
  hipLaunchKernelGGL(iota_u32_kernel, gdim(cdiv(d, TB)), gdim(TB), 0, c->stream, d, val.p);
  sort_pairs<I, uint32_t>(c, key.p, keyo.p, val.p, valo.p, d, 0, bits_for(D.dsize));
  hipLaunchKernelGGL(lexrank_from_order_kernel, gdim(cdiv(d, TB)), gdim(TB), 0, c->stream, d, valo.p, ix.lexrank.p);
  ix.wslot_lex.alloc(c, d);
  hipLaunchKernelGGL((widen_kernel<I>), gdim(cdiv(d, TB)), gdim(TB), 0, c->stream, d, keyo.p, ix.wslot_lex.p);
  PFP_HIP(hipGetLastError());
}
template void compute_lexrank<uint32_t>(pfp_ctx *, const Dictionary &, SuffixOrderT<uint32_t> &, DictIndex &);
template void compute_lexrank<uint64_t>(pfp_ctx *, const Dictionary &, SuffixOrderT<uint64_t> &, DictIndex &);

// ------------------------------------------------------------------ stage 2: BWT of the parse

// bwtparse.c:242-267: BWT(P)[j] = P[SA[j]-1]; bwlast = last of the phrase before that one
// (cyclically), bwsai = sai of that phrase; SA[j]==0 -> dummy zeros.
// From ONE record per phrase (round 4; three gathers - sym, last, sai - before): {symbol, last char of the phrase before, sai} packed in parse order by a streaming
// pass, so that the suffix-array order costs one random 16-byte (8-byte without sai) access per phrase instead of three.
__global__ void parse_pack16_kernel(uint64_t P, const uint32_t *__restrict__ sym, const uint8_t *__restrict__ last,
                                    const uint64_t *__restrict__ sai, uint4 *__restrict__ rec) {
  const uint64_t t = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (t >= P) return;
  const uint64_t v = sai[t];
  rec[t] = make_uint4(sym[t], (uint32_t)(t == 0 ? last[P - 1] : last[t - 1]), (uint32_t)v, (uint32_t)(v >> 32));
}
__global__ void parse_pack8_kernel(uint64_t P, const uint32_t *__restrict__ sym, const uint8_t *__restrict__ last, uint2 *__restrict__ rec) {
  const uint64_t t = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (t >= P) return;
  rec[t] = make_uint2(sym[t], (uint32_t)(t == 0 ? last[P - 1] : last[t - 1]));
}
template <class R>
__global__ void parse_gather_rec_kernel(uint64_t P, const uint32_t *__restrict__ sa, const R *__restrict__ rec,
                                        uint32_t *__restrict__ bwtp, uint8_t *__restrict__ bwlast,
                                        uint64_t *__restrict__ bwsai, uint32_t *__restrict__ jidx) {
  const uint64_t j = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (j > P) return;
  const uint64_t s = sa[j];
  jidx[j] = (uint32_t)j;
  if (s == 0) {
    bwtp[j] = 0; bwlast[j] = 0;
    if constexpr (sizeof(R) == 16) bwsai[j] = 0;
  } else {
    const R r = rec[s - 1];
    bwtp[j] = r.x; bwlast[j] = (uint8_t)r.y;
    if constexpr (sizeof(R) == 16) bwsai[j] = (uint64_t)r.z | ((uint64_t)r.w << 32);
  }
}

void parse_bwt(pfp_ctx *c, const uint32_t *parse_sym, uint64_t P, const uint8_t *last, const uint64_t *sai,
               const uint32_t *occ_lex, uint64_t d, ParseBWT &out, const uint32_t *sa_given) {
  PFP_REQUIRE(P >= 2, PFP_ESHORT, "parse has fewer than 2 phrases (bwtparse.c:244)");
  out.P = P;
  DBuf<uint32_t> sym(c, P + 1);
  PFP_HIP(hipMemcpyAsync(sym.p, parse_sym, P * 4, hipMemcpyDeviceToDevice, c->stream));
  PFP_HIP(hipMemsetAsync(sym.p + P, 0, 4, c->stream));
  SuffixOrder so;
  if (sa_given) {      // multi-GPU chain: the ranks sorted a share each and gathered them (pfp_dist_parse_sort)
    so.N = so.NP = P + 1;
    so.sa.alloc(c, P + 1);
    PFP_HIP(hipMemcpyAsync(so.sa.p, sa_given, (P + 1) * 4, hipMemcpyDeviceToDevice, c->stream));
  } else
    sort_int_suffixes(c, sym.p, P + 1, so, d, occ_lex, (uint32_t)d);      // symbols are 1-based word ranks <= d; occ_lex[rank] = the word's count
  if (c->debug) validate_int_sa(c, sym.p, so);
  out.rounds = so.rounds;
  out.ilist.alloc(c, P + 1);
  out.bwlast.alloc(c, P + 1);
  if (sai) out.bwsai.alloc(c, P + 1);
  DBuf<uint32_t> bwtp(c, P + 1), bwtp_s(c, P + 1), jidx(c, P + 1);
  if (sai) {
    DBuf<uint4> rec(c, P);
    { KScope ks(c, "pfp::parse_pack16_kernel", P * (13 + 16));
      hipLaunchKernelGGL(parse_pack16_kernel, gdim(cdiv(P, TB)), gdim(TB), 0, c->stream, P, (const uint32_t *)sym.p, last, sai, rec.p); }
    KScope ks(c, "pfp::parse_gather_rec_kernel", (P + 1) * (4 + 16 + 4 + 1 + 8 + 4));
    hipLaunchKernelGGL(parse_gather_rec_kernel<uint4>, gdim(cdiv(P + 1, TB)), gdim(TB), 0, c->stream, P, (const uint32_t *)so.sa.p, (const uint4 *)rec.p,
                       bwtp.p, out.bwlast.p, out.bwsai.p, jidx.p);
  } else {
    DBuf<uint2> rec(c, P);
    { KScope ks(c, "pfp::parse_pack8_kernel", P * (5 + 8));
      hipLaunchKernelGGL(parse_pack8_kernel, gdim(cdiv(P, TB)), gdim(TB), 0, c->stream, P, (const uint32_t *)sym.p, last, rec.p); }
    KScope ks(c, "pfp::parse_gather_rec_kernel", (P + 1) * (4 + 8 + 4 + 1 + 4));
    hipLaunchKernelGGL(parse_gather_rec_kernel<uint2>, gdim(cdiv(P + 1, TB)), gdim(TB), 0, c->stream, P, (const uint32_t *)so.sa.p, (const uint2 *)rec.p,
                       bwtp.p, out.bwlast.p, (uint64_t *)nullptr, jidx.p);
  }
  // bwtparse.c:281-303: positions grouped by symbol, ascending inside a group == stable sort
  { SortTag tag("inverted list"); sort_pairs_u32_u32(c, bwtp.p, bwtp_s.p, jidx.p, out.ilist.p, P + 1, 0, bits_for(d)); }
  PFP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ stage 3: merge

template <class I> struct alignas(16) Idx8 { I v[8]; };
// whole-word slots (preceding char = EndOfWord) of the block's 2048 slots -> tile_full[block]; smallest such slot -> first_full
__device__ __forceinline__ void tile_full_count(const uint32_t p8[8], int nk, uint64_t t0, uint32_t *__restrict__ tile_full,
                                                unsigned long long *__restrict__ first_full) {
  __shared__ uint32_t ws[4];
  uint32_t cfull = 0;
  int firstk = -1;
#pragma unroll
  for (int k = 7; k >= 0; k--) if (k < nk && p8[k] == kEndOfWord) { cfull++; firstk = k; }
  const unsigned long long anyf = __ballot(cfull != 0);
  for (int o = 32; o > 0; o >>= 1) cfull += __shfl_down(cfull, o, 64);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = cfull;
  // the first lane of the first wave that holds one reports it (slots ascend with the thread index)
  if (anyf && (int)(threadIdx.x & 63) == __ffsll((long long)anyf) - 1) atomicMin(first_full, (unsigned long long)(t0 + firstk));
  __syncthreads();
  if (threadIdx.x == 0) tile_full[BID] = ws[0] + ws[1] + ws[2] + ws[3];
}
// The same per-slot outputs when the records travelled in the top 16 bits of the first-round keys
// (SuffixOrder::paybits): a streaming read of skeys; only slots that later rounds re-ordered
// (refined[t]) fetch their record, and a count code of 255 its word's count.
template <class I>
__global__ __launch_bounds__(256) void slot_payload_kernel(uint64_t N, const I *__restrict__ sa,
                                                           const uint64_t *__restrict__ skeys,
                                                           const uint8_t *__restrict__ refined, const uint8_t *__restrict__ b,
                                                           WordView wv,
                                                           const uint32_t *__restrict__ wocc, uint32_t d, int w,
                                                           uint32_t *__restrict__ cnt, uint8_t *__restrict__ pc,
                                                           uint32_t *__restrict__ tile_full, unsigned long long *__restrict__ first_full) {
  if ((uint64_t)BID * 2048 >= N) return;      // a workgroup of the padded last grid row
  uint64_t t0 = ((uint64_t)BID * 256 + threadIdx.x) * 8;
  uint32_t c8[8], p8[8];
  const int nk = t0 >= N ? 0 : ((N - t0) >= 8 ? 8 : (int)(N - t0));
  uint64_t rf = 0;
  if (refined) rf = nk == 8 ? *reinterpret_cast<const uint64_t *>(refined + t0) : ~0ull;      // tail: take the slow path
#pragma unroll
  for (int k = 0; k < 8; k++) {
    uint32_t rec = 0;
    if (k < nk) {
      rec = (uint32_t)(skeys[t0 + k] >> 48);
      if ((rf >> (8 * k)) & 0xffu) {
        const I i = sa[t0 + k];
        const uint32_t wd = word_of(wv, i);
        rec = 0;
        if (wd < d && wv.wend[wd] - i > (uint64_t)w) {
          const uint32_t occ = wocc[wd];
          rec = (i == 0 ? (uint32_t)kEndOfWord : (uint32_t)b[i - 1]) | ((occ < 255u ? occ : 255u) << 8);
        }
      }
    }
    p8[k] = rec & 0xffu; c8[k] = rec >> 8;
  }
#pragma unroll
  for (int k = 0; k < 8; k++) if (c8[k] == 255u) c8[k] = wocc[word_of(wv, sa[t0 + k])];
  if (nk == 8) {
    *reinterpret_cast<uint4 *>(cnt + t0) = make_uint4(c8[0], c8[1], c8[2], c8[3]);
    *reinterpret_cast<uint4 *>(cnt + t0 + 4) = make_uint4(c8[4], c8[5], c8[6], c8[7]);
    *reinterpret_cast<uint2 *>(pc + t0) = make_uint2(p8[0] | (p8[1] << 8) | (p8[2] << 16) | (p8[3] << 24),
                                                     p8[4] | (p8[5] << 8) | (p8[6] << 16) | (p8[7] << 24));
  } else {
    for (int k = 0; k < nk; k++) { cnt[t0 + k] = c8[k]; pc[t0 + k] = (uint8_t)p8[k]; }
  }
  tile_full_count(p8, nk, t0, tile_full, first_full);
}
// ------------------------------------------------------------------ one gather per slot (BWT only / sparse SA)
//
// Random reads from HBM run at ~54 G accesses/s on this GPU whatever the element size up to 16 bytes (tools/microbench/
// gather.hip; only tables of <= 4 MB, one XCD's L2, are faster), so the merge makes exactly ONE random access per
// SA(D) slot and lets it carry everything any later kernel wants to know about the slot's suffix: a 16-byte record
// per dictionary POSITION, written by a streaming pass, gathered once per slot, and laid down again as per-SLOT
// arrays that every later kernel reads coalesced.  (Before: a 2-byte record per slot, then pos_word[sa[t]] ->
// wrec[...] per member in unit_edges_kernel and per (occurrence, member) in hard_minor_kernel: 2-3 dependent random
// sectors each, 9.5x / 34x the algorithmic bytes by the PMC counters of round 2.)
//   occ   occurrences of the position's word, 0 = the suffix emits nothing (<= w long, pfbwt.cpp:151; final 0x00)
//   first / last   smallest / largest BWT(P) position of the word's inverted list (ilist[ist], ilist[ist + occ - 1])
//   pcsl  low byte: preceding char, 1 (EndOfWord) = the suffix is a whole word (pfbwt.cpp:153);
//         bits 8..31: suffix length = distance to the word's terminator, capped at kSlCap (then: slen[] has it)
struct alignas(16) PosRec { uint32_t occ, first, last, pcsl; };
constexpr uint32_t kSlCap = 0xFFFFFFu;
// 256 positions per workgroup: the word of the block's first position is one table read (blocks start at multiples of
// 64), the words of the others follow from the terminators before them in the block (ballots); consecutive positions
// share their word's record.  dense_ist: full SA output wants the start of the word's inverted list where the sparse
// modes want its smallest position (field `first`).
// Positions [pos0, pos1), out[i - pos0] (pos0 a multiple of 256).
__global__ __launch_bounds__(256) void pprec16_kernel(WordView wv, int w, const uint4 *__restrict__ wrec /* WordRec as two uint4 */,
                                                      int dense_ist, uint64_t pos0, uint64_t pos1, PosRec *__restrict__ out) {
  __shared__ uint32_t wt[4];
  const uint64_t i = pos0 + (uint64_t)BID * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wvi = threadIdx.x >> 6;
  const uint32_t ch = i < wv.NP ? (uint32_t)wv.bytes[i] : 0u;
  const unsigned long long tm = __ballot(ch == (uint32_t)kEndOfWord);
  if (lane == 0) wt[wvi] = (uint32_t)__popcll(tm);
  __syncthreads();
  if (i >= pos1) return;
  uint32_t wd = wv.blk_word[(pos0 >> 6) + (uint64_t)BID * 4] + (uint32_t)__popcll(tm & ((1ull << lane) - 1ull));
  for (int q = 0; q < wvi; q++) wd += wt[q];
  PosRec r{0u, 0u, 0u, 0u};
  if (wd < wv.d) {
    const uint64_t sl = wv.wend[wd] - i;
    if (sl > (uint64_t)w) {
      const uint4 wr = wrec[2 * (uint64_t)wd];      // {ist, occ, first, last}
      const uint32_t pc = (i == 0) ? (uint32_t)kEndOfWord : (uint32_t)wv.bytes[i - 1];
      r = PosRec{wr.y, dense_ist ? wr.x : wr.z, wr.w, pc | ((sl < kSlCap ? (uint32_t)sl : kSlCap) << 8)};
    }
  }
  *reinterpret_cast<uint4 *>(out + (i - pos0)) = make_uint4(r.occ, r.first, r.last, r.pcsl);
}
// the same record for ONE position, computed where it is wanted: the position's 64-byte line of the dictionary (word
// lookup + preceding char) and its word's 32-byte record - two random sectors instead of one, and no 16 bytes per
// dictionary position.  A share of the suffix array (multi-GPU: N slots of NP positions, N << NP) and a dictionary whose
// records would not fit take this form.
__device__ __forceinline__ uint4 posrec_at(const WordView &wv, int w, const uint4 *__restrict__ wrec, int dense_ist, uint64_t i) {
  const uint32_t wd = word_of(wv, i);
  if (wd >= wv.d) return make_uint4(0u, 0u, 0u, 0u);
  const uint4 wr = wrec[2 * (uint64_t)wd], we = wrec[2 * (uint64_t)wd + 1];      // {ist, occ, first, last} {terminator position, -}
  const uint64_t sl = ((uint64_t)we.x | ((uint64_t)we.y << 32)) - i;
  if (sl <= (uint64_t)w) return make_uint4(0u, 0u, 0u, 0u);
  const uint32_t pc = (i == 0) ? (uint32_t)kEndOfWord : (uint32_t)wv.bytes[i - 1];
  return make_uint4(wr.y, dense_ist ? wr.x : wr.z, wr.w, pc | ((sl < kSlCap ? (uint32_t)sl : kSlCap) << 8));
}

// Per tile of 2048 slots (256 threads x 8 consecutive slots): gather the records, exclusive scan of the counts inside
// the tile (loc[], tile total -> tsum[]), preceding chars, whole-word slots per tile, and the hard-group flags: a slot
// whose preceding char differs from that of the slot before it IN THE SAME GROUP marks the group's head (members of a
// group are contiguous, so "not all chars equal" is seen at some adjacent pair; pfbwt.cpp:524-536).  Replaces
// slot_gather + slot_loc + group_flags: cnt[] (4 B per slot written and read again) is gone, pc[] is not re-read.
template <class I>
__global__ __launch_bounds__(256) void slot_records_kernel(uint64_t N, const I *__restrict__ sa, const I *__restrict__ grp,
                                                           const PosRec *__restrict__ prec, uint32_t *__restrict__ loc,
                                                           uint8_t *__restrict__ pc, uint32_t *__restrict__ sfirst,
                                                           uint32_t *__restrict__ slast, uint32_t *__restrict__ ssl,
                                                           uint64_t *__restrict__ tsum, uint32_t *__restrict__ overflow,
                                                           uint32_t *__restrict__ tile_full, unsigned long long *__restrict__ first_full,
                                                           uint8_t *__restrict__ hard, int any_multi_is_hard, WordView view, int w,
                                                           const uint4 *__restrict__ wrec, int dense_ist) {
  // prec == null: no per-position records - every slot computes its own (posrec_at)
  __shared__ uint64_t ws[4];
  __shared__ uint32_t lpc[256];
  __shared__ I lgrp[256];
  if (((uint64_t)BID << kOffTileLog) > N) return;      // tiles 0 .. N >> 11 exist (a workgroup of the padded last grid row)
  const uint64_t t0 = ((uint64_t)BID << kOffTileLog) + (uint64_t)threadIdx.x * 8;
  const int nk = t0 >= N ? 0 : ((N - t0) >= 8 ? 8 : (int)(N - t0));
  I idx[8], g8[8];
  if (nk == 8) {
    const Idx8<I> v8 = *reinterpret_cast<const Idx8<I> *>(sa + t0), w8 = *reinterpret_cast<const Idx8<I> *>(grp + t0);
#pragma unroll
    for (int k = 0; k < 8; k++) { idx[k] = v8.v[k]; g8[k] = w8.v[k]; }
  } else {
    for (int k = 0; k < 8; k++) { idx[k] = k < nk ? sa[t0 + k] : (I)0; g8[k] = k < nk ? grp[t0 + k] : IdxTraits<I>::kNone; }
  }
  uint4 r8[8];
#pragma unroll
  for (int k = 0; k < 8; k++)
    r8[k] = k >= nk ? make_uint4(0u, 0u, 0u, 0u)
                    : (prec ? *reinterpret_cast<const uint4 *>(prec + (uint64_t)idx[k]) : posrec_at(view, w, wrec, dense_ist, (uint64_t)idx[k]));
  // the slot before this thread's first one: the previous thread's last slot, or (first thread) the last slot of the tile before
  uint32_t p8[8];
#pragma unroll
  for (int k = 0; k < 8; k++) p8[k] = r8[k].w & 0xffu;
  lpc[threadIdx.x] = p8[7]; lgrp[threadIdx.x] = g8[7];
  uint32_t prev_pc = 0; I prev_g = IdxTraits<I>::kNone;
  if (threadIdx.x == 0 && t0 > 0 && nk) {
    prev_pc = (prec ? prec[(uint64_t)sa[t0 - 1]].pcsl : posrec_at(view, w, wrec, dense_ist, (uint64_t)sa[t0 - 1]).w) & 0xffu;
    prev_g = grp[t0 - 1];
  }
  // counts: exclusive scan inside the tile
  uint64_t own = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) own += r8[k].x;
  uint64_t inc = own;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int o = 1; o < 64; o <<= 1) { const uint64_t u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
  if (lane == 63) ws[wv] = inc;
  __syncthreads();
  if (threadIdx.x > 0) { prev_pc = lpc[threadIdx.x - 1]; prev_g = lgrp[threadIdx.x - 1]; }
  uint64_t run = inc - own;
  for (int q = 0; q < wv; q++) run += ws[q];
  uint32_t o8[8];
#pragma unroll
  for (int k = 0; k < 8; k++) { o8[k] = (uint32_t)run; run += r8[k].x; }
  // (loc[] is padded to whole tiles: the slots past N get the running total, slot N's is what slot_off(N) reads)
  *reinterpret_cast<uint4 *>(loc + t0) = make_uint4(o8[0], o8[1], o8[2], o8[3]);
  *reinterpret_cast<uint4 *>(loc + t0 + 4) = make_uint4(o8[4], o8[5], o8[6], o8[7]);
  if (threadIdx.x == 255) {
    tsum[BID] = run;
    if (run >> 32) atomicOr(overflow, 1u);      // a tile's offsets would not fit 32 bits
  }
  if (nk == 8) {
    *reinterpret_cast<uint2 *>(pc + t0) = make_uint2(p8[0] | (p8[1] << 8) | (p8[2] << 16) | (p8[3] << 24),
                                                     p8[4] | (p8[5] << 8) | (p8[6] << 16) | (p8[7] << 24));
    if (sfirst) {
      *reinterpret_cast<uint4 *>(sfirst + t0) = make_uint4(r8[0].y, r8[1].y, r8[2].y, r8[3].y);
      *reinterpret_cast<uint4 *>(sfirst + t0 + 4) = make_uint4(r8[4].y, r8[5].y, r8[6].y, r8[7].y);
    }
    if (slast) {
      *reinterpret_cast<uint4 *>(slast + t0) = make_uint4(r8[0].z, r8[1].z, r8[2].z, r8[3].z);
      *reinterpret_cast<uint4 *>(slast + t0 + 4) = make_uint4(r8[4].z, r8[5].z, r8[6].z, r8[7].z);
    }
    if (ssl) {
      *reinterpret_cast<uint4 *>(ssl + t0) = make_uint4(r8[0].w >> 8, r8[1].w >> 8, r8[2].w >> 8, r8[3].w >> 8);
      *reinterpret_cast<uint4 *>(ssl + t0 + 4) = make_uint4(r8[4].w >> 8, r8[5].w >> 8, r8[6].w >> 8, r8[7].w >> 8);
    }
  } else {
    for (int k = 0; k < nk; k++) {
      pc[t0 + k] = (uint8_t)p8[k];
      if (sfirst) sfirst[t0 + k] = r8[k].y;
      if (slast) slast[t0 + k] = r8[k].z;
      if (ssl) ssl[t0 + k] = r8[k].w >> 8;
    }
  }
  // hard groups: a differing char next to a member of the same group
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (k < nk && p8[k] != 0 && g8[k] == prev_g && (any_multi_is_hard || p8[k] != prev_pc)) hard[g8[k]] = 1;      // (SA output: any group of several words, pfbwt.cpp:568, 612)
    prev_pc = p8[k]; prev_g = g8[k];
  }
  tile_full_count(p8, nk, t0, tile_full, first_full);
}

// whole words are singleton groups and stand in SA(D) in the words' lexicographic order: the q-th whole-word slot
// belongs to the word of rank q.  Rank of the first whole-word slot held (0 unless the slots are one rank's range).
template <class I>
__global__ void full_base0_kernel(const unsigned long long *__restrict__ first_full, const I *__restrict__ sa,
                                  WordView wv, const uint32_t *__restrict__ lexrank, uint32_t *__restrict__ out) {
  const unsigned long long f = *first_full;
  *out = f == ~0ull ? 0u : lexrank[word_of(wv, sa[f])];
}

// a group is "hard" when its members disagree on the preceding char (pfbwt.cpp:524-536), or, with
// SA output, whenever it has more than one member (pfbwt.cpp:568, 612)
template <class I>
__global__ void group_flags_kernel(uint64_t N, const I *__restrict__ grp, const uint8_t *__restrict__ pc,
                                   int any_multi_is_hard, uint8_t *__restrict__ hard) {
  uint64_t t = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (t >= N || pc[t] == 0) return;
  I g = grp[t];
  if (g == t) return;
  if (any_multi_is_hard || pc[t] != pc[g]) hard[g] = 1;
}

// CLS_MULTI (sparse SA mode only): member of a group of several words that all have the same preceding char -
// its chars are a fill, its SA values (needed at the two ends of the group's range only) come from group_edges_kernel
enum : uint8_t { CLS_NONE = 0, CLS_FILL = 1, CLS_FULL = 2, CLS_HARD = 3, CLS_MULTI = 4 };
// SA values asked for: none (BWT only), every position (-S), or only where the sampled SA files can look:
// at run boundaries of the BWT (-s / -e)
enum : int { SA_NONE = 0, SA_DENSE = 1, SA_SPARSE = 2 };

template <class I>
__global__ void gather_idx_kernel(uint64_t n, const I *__restrict__ idx, const I *__restrict__ src, I *__restrict__ dst) {
  uint64_t i = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}
// per word: start of its inverted list, number of occurrences, smallest / largest BWT(P) position, end - one 32-byte record,
// one memory sector for everything a hard-group or unit-edge kernel wants to know about a member's word
struct alignas(32) WordRec { uint32_t ist, occ, first, last; uint64_t wend, pad; };      // wend: position of the word's terminator
__global__ void wistart_kernel(uint32_t d, const uint32_t *__restrict__ lexrank, const uint32_t *__restrict__ istart_lex,
                               const uint32_t *__restrict__ wocc, const uint32_t *__restrict__ ilist,
                               const uint64_t *__restrict__ wend, uint32_t *__restrict__ wistart, WordRec *__restrict__ wrec) {
  uint32_t j = BID * blockDim.x + threadIdx.x;
  if (j >= d) return;
  const uint32_t st = istart_lex[lexrank[j]] + 1;   // +1: ilist[0] is the EOS symbol (pfbwt.cpp:389)
  wistart[j] = st;
  if (wrec) { const uint32_t oc = wocc[j]; wrec[j] = WordRec{st, oc, ilist[st], ilist[st + oc - 1], wend[j], 0}; }
}

template <class I>
struct MergeArgsT {
  uint64_t N, n_out; uint32_t d; int w; int want_sa;
  uint64_t pos_base, n_out_global;   // global BWT position of local position 0; global n+1 (== n_out unless the slots are one rank's range)
  uint64_t out_lo, out_hi;   // this call emits BWT positions [out_lo,out_hi) only (multi-GPU slices); bwt/out_sa are indexed by global position
  const I *sa, *grp;
  WordView wv;               // word / suffix length of a dictionary position where no per-slot copy exists
  const uint32_t *ist, *wistart;
  const WordRec *wrec;
  // per-slot copies of the gathered position records (slot_records_kernel; null: look the word up through sa / pos_word):
  // smallest / largest BWT(P) position of the slot's word, suffix length (kSlCap = look it up in slen[])
  const uint32_t *sfirst, *slast, *ssl;
  const uint8_t *pc, *hard, *gmaj;     // gmaj[g]: majority char of hard group g when the minority path places the rest (else 0)
  const uint64_t *tbase; const uint32_t *loc;    // output offset of slot t = tbase[t >> 11] + loc[t] (slot_off)
  const uint32_t *ilist; const uint8_t *bwlast; const uint64_t *bwsai;
  // whole-word slots: the q-th one (in slot order) is the word of lexicographic rank q, whose inverted list starts at
  // istart_lex[q] + 1; fullbase[tile] = whole-word slots before the tile, *fullbase0 = rank of the first one held
  const uint32_t *istart_lex, *fullbase, *fullbase0;
  uint8_t *bwt; uint64_t *out_sa;
  // what a launch writes: BWT chars (bit 0), SA values (bit 1).  Dense SA does both at once; sparse SA needs the finished
  // BWT to know where values are looked at, so its kernels run twice.
  int pass;
  // sparse SA: bit (x - out_lo) of bmap is set where position x starts or ends a run of the BWT (the two ends of the
  // slice always count); bpre[k] = boundaries before bit 64 k.  SA values go to sa_c[rank of x among the boundaries]
  // (the caller keeps no SA array), or to out_sa[x] - only where the bit is set.
  const uint64_t *bmap, *bpre;
  uint64_t *sa_c;
};
enum : int { PASS_BWT = 1, PASS_SA = 2 };
template <class I>
__device__ __forceinline__ void sa_put(const MergeArgsT<I> &a, uint64_t x, uint64_t v) {
  if (x < a.out_lo || x >= a.out_hi) return;
  if (!a.bmap) { a.out_sa[x] = v; return; }
  const uint64_t r = x - a.out_lo, wv = a.bmap[r >> 6];
  if (!((wv >> (r & 63)) & 1ull)) return;
  if (a.sa_c) a.sa_c[a.bpre[r >> 6] + (uint64_t)__popcll(wv & ((1ull << (r & 63)) - 1ull))] = v;
  else a.out_sa[x] = v;
}
template <class I>
__device__ __forceinline__ bool sa_wanted(const MergeArgsT<I> &a, uint64_t x) {      // would sa_put(x) store anything?
  if (x < a.out_lo || x >= a.out_hi) return false;
  if (!a.bmap) return true;
  const uint64_t r = x - a.out_lo;
  return (a.bmap[r >> 6] >> (r & 63)) & 1ull;
}
// Exclusive prefix of the per-slot counts, kept as a 64-bit base per 2048 slots (= one expand workgroup) and a
// 32-bit offset inside the tile: 4 bytes per slot to write and read instead of 8, and the N-long scan
// becomes one streaming tile kernel plus a scan over N/2048 sums.
template <class I>
__device__ __forceinline__ uint64_t slot_off(const MergeArgsT<I> &a, uint64_t t) { return a.tbase[t >> kOffTileLog] + a.loc[t]; }


__device__ __forceinline__ uint8_t fix_char(uint8_t ch) { return ch == kDollar ? 0 : ch; }  // pfbwt.cpp:126
// start of slot t's word in ilist: stored per slot when SA values are wanted, looked up otherwise
// (BWT only needs it for whole words and hard groups)
template <class I>
__device__ __forceinline__ uint32_t slot_ist(const MergeArgsT<I> &a, uint64_t t) {
  return a.ist ? a.ist[t] : a.wistart[word_of(a.wv, a.sa[t])];
}

// what the hard-group / unit-edge kernels ask about the word of slot t: from the per-slot arrays when the slot records
// were gathered (one coalesced read), else through sa -> pos_word -> the word's record (three dependent random reads)
template <class I>
__device__ __forceinline__ uint32_t slot_first(const MergeArgsT<I> &a, uint64_t t) { return a.sfirst ? a.sfirst[t] : a.wrec[word_of(a.wv, a.sa[t])].first; }
template <class I>
__device__ __forceinline__ uint32_t slot_last(const MergeArgsT<I> &a, uint64_t t) { return a.slast ? a.slast[t] : a.wrec[word_of(a.wv, a.sa[t])].last; }
template <class I>
__device__ __forceinline__ uint64_t slot_slen(const MergeArgsT<I> &a, uint64_t t) {
  if (a.ssl) { const uint32_t v = a.ssl[t]; if (v != kSlCap) return v; }
  return slen_of(a.wv, a.sa[t]);
}

// Expansion.  One workgroup owns kSlots consecutive SA(D) slots, i.e. one contiguous range of
// the output; slot offsets, classes and chars are staged in LDS and every thread then produces 16
// consecutive BWT bytes per iteration (binary search in LDS for the first one, forward walk for
// the rest) and stores them with one 16-byte store.
//   fill entry      : the preceding char, occ times                       (pfbwt.cpp:527-533)
//   full-word entry : bwlast[ilist[..]] per occurrence                    (pfbwt.cpp:153-197)
//   hard entry      : each occurrence is ranked among all occurrences of all members of its group
//                     by BWT(P) position - own index + lower_bound in every other member's
//                     inverted list - and written to that slot of the group's range: the
//                     data-parallel form of the reference's heap merge (pfbwt.cpp:537-556).
constexpr int kHardLds = 1024;   // occurrences of a group one wave sorts in LDS (hard_sort_kernel)
constexpr int kHardRank = 512;   // occurrences of one batch ranked in LDS (hard_groups_kernel): larger groups are sorted anyway,
                                 // and 7.6 KB of tables per wave instead of 10 let five workgroups share a CU instead of three
constexpr int kHardMem = 256;    // members of one batch
constexpr int kHardSortMin = 512;   // a group with more occurrences than this (and <= kHardLds) is sorted, not ranked all-pairs
constexpr int kSlots = 2048;
constexpr uint64_t kExpandQuota = 1u << 16;   // output bytes one workgroup expands before the rest is shared

struct ExpandLds {
  uint32_t loff[kSlots + 1];      // offsets inside the tile fit 32 bits (slot_loc_kernel checks)
  uint32_t lfist[kSlots];         // whole-word slots: start of the word's inverted list
  uint8_t lpc[kSlots], lcls[kSlots];
  uint32_t ccnt[kSlots / 64 + 1];
};

// tile = index of the 2048-slot tile starting at slot t0
template <class I>
__device__ __forceinline__ void expand_stage(const MergeArgsT<I> &a, ExpandLds &L, uint64_t t0, int ns, uint64_t base, uint64_t tile) {
  for (int s = threadIdx.x; s <= ns; s += 256) L.loff[s] = (uint32_t)(slot_off(a, t0 + s) - base);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t fullmask = 0;          // bit it: my slot of iteration it is a whole word; its rank inside the 64-slot chunk
  uint64_t franks = 0;
#pragma unroll
  for (int it = 0; it < kSlots / 256; it++) {
    const int s = it * 256 + threadIdx.x;
    uint8_t cls = CLS_NONE;
    if (s < ns) {
      const uint64_t t = t0 + s;
      uint8_t ch = a.pc[t];
      cls = ch == 0 ? CLS_NONE : (ch == kEndOfWord ? CLS_FULL : CLS_FILL);
      if (cls == CLS_FILL && a.pass != PASS_SA) {      // (the SA-only round of the sparse mode looks at whole words only)
        const I g = a.grp[t];
        if (a.hard[g]) {
          cls = CLS_HARD;
          ch = a.gmaj ? a.gmaj[g] : 0;      // the group's majority char (minority path) or 0 (every position written later)
        }
      }
      L.lpc[s] = ch;
      L.lcls[s] = cls;
    }
    const unsigned long long fb = __ballot(cls == CLS_FULL);
    if (lane == 0) L.ccnt[it * 4 + wv] = (uint32_t)__popcll(fb);
    if (cls == CLS_FULL) { fullmask |= 1u << it; franks |= (uint64_t)__popcll(fb & ((1ull << lane) - 1ull)) << (8 * it); }
  }
  __syncthreads();
  if (fullmask) {
    const uint32_t q0 = a.ist ? 0u : a.fullbase[tile] + *a.fullbase0;
#pragma unroll
    for (int it = 0; it < kSlots / 256; it++)
      if ((fullmask >> it) & 1u) {
        const int s = it * 256 + threadIdx.x;
        if (a.ist) { L.lfist[s] = a.ist[t0 + s]; continue; }
        uint32_t before = 0;
        for (int c2 = 0; c2 < it * 4 + wv; c2++) before += L.ccnt[c2];
        L.lfist[s] = a.istart_lex[q0 + before + (uint32_t)((franks >> (8 * it)) & 0xffu)] + 1u;   // +1: ilist[0] is the EOS symbol
      }
  }
  __syncthreads();
}

// 16 output bytes starting at block-relative offset x0 (< Ltot)
template <class I>
__device__ __forceinline__ void expand_16(const MergeArgsT<I> &a, const ExpandLds &L, uint64_t t0, int ns, uint64_t base,
                                          uint64_t x0, uint64_t Ltot) {
  if (base + x0 + 16 <= a.out_lo || base + x0 >= a.out_hi) return;     // outside this rank's slice
  int lo = 0, hi = ns;                    // loff[lo] <= x0 < loff[hi]
  while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (L.loff[mid] <= x0) lo = mid; else hi = mid; }
  int s = lo;
  uint64_t nxt = L.loff[s + 1];
  const int nb = (Ltot - x0) >= 16 ? 16 : (int)(Ltot - x0);
  uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    if (k < nb) {
      const uint64_t x = x0 + k;
      while (x >= nxt) { s++; nxt = L.loff[s + 1]; }
      const uint8_t cl = L.lcls[s];
      uint32_t ch = 0;
      if (cl == CLS_FILL || cl == CLS_MULTI) {
        ch = fix_char(L.lpc[s]);
      } else if (cl == CLS_FULL) {
        const uint64_t pos = a.ilist[L.lfist[s] + (uint32_t)(x - L.loff[s])];
        ch = a.bwlast[pos];
      } else {      // CLS_HARD: the group's majority char (its other occurrences are placed by hard_minor_kernel), or 0 and
        ch = L.lpc[s];      // every position of the group is written by the hard-group kernels that run after this one
      }
      const uint32_t sh = (uint32_t)ch << (8 * (k & 3));
      if (k < 4) r0 |= sh; else if (k < 8) r1 |= sh; else if (k < 12) r2 |= sh; else r3 |= sh;
    }
  }
  uint8_t *dst = a.bwt + base + x0;
  if (nb == 16 && base + x0 >= a.out_lo && base + x0 + 16 <= a.out_hi) st16u(dst, make_uint4(r0, r1, r2, r3));
  else {
    const uint32_t rr[4] = {r0, r1, r2, r3};
#pragma unroll
    for (int k = 0; k < 16; k++)
      if (k < nb && base + x0 + k >= a.out_lo && base + x0 + k < a.out_hi) dst[k] = (uint8_t)(rr[k >> 2] >> (8 * (k & 3)));
  }
}

// SA value of block-relative output position x (fill and full-word entries; hard groups write their
// own).  One lane per position, so that the 8-byte stores of a wave are 512 contiguous bytes - the
// 16-positions-per-thread layout of expand_16 would put every lane's store in a different 128-byte line.
template <class I>
__device__ __forceinline__ void expand_sa_1(const MergeArgsT<I> &a, const ExpandLds &L, uint64_t t0, int ns, uint64_t base,
                                            uint64_t x) {
  if (base + x < a.out_lo || base + x >= a.out_hi) return;
  int lo = 0, hi = ns;                    // loff[lo] <= x < loff[hi]
  while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (L.loff[mid] <= x) lo = mid; else hi = mid; }
  const uint8_t cl = L.lcls[lo];
  if (cl != CLS_FULL && (cl != CLS_FILL || a.want_sa == SA_SPARSE)) return;
  if (!sa_wanted(a, base + x)) return;
  const uint64_t pos = a.ilist[(cl == CLS_FULL ? L.lfist[lo] : slot_ist(a, t0 + lo)) + (uint32_t)(x - L.loff[lo])];
  sa_put(a, base + x, (cl == CLS_FULL && a.pos_base + base + x == 0) ? a.n_out_global - 1 : a.bwsai[pos] - slot_slen(a, t0 + lo));
}

// Sparse SA mode (-s / -e without -S): SA values are only looked at where a run of the BWT starts or ends.
//   * a whole word's occurrences carry unrelated chars (bwlast): all their SA values are written here, by the
//     workgroup that expands them (P positions in all);
//   * a slot, or a group of slots, whose chars all agree fills its range with one char: only the first and
//     the last position of that range can be run boundaries - unit_edges_kernel;
//   * hard groups: the two ends (unit_edges_kernel) and what the hard-group kernels find inside.
template <class I, int SPARSE>
__global__ __launch_bounds__(256) void expand_kernel(MergeArgsT<I> a, uint32_t *__restrict__ heavy, uint32_t *__restrict__ nheavy,
                                                     uint32_t heavy_cap) {
  __shared__ ExpandLds L;
  __shared__ uint32_t nfull, fpre[SPARSE ? kSlots + 1 : 1], wsum[4];
  __shared__ uint16_t fulls[SPARSE ? kSlots : 1];
  const uint64_t t0 = (uint64_t)BID * kSlots;
  if (t0 >= a.N) return;      // a workgroup of the padded last grid row
  const int ns = (a.N - t0) >= (uint64_t)kSlots ? kSlots : (int)(a.N - t0);
  const uint64_t base = slot_off(a, t0);
  if (SPARSE && threadIdx.x == 0) nfull = 0;
  expand_stage(a, L, t0, ns, base, BID);
  const uint64_t Ltot = L.loff[ns];
  if (base + Ltot <= a.out_lo || base >= a.out_hi) return;
  // a block whose slots emit more than the quota (a word with hundreds of thousands of
  // occurrences) finishes only its first quota here; expand_heavy_kernel shares the rest
  const uint64_t mine = Ltot <= kExpandQuota ? Ltot : kExpandQuota;
  if (Ltot > kExpandQuota && threadIdx.x == 0 && (a.pass & PASS_BWT)) { uint32_t i = atomicAdd(nheavy, 1u); if (i < heavy_cap) heavy[i] = BID; }
  if (a.pass & PASS_BWT)
    for (uint64_t x0 = (uint64_t)threadIdx.x * 16; x0 < mine; x0 += 256 * 16) expand_16(a, L, t0, ns, base, x0, Ltot);
  if (!(a.pass & PASS_SA)) return;
  if (SPARSE) {
    // the whole-word slots of the block and the positions they emit, laid end to end: one lane per position (a loop over
    // the slots with 256 lanes on each - most words occur a few times - was a chain of ~20 dependent gathers per block)
    for (int s = threadIdx.x; s < ns; s += 256)
      if (L.lcls[s] == CLS_FULL) fulls[atomicAdd(&nfull, 1u)] = (uint16_t)s;
    __syncthreads();
    const uint32_t nf = nfull;
    if (nf == 0) return;
    const uint32_t chunk = (nf + 255u) / 256u, q0 = threadIdx.x * chunk, q1 = q0 + chunk < nf ? q0 + chunk : nf;
    uint32_t own = 0;
    for (uint32_t q = q0; q < q1; q++) {
      const int sl = (int)fulls[q];
      const uint64_t b = L.loff[sl], e = L.loff[sl + 1] < mine ? L.loff[sl + 1] : mine;
      const uint32_t cq = e > b ? (uint32_t)(e - b) : 0u;
      fpre[q] = cq;
      own += cq;
    }
    uint32_t inc = own;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    uint32_t run = inc - own;
    for (int q = 0; q < wv; q++) run += wsum[q];
    for (uint32_t q = q0; q < q1; q++) { const uint32_t cq = fpre[q]; fpre[q] = run; run += cq; }
    const uint32_t total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (threadIdx.x == 0) fpre[nf] = total;
    __syncthreads();
    for (uint32_t idx = threadIdx.x; idx < total; idx += 256) {
      uint32_t lo = 0, hi = nf;                    // fpre[lo] <= idx < fpre[hi]
      while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (fpre[mid] <= idx) lo = mid; else hi = mid; }
      const int sl = (int)fulls[lo];
      const uint64_t x = L.loff[sl] + (idx - fpre[lo]);
      if (!sa_wanted(a, base + x)) continue;
      const uint64_t pos = a.ilist[L.lfist[sl] + (uint32_t)(x - L.loff[sl])];
      sa_put(a, base + x, a.pos_base + base + x == 0 ? a.n_out_global - 1 : a.bwsai[pos] - slot_slen(a, t0 + sl));
    }
  } else if (a.want_sa) {
    for (uint64_t x = threadIdx.x; x < mine; x += 256) expand_sa_1(a, L, t0, ns, base, x);
  }
}

template <class I>
__global__ __launch_bounds__(256) void expand_heavy_kernel(MergeArgsT<I> a, const uint32_t *__restrict__ heavy, uint32_t nheavy) {
  __shared__ ExpandLds L;
  for (uint32_t q = 0; q < nheavy; q++) {
    const uint64_t t0 = (uint64_t)heavy[q] * kSlots;
    const int ns = (a.N - t0) >= (uint64_t)kSlots ? kSlots : (int)(a.N - t0);
    const uint64_t base = slot_off(a, t0);
    __syncthreads();
    expand_stage(a, L, t0, ns, base, heavy[q]);
    const uint64_t Ltot = L.loff[ns];
    if (a.pass & PASS_BWT)
      for (uint64_t x0 = kExpandQuota + ((uint64_t)BID * 256 + threadIdx.x) * 16; x0 < Ltot; x0 += (uint64_t)GDIM * 256 * 16)
        expand_16(a, L, t0, ns, base, x0, Ltot);
    if (a.want_sa && (a.pass & PASS_SA))      // sparse mode: expand_sa_1 writes for full-word slots only (other classes come from unit_edges_kernel)
      for (uint64_t x = kExpandQuota + (uint64_t)BID * 256 + threadIdx.x; x < Ltot; x += (uint64_t)GDIM * 256)
        expand_sa_1(a, L, t0, ns, base, x);
  }
}

// Sparse SA mode, whole words: every occurrence of a whole word carries its own char (bwlast), so any of the P positions
// they emit can be a run boundary.  One lane per occurrence e of the inverted lists laid end to end (ilist[1 + e], read
// coalesced): its word is the one of lexicographic rank q with istart_lex[q] <= e (bisection over d entries that stay in
// L2), the word's slot is wslot_lex[q], the position is that slot's offset plus the occurrence's index in its list.
// Replaces the second launch of expand_kernel over all N slots (every tile staged again to find its dozen whole words).
template <class I>
__global__ __launch_bounds__(256) void word_sa_kernel(MergeArgsT<I> a, uint64_t P, const uint64_t *__restrict__ wslot_lex,
                                                      uint64_t slot_base, const uint32_t *__restrict__ wlen_lex_word,
                                                      const uint32_t *__restrict__ word_len) {
  const uint64_t e = (uint64_t)BID * 256 + threadIdx.x;
  if (e >= P) return;
  uint32_t lo = 0, hi = a.d;                 // istart_lex[lo] <= e < istart_lex[hi]  (istart_lex[d] = P)
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if ((uint64_t)a.istart_lex[mid] <= e) lo = mid; else hi = mid; }
  const uint64_t gs = wslot_lex[lo];
  if (gs < slot_base || gs - slot_base >= a.N) return;      // (multi-GPU: the word's slot belongs to another rank's range)
  const uint64_t t = gs - slot_base;
  const uint64_t x = slot_off(a, t) + (e - (uint64_t)a.istart_lex[lo]);
  if (!sa_wanted(a, x)) return;
  const uint64_t pos = a.ilist[e + 1];       // +1: ilist[0] is the EOS symbol (pfbwt.cpp:389)
  (void)wlen_lex_word; (void)word_len;
  sa_put(a, x, a.pos_base + x == 0 ? a.n_out_global - 1 : a.bwsai[pos] - slot_slen(a, t));
}

// Sparse SA mode.  A *unit* is a slot that emits, or a group of slots of equal suffixes (several words share the
// suffix): its output range is one fill char c (unknown for hard groups), so a run of the BWT can start only at
// the unit's first position - when the unit before ends in a different char - and end only at its last position.
// The first position belongs to the smallest BWT(P) position over the members' inverted lists, the last one to the
// largest (WordRec::first / last per word): O(members) per unit instead of merging the lists (pfbwt.cpp:605-676 walks a
// heap through all of them).  One lane per slot: whether the two positions can be sampled from the boundary bitmap of
// the finished BWT, "needed" flags spread over a unit's lanes, segmented min / max by shuffles; a unit that crosses the
// wave is finished serially by its first (last) lane.
template <class I>
__global__ __launch_bounds__(256) void unit_edges_kernel(MergeArgsT<I> a) {
  const uint64_t t = (uint64_t)BID * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const uint64_t wbase = t - lane;
  const bool in = t < a.N;
  const uint32_t ch = in ? a.pc[t] : 0u;
  const bool unit = ch != 0 && ch != kEndOfWord;        // lanes that belong to a fill / hard unit (whole words: expand_kernel)
  uint64_t g = t;
  if (unit) g = a.grp[t];
  const unsigned long long um = __ballot(unit);
  // same unit as the next slot?
  const uint64_t g_next = __shfl_down(g, 1, 64);
  const bool unit_next = (um >> ((lane + 1) & 63)) & 1ull;
  bool cont;                                       // the unit goes on in slot t + 1
  if (lane < 63) cont = unit && unit_next && g_next == g;
  else cont = unit && t + 1 < a.N && a.pc[t + 1] != 0 && a.pc[t + 1] != kEndOfWord && a.grp[t + 1] == (I)g;
  const bool head = unit && g == t;
  const bool last = unit && !cont;
  // is the unit's first / last position one the sampled files can look at?
  uint64_t o_first = 0, o_last = 0;
  if (head) o_first = slot_off(a, t);
  if (last) o_last = slot_off(a, t + 1) - 1;
  const bool need_first = head && sa_wanted(a, o_first);
  const bool need_last = last && sa_wanted(a, o_last);
  // spread the two flags over the lanes of the unit that lie in this wave
  const bool head_here = unit && g >= wbase;
  const int hl = head_here ? (int)(g - wbase) : lane;
  const unsigned long long lastmask = __ballot(last);
  const unsigned long long lm = lastmask >> lane;
  const int ll = lm ? lane + (__ffsll((long long)lm) - 1) : lane;
  const bool nf = __shfl((int)need_first, hl, 64) != 0 && head_here;
  const bool nlz = __shfl((int)need_last, ll, 64) != 0 && unit && lm != 0;
  uint32_t mn = 0xFFFFFFFFu, mx = 0;
  uint64_t mysl = 0;      // length of the (common) suffix: distance to the word's terminator
  if (unit && (nf || nlz)) {
    if (nf) mn = slot_first(a, t);
    if (nlz) mx = slot_last(a, t);
    if (need_first || need_last) mysl = slot_slen(a, t);      // (equal suffixes: the same for every member; only the edge lanes store)
  }
  // segmented reductions: min towards the first lane of the unit, max towards its last lane
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t omn = __shfl_down(mn, o, 64);
    const uint64_t og = __shfl_down(g, o, 64);
    const int ou = __shfl_down((int)unit, o, 64);
    if (lane + o < 64 && unit && ou && og == g) mn = omn < mn ? omn : mn;
    const uint32_t umx = __shfl_up(mx, o, 64);
    const uint64_t ug = __shfl_up(g, o, 64);
    const int uu = __shfl_up((int)unit, o, 64);
    if (lane >= o && unit && uu && ug == g) mx = umx > mx ? umx : mx;
  }
  // At most one unit leaves the wave through lane 63 (its head is here and wants the minimum over ALL its members) and at
  // most one enters through lane 0 (its last member is here and wants the maximum): the members outside the wave are
  // fetched by all 64 lanes together, 64 at a time (one lane walking them was a chain of 3 dependent gathers per member,
  // hundreds of members long for the word families of a 1000-copy collection).
  const uint64_t g63 = __shfl(g, 63, 64);
  const bool cont63 = __shfl((int)cont, 63, 64) != 0;
  const unsigned long long fwd = __ballot(need_first && cont63 && g63 == g);       // the head lane of the unit that goes on
  if (fwd) {
    uint32_t ext = 0xFFFFFFFFu;
    for (uint64_t m0 = wbase + 64;; m0 += 64) {
      const uint64_t m = m0 + lane;
      const bool mine = m < a.N && a.grp[m] == (I)g63;       // (members are contiguous; a slot of the group emits)
      uint32_t f = 0xFFFFFFFFu;
      if (mine) f = slot_first(a, m);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { const uint32_t v = __shfl_xor(f, o, 64); f = v < f ? v : f; }
      ext = f < ext ? f : ext;
      if (__ballot(mine) != ~0ull) break;
    }
    if ((fwd >> lane) & 1ull) mn = ext < mn ? ext : mn;
  }
  const uint64_t g0 = __shfl(g, 0, 64);
  const int unit0 = __shfl((int)unit, 0, 64);
  const unsigned long long bwd = __ballot(need_last && g < wbase);                  // the last lane of the unit that came in
  if (bwd && unit0) {
    uint32_t ext = 0;
    for (uint64_t m0 = g0; m0 < wbase; m0 += 64) {
      const uint64_t m = m0 + lane;
      uint32_t l = 0;
      if (m < wbase) l = slot_last(a, m);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { const uint32_t v = __shfl_xor(l, o, 64); l = v > l ? v : l; }
      ext = l > ext ? l : ext;
    }
    if ((bwd >> lane) & 1ull) mx = ext > mx ? ext : mx;
  }
  if (need_first) sa_put(a, o_first, a.bwsai[mn] - mysl);
  if (need_last) sa_put(a, o_last, a.bwsai[mx] - mysl);
}

// Boundaries of the runs of the BWT slice [out_lo, out_hi): bit r of the map = position out_lo + r starts a run
// (differs from the byte before) or ends one (differs from the byte after); the slice's own first and last position
// always count - their neighbours belong to another rank.
constexpr int kRunTile = 4096;
__device__ __forceinline__ void run_mask16(const uint8_t *__restrict__ bwt, uint64_t base, uint64_t cnt, int left, int right,
                                           int run_end, uint32_t m[4]) {
  m[0] = m[1] = m[2] = m[3] = 0;
  if (base >= cnt) return;
  if (base + 17 <= cnt && base >= 1) {          // interior: both neighbours of all 16 positions are inside the slice
    const uint4 x = ld16u(bwt + base);
    const uint4 y = run_end ? ld16u(bwt + base + 1) : ld16u(bwt + base - 1);
    m[0] = nonzero_bytes(x.x ^ y.x); m[1] = nonzero_bytes(x.y ^ y.y); m[2] = nonzero_bytes(x.z ^ y.z); m[3] = nonzero_bytes(x.w ^ y.w);
    return;
  }
  for (int k = 0; k < 16; k++) {
    const uint64_t i = base + k;
    if (i >= cnt) break;
    const int b = bwt[i];
    int nb;
    if (run_end) nb = i + 1 < cnt ? (int)bwt[i + 1] : right;
    else nb = i ? (int)bwt[i - 1] : left;
    if (nb < 0 || nb != b) m[k >> 2] |= 1u << (8 * (k & 3));
  }
}
struct __attribute__((packed, aligned(1))) U16u { uint16_t v; };
__device__ __forceinline__ uint32_t pack_byte_flags(uint32_t m) {      // flags at bits 0, 8, 16, 24 -> bits 0..3
  return (m & 1u) | ((m >> 7) & 2u) | ((m >> 14) & 4u) | ((m >> 21) & 8u);
}
// one lane per 16 positions (one 16-byte load and its two shifted neighbours), four lanes make one word of the map;
// smap / emap (optional): the starts and the ends on their own, with their counts per word
__global__ __launch_bounds__(256) void run_bitmap_kernel(const uint8_t *__restrict__ bwt, uint64_t cnt, uint64_t *__restrict__ bmap,
                                                         uint32_t *__restrict__ wcnt, uint64_t *__restrict__ smap,
                                                         uint32_t *__restrict__ scnt, uint64_t *__restrict__ emap,
                                                         uint32_t *__restrict__ ecnt) {
  const uint64_t ch = (uint64_t)BID * 256 + threadIdx.x;
  const uint64_t base = ch * 16;
  uint32_t ms[4], me[4];
  run_mask16(bwt, base, cnt, -1, -1, 0, ms);
  run_mask16(bwt, base, cnt, -1, -1, 1, me);
  uint32_t bs = 0, be = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) { bs |= pack_byte_flags(ms[q]) << (4 * q); be |= pack_byte_flags(me[q]) << (4 * q); }
  uint64_t vs = (uint64_t)bs << (16 * (threadIdx.x & 3)), ve = (uint64_t)be << (16 * (threadIdx.x & 3));
  vs |= __shfl_xor(vs, 1, 64); vs |= __shfl_xor(vs, 2, 64);
  ve |= __shfl_xor(ve, 1, 64); ve |= __shfl_xor(ve, 2, 64);
  if ((threadIdx.x & 3) == 0 && base < cnt) {
    const uint64_t w = ch >> 2;
    bmap[w] = vs | ve; wcnt[w] = (uint32_t)__popcll(vs | ve);
    if (smap) { smap[w] = vs; scnt[w] = (uint32_t)__popcll(vs); }
    if (emap) { emap[w] = ve; ecnt[w] = (uint32_t)__popcll(ve); }
  }
}
// .ssa / .esa pairs of the whole BWT from the bitmaps: one thread per word of the start (end) map, the SA value of a
// set bit from its rank among all boundaries
__global__ __launch_bounds__(256) void bitmap_place_kernel(const uint64_t *__restrict__ map, const uint64_t *__restrict__ pre,
                                                           const uint64_t *__restrict__ bmap, const uint64_t *__restrict__ bpre,
                                                           const uint64_t *__restrict__ sa_c, uint64_t nw, uint8_t *__restrict__ out10,
                                                           uint64_t pos_base, int drop_first, uint64_t drop_pos) {
  const uint64_t w = (uint64_t)BID * 256 + threadIdx.x;
  if (w >= nw) return;
  uint64_t m = map[w];
  uint64_t o = pre[w];
  // a slice's edge that is no boundary after all (multi-GPU): position 0 leaves the list and every later pair moves
  // up by one; drop_pos (the slice's last position, or ~0) just leaves
  if (drop_first) { if (w == 0) m &= ~1ull; else o -= 1; }
  if ((drop_pos >> 6) == w) m &= ~(1ull << (drop_pos & 63));
  if (!m) return;
  const uint64_t all = bmap[w], rb = bpre[w];
  while (m) {
    const int b = __builtin_ctzll(m);
    m &= m - 1;
    const uint64_t x = pos_base + w * 64 + b, v = sa_c[rb + (uint64_t)__popcll(all & ((1ull << b) - 1ull))];
    uint8_t *dst = out10 + 10 * o++;
    reinterpret_cast<U64u *>(dst)->v = (x & 0xFFFFFFFFFFull) | (v << 40);       // 5 bytes of x, 3 low bytes of v
    reinterpret_cast<U16u *>(dst + 8)->v = (uint16_t)(v >> 24);                  // bytes 3, 4 of v
  }
}
__global__ void count_unset_kernel(const uint64_t *__restrict__ v, uint64_t n, unsigned long long *__restrict__ total) {
  const uint64_t i = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (i < n && v[i] == ~0ull) atomicAdd(total, 1ull);
}
// Hard groups, BWT-only and sparse-SA modes: majority fill + minority placement.  The members of a group of equal
// suffixes whose preceding chars disagree are, in a collection of similar sequences, one or a few words with most of
// the occurrences and one char (the base phrase and the variants that differ elsewhere) plus a handful of
// occurrences with another char (the variants that differ right here).  The occurrences of the majority char need no
// ranks: whatever their order, they write the same byte.  So the group's range is filled with the majority char (by
// expand_kernel, through gmaj[]) and only the minority occurrences are ranked - own index + lower_bound in every other
// member's inverted list, straight from global memory - and written over the fill.  In sparse SA mode a minority
// occurrence also gives the SA values of its two neighbours in the merged order (predecessor / successor of its
// BWT(P) position over all lists): those and the group's two ends are the only places a run can start or end.
// Groups where no char dominates and that fit the LDS kernels keep the old path (fallback list).
struct HardGroupInfo { uint64_t g; uint64_t E; uint32_t k; uint32_t minor; };
// member m (of k) of the hard group at head slot g does not carry the majority char; roff: index of its first occurrence among
// all minority occurrences (where hard_minor_kernel keeps its record for the SA round - no counter to contend for)
// base / ist / occ / sl / ch: the group's first output position and the member's own inverted list, char and suffix length, looked up
// once where the list is laid out (eight lanes per group) instead of at the head of every ranking wave's chain of dependent loads
struct MinorMember { uint64_t g; uint32_t k, m; uint64_t roff; uint64_t base; uint32_t ist, occ; uint32_t sl; uint32_t ch; };
// sparse SA: what the second pass needs of a placed minority occurrence - its output position, its BWT(P) position
// and those of its neighbours in the merged order (flags bit 0 / 1: there is a predecessor / successor), suffix length
struct MinorRec { uint64_t o; uint32_t pos, pred, succ, flags, sl, pad; };
// (One thread per group.  Eight lanes per group - members strided over the lanes, tables merged by a butterfly - was
//  tried in round 3 and ran twice as long, 10 -> 20 ms on 1024 copies: most groups have a handful of members, and the
//  chain of dependent loads that finds a group's extent is the same for eight lanes as for one.)
template <class I>
__global__ __launch_bounds__(256) void hard_classify_kernel(MergeArgsT<I> a, const I *__restrict__ heads, uint64_t nH,
                                                            uint8_t *__restrict__ gmaj, HardGroupInfo *__restrict__ info,
                                                            uint32_t *__restrict__ minor_cnt, uint8_t *__restrict__ fallback,
                                                            unsigned long long *__restrict__ chars_total, uint32_t *__restrict__ minor_members) {
  const uint64_t h = (uint64_t)BID * 256 + threadIdx.x;
  unsigned long long mychars = 0, myk = 0;
  if (h < nH) {
  const uint64_t g = heads[h];
  // members are the slots g .. g+k-1 (grp == g): gallop, then bisect
  uint64_t lo = 1, hi = 2;                   // offset lo is inside (a hard group has two members), hi will be outside
  while (g + hi < a.N && a.grp[g + hi] == (I)g) { lo = hi; hi *= 2; }
  if (g + hi > a.N) hi = a.N - g;            // slot N counts as outside
  while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (a.grp[g + mid] == (I)g) lo = mid; else hi = mid; }
  const uint64_t k = hi;                     // first offset outside = number of members
  const uint64_t base = slot_off(a, g);
  const uint64_t E = slot_off(a, g + k) - base;
  // occurrences per distinct char (up to 4 kept; more distinct chars: fallback)
  uint32_t c0 = 0x100, c1 = 0x100, c2 = 0x100, c3 = 0x100;
  uint64_t n0 = 0, n1 = 0, n2 = 0, n3 = 0;
  bool many = false;
  uint64_t prev = base;
  for (uint64_t m = 0; m < k; m++) {
    const uint64_t nxt = slot_off(a, g + m + 1);
    const uint64_t occ = nxt - prev;
    prev = nxt;
    const uint32_t c = fix_char(a.pc[g + m]);
    if (c == c0 || c0 == 0x100) { c0 = c; n0 += occ; }
    else if (c == c1 || c1 == 0x100) { c1 = c; n1 += occ; }
    else if (c == c2 || c2 == 0x100) { c2 = c; n2 += occ; }
    else if (c == c3 || c3 == 0x100) { c3 = c; n3 += occ; }
    else many = true;
  }
  uint32_t mc = c0; uint64_t mnn = n0;
  if (n1 > mnn) { mc = c1; mnn = n1; }
  if (n2 > mnn) { mc = c2; mnn = n2; }
  if (n3 > mnn) { mc = c3; mnn = n3; }
  const uint64_t minor = E - mnn;
  uint32_t mm = 0;                               // members that do not carry the majority char, and the longest of their lists
  uint64_t mm_max = 0;
  prev = base;
  for (uint64_t m = 0; m < k; m++) {
    const uint64_t nxt = slot_off(a, g + m + 1);
    if ((uint32_t)fix_char(a.pc[g + m]) != mc) { mm++; mm_max = nxt - prev > mm_max ? nxt - prev : mm_max; }
    prev = nxt;
  }
  // fallback (the kernels that rank every occurrence): no dominating char, a long minority list (its occurrences
  // would be ranked one after the other), too many distinct chars
  // ... or a minority that is cheap per occurrence but meets too many members: every minority occurrence is ranked against every
  // member of the group (hard_minor_kernel: minor x k list checks), which is nothing for a phrase and its variants and quadratic
  // for the words of a satellite array - 17 K near-identical monomers, 2 % of them with another preceding char: 0.02 k^2 = 6 M
  // checks per group where ranking all k occurrences by sorting costs k log k (round 4, the c2r workload: 200 ms -> 3 ms)
  const bool fb = many || minor * 4 > E || mm_max > 1024 || minor * k > 16 * E + 4096;
  const bool in_slice = !(base + E <= a.out_lo || base >= a.out_hi);
  fallback[h] = (fb && in_slice) ? 1 : 0;
  gmaj[g] = fb ? 0 : (uint8_t)mc;
  minor_cnt[h] = (fb || !in_slice) ? 0u : (uint32_t)minor;
  minor_members[h] = (fb || !in_slice) ? 0u : mm;
  info[h] = HardGroupInfo{g, E, (uint32_t)k, (uint32_t)(fb ? 0 : minor)};
  if (!fb && in_slice) { mychars = E; myk = k; }
  }
  for (int o = 32; o > 0; o >>= 1) { mychars += __shfl_down(mychars, o, 64); myk += __shfl_down(myk, o, 64); }      // "Hard bwt chars" (pfbwt.cpp:231-233) of these groups
  if ((threadIdx.x & 63) == 0 && mychars) { atomicAdd(chars_total, mychars); atomicAdd(chars_total + 1, myk); }      // [1]: members, for the lane choice of hard_minor_kernel
}

// the minority members of all groups, laid end to end (offsets = prefix sums of their count per group); eight lanes per
// group, eight members per step, order and record offsets from prefix sums over the lanes
template <class I>
__global__ __launch_bounds__(256) void hard_minor_fill_kernel(MergeArgsT<I> a, const HardGroupInfo *__restrict__ info, uint64_t nH,
                                                              const uint64_t *__restrict__ mm_off, const uint64_t *__restrict__ minor_off,
                                                              const uint8_t *__restrict__ gmaj, MinorMember *__restrict__ out) {
  const uint64_t tq = (uint64_t)BID * 256 + threadIdx.x;
  const uint64_t h = tq >> 3;
  const int l8 = (int)(tq & 7);
  if (h >= nH) return;
  uint64_t o = mm_off[h];
  if (mm_off[h + 1] == o) return;
  const HardGroupInfo gi = info[h];
  const uint32_t maj = gmaj[gi.g];
  uint64_t r = minor_off[h];
  const uint64_t gbase = slot_off(a, gi.g);
  for (uint32_t m0 = 0; m0 < gi.k; m0 += 8) {
    const uint32_t m = m0 + (uint32_t)l8;
    const bool is_minor = m < gi.k && (uint32_t)fix_char(a.pc[gi.g + m]) != maj;
    uint32_t cnt = is_minor ? 1u : 0u;
    uint64_t oc = is_minor ? slot_off(a, gi.g + m + 1) - slot_off(a, gi.g + m) : 0ull;
    const uint32_t own_c = cnt; const uint64_t own_o = oc;
#pragma unroll
    for (int d2 = 1; d2 < 8; d2 <<= 1) {
      const uint32_t vc = __shfl_up(cnt, d2, 8); const uint64_t vo = __shfl_up(oc, d2, 8);
      if (l8 >= d2) { cnt += vc; oc += vo; }
    }
    if (is_minor) {
      const uint64_t myi = a.sa[gi.g + m];
      const WordRec wr = a.wrec[word_of(a.wv, myi)];
      out[o + cnt - own_c] = MinorMember{gi.g, gi.k, m, r + oc - own_o, gbase, wr.ist, wr.occ, (uint32_t)(wr.wend - myi),
                                         (uint32_t)fix_char(a.pc[gi.g + m])};
    }
    o += __shfl(cnt, 7, 8); r += __shfl(oc, 7, 8);
  }
}
// LPM lanes per minority member (8, or the whole wave where groups have many members - the word families of a
// collection of hundreds of copies): its occurrences (usually one) are ranked one after the other - own index +
// lower_bound in every other member's inverted list, the group's members shared among the lanes - rank, predecessor
// and successor combined by shuffles.  What a member's list looks like from outside (smallest / largest position,
// length) comes from the per-slot arrays, LPM consecutive slots per load.
template <class I, int LPM>
__global__ __launch_bounds__(256) void hard_minor_kernel(MergeArgsT<I> a, const MinorMember *__restrict__ mem, uint64_t total,
                                                         MinorRec *__restrict__ recs) {
  constexpr int PER = 256 / LPM;             // members per workgroup and round
  const int ll = threadIdx.x & (LPM - 1);
  const uint64_t rounds = (total + GDIM * PER - 1) / (GDIM * PER);      // every lane runs the same number of rounds (shuffles inside)
  for (uint64_t it = 0; it < rounds; it++) {
    const uint64_t q = (it * GDIM + BID) * PER + (threadIdx.x / LPM);
    const bool live = q < total;
    MinorMember mmv{0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (live) mmv = mem[q];
    const uint64_t g = mmv.g;
    const uint32_t k = mmv.k, me = mmv.m;
    const uint32_t my_occ = mmv.occ, my_ist = mmv.ist;
    const uint64_t base = mmv.base, sl = mmv.sl;          // (equal suffixes: the same length for every member)
    const uint8_t mych = (uint8_t)mmv.ch;
    // longest occurrence count among the members of the wave decides the trip count (shuffles inside the loop)
    uint32_t trips = my_occ;
#pragma unroll
    for (int o = LPM; o < 64; o <<= 1) { const uint32_t v = __shfl_xor(trips, o, 64); trips = v > trips ? v : trips; }
    for (uint32_t j = 0; j < trips; j++) {
      const bool act = live && j < my_occ;
      uint64_t r = 0;
      uint32_t pos = 0, pred = 0, succ = 0xFFFFFFFFu, has_pred = 0, has_succ = 0;
      if (act) {
        pos = a.ilist[my_ist + j];
        for (uint32_t m = (uint32_t)ll; m < k; m += LPM) {
          uint32_t lb, pv = 0, sv = 0;
          bool hp = false, hs = false;
          if (m == me) {
            lb = j;
            if (a.want_sa) {
              if (j > 0) { pv = a.ilist[my_ist + j - 1]; hp = true; }
              if (j + 1 < my_occ) { sv = a.ilist[my_ist + j + 1]; hs = true; }
            }
          } else {
            // a list that lies wholly on one side of pos (a variant that occurs once: always) needs no look inside;
            // only a list that straddles pos is looked up and bisected
            const uint32_t mfirst = slot_first(a, g + m), mlast = slot_last(a, g + m);
            if (pos < mfirst) { lb = 0; sv = mfirst; hs = true; }
            else if (pos > mlast) { lb = (uint32_t)(slot_off(a, g + m + 1) - slot_off(a, g + m)); pv = mlast; hp = true; }
            else {
              const WordRec wr = a.wrec[word_of(a.wv, a.sa[g + m])];
              const uint32_t *lst = a.ilist + wr.ist;
              uint32_t l2 = 1, h2 = wr.occ - 1;          // # entries < pos: lst[0] < pos < lst[occ - 1]
              while (l2 < h2) { const uint32_t mid = (l2 + h2) >> 1; if (lst[mid] < pos) l2 = mid + 1; else h2 = mid; }
              lb = l2;
              if (a.want_sa) { pv = lst[lb - 1]; hp = true; sv = lst[lb]; hs = true; }
            }
          }
          r += lb;
          if (hp && (!has_pred || pv > pred)) { pred = pv; has_pred = 1; }
          if (hs && (!has_succ || sv < succ)) { succ = sv; has_succ = 1; }
        }
      }
#pragma unroll
      for (int o = 1; o < LPM; o <<= 1) {
        r += __shfl_xor(r, o, 64);
        const uint32_t op = __shfl_xor(pred, o, 64), ohp = __shfl_xor(has_pred, o, 64);
        const uint32_t os = __shfl_xor(succ, o, 64), ohs = __shfl_xor(has_succ, o, 64);
        if (ohp && (!has_pred || op > pred)) { pred = op; has_pred = 1; }
        if (ohs && (!has_succ || os < succ)) { succ = os; has_succ = 1; }
      }
      if (!act || ll != 0) continue;
      const uint64_t o = base + r;
      if (o >= a.out_lo && o < a.out_hi) a.bwt[o] = mych;
      if (a.want_sa)       // (also when the occurrence lies just outside this rank's slice: its neighbours may be inside)
        recs[mmv.roff + j] = MinorRec{o, pos, pred, succ, has_pred | (has_succ << 1), (uint32_t)sl, 0u};
    }
  }
}

template <class I>
__global__ void hard_minor_sa_kernel(MergeArgsT<I> a, const MinorRec *__restrict__ recs, uint64_t n) {
  const uint64_t q = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const MinorRec r = recs[q];
  sa_put(a, r.o, a.bwsai[r.pos] - (uint64_t)r.sl);
  if ((r.flags & 1u) && r.o >= 1) sa_put(a, r.o - 1, a.bwsai[r.pred] - (uint64_t)r.sl);
  if (r.flags & 2u) sa_put(a, r.o + 1, a.bwsai[r.succ] - (uint64_t)r.sl);
}

// Hard groups.  hard[] is 1 exactly at the head slot of every hard group; the heads are compacted
// into a list and every wave takes 64 of them at a time, one per lane.  A single group is a chain
// of dependent loads (members, offsets, inverted-list starts, positions), so the work of a batch
// is laid out flat and every phase runs with all lanes busy on independent loads:
//   A  lane = group      : member count, output base, number of occurrences E
//   B  lane = member     : offset of its occurrences, inverted-list start, char, suffix length
//   C  lane = occurrence : its BWT(P) position from the inverted list  -> LDS
//   D  lane = occurrence : rank among the E positions of its own group -> output slot
// Every occurrence (member, j) must land at the rank of its BWT(P) position among all occurrences
// of the group - the reference pops a heap (pfbwt.cpp:537-556).  Ranking is all-pairs over the
// group's LDS segment, or, when the group is a few long sorted lists, own index + lower_bound in
// each other member's segment.  Groups that do not fit the LDS tables are queued for
// hard_big_kernel.  (One group at a time per wave took 7.7 us per group, 22 ms at 8.8 M groups.)
struct BigGroup { uint64_t g; uint64_t E; uint32_t k; uint32_t pad; };
struct HardLds {
  uint32_t lpos[kHardRank];
  uint8_t lq[kHardRank];
  uint32_t lmoff[kHardMem + 1], lmist[kHardMem], lmsl[kHardMem];
  uint8_t lmch[kHardMem], lmg[kHardMem];
  uint64_t gbase[64], ghead[64];
  uint32_t geoff[65], gk0[65];
};
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const uint32_t x = __shfl_xor(v, o, 64); v = x > v ? x : v; }
  return v;
}
template <class I>
__global__ __launch_bounds__(256) void hard_groups_kernel(MergeArgsT<I> a, const I *__restrict__ heads,
                                                          const uint64_t *__restrict__ nheads_p,
                                                          unsigned long long *__restrict__ stats,
                                                          BigGroup *__restrict__ big, uint32_t big_cap,
                                                          BigGroup *__restrict__ mid, uint32_t mid_cap, int gpb) {
  __shared__ HardLds S[4];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  HardLds &L = S[wv];
  const uint64_t nH = *nheads_p;
  // gpb groups per wave and batch: 64 when there are groups enough to keep every wave of the chip busy, fewer
  // otherwise - a batch is one long chain of dependent phases, so the chip wants many of them in flight
  const uint64_t nbatch = (nH + gpb - 1) / gpb;
  unsigned long long my_chars = 0, my_groups = 0;
  for (uint64_t b = BID * 4 + wv; b < nbatch; b += GDIM * 4) {
    // ---- A: one group per lane
    const uint64_t hidx = b * gpb + lane;
    uint64_t g = 0, base = 0;
    uint32_t k = 0, E = 0;
    bool live = false;
    if (lane < gpb && hidx < nH) {
      g = heads[hidx];
      uint32_t kk = 0;
      while (g + kk < a.N && a.grp[g + kk] == (I)g && a.pc[g + kk] != 0) kk++;
      base = slot_off(a, g);
      const uint64_t Eg = slot_off(a, g + kk) - base;
      if (kk && !(base + Eg <= a.out_lo || base >= a.out_hi)) {      // inside this rank's slice
        my_chars += Eg; my_groups += 1;
        const bool sorted_path = Eg > (uint64_t)kHardSortMin;
        if (Eg <= (uint64_t)kHardRank && kk <= (uint32_t)kHardMem && !sorted_path) { live = true; k = kk; E = (uint32_t)Eg; }
        else if (Eg <= (uint64_t)kHardLds && sorted_path) {      // one wave sorts it in LDS: hard_sort_kernel
          const unsigned long long idx = atomicAdd(&stats[3], 1ull);
          if (idx < mid_cap) mid[idx] = BigGroup{g, Eg, kk, 0};
        } else {     // too large for the LDS tables: hard_big_kernel (whole grid, one thread per occurrence)
          const unsigned long long idx = atomicAdd(&stats[2], 1ull);
          if (idx < big_cap) big[idx] = BigGroup{g, Eg, kk, 0};
        }
      }
    }
    unsigned long long todo = __ballot(live);
    while (todo) {
      // the longest prefix of the remaining groups whose members and occurrences fit the tables
      const bool mine = (todo >> lane) & 1ull;
      uint32_t pk = mine ? k : 0u, pE = mine ? E : 0u;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t vk = __shfl_up(pk, o, 64), vE = __shfl_up(pE, o, 64);
        if (lane >= o) { pk += vk; pE += vE; }
      }
      const bool fit = mine && pk <= (uint32_t)kHardMem && pE <= (uint32_t)kHardRank;
      const unsigned long long take = __ballot(fit);       // never empty: a single live group fits
      todo &= ~take;
      const int nG = __popcll(take);
      const int lastl = 63 - __clzll((long long)take);
      const uint32_t M = __shfl(pk, lastl, 64), Eb = __shfl(pE, lastl, 64);
      if (fit) {
        const int gi = __popcll(take & ((1ull << lane) - 1ull));
        L.ghead[gi] = g; L.gbase[gi] = base; L.geoff[gi] = pE - E; L.gk0[gi] = pk - k;
      }
      if (lane == 0) { L.geoff[nG] = Eb; L.gk0[nG] = M; }
      wave_lds_sync();
      // ---- B: one member per lane
      for (uint32_t q = lane; q < M; q += 64) {
        int lo = 0, hi = nG;                       // gk0[lo] <= q < gk0[hi]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (L.gk0[mid] <= q) lo = mid; else hi = mid; }
        const uint64_t t = L.ghead[lo] + (q - L.gk0[lo]);
        L.lmoff[q] = L.geoff[lo] + (uint32_t)(slot_off(a, t) - L.gbase[lo]);
        L.lmist[q] = slot_ist(a, t);
        uint32_t sl = 0;
        if (a.want_sa) sl = (uint32_t)slot_slen(a, t);
        L.lmsl[q] = sl;
        L.lmch[q] = fix_char(a.pc[t]);
        L.lmg[q] = (uint8_t)lo;
      }
      if (lane == 0) L.lmoff[M] = Eb;
      wave_lds_sync();
      // ---- C: one occurrence per lane, independent inverted-list gathers
      for (uint32_t e = lane; e < Eb; e += 64) {
        uint32_t lo = 0, hi = M;                   // lmoff[lo] <= e < lmoff[hi]
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (L.lmoff[mid] <= e) lo = mid; else hi = mid; }
        L.lpos[e] = a.ilist[L.lmist[lo] + (e - L.lmoff[lo])];
        L.lq[e] = (uint8_t)lo;
      }
      wave_lds_sync();
      // ---- D: rank inside the own group's segment
      for (uint32_t e = lane; e < Eb; e += 64) {
        const uint32_t q = L.lq[e], gi = L.lmg[q];
        const uint32_t s0 = L.geoff[gi], s1 = L.geoff[gi + 1], k0 = L.gk0[gi], k1 = L.gk0[gi + 1];
        const uint32_t Eg = s1 - s0, kg = k1 - k0;
        const uint32_t pos = L.lpos[e];
        uint32_t lg = 1;
        while ((kg << lg) < Eg) lg++;
        const bool by_search = (uint64_t)kg * (lg + 1) * 3 < Eg;
        uint32_t r = 0;
        if (!by_search) {
          for (uint32_t x = s0; x < s1; x++) r += L.lpos[x] < pos;
        } else {
          for (uint32_t m2 = k0; m2 < k1; m2++) {
            uint32_t l2 = L.lmoff[m2], h2 = L.lmoff[m2 + 1];       // # entries < pos in member m2's segment
            const uint32_t b2 = l2;
            while (l2 < h2) { const uint32_t mid = (l2 + h2) >> 1; if (L.lpos[mid] < pos) l2 = mid + 1; else h2 = mid; }
            r += l2 - b2;
          }
        }
        const uint64_t o = L.gbase[gi] + r;
        if (o >= a.out_lo && o < a.out_hi) {
          if (a.pass & PASS_BWT) a.bwt[o] = L.lmch[q];
          if (a.want_sa && (a.pass & PASS_SA)) sa_put(a, o, a.bwsai[pos] - (uint64_t)L.lmsl[q]);
        }
      }
      wave_lds_sync();
    }
  }
  // statistics (pfbwt.cpp:231-233 "Hard bwt chars")
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { my_chars += __shfl_down(my_chars, o, 64); my_groups += __shfl_down(my_groups, o, 64); }
  if (lane == 0 && (my_chars | my_groups)) { atomicAdd(&stats[0], my_chars); atomicAdd(&stats[1], my_groups); }
}

// Groups of 513..1024 occurrences (a word family of a collection of hundreds of copies): ranking every
// occurrence against all others costs E^2 LDS reads; one wave sorts the group's positions with a bitonic
// network instead (E log^2 E / 2 exchanges; 12.6 GB / 1024 copies: 374 -> ~250 ms).  Kept out of
// hard_groups_kernel so that its 10 KB of sort buffers per wave do not halve that kernel's occupancy
// (they did: 11 -> 20 ms on the 64-copy workload).  The member search is by position here (a member table
// would not fit for k > 256): every occurrence finds its member by bisection over the slots' offsets.
template <class I>
__global__ __launch_bounds__(256) void hard_sort_kernel(MergeArgsT<I> a, const BigGroup *__restrict__ mid, uint32_t nmid) {
  __shared__ uint64_t skey[4][kHardLds];       // (position << 16 | occurrence index)
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint64_t *K = skey[wv];
  for (uint32_t qi = BID * 4 + wv; qi < nmid; qi += GDIM * 4) {
    const uint64_t g = mid[qi].g;
    const uint32_t E = (uint32_t)mid[qi].E, k = mid[qi].k;
    const uint64_t base = slot_off(a, g);
    uint32_t n = 64;
    while (n < E) n <<= 1;
    for (uint32_t e = lane; e < n; e += 64) {
      uint64_t key = ~0ull;
      if (e < E) {
        uint32_t lo = 0, hi = k;                 // member holding occurrence e: off[g+lo] - base <= e
        while (hi - lo > 1) { const uint32_t mdl = (lo + hi) >> 1; if (slot_off(a, g + mdl) - base <= e) lo = mdl; else hi = mdl; }
        const uint32_t pos = a.ilist[slot_ist(a, g + lo) + (e - (uint32_t)(slot_off(a, g + lo) - base))];
        key = ((uint64_t)pos << 16) | e;
      }
      K[e] = key;
    }
    wave_lds_sync();
    for (uint32_t k2 = 2; k2 <= n; k2 <<= 1)
      for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
        for (uint32_t i = lane; i < (n >> 1); i += 64) {
          const uint32_t x = ((i & ~(j - 1)) << 1) | (i & (j - 1)), y = x | j;
          const uint64_t kx = K[x], ky = K[y];
          if ((kx > ky) == ((x & k2) == 0)) { K[x] = ky; K[y] = kx; }
        }
        wave_lds_sync();
      }
    for (uint32_t r = lane; r < E; r += 64) {
      const uint64_t key = K[r];
      const uint32_t pos = (uint32_t)(key >> 16), e = (uint32_t)(key & 0xffffu);
      const uint64_t o = base + r;
      if (o < a.out_lo || o >= a.out_hi) continue;
      uint32_t lo = 0, hi = k;
      while (hi - lo > 1) { const uint32_t mdl = (lo + hi) >> 1; if (slot_off(a, g + mdl) - base <= e) lo = mdl; else hi = mdl; }
      const uint64_t t = g + lo;
      if (a.pass & PASS_BWT) a.bwt[o] = fix_char(a.pc[t]);
      if (a.want_sa && (a.pass & PASS_SA)) sa_put(a, o, a.bwsai[pos] - slot_slen(a, t));
    }
    wave_lds_sync();
  }
}

// large hard groups: one thread per occurrence, rank = own index + lower_bound in every other
// member's inverted list
template <class I>
__global__ __launch_bounds__(256) void hard_big_kernel(MergeArgsT<I> a, const BigGroup *__restrict__ big, uint32_t nbig,
                                                       const uint64_t *__restrict__ estart, uint64_t total) {
  const uint64_t e0 = estart[0];              // (the queue's offsets are absolute: this launch may start in its middle)
  for (uint64_t gi = (uint64_t)BID * 256 + threadIdx.x; gi < total; gi += (uint64_t)GDIM * 256) {
    const uint64_t ge = e0 + gi;
    uint32_t lo = 0, hi = nbig;               // estart[lo] <= ge < estart[hi]
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (estart[mid] <= ge) lo = mid; else hi = mid; }
    const uint64_t g = big[lo].g, e = ge - estart[lo];
    const uint32_t k = big[lo].k;
    const uint64_t base = slot_off(a, g);
    uint32_t ml = 0, mh = k;                  // member holding occurrence e
    while (mh - ml > 1) { uint32_t mid = (ml + mh) >> 1; if (slot_off(a, g + mid) - base <= e) ml = mid; else mh = mid; }
    const uint64_t t = g + ml;
    const uint32_t j = (uint32_t)(e - (slot_off(a, t) - base));
    const uint32_t pos = a.ilist[slot_ist(a, t) + j];
    uint64_t r = j;
    for (uint64_t t2 = g; t2 < g + k; t2++) {
      if (t2 == t) continue;
      const uint32_t *lst = a.ilist + slot_ist(a, t2);
      uint32_t l2 = 0, h2 = (uint32_t)(slot_off(a, t2 + 1) - slot_off(a, t2));      // # entries < pos
      while (l2 < h2) { uint32_t mid = (l2 + h2) >> 1; if (lst[mid] < pos) l2 = mid + 1; else h2 = mid; }
      r += l2;
    }
    if (base + r >= a.out_lo && base + r < a.out_hi) {
      if (a.pass & PASS_BWT) a.bwt[base + r] = fix_char(a.pc[t]);
      if (a.want_sa && (a.pass & PASS_SA)) sa_put(a, base + r, a.bwsai[pos] - slot_slen(a, t));
    }
  }
}

// Large hard groups, the usual way: the occurrences of a chunk of queued groups become keys (group, BWT(P) position)
// with the member's slot as value, one device-wide radix sort merges every group's lists at once (the reference pops a
// heap, pfbwt.cpp:537-556), and position i of a group's sorted range is its i-th output.  Ranking every occurrence
// against every member's list (hard_big_kernel) is E k log(E / k) dependent loads; on 1024 copies of a genome with
// repeats - k words of ~1000 occurrences each - that was the longest kernel of the chain.
template <class I>
__global__ __launch_bounds__(256) void big_keys_kernel(MergeArgsT<I> a, const BigGroup *__restrict__ big, uint32_t q0, uint32_t q1,
                                                       const uint64_t *__restrict__ estart, uint64_t cnt,
                                                       uint64_t *__restrict__ keys, uint64_t *__restrict__ vals) {
  const uint64_t i = (uint64_t)BID * 256 + threadIdx.x;
  if (i >= cnt) return;
  const uint64_t ge = estart[q0] + i;
  uint32_t lo = q0, hi = q1;                // estart[lo] <= ge < estart[hi]
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (estart[mid] <= ge) lo = mid; else hi = mid; }
  const uint64_t g = big[lo].g, e = ge - estart[lo];
  const uint32_t k = big[lo].k;
  const uint64_t base = slot_off(a, g);
  uint32_t ml = 0, mh = k;                  // member holding occurrence e
  while (mh - ml > 1) { const uint32_t mid = (ml + mh) >> 1; if (slot_off(a, g + mid) - base <= e) ml = mid; else mh = mid; }
  const uint64_t t = g + ml;
  const uint32_t j = (uint32_t)(e - (slot_off(a, t) - base));
  keys[i] = ((uint64_t)(lo - q0) << 32) | a.ilist[slot_ist(a, t) + j];
  vals[i] = t;
}
template <class I>
__global__ __launch_bounds__(256) void big_place_kernel(MergeArgsT<I> a, const BigGroup *__restrict__ big, uint32_t q0,
                                                        const uint64_t *__restrict__ estart, uint64_t cnt,
                                                        const uint64_t *__restrict__ keys, const uint64_t *__restrict__ vals) {
  const uint64_t i = (uint64_t)BID * 256 + threadIdx.x;
  if (i >= cnt) return;
  const uint64_t key = keys[i], t = vals[i];
  const uint32_t lo = q0 + (uint32_t)(key >> 32), pos = (uint32_t)key;
  const uint64_t o = slot_off(a, big[lo].g) + (estart[q0] + i - estart[lo]);
  if (o < a.out_lo || o >= a.out_hi) return;
  if (a.pass & PASS_BWT) a.bwt[o] = fix_char(a.pc[t]);
  if (a.want_sa && (a.pass & PASS_SA) && sa_wanted(a, o)) sa_put(a, o, a.bwsai[pos] - slot_slen(a, t));
}

// loc[t] = sum of cnt over the slots of t's tile before t; tsum[tile] = the tile's total.  256 threads x 8 slots.
__global__ __launch_bounds__(256) void slot_loc_kernel(const uint32_t *__restrict__ cnt, uint64_t N, uint32_t *__restrict__ loc,
                                                       uint64_t *__restrict__ tsum, uint32_t *__restrict__ overflow) {
  __shared__ uint64_t ws[4];
  if (((uint64_t)BID << kOffTileLog) > N) return;      // tiles 0 .. N >> 11 exist (a workgroup of the padded last grid row)
  const uint64_t t0 = ((uint64_t)BID << kOffTileLog) + (uint64_t)threadIdx.x * 8;
  uint32_t v[8];
  if (t0 + 8 <= N) {
    const uint4 x = *reinterpret_cast<const uint4 *>(cnt + t0), y = *reinterpret_cast<const uint4 *>(cnt + t0 + 4);
    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
  } else {
    for (int k = 0; k < 8; k++) v[k] = t0 + k < N ? cnt[t0 + k] : 0u;
  }
  uint64_t own = 0;
  for (int k = 0; k < 8; k++) own += v[k];
  uint64_t inc = own;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int o = 1; o < 64; o <<= 1) { const uint64_t u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
  if (lane == 63) ws[wv] = inc;
  __syncthreads();
  uint64_t run = inc - own;
  for (int q = 0; q < wv; q++) run += ws[q];
  uint32_t o8[8];
  for (int k = 0; k < 8; k++) { o8[k] = (uint32_t)run; run += v[k]; }
  *reinterpret_cast<uint4 *>(loc + t0) = make_uint4(o8[0], o8[1], o8[2], o8[3]);
  *reinterpret_cast<uint4 *>(loc + t0 + 4) = make_uint4(o8[4], o8[5], o8[6], o8[7]);
  if (threadIdx.x == 255) {
    tsum[BID] = run;
    if (run >> 32) atomicOr(overflow, 1u);      // a tile's offsets would not fit 32 bits
  }
}

template <class I>
void merge_bwt(pfp_ctx *c, const Dictionary &D, const DictIndex &ix, const SuffixOrderT<I> &so, const ParseBWT &pb,
               const uint32_t *occ_lex, int w, int flags, uint64_t expect_n_out, BwtOutputs &out, uint64_t out_lo,
               uint64_t out_hi, uint64_t pos_base, uint64_t n_out_global) {
  // NP dictionary positions; N suffix-array slots held by `so` (all of them, or - multi-GPU, key-range
  // sharded sort - one contiguous range of SA(D) whose first output position is pos_base)
  const uint64_t NP = D.dsize;
  const uint64_t N = so.N;
  const uint32_t d = (uint32_t)D.d;
  PFP_REQUIRE(!flags || pb.bwsai.p, PFP_EINVAL, "SA output requested without sa info");
  // istart in lexicographic order (pfbwt.cpp:388-396), looked up per word
  DBuf<uint32_t> istart_lex(c, d), wistart(c, d);
  exclusive_sum_u32(c, occ_lex, istart_lex.p, d);
  // SA values: none, all (-S), or only where the sampled files can look (-s / -e): run boundaries of the BWT
  const int sa_mode = !flags ? SA_NONE : ((flags & PFP_FLAG_SA) ? SA_DENSE : SA_SPARSE);
  const int samode = sa_mode;
  const bool dense = samode == SA_DENSE;
  DBuf<WordRec> wrec(c, d);      // per word: list start, occurrences, smallest / largest BWT(P) position, terminator
  hipLaunchKernelGGL(wistart_kernel, gdim(cdiv(d, TB)), gdim(TB), 0, c->stream, d, ix.lexrank.p, istart_lex.p, D.wocc.p, pb.ilist.p,
                     ix.wend.p, wistart.p, wrec.p);
  const WordView wv = word_view(D, ix);
  DBuf<PosRec> prec;
  DBuf<uint32_t> cnt, sfirst, slast, ssl;
  // records already sit at their slots - unless most slots were re-ordered after the first round (a
  // dictionary of near-identical variants), where fetching each such record costs more than the gather
  const bool from_keys = !dense && so.paybits == 16 && so.skeys.p && so.n_refined * 5 < N;
  // everything else: ONE 16-byte gather per slot (slot_records_kernel); full SA output included - its per-slot inverted-list
  // start travels in the record's `first` field
  const bool fused = !from_keys;
  if (!fused) { cnt.alloc(c, N + 8); PFP_HIP(hipMemsetAsync(cnt.p + N, 0, 4, c->stream)); }
  DBuf<uint8_t> pc(c, N + 8), hard(c, N);
  const uint64_t ntile = (N >> kOffTileLog) + 1;        // covers slot index N (one past the last)
  DBuf<uint32_t> loc(c, ntile << kOffTileLog), ovf(c, 1);
  DBuf<uint64_t> tsum(c, ntile + 1), tbase(c, ntile + 1);
  hard.zero();
  DBuf<uint32_t> tile_full(c, ntile + 1), fullbase(c, ntile + 1), fullbase0(c, 1);
  DBuf<unsigned long long> first_full(c, 1);
  tile_full.zero();
  PFP_HIP(hipMemsetAsync(first_full.p, 0xFF, 8, c->stream));
  ovf.zero();
  PFP_HIP(hipMemsetAsync(tsum.p + ntile, 0, 8, c->stream));
  if (fused) {
    // 16 bytes per dictionary position, written by one streaming pass and gathered once per slot - unless the slots are a
    // small share of the positions (a rank's range of the multi-GPU chain: the pass over all NP positions is replicated
    // work, 14 GB of records for a 0.9 GB dictionary) or the records would take more than an eighth of the device
    // (a 30 GB dictionary: 480 GB): then every slot computes its record itself from the dictionary line and its word's
    // entry (posrec_at: two random sectors per slot instead of one).  PFP_PREC_DIRECT=1/0 forces the choice (tests).
    bool direct = N * 4 <= NP;
    {
      const uint64_t dev_bytes = c->pool.soft_limit ? c->pool.soft_limit / 7 * 10 : (64ull << 30);
      if (NP * sizeof(PosRec) > dev_bytes / 8) direct = true;
      const char *e = getenv("PFP_PREC_DIRECT");
      if (e) direct = atoi(e) != 0;
    }
    if (!direct) {
      const uint64_t np256 = cdiv64(NP, 256) * 256;
      prec.alloc(c, np256);
      KScope ks(c, "pfp::pprec16_kernel", NP * (1 + 16) + (uint64_t)d * 40);
      hipLaunchKernelGGL(pprec16_kernel, gdim(cdiv(np256, TB)), gdim(TB), 0, c->stream, wv, w, reinterpret_cast<const uint4 *>(wrec.p),
                         dense ? 1 : 0, (uint64_t)0, NP, prec.p);
    }
    // per-slot copies of what the unit-edge / minority kernels ask about a slot's word (read coalesced there); full SA:
    // the list start of every slot's word and its suffix length
    sfirst.alloc(c, ntile << kOffTileLog);
    if (!dense) slast.alloc(c, ntile << kOffTileLog);
    if (samode != SA_NONE) ssl.alloc(c, ntile << kOffTileLog);
    { KScope ks(c, "pfp::slot_records_kernel", N * (2 * sizeof(I) + (direct ? 64 + 32 : 16) + 4 + 1 + 4 + (slast.p ? 4 : 0) + (ssl.p ? 4 : 0)));
      hipLaunchKernelGGL(slot_records_kernel<I>, gdim((unsigned)ntile), gdim(256), 0, c->stream, N, so.sa.p, so.grp.p,
                         direct ? (const PosRec *)nullptr : prec.p, loc.p, pc.p, sfirst.p, slast.p, ssl.p, tsum.p, ovf.p, tile_full.p,
                         first_full.p, hard.p, dense ? 1 : 0, wv, w, reinterpret_cast<const uint4 *>(wrec.p), dense ? 1 : 0); }
    prec.release();
  } else {
    { KScope ks(c, "pfp::slot_payload_kernel", N * (8 + 1 + 5));
      hipLaunchKernelGGL(slot_payload_kernel<I>, gdim(cdiv(cdiv64(N, 8), 256)), gdim(256), 0, c->stream, N, so.sa.p, so.skeys.p,
                         so.refined.p, D.bytes.p, wv, D.wocc.p, d, w, cnt.p, pc.p, tile_full.p, first_full.p); }
    { KScope ks(c, "pfp::slot_loc_kernel", N * 8);
      hipLaunchKernelGGL(slot_loc_kernel, gdim((unsigned)ntile), gdim(256), 0, c->stream, cnt.p, N, loc.p, tsum.p, ovf.p); }
    { KScope ks(c, "pfp::group_flags_kernel", N * 5);
      hipLaunchKernelGGL(group_flags_kernel<I>, gdim(cdiv(N, TB)), gdim(TB), 0, c->stream, N, so.grp.p, pc.p, 0, hard.p); }
  }
  exclusive_sum_u64(c, tsum.p, tbase.p, ntile + 1);
  exclusive_sum_u32(c, tile_full.p, fullbase.p, ntile + 1);
  hipLaunchKernelGGL(full_base0_kernel<I>, dim3(1), dim3(1), 0, c->stream, first_full.p, so.sa.p, wv, ix.lexrank.p, fullbase0.p);
  PFP_REQUIRE(read_scalar(c, ovf.p) == 0, PFP_ELIMIT, "2048 consecutive suffix-array slots emit 2^32 or more BWT positions");
  const uint64_t n_out = read_scalar(c, tbase.p + ntile);
  PFP_REQUIRE(expect_n_out == 0 || n_out == expect_n_out, PFP_EFORMAT,
              "merge: sum of occurrence counts (" + std::to_string(n_out) + ") != text length + 1 (" +
                  std::to_string(expect_n_out) + ")");
  out.n_out = n_out;
  cnt.release();
  MergeArgsT<I> a{};
  a.N = N; a.n_out = n_out; a.d = d; a.w = w; a.want_sa = samode;
  a.pos_base = pos_base; a.n_out_global = n_out_global ? n_out_global : n_out;
  a.sa = so.sa.p; a.grp = so.grp.p; a.wv = wv;
  a.ist = dense ? sfirst.p : nullptr;          // full SA: the record's `first` field held the list start
  a.wistart = wistart.p; a.wrec = wrec.p;
  a.sfirst = dense ? nullptr : sfirst.p; a.slast = slast.p; a.ssl = ssl.p;
  a.pc = pc.p; a.hard = hard.p; a.tbase = tbase.p; a.loc = loc.p;
  a.ilist = pb.ilist.p; a.bwlast = pb.bwlast.p; a.bwsai = pb.bwsai.p;
  a.istart_lex = istart_lex.p; a.fullbase = fullbase.p; a.fullbase0 = fullbase0.p;
  // the caller's buffers hold positions [out_lo, out_hi): rebase so that kernels index by global position
  a.out_lo = out_lo; a.out_hi = out_hi < n_out ? out_hi : n_out;
  a.bwt = out.d_bwt - out_lo; a.out_sa = out.d_sa ? out.d_sa - out_lo : nullptr;
  DBuf<unsigned long long> hstats(c, 5);
  hstats.zero();
  // hard[] marks exactly the heads of the hard groups (group_flags_kernel): count, then compact them once
  const uint64_t n_heads = count_flags(c, hard.p, N);
  // queue of the groups of more than 1024 occurrences: each emits more than 1024 positions and is a hard group; redone larger
  // if it overflows all the same (PFP_BIG_CAP: tests)
  uint32_t big_cap = (uint32_t)std::min<uint64_t>({(uint64_t)1u << 20, n_heads + 1, n_out / 1024 + 1});
  { const char *e = getenv("PFP_BIG_CAP"); if (e && atoll(e) > 0) big_cap = (uint32_t)atoll(e); }
  DBuf<BigGroup> big(c, big_cap);
  // such a group emits > kHardSortMin positions
  const uint32_t mid_cap = (uint32_t)std::min<uint64_t>(n_out / (kHardSortMin + 1) + 64, 0x7FFFFFFFull);
  DBuf<BigGroup> mid(c, mid_cap);
  DBuf<I> heads(c, n_heads + 1);
  DBuf<uint64_t> nheads(c, 2);
  select_index<I>(c, hard.p, heads.p, nheads.p, N);
  // BWT only / sparse SA: majority fill + minority placement; groups it leaves (no dominating char) go to the LDS kernels
  DBuf<uint8_t> gmaj, fallback;
  DBuf<unsigned long long> mstat(c, 2);      // chars and members of the hard groups the minority path takes
  mstat.zero();
  DBuf<HardGroupInfo> ginfo;
  DBuf<uint32_t> minor_cnt, mm_cnt;
  DBuf<uint64_t> minor_off, mm_off;
  DBuf<MinorMember> mm_list;
  uint64_t n_mm = 0, sum_k = 0;
  DBuf<I> fb_heads;
  const I *hard_list = heads.p;             // what hard_groups_kernel works through
  const uint64_t *hard_list_n = nheads.p;
  uint64_t n_minor = 0, n_fallback = n_heads;
  if (!dense && n_heads) {
    gmaj.alloc(c, N); fallback.alloc(c, n_heads); ginfo.alloc(c, n_heads); minor_cnt.alloc(c, n_heads + 1); minor_off.alloc(c, n_heads + 1);
    mm_cnt.alloc(c, n_heads + 1); mm_off.alloc(c, n_heads + 1);
    PFP_HIP(hipMemsetAsync(minor_cnt.p + n_heads, 0, 4, c->stream));
    PFP_HIP(hipMemsetAsync(mm_cnt.p + n_heads, 0, 4, c->stream));
    a.gmaj = gmaj.p;
    { KScope ks(c, "pfp::hard_classify_kernel", n_heads * 40);
      hipLaunchKernelGGL(hard_classify_kernel<I>, gdim(cdiv(n_heads, 256)), gdim(256), 0, c->stream, a, heads.p, n_heads, gmaj.p,
                         ginfo.p, minor_cnt.p, fallback.p, mstat.p, mm_cnt.p); }
    exclusive_sum_u32_u64(c, minor_cnt.p, minor_off.p, n_heads + 1);
    exclusive_sum_u32_u64(c, mm_cnt.p, mm_off.p, n_heads + 1);
    n_mm = read_scalar(c, mm_off.p + n_heads);
    sum_k = read_scalar(c, (const uint64_t *)mstat.p + 1);
    if (n_mm) {
      mm_list.alloc(c, n_mm);
      KScope ks(c, "pfp::hard_minor_fill_kernel", n_heads * 40 + n_mm * 16);
      hipLaunchKernelGGL(hard_minor_fill_kernel<I>, gdim(cdiv(n_heads * 8, 256)), gdim(256), 0, c->stream, a, ginfo.p, n_heads, mm_off.p,
                         minor_off.p, gmaj.p, mm_list.p);
    }
    n_fallback = count_flags(c, fallback.p, n_heads);
    n_minor = read_scalar(c, minor_off.p + n_heads);
    fb_heads.alloc(c, n_fallback + 1);
    {  // heads of the fallback groups = heads[] where fallback[] is set
      DBuf<I> fidx(c, n_fallback + 1);
      select_index<I>(c, fallback.p, fidx.p, nheads.p + 1, n_heads);
      if (n_fallback) hipLaunchKernelGGL(gather_idx_kernel<I>, gdim(cdiv(n_fallback, 256)), gdim(256), 0, c->stream, n_fallback, fidx.p,
                                         heads.p, fb_heads.p);
      PFP_HIP(hipGetLastError());
      sync(c);
    }
    hard_list = fb_heads.p; hard_list_n = nheads.p + 1;
  }
  const bool sparse = samode == SA_SPARSE;
  // what one launch writes: chars and SA values together (dense), or chars first and - once the finished BWT says where
  // the sampled files can look - SA values in a second round of the same kernels (sparse)
  a.pass = sparse ? PASS_BWT : (PASS_BWT | PASS_SA);
  const uint32_t nblk = (uint32_t)cdiv64(N, kSlots);
  DBuf<uint32_t> heavy(c, nblk), nheavy(c, 1);
  nheavy.zero();
  uint32_t nh = 0;
  auto run_expand = [&]() {
    { KScope ks(c, "pfp::expand_kernel", a.pass == PASS_SA ? N * 5 + pb.P * 28 : N * 14 + n_out * (dense ? 17 : 1));
      // (the whole-word list of the sparse SA round costs 12 KB of LDS: only that round's launch carries it)
      if (sparse && (a.pass & PASS_SA)) hipLaunchKernelGGL((expand_kernel<I, 1>), gdim(nblk), gdim(256), 0, c->stream, a, heavy.p, nheavy.p, nblk);
      else hipLaunchKernelGGL((expand_kernel<I, 0>), gdim(nblk), gdim(256), 0, c->stream, a, heavy.p, nheavy.p, nblk); }
    if (a.pass & PASS_BWT) nh = read_scalar(c, nheavy.p);
    KScope ks2(c, "pfp::expand_heavy_kernel", 0);   // bytes are accounted in expand_kernel's n_out term
    if (nh) hipLaunchKernelGGL(expand_heavy_kernel<I>, gdim(c->n_cu * 4), gdim(256), 0, c->stream, a, heavy.p, nh);
    PFP_HIP(hipGetLastError());
  };
  run_expand();
  DBuf<MinorRec> recs;
  if (n_mm) {
    if (sparse) recs.alloc(c, n_minor + 1);
    KScope ks(c, "pfp::hard_minor_kernel", n_minor * 64);
    // lanes per minority member: eight, or the wave where the groups average more than 32 members
    const uint64_t avg_k = sum_k / std::max<uint64_t>(n_heads - n_fallback, 1);
    if (avg_k > 32)
      hipLaunchKernelGGL((hard_minor_kernel<I, 64>), gdim((unsigned)std::min<uint64_t>(cdiv64(n_mm, 4), (uint64_t)c->n_cu * 64)), gdim(256), 0,
                         c->stream, a, mm_list.p, n_mm, recs.p);
    else
      hipLaunchKernelGGL((hard_minor_kernel<I, 8>), gdim((unsigned)std::min<uint64_t>(cdiv64(n_mm, 32), (uint64_t)c->n_cu * 64)), gdim(256), 0,
                         c->stream, a, mm_list.p, n_mm, recs.p);
  }
  PFP_HIP(hipGetLastError());
  int gpb = 64;      // groups per wave batch: down to 8 while that still leaves every wave of the launch a batch
  while (gpb > 8 && n_fallback / gpb < (uint64_t)c->n_cu * 32) gpb >>= 1;
  for (;;) {
    if (!n_fallback) { PFP_HIP(hipMemsetAsync(hstats.p, 0, 40, c->stream)); }
    else { KScope ks(c, "pfp::hard_groups_kernel", N * 5);
      hipLaunchKernelGGL(hard_groups_kernel<I>, gdim(c->n_cu * 8), gdim(256), 0, c->stream, a, hard_list, hard_list_n, hstats.p, big.p,
                         big_cap, mid.p, mid_cap, gpb); }
    PFP_HIP(hipGetLastError());
    PFP_HIP(hipMemcpyAsync(c->h_scalars, hstats.p, 40, hipMemcpyDeviceToHost, c->stream));
    sync(c);
    if (c->h_scalars[2] <= big_cap) break;
    big_cap = (uint32_t)std::min<uint64_t>(c->h_scalars[2], 0xFFFFFFFFull);   // rare: redo with a queue that fits
    big.alloc(c, big_cap);
    hstats.zero();
  }
  out.hard_chars = c->h_scalars[0];
  out.hard_groups = c->h_scalars[1];
  out.hard_minor_groups = n_heads - n_fallback; out.hard_minor_chars = n_minor;
  out.hard_big_groups = c->h_scalars[2]; out.hard_max_chars = 0; out.hard_max_members = c->h_scalars[4];
  const uint32_t nmid = (uint32_t)std::min<uint64_t>(c->h_scalars[3], mid_cap);
  PFP_REQUIRE(c->h_scalars[3] <= mid_cap, PFP_EHIP, "more sorted-path hard groups than the output can hold");
  const uint32_t nbig = (uint32_t)c->h_scalars[2];
  out.hard_chars += read_scalar(c, (const uint64_t *)mstat.p);      // all chars of hard groups, whichever path wrote them
  out.hard_groups += n_heads - n_fallback;
  DBuf<uint64_t> estart;
  std::vector<uint64_t> es(nbig + 1, 0);
  // (read per call: the tests switch them inside one process)
  const uint64_t big_budget = [] { const char *e = getenv("PFP_BIG_BUDGET"); return e ? strtoull(e, nullptr, 10) : (1ull << 27); }();
  if (nbig) {
    // occurrences of the queued groups, laid end to end
    std::vector<BigGroup> hb(nbig);
    PFP_HIP(hipMemcpyAsync(hb.data(), big.p, nbig * sizeof(BigGroup), hipMemcpyDeviceToHost, c->stream));
    sync(c);
    for (uint32_t q = 0; q < nbig; q++) es[q + 1] = es[q] + hb[q].E;
    estart.alloc(c, nbig + 1);
    PFP_HIP(hipMemcpyAsync(estart.p, es.data(), (nbig + 1) * 8, hipMemcpyHostToDevice, c->stream));
    sync(c);
  }
  auto run_queued = [&]() {
    if (nmid) {
      KScope ks(c, "pfp::hard_sort_kernel", (uint64_t)nmid * (kHardSortMin + 1) * (samode ? 21 : 5));      // lower bound: ilist entry + char (+ SA) per occurrence
      hipLaunchKernelGGL(hard_sort_kernel<I>, gdim(c->n_cu * 4), gdim(256), 0, c->stream, a, mid.p, nmid);
      PFP_HIP(hipGetLastError());
    }
    // queued groups in chunks of at most big_budget occurrences: keys, one sort, placement (a single group beyond the
    // budget is ranked occurrence by occurrence - its sort scratch is not worth the memory)
    for (uint32_t q0 = 0; q0 < nbig;) {
      uint32_t q1 = q0 + 1;
      while (q1 < nbig && es[q1 + 1] - es[q0] <= big_budget) q1++;
      const uint64_t cnt = es[q1] - es[q0];
      if (cnt > big_budget) {
        const int nb = (int)std::min<uint64_t>(cdiv64(cnt, 256), (uint64_t)c->n_cu * 32);
        KScope ks(c, "pfp::hard_big_kernel", cnt * (samode ? 21 : 5));
        hipLaunchKernelGGL(hard_big_kernel<I>, gdim(nb), gdim(256), 0, c->stream, a, big.p + q0, q1 - q0, estart.p + q0, cnt);
      } else {
        DBuf<uint64_t> bk(c, cnt), bka(c, cnt), bv(c, cnt), bva(c, cnt);
        { KScope ks(c, "pfp::big_keys_kernel", cnt * 28);
          hipLaunchKernelGGL(big_keys_kernel<I>, gdim(cdiv(cnt, 256)), gdim(256), 0, c->stream, a, big.p, q0, q1, estart.p, cnt, bk.p, bv.p); }
        { SortTag tag("large hard groups"); sort_pairs_db<uint64_t, uint64_t>(c, bk, bka, bv, bva, cnt, 0, 32 + bits_for(q1 - q0)); }
        { KScope ks(c, "pfp::big_place_kernel", cnt * (samode ? 34 : 18));
          hipLaunchKernelGGL(big_place_kernel<I>, gdim(cdiv(cnt, 256)), gdim(256), 0, c->stream, a, big.p, q0, estart.p, cnt, bk.p, bv.p); }
      }
      PFP_HIP(hipGetLastError());
      q0 = q1;
    }
  };
  run_queued();
  if (!sparse) { sync(c); return; }

  // ---- sparse SA, second round: where can the sampled files look?
  const uint64_t cnt_slice = a.out_hi > a.out_lo ? a.out_hi - a.out_lo : 0;
  const uint64_t nw = cdiv64(cnt_slice, 64);
  out.bmap.alloc(c, nw + 1); out.bpre.alloc(c, nw + 1);
  // ... and no SA array to write to: the sampled files come from bitmaps (a slice's two edge positions count as run
  // starts / ends here; the multi-GPU caller drops them again when the neighbour's halo byte says so)
  const bool want_s = !out.d_sa && (flags & PFP_FLAG_SSA), want_e = !out.d_sa && (flags & PFP_FLAG_ESA);
  out.slice_n = cnt_slice;
  {
    DBuf<uint32_t> wcnt(c, nw + 1), scnt, ecnt;
    PFP_HIP(hipMemsetAsync(wcnt.p + nw, 0, 4, c->stream));
    PFP_HIP(hipMemsetAsync(out.bmap.p + nw, 0, 8, c->stream));
    if (want_s) { out.smap.alloc(c, nw + 1); out.spre.alloc(c, nw + 1); scnt.alloc(c, nw + 1); PFP_HIP(hipMemsetAsync(scnt.p + nw, 0, 4, c->stream)); }
    if (want_e) { out.emap.alloc(c, nw + 1); out.epre.alloc(c, nw + 1); ecnt.alloc(c, nw + 1); PFP_HIP(hipMemsetAsync(ecnt.p + nw, 0, 4, c->stream)); }
    if (nw) {
      KScope ks(c, "pfp::run_bitmap_kernel", cnt_slice + nw * (12 + (want_s ? 12 : 0) + (want_e ? 12 : 0)));
      hipLaunchKernelGGL(run_bitmap_kernel, gdim(cdiv(cdiv64(cnt_slice, 16), 256)), gdim(256), 0, c->stream, out.d_bwt, cnt_slice,
                         out.bmap.p, wcnt.p, out.smap.p, scnt.p, out.emap.p, ecnt.p);
    }
    exclusive_sum_u32_u64(c, wcnt.p, out.bpre.p, nw + 1);
    if (want_s) exclusive_sum_u32_u64(c, scnt.p, out.spre.p, nw + 1);
    if (want_e) exclusive_sum_u32_u64(c, ecnt.p, out.epre.p, nw + 1);
    out.n_bound = read_scalar(c, out.bpre.p + nw);
    if (want_s) out.n_starts = read_scalar(c, out.spre.p + nw);
    if (want_e) out.n_ends = read_scalar(c, out.epre.p + nw);
  }
  a.bmap = out.bmap.p; a.bpre = out.bpre.p;
  if (!out.d_sa) {      // the caller keeps no SA array: values go to their rank among the boundaries
    out.sa_c.alloc(c, out.n_bound + 1);
    if (c->debug) PFP_HIP(hipMemsetAsync(out.sa_c.p, 0xFF, (out.n_bound + 1) * 8, c->stream));
    a.sa_c = out.sa_c.p;
  }
  a.pass = PASS_SA;
  if (ix.wslot_lex.p && pb.P) {      // whole words: one lane per occurrence of the inverted lists
    KScope ks(c, "pfp::word_sa_kernel", pb.P * (4 + 8 + 8 + 8));
    hipLaunchKernelGGL(word_sa_kernel<I>, gdim(cdiv(pb.P, 256)), gdim(256), 0, c->stream, a, pb.P, ix.wslot_lex.p, so.slot_base,
                       (const uint32_t *)nullptr, (const uint32_t *)nullptr);
  } else run_expand();
  { KScope ks(c, "pfp::unit_edges_kernel", N * (1 + sizeof(I) * 2 + 4));
    hipLaunchKernelGGL(unit_edges_kernel<I>, gdim(cdiv(N, 256)), gdim(256), 0, c->stream, a); }
  if (n_mm) {
    const uint64_t nr = n_minor;      // one record per minority occurrence, at its precomputed index (MinorMember::roff)
    if (nr) hipLaunchKernelGGL(hard_minor_sa_kernel<I>, gdim(cdiv(nr, 256)), gdim(256), 0, c->stream, a, recs.p, nr);
  }
  if (n_fallback) {      // the groups the LDS kernels ranked: the same ranks again, SA values this time (queues as they stand)
    DBuf<unsigned long long> scratch(c, 5);
    scratch.zero();
    KScope ks(c, "pfp::hard_groups_kernel", N * 5);
    hipLaunchKernelGGL(hard_groups_kernel<I>, gdim(c->n_cu * 8), gdim(256), 0, c->stream, a, hard_list, hard_list_n, scratch.p, big.p,
                       0u, mid.p, 0u, gpb);
    PFP_HIP(hipGetLastError());
    run_queued();
    sync(c);
  }
  PFP_HIP(hipGetLastError());
  if (c->debug && a.sa_c) {      // every boundary of the BWT must have received its value
    DBuf<unsigned long long> unset(c, 1);
    unset.zero();
    // (a slice's own two edge positions are marked as boundaries whatever their neighbours in the other slices hold:
    //  when they lie inside a run of the whole BWT nobody owes them a value - and nobody will sample them)
    const uint64_t skip_lo = (a.pos_base + a.out_lo > 0) ? 1 : 0, skip_hi = (a.pos_base + a.out_hi < a.n_out_global) ? 1 : 0;
    if (out.n_bound > skip_lo + skip_hi)
      hipLaunchKernelGGL(count_unset_kernel, gdim(cdiv(out.n_bound, 256)), gdim(256), 0, c->stream, out.sa_c.p + skip_lo,
                         out.n_bound - skip_lo - skip_hi, unset.p);
    const uint64_t u = read_scalar(c, (const uint64_t *)unset.p);
    PFP_REQUIRE(u == 0, PFP_EHIP, "sparse SA: " + std::to_string(u) + " run boundaries of the BWT received no SA value");
  }
  sync(c);
}
template void merge_bwt<uint32_t>(pfp_ctx *, const Dictionary &, const DictIndex &, const SuffixOrderT<uint32_t> &, const ParseBWT &,
                                  const uint32_t *, int, int, uint64_t, BwtOutputs &, uint64_t, uint64_t, uint64_t, uint64_t);
template void merge_bwt<uint64_t>(pfp_ctx *, const Dictionary &, const DictIndex &, const SuffixOrderT<uint64_t> &, const ParseBWT &,
                                  const uint32_t *, int, int, uint64_t, BwtOutputs &, uint64_t, uint64_t, uint64_t, uint64_t);

// ------------------------------------------------------------------ output packing

// utils.c:112-129: low 5 bytes, little endian.  One thread per 16 OUTPUT bytes (3.2 values): it reads the
// (up to) four values its chunk overlaps - neighbouring lanes read overlapping, consecutive values - lays
// their 5-byte fields end to end and cuts its 16 bytes out; every store is one aligned-size 16-byte store.
__global__ __launch_bounds__(256) void pack5_kernel(const uint64_t *__restrict__ v, uint64_t cnt, uint8_t *__restrict__ out) {
  const uint64_t q = (uint64_t)BID * 256 + threadIdx.x;
  const uint64_t total = cnt * 5, b0 = q * 16;
  if (b0 >= total) return;
  const uint64_t v0 = b0 / 5;
  const uint32_t s = (uint32_t)(b0 - v0 * 5) * 8;
  const uint64_t M = 0xFFFFFFFFFFull;
  const uint64_t a = v[v0] & M, b = v0 + 1 < cnt ? v[v0 + 1] & M : 0, c2 = v0 + 2 < cnt ? v[v0 + 2] & M : 0,
                 d = v0 + 3 < cnt ? v[v0 + 3] & M : 0;
  const uint64_t lo = a | (b << 40), mid = (b >> 24) | (c2 << 16) | (d << 56), hi = d >> 8;
  const uint64_t olo = s ? (lo >> s) | (mid << (64 - s)) : lo, ohi = s ? (mid >> s) | (hi << (64 - s)) : mid;
  if (b0 + 16 <= total) st16u(out + b0, make_uint4((uint32_t)olo, (uint32_t)(olo >> 32), (uint32_t)ohi, (uint32_t)(ohi >> 32)));
  else
    for (uint64_t k = 0; b0 + k < total; k++) out[b0 + k] = (uint8_t)((k < 8 ? olo >> (8 * k) : ohi >> (8 * (k - 8))) & 0xff);
}
__global__ void unpack5_kernel(const uint8_t *__restrict__ in, uint64_t cnt, uint64_t *__restrict__ v) {
  uint64_t i = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (i >= cnt) return;
  const uint8_t *p = in + 5 * i;
  uint64_t x = 0;
#pragma unroll
  for (int b = 0; b < 5; b++) x |= (uint64_t)p[b] << (8 * b);
  v[i] = x;
}
void pack5_dev(pfp_ctx *c, const uint64_t *vals, uint64_t cnt, uint8_t *out5) {
  if (!cnt) return;
  KScope ks(c, "pfp::pack5_kernel", cnt * 13);
  hipLaunchKernelGGL(pack5_kernel, gdim((unsigned)cdiv64(cdiv64(cnt * 5, 16), TB)), gdim(TB), 0, c->stream, vals, cnt, out5);
  PFP_HIP(hipGetLastError());
}
void unpack5_dev(pfp_ctx *c, const uint8_t *in5, uint64_t cnt, uint64_t *vals) {
  if (!cnt) return;
  hipLaunchKernelGGL(unpack5_kernel, gdim(cdiv(cnt, TB)), gdim(TB), 0, c->stream, in5, cnt, vals);
  PFP_HIP(hipGetLastError());
}

// run boundaries: .ssa = <j,SA[j]> for BWT[j] != BWT[j-1] incl. j=0 (pfbwt.cpp:169-174,184-189);
//                 .esa = <j,SA[j]> for BWT[j] != BWT[j+1] incl. j=n (pfbwt.cpp:175-179,225-229)
// Two streaming passes over the BWT bytes of a slice [pos_base, pos_base+cnt) (16 positions per thread from
// one unaligned 16-byte load, the neighbour byte shifted in): boundaries counted per tile of 4096
// positions, tile offsets scanned, then every thread places its pairs <position, SA value> as 10 bytes.
// left / right: the BWT byte just outside the slice, or -1 at the ends of the whole BWT (then the
// edge position is a boundary by definition).
__global__ __launch_bounds__(256) void run_count_kernel(const uint8_t *__restrict__ bwt, uint64_t cnt, int left, int right,
                                                        int run_end, uint32_t *__restrict__ tile_cnt) {
  __shared__ uint32_t ws[4];
  if ((uint64_t)BID * kRunTile >= cnt) return;      // a workgroup of the padded last grid row
  const uint64_t base = (uint64_t)BID * kRunTile + (uint64_t)threadIdx.x * 16;
  uint32_t m[4];
  run_mask16(bwt, base, cnt, left, right, run_end, m);
  uint32_t c = __popc(m[0]) + __popc(m[1]) + __popc(m[2]) + __popc(m[3]);
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) tile_cnt[BID] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ __launch_bounds__(256) void run_place_kernel(const uint8_t *__restrict__ bwt, SaView sa,
                                                        uint64_t cnt, uint64_t pos_base, int left, int right, int run_end,
                                                        const uint64_t *__restrict__ tile_off, uint8_t *__restrict__ out10) {
  __shared__ uint32_t ws[4];
  if ((uint64_t)BID * kRunTile >= cnt) return;      // a workgroup of the padded last grid row
  const uint64_t base = (uint64_t)BID * kRunTile + (uint64_t)threadIdx.x * 16;
  uint32_t m[4];
  run_mask16(bwt, base, cnt, left, right, run_end, m);
  const uint32_t c = __popc(m[0]) + __popc(m[1]) + __popc(m[2]) + __popc(m[3]);
  uint32_t inc = c;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
  if (lane == 63) ws[wv] = inc;
  __syncthreads();
  if (!c) return;
  uint64_t o = tile_off[BID] + inc - c;
  for (int q = 0; q < wv; q++) o += ws[q];
#pragma unroll
  for (int k = 0; k < 16; k++)
    if ((m[k >> 2] >> (8 * (k & 3))) & 1u) {
      const uint64_t x = pos_base + base + k, r = base + k;
      uint64_t v;
      if (sa.dense) v = sa.dense[r];
      else { const uint64_t wv = sa.bmap[r >> 6]; v = sa.sa_c[sa.bpre[r >> 6] + (uint64_t)__popcll(wv & ((1ull << (r & 63)) - 1ull))]; }
      uint8_t *dst = out10 + 10 * o;
      reinterpret_cast<U64u *>(dst)->v = (x & 0xFFFFFFFFFFull) | (v << 40);       // 5 bytes of x, 3 low bytes of v
      reinterpret_cast<U16u *>(dst + 8)->v = (uint16_t)(v >> 24);                  // bytes 3, 4 of v
      o++;
    }
}

RunSampler::RunSampler(pfp_ctx *c_, const uint8_t *bwt_, uint64_t cnt_, int left_, int right_, bool run_end_)
    : c(c_), bwt(bwt_), cnt(cnt_), left(left_), right(right_), run_end(run_end_) {
  ntile = cdiv64(cnt, kRunTile);
  tile_cnt.alloc(c, ntile + 1);
  tile_off.alloc(c, ntile + 1);
  PFP_HIP(hipMemsetAsync(tile_cnt.p + ntile, 0, 4, c->stream));
  if (ntile) {
    KScope ks(c, "pfp::run_count_kernel", cnt);
    hipLaunchKernelGGL(run_count_kernel, gdim((unsigned)ntile), gdim(256), 0, c->stream, bwt, cnt, left, right, run_end ? 1 : 0,
                       tile_cnt.p);
  }
  exclusive_sum_u32_u64(c, tile_cnt.p, tile_off.p, ntile + 1);
  PFP_HIP(hipGetLastError());
  pairs = read_scalar(c, tile_off.p + ntile);
}
void RunSampler::place(const SaView &sa, uint64_t pos_base, uint8_t *out10) {
  if (!ntile || !pairs) return;
  KScope ks(c, "pfp::run_place_kernel", cnt + pairs * 18);
  hipLaunchKernelGGL(run_place_kernel, gdim((unsigned)ntile), gdim(256), 0, c->stream, bwt, sa, cnt, pos_base, left, right,
                     run_end ? 1 : 0, tile_off.p, out10);
  PFP_HIP(hipGetLastError());
}

uint64_t sample_runs_dev(pfp_ctx *c, const uint8_t *bwt, const SaView &sa, uint64_t n_out, bool run_end,
                         DBuf<uint8_t> &out10) {
  const uint64_t *map = run_end ? sa.emap : sa.smap;
  if (map && sa.sa_c) {      // the merge left the run starts / ends as bitmaps
    const uint64_t pairs = run_end ? sa.n_ends : sa.n_starts;
    out10.alloc(c, pairs * 10 + 16);
    if (sa.n_words) {
      KScope ks(c, "pfp::bitmap_place_kernel", sa.n_words * 24 + pairs * 18);
      hipLaunchKernelGGL(bitmap_place_kernel, gdim(cdiv(sa.n_words, 256)), gdim(256), 0, c->stream, map, run_end ? sa.epre : sa.spre, sa.bmap,
                         sa.bpre, sa.sa_c, sa.n_words, out10.p, (uint64_t)0, 0, ~0ull);
      PFP_HIP(hipGetLastError());
    }
    return pairs;
  }
  RunSampler rs(c, bwt, n_out, -1, -1, run_end);
  out10.alloc(c, rs.pairs * 10 + 16);
  rs.place(sa, 0, out10.p);
  return rs.pairs;
}

uint64_t sample_runs_maps(pfp_ctx *c, const SaView &sa, uint64_t slice_n, bool run_end, bool drop_edge, uint64_t pos_base, uint8_t *out10) {
  const uint64_t *map = run_end ? sa.emap : sa.smap;
  PFP_REQUIRE(map && sa.sa_c, PFP_EINVAL, "no run maps: the merge was not run for a sampled SA without an SA array");
  const uint64_t all = run_end ? sa.n_ends : sa.n_starts;
  const uint64_t pairs = all - ((drop_edge && all) ? 1 : 0);
  if (!out10 || !pairs || !sa.n_words) return pairs;
  KScope ks(c, "pfp::bitmap_place_kernel", sa.n_words * 24 + pairs * 18);
  hipLaunchKernelGGL(bitmap_place_kernel, gdim(cdiv(sa.n_words, 256)), gdim(256), 0, c->stream, map, run_end ? sa.epre : sa.spre, sa.bmap,
                     sa.bpre, sa.sa_c, sa.n_words, out10, pos_base, (drop_edge && !run_end) ? 1 : 0,
                     (drop_edge && run_end && slice_n) ? slice_n - 1 : ~0ull);
  PFP_HIP(hipGetLastError());
  return pairs;
}

}  // namespace pfp
