// kernels.hpp -- declarations shared by the pipeline translation units.
#pragma once
#include "common.hpp"
#include "wordview.hpp"

namespace pfp {

// ---------------------------------------------------------------- text staging
// T' = Dollar . T . Dollar^w lives in HBM with a zeroed 64-byte front pad so that T[0] is
// 16-byte aligned, T'[x] = tprime()[x], and short unaligned over-reads stay inside the buffer.
struct StagedText {
  static constexpr size_t kFront = 64, kBack = 128;
  DBuf<uint8_t> buf;
  uint64_t n = 0;
  int w = 0;
  const uint8_t *tbase() const { return buf.p + kFront; }        // T[i]
  const uint8_t *tprime() const { return buf.p + kFront - 1; }   // T'[x]
  void stage(pfp_ctx *c, const void *src, bool src_on_device, uint64_t n_, int w_);
  void stage_fd(pfp_ctx *c, int fd, uint64_t file_off, uint64_t n_, int w_);      // host text read from an open file (parallel pread)
  void restage_tail(pfp_ctx *c, uint64_t new_n, int w_) const;    // Dollars at [new_n,new_n+w), zeros after
};

// ---------------------------------------------------------------- stage 1a (scan.hip)
struct KRParams {
  static constexpr uint32_t kMaxExtra = 32;
  uint32_t negpw, pinv, pshift, plimit;
  float fdens = 1.0f;                          // the density the scan cuts at (x nominal)
  uint32_t fthr_nom = 0, fauto = 0;            // nominal threshold (density 1 / p); fauto: the chain chooses between fthr_nom and fthr = twice that
  uint32_t fast = 0, fseed = 0, fthr = 0;      // fused chain only (round 4): the cheap window hash of scan.hip instead of Karp-Rabin
  uint32_t nextra;            // extra trigger hashes (fused chain only), 0 in the reference-exact scan
  uint64_t bloom;             // bit (h & 63) set for every extra hash
  uint32_t extra[kMaxExtra];
};
KRParams make_kr_params(int w, uint64_t p);
KRParams make_fast_params(pfp_ctx *c, const StagedText &tx, uint64_t n, int w, uint64_t p);
KRParams make_fast_params_host(const uint8_t *first_window, int w, uint64_t p, double density_setting);
// phrase length by repetitiveness (scan.hip): the pieces the single-GPU chain runs in one go and the multi-GPU chain with an
// all-gather of the ranks' samples in between
struct CutSample { DBuf<uint8_t> nominal; DBuf<uint64_t> hashes; uint64_t ns = 0; };
void classify_cuts(pfp_ctx *c, const StagedText &tx, int w, const DBuf<uint64_t> &d_ends, uint64_t ne, const KRParams &kp, uint64_t min_end,
                   CutSample &out);
bool sample_says_dense(pfp_ctx *c, const uint64_t *d_sorted, uint64_t ns, uint64_t p);
uint64_t keep_nominal_cuts(pfp_ctx *c, DBuf<uint64_t> &d_ends, uint64_t ne, const DBuf<uint8_t> &nominal);
uint32_t window_hash_host(const uint8_t *win, int w, uint32_t seed);      // the seeded window hash of scan.hip, on the host
void scan_flags(pfp_ctx *c, const uint8_t *tbase, uint64_t n, int w, uint64_t p, uint16_t *flags16,
                uint32_t *block_counts, unsigned long long *first_bad, const KRParams *kp_override = nullptr);
uint64_t scan_text(pfp_ctx *c, const StagedText &tx, uint64_t n, int w, uint64_t p, DBuf<uint64_t> &d_ends,
                   uint64_t *n_used, const KRParams *kp_override = nullptr);
uint32_t propose_extra_triggers(pfp_ctx *c, const StagedText &tx, uint64_t n_used, int w, uint64_t max_phrase,
                                const DBuf<uint64_t> &d_ends, uint64_t ne, KRParams &kp);
// fused chain: reference triggers plus a few extra window hashes that split phrases longer than max_phrase
uint64_t scan_text_adaptive(pfp_ctx *c, const StagedText &tx, uint64_t n, int w, uint64_t p, uint64_t max_phrase,
                            DBuf<uint64_t> &d_ends, uint64_t *n_used, uint32_t *n_extra);

// ---------------------------------------------------------------- stage 1b (phrase.hip)
// Distinct phrases (the dictionary), most frequent first (ties: first occurrence), plus the parse as word ids.
struct Dictionary {
  uint64_t P = 0;          // # phrases
  uint64_t d = 0;          // # distinct words
  uint64_t dsize = 0;      // bytes of dict incl. one 0x01 per word and the final 0x00
  DBuf<uint8_t> bytes;     // D (padded with zeros: +64 front is not needed, +64 back)
  DBuf<uint64_t> woff;     // [d+1] start of word j in bytes; woff[d] = dsize-1
  DBuf<uint32_t> wlen;     // [d]
  DBuf<uint32_t> wocc;     // [d]
  DBuf<uint32_t> pid;      // [P] word id of phrase k (empty when built from a .dict file)
  DBuf<uint8_t> last;      // [P] .last
  DBuf<uint64_t> sai;      // [P] .sai values
  uint64_t reseeds = 0;
};
void build_dictionary(pfp_ctx *c, const StagedText &tx, uint64_t n, int w, const DBuf<uint64_t> &ends, uint64_t n_ends,
                      bool want_sai, Dictionary &D);
// multi-GPU: a shard owns phrases [k0,k0+P) of its local scan; sai values are shifted by sai_base
void build_dictionary_shard(pfp_ctx *c, const StagedText &tx, uint64_t n, int w, const DBuf<uint64_t> &ends,
                            uint64_t n_ends, uint64_t k0, uint64_t P, bool want_sai, uint64_t sai_base, Dictionary &D);
// multi-GPU: dictionary of a union of word lists; D.pid[u] = id of union word u, occ = sum of weights
void build_dictionary_words(pfp_ctx *c, const uint8_t *bytes, const uint64_t *wstart, const uint32_t *wlen, uint64_t U,
                            const uint32_t *weight, uint64_t total_bytes, Dictionary &D);
// identity hash of every word of a list (same function the dedup sorts by)
void reorder_dictionary_by_occ(pfp_ctx *c, Dictionary &D, DBuf<uint32_t> &perm);      // words by descending occurrence; perm[old] = new
void hash_word_list(pfp_ctx *c, const uint8_t *bytes, const uint64_t *wstart, const uint32_t *wlen, uint64_t U, uint64_t seed,
                    uint64_t *d_hash);

// ---------------------------------------------------------------- suffix sorting (sufsort.hip)
// Index width.  Positions inside the dictionary and slots of its suffix array are `I` = uint32_t while the
// dictionary is shorter than 4 GiB and uint64_t beyond (the reference makes the same switch between its
// 32-bit and -DM64 builds, bigbwt:109-151, gsa/gsacak.h:42-60).  Everything that indexes the parse (P <=
// 2^32-2, bwtparse.c:93), words (d < 2^32) or a distance inside one word stays 32 bits wide in both builds.
template <class I> struct IdxTraits;
template <> struct IdxTraits<uint32_t> { using DKey = uint64_t; static constexpr uint32_t kNone = 0xFFFFFFFFu; static constexpr uint32_t kTop = 0x80000000u; };
template <> struct IdxTraits<uint64_t> { using DKey = unsigned __int128; static constexpr uint64_t kNone = ~0ull; static constexpr uint64_t kTop = 1ull << 63; };
// true when a dictionary of dsize bytes needs (or PFP_FORCE_IDX64 asks for) the wide build
// (the reference goes to its 64-bit build at 2^31 - 4 already, bigbwt:130; unsigned 32-bit positions reach twice as far, and
// the wide build needs 8 bytes where this one needs 4)
inline bool use_wide_index(const pfp_ctx *c, uint64_t dsize) { return c->force_wide || dsize >= 0xFFFFFFF0ull; }
// Dictionaries of 2^31 .. 2^32 bytes on one GPU (round 4): 32-bit positions still fit, but not with the sorter's "settled" flag in
// their top bit - the narrow build then has doubling rounds only (2.0 s for a 2.4 GB dictionary of 3 % variants); the wide build
// keeps its pivot rounds (1.25 s) at ~96 bytes of device memory per dictionary byte.  Width by cost: wide where that much memory
// is there (free on the device + cached in the pool), narrow otherwise.  profiles/r04_dictionary_2_to_4_gib.json
inline bool prefer_wide_index(const pfp_ctx *c, uint64_t dsize) {
  if (use_wide_index(c, dsize)) return true;
  if (dsize < (1ull << 31) || c->force_narrow) return false;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return false; }
  const uint64_t avail = (uint64_t)free_b + (c->pool.total_bytes - c->pool.live_bytes);
  return avail >= dsize * 100ull;
}

template <class I>
struct SuffixOrderT {
  double rep_hint = 0;   // set by the caller before sort_dict_suffixes: text bytes per dictionary byte (>= 2: a repetitive collection)
  uint64_t N = 0;        // slots held here: all NP suffixes, or (key-range sharded sort) one contiguous range of SA(D)
  uint64_t NP = 0;       // positions of the sorted string
  uint64_t slot_base = 0;        // range mode: SA(D) slot of local slot 0
  uint64_t range_emits = 0;      // range mode: BWT positions the range emits (when a count source was given)
  uint64_t klo = 0, khi = ~0ull; // range mode: first-round keys in [klo, khi) (khi == ~0: no upper bound)
  bool range = false, complete = true;   // range mode stops (complete = false) where it would need other ranks' ranks
  DBuf<I> sa;            // [N] suffix start positions in sorted order (ties: position order)
  DBuf<I> grp;           // [N] grp[t] = first sa slot of slot t's group (equal strings share it)
  DBuf<I> rank;          // [NP] rank[i] = grp[slot of i]; kNone where the sorter never needed it (see RankView); not allocated
                         // where no round ever read it (then: wordrank)
  DBuf<I> wordrank;      // [d+1] rank of word j's whole-word suffix where a round after the first settled it, else kNone
  uint64_t rounds = 0;
  // Dictionary mode keeps the sorted first-round keys: the rank of a suffix that the first round
  // already settled is the lower bound of its packed key among them, found on demand instead of
  // being scattered to rank[] for all N positions (the scatter was 9 ms of 56 at N = 260 M).
  DBuf<uint64_t> skeys;  // [N] packed keys of the first round in sorted order
  DBuf<I> tab;           // [T] kNone - (first slot whose key has top bits >= T-1-r), r reversed bucket
  DBuf<uint32_t> lut;    // [256] byte -> (alphabetic code << 6) | code length
  const uint8_t *bytes = nullptr;
  int kbits = 0, shift = 0;   // kbits code bits per key (+ 1 flag bit)
  uint32_t T = 0;
  I finbit = 0;          // dictionary mode: rank[] values carry this bit once their suffix is settled (0: N too large for a spare bit)
  // When the first-round keys leave 16 bits free, every key carries its suffix's merge record {preceding
  // char, count code} there (SlotPayloadSrc): it arrives at the suffix's slot with the sort, and the merge
  // reads it from skeys instead of gathering 2 bytes per slot at random (4.9 ms of 38 at N = 260 M).
  // refined[t] != 0: slot t was re-ordered after the first round, its record has to be fetched.
  int paybits = 0;
  uint64_t keymask = ~0ull;
  DBuf<uint8_t> refined;
  uint64_t n_refined = 0;
};
using SuffixOrder = SuffixOrderT<uint32_t>;
using SuffixOrder64 = SuffixOrderT<uint64_t>;
// what a merge record is made from: the word of a position (WordView), the words' occurrence counts, the window
struct SlotPayloadSrc { WordView wv; const uint32_t *wocc; int w; };
// device-side view for rank lookups (sufsort.hip: rank_at)
template <class I>
struct RankViewT {
  const I *rank; const uint64_t *skeys; const I *tab; const uint32_t *lut; const uint8_t *bytes;
  uint64_t N; int kbits, shift; uint32_t T; I finbit; uint64_t keymask;
  const I *wordrank;      // gather_ranks over the word starts: see SuffixOrderT::wordrank
};
template <class I> RankViewT<I> rank_view(const SuffixOrderT<I> &so);
// out[k] = rank of the suffix starting at pos[k]
template <class I> void gather_ranks(pfp_ctx *c, const SuffixOrderT<I> &so, const uint64_t *d_pos, uint64_t count, I *d_out);
// Suffixes of the dictionary as 0x01-terminated strings (gsacak semantics); wv answers where the
// word of a position ends (the final 0x00 is its own word).
template <class I>
void sort_dict_suffixes(pfp_ctx *c, const uint8_t *bytes, uint64_t N, const WordView &wv, SuffixOrderT<I> &out,
                        const SlotPayloadSrc *pay = nullptr);
// multi-GPU: rank `part` of `parts` sorts the suffixes whose first-round key lies in its share of the key
// space (splitters from a deterministic key sample: every rank derives the same ones, no exchange);
// groups never straddle shares, and pivot rounds compare strings, not ranks, so a share is finished
// without its neighbours.  out.complete == false: a group was left that only doubling could settle.
template <class I>
void sort_dict_suffixes_range(pfp_ctx *c, const uint8_t *bytes, uint64_t N, const WordView &wv, uint32_t part,
                              uint32_t parts, SuffixOrderT<I> &out, const SlotPayloadSrc *pay = nullptr,
                              const SlotPayloadSrc *count = nullptr);
// range mode: out[j] = 1 + SA(D) slot of the whole-word suffix of word j if it belongs to this share, else 0 (count = d)
template <class I>
void gather_slots_range(pfp_ctx *c, const SuffixOrderT<I> &so, const WordView &wv, uint64_t count, uint64_t *d_out);
// plain suffix array of an integer string with unique smallest last symbol (sacak_int)
// max_sym: largest symbol value (spare key bits then describe runs of equal symbols)
// one rank's share of the same suffix array (the suffixes whose first-round key lies in share `part` of `parts`): out.N slots from
// out.slot_base on, out.complete = false where only a doubling round could go on
void sort_int_suffixes_range(pfp_ctx *c, const uint32_t *sym, uint64_t N, uint32_t max_sym, const uint32_t *occ, uint32_t n_sym,
                             uint32_t part, uint32_t parts, SuffixOrder &out);
// occ (optional, n_sym entries): occ[x - 1] = occurrences of symbol x - lets the sorter pick its pivots (the parse of a collection)
void sort_int_suffixes(pfp_ctx *c, const uint32_t *sym, uint64_t N, SuffixOrder &out, uint32_t max_sym = 0xFFFFFFFFu,
                       const uint32_t *occ = nullptr, uint32_t n_sym = 0);
// plain suffix array of a byte string with s[N-1]==0 unique smallest (sacak)
template <class I> void sort_byte_suffixes(pfp_ctx *c, const uint8_t *bytes, uint64_t N, SuffixOrderT<I> &out);

// ---------------------------------------------------------------- stage 2+3 (merge.hip)
struct DictIndex {        // word lookup over the dictionary (wordview.hpp): |D| / 16 + 8 d bytes, no per-position array
  DBuf<uint32_t> blk_word;   // [dsize / 64 + 1] word containing position 64 b
  DBuf<uint64_t> wend;       // [d+1] terminator position of word j (wend[d] = dsize-1)
  DBuf<uint32_t> lexrank;    // [d] 0-based lexicographic rank of word j
  DBuf<uint64_t> wslot_lex;  // [d] SA(D) slot (global) of the whole-word suffix of the word of lexicographic rank q
};
void build_dict_index(pfp_ctx *c, const Dictionary &D, DictIndex &ix);      // needs D.woff / D.wlen
inline WordView word_view(const Dictionary &D, const DictIndex &ix) {
  return WordView{D.bytes.p, ix.blk_word.p, ix.wend.p, (uint32_t)D.d, D.dsize};
}
// D.bytes/D.dsize given (words + 0x01, final 0x00): fills D.d, D.woff, D.wlen; at most max_words words expected
void word_table_from_bytes(pfp_ctx *c, Dictionary &D, uint64_t max_words);
template <class I> void compute_lexrank(pfp_ctx *c, const Dictionary &D, SuffixOrderT<I> &so, DictIndex &ix);
// d_wslot_all: `parts` arrays of d entries (u64), 1 + SA(D) slot of every word's first suffix or 0
void compute_lexrank_from_slots(pfp_ctx *c, const Dictionary &D, const uint64_t *d_wslot_all, uint32_t parts, DictIndex &ix);
template <class I> uint64_t count_slot_outputs(pfp_ctx *c, const Dictionary &D, const DictIndex &ix, const SuffixOrderT<I> &so, int w);

struct ParseBWT {         // outputs of bwtparse.c in HBM
  uint64_t P = 0;
  DBuf<uint32_t> ilist;    // [P+1]
  DBuf<uint8_t> bwlast;    // [P+1]
  DBuf<uint64_t> bwsai;    // [P+1] (only with SA flags)
  uint64_t rounds = 0;
};
// parse symbols are 1-based lexicographic ranks; occ_lex[r] = occurrences of rank r
// sa_given (optional): the suffix array of the parse + its end symbol, P + 1 entries, computed elsewhere
void parse_bwt(pfp_ctx *c, const uint32_t *parse_sym, uint64_t P, const uint8_t *last, const uint64_t *sai,
               const uint32_t *occ_lex, uint64_t d, ParseBWT &out, const uint32_t *sa_given = nullptr);

struct BwtOutputs {
  uint64_t n_out = 0;      // n+1
  uint8_t *d_bwt = nullptr;    // [n+1] device, caller-provided
  uint64_t *d_sa = nullptr;    // [n+1] device, caller-provided when flags != 0
  uint64_t hard_groups = 0, hard_chars = 0, hard_big_groups = 0, hard_max_chars = 0, hard_max_members = 0;
  uint64_t hard_minor_groups = 0, hard_minor_chars = 0;   // groups handled by majority fill; occurrences ranked for them
  // -s / -e without -S (sparse SA): the run boundaries of the emitted BWT slice - bit r of bmap = position r starts or ends
  // a run, bpre[k] = boundaries before bit 64 k, n_bound of them in all.  With d_sa == nullptr the SA values live in
  // sa_c[rank of the position among the boundaries] (16 bytes per run instead of 8 per text byte); with d_sa given they
  // are stored there, at boundary positions only.
  DBuf<uint64_t> bmap, bpre, sa_c;
  uint64_t n_bound = 0;
  // ... and, when the emitted slice is the whole BWT, where runs start / end (the .ssa / .esa positions) with their
  // prefix counts: the sampled files are then written from the bitmaps alone, without another pass over the BWT bytes
  DBuf<uint64_t> smap, spre, emap, epre;
  uint64_t n_starts = 0, n_ends = 0;
  uint64_t slice_n = 0;        // positions the maps cover (the emitted slice; a slice's two edge positions count as boundaries)
};
// where the SA value of BWT position i (relative to the slice the arrays describe) is found
struct SaView {
  const uint64_t *dense = nullptr;                       // dense[i], or
  const uint64_t *bmap = nullptr, *bpre = nullptr, *sa_c = nullptr;      // sa_c[rank of i among the set bits of bmap]
  // whole BWT only: run starts / ends as bitmaps with prefix counts (null: find them in the BWT bytes)
  const uint64_t *smap = nullptr, *spre = nullptr, *emap = nullptr, *epre = nullptr;
  uint64_t n_starts = 0, n_ends = 0, n_words = 0;
};
inline SaView sa_view(const BwtOutputs &o) {
  SaView v;
  if (o.sa_c.p) {
    v.bmap = o.bmap.p; v.bpre = o.bpre.p; v.sa_c = o.sa_c.p;
    v.smap = o.smap.p; v.spre = o.spre.p; v.emap = o.emap.p; v.epre = o.epre.p;
    v.n_starts = o.n_starts; v.n_ends = o.n_ends; v.n_words = ((o.slice_n ? o.slice_n : o.n_out) + 63) / 64;
  } else v.dense = o.d_sa;
  return v;
}
// emits BWT positions [out_lo,out_hi) into out.d_bwt[0..) / out.d_sa[0..) (default: everything)
template <class I>
void merge_bwt(pfp_ctx *c, const Dictionary &D, const DictIndex &ix, const SuffixOrderT<I> &so, const ParseBWT &pb,
               const uint32_t *occ_lex, int w, int flags, uint64_t expect_n_out, BwtOutputs &out, uint64_t out_lo = 0,
               uint64_t out_hi = ~0ull, uint64_t pos_base = 0, uint64_t n_out_global = 0);

// 5-byte packing and run sampling of finished device outputs
void pack5_dev(pfp_ctx *c, const uint64_t *vals, uint64_t cnt, uint8_t *out5);
void unpack5_dev(pfp_ctx *c, const uint8_t *in5, uint64_t cnt, uint64_t *vals);
// pairs (pos,sa) packed as 10 bytes each; returns pair count; out buffer allocated inside
uint64_t sample_runs_dev(pfp_ctx *c, const uint8_t *bwt, const SaView &sa, uint64_t n_out, bool run_end,
                         DBuf<uint8_t> &out10);
// the same for a slice whose maps a merge left (multi-GPU): pairs of the run starts (ends) at slice position r as
// <pos_base + r, SA value>; drop_edge: the slice's first (last) position is NOT a run start (end) after all - its
// neighbour in the adjacent slice carries the same byte.  out10 == nullptr: count only.
uint64_t sample_runs_maps(pfp_ctx *c, const SaView &sa, uint64_t slice_n, bool run_end, bool drop_edge, uint64_t pos_base, uint8_t *out10);
// the same in two steps over a slice of the BWT (multi-GPU): count the boundaries, then place the pairs.
// left / right = the BWT byte just before / after the slice, -1 at the ends of the whole BWT.
struct RunSampler {
  pfp_ctx *c; const uint8_t *bwt; uint64_t cnt; int left, right; bool run_end;
  uint64_t ntile = 0, pairs = 0;
  DBuf<uint32_t> tile_cnt; DBuf<uint64_t> tile_off;
  RunSampler(pfp_ctx *c, const uint8_t *bwt, uint64_t cnt, int left, int right, bool run_end);   // counts (one sync)
  void place(const SaView &sa, uint64_t pos_base, uint8_t *out10);                                 // SA value i belongs to bwt[i]
};


// ---------------------------------------------------------------- PFP_DEBUG=1 (validate.hip)
void validate_scan(pfp_ctx *c, const DBuf<uint64_t> &ends, uint64_t n_ends, uint64_t n, int w);
void validate_dictionary(pfp_ctx *c, const Dictionary &D, int w);
void validate_index(pfp_ctx *c, const Dictionary &D, const DictIndex &ix);
template <class I> void validate_suffix_order(pfp_ctx *c, const uint8_t *bytes, SuffixOrderT<I> &so, bool dict_mode, const char *what);
void validate_int_sa(pfp_ctx *c, const uint32_t *sym, const SuffixOrder &so);
void validate_lexrank(pfp_ctx *c, const Dictionary &D, const DictIndex &ix);
void validate_parse_bwt(pfp_ctx *c, const ParseBWT &pb);

}  // namespace pfp
