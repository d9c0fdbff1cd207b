// sufsort.hip -- GPU suffix sorting by prefix doubling on group-head ranks.
//
// Replaces the reference's serial induced sorting (gsa/gsacak.c: sacak_int :2497-2500 used by
// bwtparse.c:167; gsacak :2502-2522 used by pfbwt.cpp:495; sacak :2492-2495 used by
// simplebwt.c:77).  Induced sorting is one long left-to-right/right-to-left dependency chain
// (gsacak.c:262-289); by SURVEY.md 2.2-Q11 any correct sorter yields the same arrays, so the
// MI355X build sorts by prefix doubling: every round is a device-wide radix sort of the still
// unresolved suffixes keyed by (group head, rank of the suffix h positions on), O(log maxlen)
// rounds, all state resident in HBM.
//
// Dictionary mode compares 0x01-terminated strings: the key of a suffix never looks past its
// own word's terminator, a group whose whole string is already inside the sorted prefix is
// final, and equal strings keep one shared rank.  That rank equality *is* the "same suffix"
// test the reference derives from the LCP array (pfbwt.cpp:204-209), so no LCP array is built.
// Ties stay in position order (stable sorts), which reproduces gsacak's separator order
// (gsacak.c:1559-1561).
#include "kernels.hpp"
#include "prims.hpp"
#include "devutil.hpp"
#include <algorithm>
#include <cmath>
#include <optional>
#include <vector>
#include <cstdio>
#include <cstdlib>

namespace pfp {

enum : int { MODE_DICT = 0, MODE_PLAIN = 1 };

struct SufGeom {
  int mode; uint64_t N; WordView wv;            // dictionary mode: where the word of a position ends
  const uint32_t *sym = nullptr;                // plain mode on an integer string (unique smallest last symbol): the string
  const uint32_t *dist = nullptr;               // ... and, where the caller knows the symbols' frequencies (the parse), the distance from every
                                                // position to the next RARE symbol at or after it: see the parse's pivot rounds below
};
// length of the suffix string starting at i, terminator included
__device__ __forceinline__ uint64_t suf_len(const SufGeom &g, uint64_t i) {
  if (g.mode == MODE_DICT) return slen_of(g.wv, i) + 1;
  return g.N - i;
}

// First-round keys.  The bytes that occur in the dictionary get an order-preserving prefix-free
// (alphabetic) code whose lengths follow their frequencies - A,C,G,T of a FASTA text cost 2-3 bits,
// header characters and other rare bytes a dozen - and the codes of the first characters of every
// suffix are concatenated into kbits (<= 63) bits, cut off where the bits run out; the last bit
// of the key tells whether the suffix's terminator lies inside.  Concatenated alphabetic codes
// compare like the strings they encode, so the first device-wide sort orders the suffixes by a
// prefix of 20-odd characters for DNA text with any number of rare symbols mixed in (a fixed
// width code gave 16 characters for <= 14 distinct bytes and 12 beyond that), and no suffix is
// covered by fewer than hmin = kbits / (longest code) characters.
// lut[b] = (code << 6) | length.
struct KeyCode {
  uint32_t lut[256];
  int kbits, hmin, maxlen;
};

__global__ __launch_bounds__(256) void byte_histogram_kernel(const uint8_t *__restrict__ s, uint64_t N,
                                                             unsigned long long *__restrict__ hist /*[256]*/) {
  __shared__ uint32_t h[4][256];
  for (int q = threadIdx.x; q < 1024; q += 256) (&h[0][0])[q] = 0;
  __syncthreads();
  uint32_t *mine = h[threadIdx.x >> 6];
  for (uint64_t i = ((uint64_t)BID * 256 + threadIdx.x) * 16; i < N; i += (uint64_t)GDIM * 256 * 16) {
    uint4 v = *reinterpret_cast<const uint4 *>(s + i);     // buffer is zero padded past N
    uint32_t w4[4] = {v.x, v.y, v.z, v.w};
    const int nb = (N - i) >= 16 ? 16 : (int)(N - i);
#pragma unroll
    for (int q = 0; q < 16; q++)
      if (q < nb) atomicAdd(&mine[(w4[q >> 2] >> (8 * (q & 3))) & 0xff], 1u);
  }
  __syncthreads();
  const uint32_t t = h[0][threadIdx.x] + h[1][threadIdx.x] + h[2][threadIdx.x] + h[3][threadIdx.x];
  if (t) atomicAdd(&hist[threadIdx.x], (unsigned long long)t);
}

// Alphabetic code by recursive weight-balanced splitting of the symbols in byte order (every split
// point minimises |left weight - right weight|): within 2 bits of the entropy bound, always a
// complete prefix-free order-preserving code.  Rare symbols are floored at 1/4096 of the total so
// that no code grows beyond ~14 bits.
static void balanced_code(const std::vector<uint64_t> &w, int lo, int hi, uint32_t code, int len, std::vector<uint32_t> &codes,
                          std::vector<int> &lens) {
  if (hi - lo == 1) { codes[lo] = code; lens[lo] = len; return; }
  uint64_t tot = 0;
  for (int k = lo; k < hi; k++) tot += w[k];
  uint64_t acc = 0, best = ~0ull;
  int cut = lo + 1;
  for (int k = lo; k + 1 < hi; k++) {
    acc += w[k];
    const uint64_t l = acc, r = tot - acc, diff = l > r ? l - r : r - l;
    if (diff < best) { best = diff; cut = k + 1; }
  }
  balanced_code(w, lo, cut, code << 1, len + 1, codes, lens);
  balanced_code(w, cut, hi, (code << 1) | 1u, len + 1, codes, lens);
}
static KeyCode make_key_code(const uint64_t hist[256], double rep_hint = 0) {
  KeyCode kc{};
  std::vector<int> sym;
  for (int b2 = 0; b2 < 256; b2++) if (hist[b2] || b2 <= 1) sym.push_back(b2);       // 0x00 / 0x01 always coded
  uint64_t total = 0;
  for (int b2 : sym) total += hist[b2];
  const uint64_t floor_w = total / 4096 + 1;
  std::vector<uint64_t> w(sym.size());
  for (size_t k = 0; k < sym.size(); k++) w[k] = std::max<uint64_t>(hist[sym[k]], floor_w);
  std::vector<uint32_t> codes(sym.size());
  std::vector<int> lens(sym.size());
  if (sym.size() == 1) { codes[0] = 0; lens[0] = 1; }
  else balanced_code(w, 0, (int)sym.size(), 0u, 0, codes, lens);
  int maxlen = 1;
  for (int l : lens) maxlen = std::max(maxlen, l);
  if (maxlen > 24) {        // cannot happen with the weight floor; fixed width keeps the table format valid regardless
    const int bits = bits_for((uint64_t)sym.size() - 1);
    for (size_t k = 0; k < sym.size(); k++) { codes[k] = (uint32_t)k; lens[k] = bits; }
    maxlen = bits;
  }
  for (size_t k = 0; k < sym.size(); k++) kc.lut[sym[k]] = (codes[k] << 6) | (uint32_t)lens[k];
  // bytes that do not occur still need a valid entry (reads past the terminator are masked, not skipped)
  for (int b2 = 0; b2 < 256; b2++) if (!kc.lut[b2]) kc.lut[b2] = (uint32_t)maxlen;
  // Key width: every 8 bits less is one radix pass less over all N suffixes.  A key of kbits code bits
  // distinguishes about 2^(kbits * H / L) strings (H = order-0 entropy of the bytes, L = mean code
  // length); 2^8 times more than there are suffixes leaves well under 1 % of them tied by chance, and
  // the ties that remain - true repeats - are what the later rounds are for.  (253 MB FASTA: 47 bits,
  // 6 passes instead of 8, 2 M of 260 M suffixes tied after the first round: 48.3 -> 45.0 ms.)
  double H = 0, Lm = 0;
  for (size_t k = 0; k < sym.size(); k++) {
    if (!hist[sym[k]]) continue;
    const double pr = (double)hist[sym[k]] / (double)total;
    H -= pr * std::log2(pr); Lm += pr * lens[k];
  }
  // (a repetitive collection's dictionary holds about total / rep distinct contexts: the rest are family members no key width separates)
  const double need = std::log2(std::max((double)total / std::max(rep_hint, 1.0), 2.0)) + 6.0;
  kc.kbits = 63;
  for (int kb : {39, 47, 55}) if (Lm > 0 && kb * H / Lm >= need) { kc.kbits = kb; break; }
  { const char *e = getenv("PFP_KEYBITS"); if (e) kc.kbits = std::max(8, std::min(63, atoi(e))); }
  kc.hmin = std::max(1, std::min(32, kc.kbits / maxlen));
  kc.maxlen = maxlen;
  return kc;
}

// key of the suffix at i: codes of its first characters, most significant first, in kbits bits; then
// one bit "the terminator's code is inside".  `lut` may live in LDS (bulk kernel) or global memory.
__device__ __forceinline__ uint64_t packed_key_at(const uint8_t *__restrict__ s, uint64_t i, int kbits, const uint32_t *lut) {
  uint4 a = ld16u(s + i), b = ld16u(s + i + 16);
  const uint32_t w8[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  uint64_t k = 0;
  int pos = kbits;
  uint32_t term = 0;
#pragma unroll
  for (int j = 0; j < 32; j++) {
    if ((j == 16 || j == 20 || j == 24 || j == 28) && !__any(pos > 0)) break;     // every active lane has filled its key
    if (pos > 0) {
      const uint32_t c = (w8[j >> 2] >> (8 * (j & 3))) & 0xff;
      const uint32_t e = lut[c];
      const int l = (int)(e & 63u);
      const uint64_t cd = e >> 6;
      if (pos >= l) {
        pos -= l;
        k |= cd << pos;
        if (c <= kEndOfWord) { term = 1; pos = 0; }
      } else {
        k |= cd >> (l - pos);
        pos = 0;
      }
    }
  }
  return (k << 1) | term;
}

// Rank of the suffix at j.  rank[j] holds it when some round after the first refined j's group;
// otherwise the first round settled j and its rank is the first slot holding j's packed key: the
// bucket table narrows the range to the slots that share the key's top bits, a binary search
// over the sorted keys finishes (a handful of sectors instead of one scattered write per suffix).
//
// `settled` (dictionary mode, finbit != 0): j's group will never be refined again - it is a
// singleton or a set of identical strings.  An unresolved tie whose continuations are settled is
// itself a set of identical strings (their sorted prefixes reach the terminator), which is how
// write_back retires groups without gathering the suffix length of every member every round.
template <class I>
__device__ __forceinline__ I rank_at(const RankViewT<I> &L, uint64_t j, bool &settled) {
  const I r = L.rank ? L.rank[j] : IdxTraits<I>::kNone;
  if (L.skeys == nullptr || r != IdxTraits<I>::kNone) { settled = (r & L.finbit) != 0; return r & ~L.finbit; }
  settled = true;
  const uint64_t k = packed_key_at(L.bytes, j, L.kbits, L.lut);
  const uint32_t b = (uint32_t)(k >> L.shift);
  I lo = IdxTraits<I>::kNone - L.tab[L.T - 1 - b];
  I hi = (b + 1 < L.T) ? IdxTraits<I>::kNone - L.tab[L.T - 2 - b] : (I)L.N;
  while (lo < hi) {
    const I mid = lo + ((hi - lo) >> 1);
    if ((L.skeys[mid] & L.keymask) < k) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// merge record of position i: low byte = preceding char (1 = whole word, pfbwt.cpp:153; 0 = emits
// nothing, <= w long, pfbwt.cpp:151), high byte = occurrences of its word, 255 = "255 or more"
__device__ __forceinline__ uint32_t slot_record(const uint8_t *__restrict__ b, uint64_t i, uint32_t wd, const SlotPayloadSrc &P) {
  if (!(wd < P.wv.d && P.wv.wend[wd] - i > (uint64_t)P.w)) return 0u;
  const uint32_t pc = (i == 0) ? kEndOfWord : b[i - 1];
  const uint32_t occ = P.wocc[wd];
  return pc | ((occ < 255u ? occ : 255u) << 8);
}
// The same keys for ALL positions, without walking 20-odd characters per suffix: a workgroup lays the
// codes of 256 consecutive characters down as one bit stream in LDS (bit offset = prefix sum of the code
// lengths), and the key of position t is the kbits bits that start at t's offset, cut at its word's
// terminator (endpos) and at 32 characters - exactly what packed_key_at assembles character by
// character (checked against it under PFP_DEBUG).  224 positions per workgroup, 32 characters of overlap.
constexpr int kKeyPos = 224;
struct KeyStreamLds { uint32_t off[257]; uint32_t bs[200]; uint32_t wsum[4]; unsigned long long tmask[4]; uint32_t wbase; };
// workgroup-cooperative: returns the key (code bits << 1 | terminator flag) of position B0 + threadIdx.x
// for threadIdx.x < kKeyPos and position < N; every thread of the workgroup must call it
// Where a position's word ends is read off the block's own characters (the first byte <= 1 at or after it, looked for
// over the 32 characters a key can cover): no per-position length array is read.
// word_out (optional, needs wv): the word of position B0 + threadIdx.x - the word of the block's first position (one lookup
// per block) plus the terminators the ballots count before the lane.
__device__ __forceinline__ uint64_t block_stream_key(KeyStreamLds &L, const uint8_t *__restrict__ s, uint64_t N,
                                                     const KeyCode &kp, uint64_t B0, const WordView *view = nullptr,
                                                     uint32_t *word_out = nullptr) {
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const uint64_t pos = B0 + t;
  if (t < 200) L.bs[t] = 0;
  const uint32_t c = pos < N ? s[pos] : 0u;
  const uint32_t e = kp.lut[c];
  const uint32_t l = e & 63u, cd = e >> 6;
  uint32_t inc = l;
  for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
  if (lane == 63) L.wsum[wv] = inc;
  const unsigned long long tmine = __ballot(c <= (uint32_t)kEndOfWord);      // terminators (0x01) and the final 0x00 / the padding behind it
  if (lane == 0) L.tmask[wv] = tmine;
  if (word_out && t == 0) L.wbase = B0 < N ? word_of(*view, B0) : 0u;
  __syncthreads();
  if (word_out) {
    uint32_t before = (uint32_t)__popcll(tmine & ((1ull << lane) - 1ull));
    for (int q = 0; q < wv; q++) before += (uint32_t)__popcll(L.tmask[q]);
    *word_out = L.wbase + before;
  }
  uint32_t base = 0;
  for (int q = 0; q < wv; q++) base += L.wsum[q];
  const uint32_t p = base + inc - l;            // bit offset of this character's code
  L.off[t] = p;
  if (t == 255) L.off[256] = p + l;
  {
    const uint32_t w = p >> 5, o = p & 31u;
    if (o + l <= 32) atomicOr(&L.bs[w], cd << (32 - o - l));
    else {
      const uint32_t r = o + l - 32;            // bits that spill into the next word
      atomicOr(&L.bs[w], cd >> r);
      atomicOr(&L.bs[w + 1], (cd & ((1u << r) - 1u)) << (32 - r));
    }
  }
  __syncthreads();
  if (t >= kKeyPos || pos >= N) return 0;
  // characters up to and including the terminator of this position's word (33 = "more than a key can cover")
  uint32_t toterm = 33u;
  {
    const unsigned long long m0 = tmine >> lane;
    if (m0) toterm = (uint32_t)__builtin_ctzll(m0) + 1u;
    else if (wv < 3) { const unsigned long long m1 = L.tmask[wv + 1]; if (m1) toterm = (uint32_t)(64 - lane) + (uint32_t)__builtin_ctzll(m1) + 1u; }
    if (toterm > 33u) toterm = 33u;
  }
  const uint32_t nch = toterm < 32u ? toterm : 32u;
  const uint32_t avail = L.off[t + nch] - p;
  const uint32_t kb = (uint32_t)kp.kbits;
  const uint32_t take = avail < kb ? avail : kb;
  const uint32_t w = p >> 5, sh = p & 31u;
  const uint64_t hi = ((uint64_t)L.bs[w] << 32) | L.bs[w + 1];
  uint64_t x = hi << sh;
  if (sh) x |= (uint64_t)(L.bs[w + 2] >> (32 - sh));
  uint64_t k = x >> (64 - kb);
  if (take < kb) k &= ~((1ull << (kb - take)) - 1ull);
  const uint32_t term = (toterm <= 32u && L.off[t + toterm] - p <= kb) ? 1u : 0u;
  return (k << 1) | term;
}
template <class I>
__global__ __launch_bounds__(256) void init_keys_packed_kernel(const uint8_t *__restrict__ s, uint64_t N, KeyCode kp,
                                                               SlotPayloadSrc P, int paybits, uint64_t *__restrict__ key,
                                                               I *__restrict__ val, int idx_bits) {
  __shared__ KeyStreamLds L;
  const uint64_t B0 = (uint64_t)BID * kKeyPos;
  uint32_t wd = 0;
  uint64_t k = paybits ? block_stream_key(L, s, N, kp, B0, &P.wv, &wd) : block_stream_key(L, s, N, kp, B0);
  const uint64_t pos = B0 + threadIdx.x;
  if (threadIdx.x >= kKeyPos || pos >= N) return;
  if (idx_bits) { key[pos] = (k << idx_bits) | pos; return; }      // keys-only sort: the position rides in the low bits
  if (paybits) k |= (uint64_t)slot_record(s, pos, wd, P) << 48;
  key[pos] = k; val[pos] = (I)pos;
}
// sorted (key << idx_bits | position) words -> the sorted keys and the sorted positions
template <class I>
__global__ void split_keys_kernel(const uint64_t *__restrict__ comb, uint64_t n, int idx_bits, uint64_t *__restrict__ key,
                                  I *__restrict__ val) {
  const uint64_t a = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (a >= n) return;
  const uint64_t x = comb[a];
  key[a] = x >> idx_bits;
  val[a] = (I)(x & ((1ull << idx_bits) - 1ull));
}
// PFP_DEBUG: the bit-stream keys against the character-by-character ones
__global__ void check_keys_kernel(const uint8_t *__restrict__ s, uint64_t N, KeyCode kp, const uint64_t *__restrict__ key,
                                  uint64_t keymask, unsigned long long *__restrict__ bad, int idx_bits) {
  uint64_t i = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (i >= N) return;
  if (((key[i] >> idx_bits) & keymask) != packed_key_at(s, i, kp.kbits, kp.lut)) atomicAdd(bad, 1ull);
}
template <class I>
__global__ void init_keys_bytes_kernel(const uint8_t *__restrict__ s, uint64_t N, uint64_t *__restrict__ key,
                                       I *__restrict__ val) {
  uint64_t i = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (i >= N) return;
  key[i] = __builtin_bswap64(ld8u(s + i));         // buffer is zero padded past N
  val[i] = (I)i;
}
__global__ void init_keys_int_kernel(const uint32_t *__restrict__ s, uint64_t N, uint64_t *__restrict__ key,
                                     uint32_t *__restrict__ val) {
  uint64_t i = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (i >= N) return;
  uint64_t a = s[i], b = (i + 1 < N) ? s[i + 1] : 0;
  key[i] = (a << 32) | b; val[i] = (uint32_t)i;
}
// Integer strings (the parse): the first-round key holds two symbols and, where they are equal, how the
// run of that symbol ends.  Suffixes inside a run x^r y order by (y < x: shorter run first | y > x:
// longer run first), which plain doubling needs log2(r) rounds to find out - a parse of a text with
// a long N run holds a run of 300 k equal phrases (18 rounds; with the run key 8).
__global__ __launch_bounds__(256) void max_u32_kernel(const uint32_t *__restrict__ s, uint64_t N, uint32_t *__restrict__ out) {
  uint32_t m = 0;
  for (uint64_t i = (uint64_t)BID * 256 + threadIdx.x; i < N; i += (uint64_t)GDIM * 256) m = s[i] > m ? s[i] : m;
  for (int o = 32; o > 0; o >>= 1) { const uint32_t v = __shfl_down(m, o, 64); m = v > m ? v : m; }
  if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}
__global__ void run_marks_kernel(const uint32_t *__restrict__ s, uint32_t N, uint32_t *__restrict__ v) {
  uint32_t j = BID * blockDim.x + threadIdx.x;      // reversed index
  if (j >= N) return;
  const uint32_t i = N - 1 - j;
  v[j] = (i == N - 1 || s[i] != s[i + 1]) ? j : 0u;
}
__global__ void init_keys_int_run_kernel(const uint32_t *__restrict__ s, uint32_t N, const uint32_t *__restrict__ pm, int sb,
                                         uint64_t *__restrict__ key, uint32_t *__restrict__ val) {
  uint32_t i = BID * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const uint64_t a = s[i], b = (i + 1 < N) ? s[i + 1] : 0;
  const int eb = 64 - 2 * sb;
  uint64_t e = 0;
  if (b == a && i + 1 < N) {
    const uint32_t re = N - 1 - pm[N - 1 - i];              // last position of the run that holds i
    const uint64_t cap = (1ull << (eb - 1)) - 1;
    uint64_t r = (uint64_t)re - i + 1;
    if (r > cap) r = cap;
    const bool up = re + 1 < N && s[re + 1] > a;            // the run is followed by a larger symbol
    e = up ? ((1ull << (eb - 1)) | (cap - r)) : r;
  }
  key[i] = (a << (64 - sb)) | (b << (64 - 2 * sb)) | e;
  val[i] = i;
}
template <class I>
__global__ void iota_kernel(I *p, uint64_t n) {
  uint64_t i = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (I)i;
}

// key of an unresolved suffix: (its group's head slot, 1 + rank of the suffix h further on).
// The suffix position and its group travel with the active list (active_place_kernel), so the only
// random access is rank[i+h].  In dictionary mode an unresolved suffix is always longer than the
// sorted prefix h (otherwise write_back would have retired it), so i+h stays inside its word.
template <class I>
__global__ void build_keys_kernel(SufGeom g, uint64_t m, uint64_t h, const I *__restrict__ act_i,
                                  const I *__restrict__ act_grp, RankViewT<I> L, int nb,
                                  typename IdxTraits<I>::DKey *__restrict__ key, I *__restrict__ val) {
  using K = typename IdxTraits<I>::DKey;
  uint64_t a = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (a >= m) return;
  const I i = act_i[a];
  const K grp = act_grp[a];
  bool settled = false;
  const uint64_t nxt = (g.mode == MODE_DICT || (uint64_t)i + h < g.N) ? (uint64_t)rank_at(L, (uint64_t)i + h, settled) + 1 : 0;
  key[a] = (grp << nb) | (K)nxt;
  val[a] = i | (settled ? L.finbit : (I)0);
}

// segmented variant of a doubling round: the unresolved suffixes are already grouped (the active
// list is in slot order), so only the 32-bit "next" key has to be sorted, inside every group
template <class I>
__global__ void group_starts_kernel(uint64_t m, const I *__restrict__ act_grp, uint8_t *__restrict__ gs) {
  uint64_t a = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (a < m) gs[a] = (a == 0 || act_grp[a] != act_grp[a - 1]) ? 1 : 0;
}
__global__ void build_keys32_kernel(SufGeom g, uint64_t m, uint64_t h, const uint32_t *__restrict__ act_i,
                                    RankViewT<uint32_t> L, uint32_t *__restrict__ key, uint32_t *__restrict__ val) {
  uint64_t a = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (a >= m) return;
  const uint32_t i = act_i[a];
  bool settled = false;
  key[a] = (g.mode == MODE_DICT || (uint64_t)i + h < g.N) ? rank_at(L, (uint64_t)i + h, settled) + 1u : 0u;
  val[a] = i | (settled ? L.finbit : 0u);
}
// Small segments.  The unresolved suffixes of a dictionary of variants come in families of a handful of members
// (a phrase and the variants that agree with it over the first-round prefix): a device-wide radix sort of
// (group, order key) moves every element 7 times to reorder it inside a group of six.  Here every element
// finds its place directly: the group ids and keys of 256 list positions plus kSmallSeg on either side are
// staged in LDS, an element walks back to the start of its group and forward over its members, counting the
// keys that sort before its own (stable: equal keys keep their list order).  A group longer than kSmallSeg does
// not fit the window: the kernel then only raises *overflow and the caller sorts that round with the library.
// Also writes the group-start flags gs[] the head detection of the round reads.
// (Tried in round 3: the same kernel with a 2048-member window and 1024 positions per workgroup for the families of a
//  1000-copy collection, in place of the library's segmented sort.  Bit-exact, and 5x slower: 12.6 GB 58 -> 308 ms for these
//  sorts.  Counting is quadratic in the group size and the count loop is a chain of dependent LDS reads with four waves per
//  SIMD to hide them.  A real sort of whole groups in LDS - block radix, bitonic - was tried too: no faster than the library's
//  0.024 ns per element and round)
constexpr uint32_t kSmallSeg = 64;
template <class I, class KT>
__global__ __launch_bounds__(256) void seg_small_sort_kernel(uint64_t m, const I *__restrict__ act_grp,
                                                             const KT *__restrict__ key, const I *__restrict__ val,
                                                             uint8_t *__restrict__ gs, KT *__restrict__ keyo,
                                                             I *__restrict__ valo, uint32_t *__restrict__ overflow,
                                                             uint8_t *__restrict__ big) {
  constexpr int K = (int)kSmallSeg, W = 256 + 2 * K;
  __shared__ KT lkey[W];
  __shared__ I lgrp[W];
  const uint64_t B = (uint64_t)BID * 256;
  for (int idx = threadIdx.x; idx < W; idx += 256) {
    const uint64_t j = B + idx;                   // list position + K
    const bool ok = j >= (uint64_t)K && j - K < m;
    lkey[idx] = ok ? key[j - K] : (KT)0;
    lgrp[idx] = ok ? act_grp[j - K] : IdxTraits<I>::kNone;      // outside the list: no group
  }
  __syncthreads();
  const uint64_t a = B + threadIdx.x;
  if (a >= m) return;
  const int la = (int)threadIdx.x + K;
  const I g = lgrp[la];
  gs[a] = lgrp[la - 1] != g ? 1 : 0;
  // a group of more than K members: with `big` its elements are passed through in place and flagged (the caller
  // sorts just those), without it the round is the library's.  Groups are contiguous, so two probes tell: the
  // position K back, and - once the start is known - the position K past it.
  const KT ka = lkey[la];
  bool too_long = lgrp[la - K] == g;      // the group starts K or more positions back
  int ls = la;
  if (!too_long) {
    while (lgrp[ls - 1] == g) ls--;       // (stops within K - 1 steps)
    too_long = ls + K < W && lgrp[ls + K] == g;      // more than K members
  }
  if (too_long) {
    atomicOr(overflow, 1u);
    if (big) { big[a] = 1; keyo[a] = ka; valo[a] = val[a]; }
    return;
  }
  uint32_t r = 0;
  for (int j = ls; j < ls + K && lgrp[j] == g; j++) {
    const KT kj = lkey[j];
    r += (kj < ka || (kj == ka && j < la)) ? 1u : 0u;
  }
  if (big) big[a] = 0;
  const uint64_t s0 = a - (uint64_t)(la - ls);
  keyo[s0 + r] = ka;
  valo[s0 + r] = val[a];
}
// the elements of the groups seg_small_sort_kernel passed through: (group, key) words for one sort of just those,
// and the way back - the sorted side list has the groups in list order, so its j-th element belongs at idx[j]
template <class I>
__global__ void big_seg_gather_kernel(uint64_t nb, const I *__restrict__ idx, const I *__restrict__ act_grp, const uint32_t *__restrict__ keyo,
                                      const I *__restrict__ valo, uint64_t *__restrict__ k64, I *__restrict__ v) {
  const uint64_t j = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (j >= nb) return;
  const I a = idx[j];
  k64[j] = ((uint64_t)act_grp[a] << 32) | keyo[a];
  v[j] = valo[a];
}
template <class I>
__global__ void big_seg_scatter_kernel(uint64_t nb, const I *__restrict__ idx, const uint64_t *__restrict__ k64, const I *__restrict__ v,
                                       uint32_t *__restrict__ keyo, I *__restrict__ valo) {
  const uint64_t j = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (j >= nb) return;
  const I a = idx[j];
  keyo[a] = (uint32_t)k64[j];
  valo[a] = v[j];
}
__global__ void seg_end_kernel(uint32_t ng, uint32_t m, const uint32_t *__restrict__ seg_begin, uint32_t *__restrict__ seg_end,
                               uint32_t *__restrict__ maxlen) {
  uint32_t k = BID * blockDim.x + threadIdx.x;
  uint32_t len = 0;
  if (k < ng) { uint32_t e = (k + 1 < ng) ? seg_begin[k + 1] : m; seg_end[k] = e; len = e - seg_begin[k]; }
  for (int o = 32; o > 0; o >>= 1) { uint32_t v = __shfl_down(len, o, 64); len = v > len ? v : len; }
  if ((threadIdx.x & 63) == 0 && len) atomicMax(maxlen, len);
}
template <class I>
__global__ void heads32_kernel(uint64_t m, const uint8_t *__restrict__ gs, const uint32_t *__restrict__ key,
                               const I *__restrict__ aslot, uint8_t *__restrict__ hd, I *__restrict__ hv) {
  uint64_t a = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (a >= m) return;
  bool h = gs[a] || key[a] != key[a - 1];
  hd[a] = h ? 1 : 0;
  hv[a] = h ? aslot[a] : (I)0;
}

// ---- pivot rounds for the suffixes of the PARSE of a collection of near-identical sequences.
// After the first round a group is one place of the genome in all its copies; two members agree until one of them meets a phrase
// that only its own copy has (a variant word: a rare symbol).  Comparing every member with the group's FIRST member only halves
// the group per round - that pivot has a variant of its own a few phrases on, and everybody who agrees with the consensus up to
// there ties at that position (round 2 tried it: 13 rounds instead of doubling's 8).  The frequencies of the symbols tell which
// member to compare with: the one whose next rare symbol is FARTHEST (dist[], one backward scan over the parse).  Every other
// member then leaves the pivot at its own variant - a different (offset, symbol) for each - and one round places 99 % of a
// group where doubling needs log2(offset) rounds (64 copies: 160 K unresolved -> 1.1 K after one round; first-member pivot: 81 K).
// Order key, as for the dictionary's pivot rounds: smaller than the pivot: ascending offset, then symbol; the pivot's class; greater:
// descending offset, then symbol - two 32-bit halves of a 64-bit key sorted inside every group.
constexpr uint32_t kIntPivCap = 1024;      // symbols compared per member and round
// smallest list such a round is tried on (PFP_PARSE_PIVOT_MIN: tests run it on small inputs; 0 = never)
static uint64_t parse_pivot_min() {
  const char *e = getenv("PFP_PARSE_PIVOT_MIN");
  if (!e) return 1u << 16;
  const uint64_t v = strtoull(e, nullptr, 10);
  return v ? v : ~0ull;
}
template <class I>
__global__ void ipivot_select_kernel(uint64_t m, uint64_t N, uint64_t h, const I *__restrict__ act_i, const I *__restrict__ act_grp,
                                     const uint32_t *__restrict__ dist, unsigned long long *__restrict__ best) {
  const uint64_t a = (uint64_t)BID * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  unsigned long long v = 0;
  uint64_t g = ~0ull;
  if (a < m) {
    const uint64_t i = act_i[a];
    g = act_grp[a];
    const uint32_t dv = i + h < N ? dist[i + h] : 0u;
    v = ((unsigned long long)dv << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)a);      // ties: the first member
  }
  // the members of a group are neighbours: maximum over the run of equal groups that ends at this lane, one atomic per run and wave
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long u = __shfl_up(v, o, 64);
    const uint64_t ug = __shfl_up(g, o, 64);
    if (lane >= o && ug == g && u > v) v = u;
  }
  const uint64_t gn = __shfl_down(g, 1, 64);
  if (a < m && (lane == 63 || gn != g)) atomicMax(&best[g], v);
}
template <class I>
__global__ void ipivot_keys_kernel(uint64_t m, uint64_t N, uint64_t h, uint32_t cap, const uint32_t *__restrict__ sym,
                                   const I *__restrict__ act_i, const I *__restrict__ act_grp,
                                   const unsigned long long *__restrict__ best, uint64_t *__restrict__ key, I *__restrict__ val) {
  const uint64_t a = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (a >= m) return;
  const uint64_t i = act_i[a];
  const uint64_t piv = act_i[0xFFFFFFFFu - (uint32_t)(best[act_grp[a]] & 0xFFFFFFFFull)];
  uint64_t k = (uint64_t)(cap + 1) << 32;      // the pivot itself, and whoever equals it for cap symbols
  if (i != piv) {
    // four symbols a step (two unaligned 16-byte loads; round 4: one symbol a step was a chain of ~20 dependent 4-byte loads per
    // member - 52 ms of the 12.6 GB chain once the phrases got shorter); the last steps before the end of the string one by one
    bool done = false;
    for (uint32_t q = 0; q < cap && !done; q += 4) {
      const uint64_t pi = i + h + q, pp = piv + h + q;
      uint32_t x[4], y[4];
      if (pi + 4 <= N && pp + 4 <= N) {
        const uint4 X = ld16u(reinterpret_cast<const uint8_t *>(sym + pi)), Y = ld16u(reinterpret_cast<const uint8_t *>(sym + pp));
        if (X.x == Y.x && X.y == Y.y && X.z == Y.z && X.w == Y.w) continue;
        x[0] = X.x; x[1] = X.y; x[2] = X.z; x[3] = X.w; y[0] = Y.x; y[1] = Y.y; y[2] = Y.z; y[3] = Y.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) { x[j] = pi + j < N ? sym[pi + j] : 0u; y[j] = pp + j < N ? sym[pp + j] : 0u; }
      }
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (!done && q + j < cap && x[j] != y[j]) {
          const uint32_t qq = q + (uint32_t)j;
          k = ((uint64_t)(x[j] < y[j] ? qq : 2 * cap + 2 - qq) << 32) | x[j];
          done = true;
        }
    }
  }
  key[a] = k;
  val[a] = (I)i;
}
template <class I>
__global__ void heads64seg_kernel(uint64_t m, const uint8_t *__restrict__ gs, const uint64_t *__restrict__ key,
                                  const I *__restrict__ aslot, uint8_t *__restrict__ hd, I *__restrict__ hv) {
  uint64_t a = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (a >= m) return;
  bool h = gs[a] || key[a] != key[a - 1];
  hd[a] = h ? 1 : 0;
  hv[a] = h ? aslot[a] : (I)0;
}
// distance to the next rare symbol: marks over the reversed string (0 = common; else reversed index + 1), running maximum, difference
__global__ void rare_marks_kernel(const uint32_t *__restrict__ s, uint32_t N, const uint32_t *__restrict__ occ, uint32_t n_sym,
                                  uint32_t below, uint32_t *__restrict__ v) {
  uint32_t j = BID * blockDim.x + threadIdx.x;      // reversed index
  if (j >= N) return;
  const uint32_t x = s[N - 1 - j];
  const bool rare = x == 0 || x > n_sym || occ[x - 1] < below;
  v[j] = rare ? j + 1 : 0u;
}
__global__ void rare_dist_kernel(uint32_t N, const uint32_t *__restrict__ pm, uint32_t *__restrict__ dist) {
  uint32_t i = BID * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const uint32_t j = N - 1 - i, mk = pm[j];
  dist[i] = mk ? j + 1 - mk : 0xFFFFFFFFu;      // (the unique last symbol is rare: every position has one ahead)
}

// Compaction of the active list: kept suffixes and kept group heads counted per 256 list positions
// from 16-byte loads of the flag bytes, tiny scans over those sums, then one thread per list position
// places its (slot, suffix, group) triple (coalesced reads, consecutive writes).  Replaces an N-long
// inclusive scan into a 4-byte array and its re-read.
constexpr int kTile = 256;
__global__ __launch_bounds__(256) void active_count_kernel(const uint8_t *__restrict__ keep, const uint8_t *__restrict__ hd,
                                                           uint64_t m, uint32_t *__restrict__ tile_keep,
                                                           uint32_t *__restrict__ tile_heads) {
  // thread = 16 flags, 16 threads = one tile of 256
  const uint64_t t = (uint64_t)BID * 256 + threadIdx.x;
  const uint64_t base = t * 16;
  uint32_t k[4] = {0, 0, 0, 0}, h[4] = {0, 0, 0, 0};
  if (base < m) { load_flags16(keep, base, m, k); load_flags16(hd, base, m, h); }
  uint32_t ck = __popc(k[0]) + __popc(k[1]) + __popc(k[2]) + __popc(k[3]);
  uint32_t ch = __popc(k[0] & h[0]) + __popc(k[1] & h[1]) + __popc(k[2] & h[2]) + __popc(k[3] & h[3]);
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) { ck += __shfl_down(ck, o, 16); ch += __shfl_down(ch, o, 16); }
  if ((threadIdx.x & 15) == 0 && base < m) { tile_keep[t >> 4] = ck; tile_heads[t >> 4] = ch; }
}
template <class I>
__global__ __launch_bounds__(256) void active_place_kernel(const uint8_t *__restrict__ keep, uint64_t m,
                                                           const uint32_t *__restrict__ tile_keep,
                                                           const I *__restrict__ tile_off,
                                                           const I *__restrict__ aslot, const I *__restrict__ val,
                                                           const I *__restrict__ newhead, I finbit,
                                                           I *__restrict__ aslot2, I *__restrict__ act_i,
                                                           I *__restrict__ act_grp) {
  __shared__ uint32_t ws[4];
  if ((uint64_t)BID * kTile >= m) return;      // a workgroup of the padded last grid row
  if (tile_keep[BID] == 0) return;       // nothing kept in these 256 positions (most tiles after the first round)
  const uint64_t a = (uint64_t)BID * kTile + threadIdx.x;
  const bool k = a < m && keep[a];
  const unsigned long long mask = __ballot(k);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) ws[wv] = (uint32_t)__popcll(mask);
  __syncthreads();
  if (!k) return;
  I pos = tile_off[BID] + (I)__popcll(mask & ((1ull << lane) - 1ull));
  for (int q = 0; q < wv; q++) pos += ws[q];
  aslot2[pos] = aslot ? aslot[a] : (I)a; act_i[pos] = val[a] & ~finbit; act_grp[pos] = newhead[a];      // no list yet: slot == index
}

template <class I, class K>
__global__ void heads_kernel(uint64_t m, const K *__restrict__ key, const I *__restrict__ aslot,
                             uint8_t *__restrict__ hd, I *__restrict__ hv) {
  uint64_t a = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (a >= m) return;
  bool h = (a == 0) || key[a] != key[a - 1];
  hd[a] = h ? 1 : 0;
  hv[a] = h ? aslot[a] : (I)0;
}

// First round of dictionary mode (slot == index): group heads, and the bucket table of the sorted
// keys for rank_at: the first slot of every occupied top-bits bucket marks its reversed entry.
// Per workgroup of 256 slots also the index of its last head (0 = none, slot 0 is always a head): a
// max-scan over those 4 bytes per 256 slots gives every workgroup of write_back0 its carry-in, and
// the head of each slot is found from the head flags with ballots - no N-long scan of head indices.
template <class I>
__global__ __launch_bounds__(256) void heads0_kernel(uint64_t m, const uint64_t *__restrict__ key, uint64_t keymask, int shift,
                                                     uint32_t T, uint8_t *__restrict__ hd, I *__restrict__ tile_last,
                                                     I *__restrict__ tab) {
  __shared__ I wl[4];
  if ((uint64_t)BID * 256 >= m) return;      // a workgroup of the padded last grid row
  const uint64_t a = (uint64_t)BID * 256 + threadIdx.x;
  bool h = false;
  if (a < m) {
    const uint64_t k = key[a] & keymask, kprev = a ? (key[a - 1] & keymask) : ~k;
    h = k != kprev;
    hd[a] = h ? 1 : 0;
    if (a == 0 || (k >> shift) != (kprev >> shift)) tab[T - 1 - (uint32_t)(k >> shift)] = IdxTraits<I>::kNone - (I)a;
  }
  const unsigned long long mask = __ballot(h);
  if ((threadIdx.x & 63) == 0)
    wl[threadIdx.x >> 6] = mask ? (I)(BID * 256ull + (threadIdx.x & ~63) + (63 - __clzll((long long)mask))) : (I)0;
  __syncthreads();
  if (threadIdx.x == 0) {
    I v = wl[0];
    for (int q = 1; q < 4; q++) v = wl[q] > v ? wl[q] : v;
    tile_last[BID] = v;
  }
}
template <class I>
__global__ void fill_kernel(I *p, uint64_t n, I v) {
  uint64_t i = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
// First round of dictionary mode: sa and grp are the sorted values / scanned heads themselves
// (streaming copies); rank[] is written only for the suffixes that stay unresolved.
template <class I>
__global__ __launch_bounds__(256) void write_back0_kernel(uint64_t m, const I *__restrict__ val,
                                                          const I *__restrict__ tile_scan,
                                                          const uint8_t *__restrict__ hd, const uint64_t *__restrict__ key0,
                                                          I *__restrict__ rank,
                                                          I *__restrict__ grp, uint8_t *__restrict__ keep, int write_ranks) {
  __shared__ I wl[4];
  if ((uint64_t)BID * 256 >= m) return;      // a workgroup of the padded last grid row
  const uint64_t a = (uint64_t)BID * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // all loads first: nothing below the barrier waits on memory except the carry-in
  const bool in = a < m;
  const bool h = in && hd[a];
  const bool hnext = in && (a + 1 == m || hd[a + 1]);
  const I i = in ? val[a] : (I)0;
  const uint32_t flagbit = in ? (uint32_t)(key0[a] & 1ull) : 0u;
  const I carry = BID ? tile_scan[BID - 1] : (I)0;
  const unsigned long long mask = __ballot(h);
  const I wbase = (I)(BID * 256ull + (threadIdx.x & ~63));
  if (lane == 0) wl[wv] = mask ? wbase + (63 - __clzll((long long)mask)) : (I)0;
  __syncthreads();
  if (!in) return;
  // head of slot a: the last head flag at or before it - in its wave, else in an earlier wave of the
  // workgroup, else the carry-in (last head of all earlier workgroups)
  const unsigned long long upto = mask & (~0ull >> (63 - lane));
  I head;
  if (upto) head = wbase + (63 - __clzll((long long)upto));
  else {
    head = carry;
    for (int q = 0; q < wv; q++) head = wl[q] > head ? wl[q] : head;
  }
  grp[a] = head;       // sa[] is the sorted value array itself (moved into place by the host)
  const bool single = h && hnext;
  const bool fin = !single && flagbit;       // the terminator is inside the key: the tied strings are identical
  const bool k = !single && !fin;
  if (k && write_ranks) rank[i] = head;       // lazy mode: rank[] of an unresolved suffix is filled in when a doubling round asks
  keep[a] = k ? 1 : 0;
}

// dense fallback after the first round: many suffixes stay unresolved, so most lookups would need
// the search; scatter the settled ranks once instead (the pre-lazy behaviour)
template <class I>
__global__ void scatter_settled_kernel(uint64_t N, const I *__restrict__ sa, const I *__restrict__ grp,
                                       const uint8_t *__restrict__ keep, I finbit, I *__restrict__ rank) {
  uint64_t t = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (t < N && !keep[t]) rank[sa[t]] = grp[t] | finbit;
}

// Pivot rounds.  The unresolved groups of a dictionary of near-identical phrases (a collection of
// variants) are families: most members are identical to each other up to the terminator or differ
// from each other at one position a few dozen bytes on.  Doubling needs log2(lcp) more rounds for
// them - each a device-wide sort plus a scattered rank update.  A pivot round instead compares
// every member directly with its group's first member P (= sa[head slot]), from the offset already
// known equal, 16 bytes per step, for at most `cap` bytes, and sorts by what it found:
//     smaller than P : (first difference at lcp, own byte there)   ascending lcp, then byte
//     identical to P up to the terminator
//     greater than P : descending lcp, then ascending byte
// Two members that differ from P at different offsets are thereby ordered against each other, two
// that differ at the same offset by their bytes; the same (side, lcp, byte) is a still unresolved
// tie.  A member still equal to P after cap bytes joins P's class unresolved (no settled bit) and
// vetoes the settling of that whole class (pivot_veto_kernel).  Order-key layout (kPivBits bits):
constexpr int kPivBits = 23;
constexpr uint32_t kPivEq = 1u << 21;
constexpr uint32_t kPivCapMax = 8192;
// (Tried in round 3: eight lanes per member, 128 contiguous bytes of the member's string per load instruction, the pivot's
//  bytes coalesced, ballot for the first differing chunk.  Same lines fetched, 3x fewer address-coalescer cycles - and 17 %
//  SLOWER (c3 6.8 -> 7.9 ms, 12.6 GB 107 -> 120 ms): the kernel is bound by the rate at which HBM serves random lines
//  (~54 G/s, tools/microbench/gather.hip; 112 M members x ~1.6 lines + their pivots' = 3.9 ms at that rate), not by
//  its instruction stream, and eight times the waves only add ballots.  One thread per member stays.
//  Also tried: 64 bytes per step instead of 32, the second half's loads in flight with the first's - c3 6.5 -> 7.1 ms,
//  12.6 GB 101 -> 102.5 ms: members that differ or end inside the first 32 bytes pay for a second line they do not need.)
template <class I>
__global__ void build_keys_pivot_kernel(const uint8_t *__restrict__ s, uint64_t m, uint64_t from, uint32_t cap,
                                        const I *__restrict__ act_i, const I *__restrict__ act_grp,
                                        const I *__restrict__ sa, I finbit, uint64_t *__restrict__ key,
                                        uint32_t *__restrict__ key32, I *__restrict__ val, int both = 0) {
  uint64_t a = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (a >= m) return;
  const I i = act_i[a], grp = act_grp[a];
  const I piv = sa[grp];
  uint32_t ok = kPivEq;
  I settled = finbit;
  if (i != piv) {
    settled = 0;
    const uint8_t *pa = s + i + from, *pp = s + piv + from;
    // skim 32 bytes per step until the strings differ or the member ends, then look inside that chunk
    uint32_t off = 0;
    uint4 xa0, xa1, xp0, xp1;
    bool hit = false;
    for (; off < cap; off += 32) {
      xa0 = ld16u(pa + off); xa1 = ld16u(pa + off + 16); xp0 = ld16u(pp + off); xp1 = ld16u(pp + off + 16);
      const uint32_t diff = (xa0.x ^ xp0.x) | (xa0.y ^ xp0.y) | (xa0.z ^ xp0.z) | (xa0.w ^ xp0.w) | (xa1.x ^ xp1.x) |
                            (xa1.y ^ xp1.y) | (xa1.z ^ xp1.z) | (xa1.w ^ xp1.w);
#define PFP_LT2(v) (((v) - 0x02020202u) & ~(v) & 0x80808080u)
      const uint32_t term = PFP_LT2(xa0.x) | PFP_LT2(xa0.y) | PFP_LT2(xa0.z) | PFP_LT2(xa0.w) | PFP_LT2(xa1.x) | PFP_LT2(xa1.y) |
                            PFP_LT2(xa1.z) | PFP_LT2(xa1.w);
#undef PFP_LT2
      if (diff | term) { hit = true; break; }
    }
    if (hit) {
      const uint64_t x[4] = {(uint64_t)xa0.x | ((uint64_t)xa0.y << 32), (uint64_t)xa0.z | ((uint64_t)xa0.w << 32),
                             (uint64_t)xa1.x | ((uint64_t)xa1.y << 32), (uint64_t)xa1.z | ((uint64_t)xa1.w << 32)};
      const uint64_t y[4] = {(uint64_t)xp0.x | ((uint64_t)xp0.y << 32), (uint64_t)xp0.z | ((uint64_t)xp0.w << 32),
                             (uint64_t)xp1.x | ((uint64_t)xp1.y << 32), (uint64_t)xp1.z | ((uint64_t)xp1.w << 32)};
      bool done = false;
#pragma unroll
      for (int q = 0; q < 4 && !done; q++) {
        const uint64_t tb = (x[q] - 0x0202020202020202ull) & ~x[q] & 0x8080808080808080ull;   // bytes < 2; lowest flag exact
        if (x[q] == y[q]) { if (tb) { settled = finbit; done = true; } continue; }
        const int fd = __builtin_ctzll(x[q] ^ y[q]) >> 3;
        const int ft = tb ? (__builtin_ctzll(tb) >> 3) : 8;
        if (ft < fd) { settled = finbit; done = true; continue; }        // both end before they differ: identical
        const uint32_t bx = (uint32_t)(x[q] >> (8 * fd)) & 0xffu, by = (uint32_t)(y[q] >> (8 * fd)) & 0xffu;
        const uint32_t lcp = off + 8 * q + fd;
        ok = bx < by ? ((lcp << 8) | bx) : ((2u << 21) | ((kPivCapMax - 1 - lcp) << 8) | bx);
        if (bx <= kEndOfWord) settled = finbit;      // the member ends here: whoever shares this key is the same string
        done = true;
      }
    }
    // not done: still equal to P after cap bytes -> P's class, without the settled bit
  }
  if (key32) key32[a] = ok;                 // segmented sort: the group is the segment, only the order key is sorted
  if (!key32 || both) key[a] = ((uint64_t)grp << kPivBits) | ok;
  val[a] = i | settled;
}
template <class I>
__global__ void veto_clear_kernel(uint64_t m, const I *__restrict__ newhead, uint8_t *__restrict__ veto) {
  uint64_t a = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (a < m) veto[newhead[a]] = 0;
}
// a member of P's class without the settled bit (equal to P for cap bytes, unknown beyond): nobody in
// that class may settle this round
template <class I>
__global__ void pivot_veto_kernel(uint64_t m, const uint64_t *__restrict__ key, const uint32_t *__restrict__ key32,
                                  const I *__restrict__ val, const I *__restrict__ newhead, I finbit,
                                  uint8_t *__restrict__ veto) {
  uint64_t a = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (a >= m) return;
  const uint32_t ok = key32 ? key32[a] : ((uint32_t)key[a] & ((1u << kPivBits) - 1));
  if (ok == kPivEq && !(val[a] & finbit)) veto[newhead[a]] = 1;
}

// write the refined order back and decide which suffixes stay unresolved.
// sorted_len = prefix length that is sorted after this round.
// Round 0 (key0 != nullptr) decides "whole string inside the sorted prefix" from the packed key
// itself (some field holds the terminator code 1): no gather.
template <class I, class K>
__global__ void write_back_kernel(SufGeom g, uint64_t m, uint64_t sorted_len, const I *__restrict__ aslot,
                                  const I *__restrict__ val, const I *__restrict__ newhead,
                                  const uint8_t *__restrict__ hd, I finbit, const K *__restrict__ prevkey,
                                  int prevshift, const I *__restrict__ prevgrp, const uint8_t *__restrict__ veto,
                                  const uint32_t *__restrict__ wstart_bits, I *__restrict__ sa,
                                  I *__restrict__ rank, I *__restrict__ wordrank, I *__restrict__ grp, uint8_t *__restrict__ keep) {
  uint64_t a = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (a >= m) return;
  const I iv = val[a], i = iv & ~finbit;
  const I slot = aslot[a];
  sa[slot] = i;
  grp[slot] = newhead[a];       // slot-side copy of the group head: the merge never gathers rank[]
  bool single = hd[a] && (a + 1 == m || hd[a + 1]);
  bool fin = false;
  if (!single && g.mode == MODE_DICT) {
    if (finbit) fin = (iv & finbit) != 0 && !(veto && veto[newhead[a]]);   // continuation settled / identical to the pivot
    else fin = suf_len(g, i) <= sorted_len;            // N >= 2^31: no spare bit, gather the length
  }
  const bool k = !single && !fin;
  // rank[i] already holds the old group head: a suffix that stays unresolved in a group that kept
  // its head needs no write - the scattered 4-byte store is the expensive access of this kernel
  const I old = prevkey ? (I)(prevkey[a] >> prevshift) : (prevgrp ? prevgrp[a] : ~newhead[a]);
  // Pivot rounds (lazy_bytes != null) do not read rank[], so the rank of a suffix that settles there is
  // written only if somebody will ask for it - the whole words, whose ranks order the dictionary
  // (compute_lexrank); every other settled rank is filled in by repair_ranks_kernel if a doubling
  // round follows after all.  One byte read replaces the scattered 4-byte store.
  bool wr = k ? old != newhead[a] : true;
  // (a whole word is a singleton group: only a suffix alone in its group can be one; which positions start a word is
  // read from a bitmap of N bits - it stays cache resident where the dictionary bytes would not)
  if (wstart_bits) wr = !k && single && ((wstart_bits[i >> 5] >> (i & 31)) & 1u);
  if (wr && rank) rank[i] = newhead[a] | (k ? (I)0 : finbit);
  else if (wr && wordrank && wstart_bits) wordrank[word_of(g.wv, i)] = newhead[a] | finbit;      // (a settled whole word: see `wr` above)
  keep[a] = k ? 1 : 0;
}

// ranks that pivot rounds left unwritten: every slot re-ordered after the first round that is not in
// the active list any more holds a settled suffix
// lazy ranks: the unresolved suffixes get theirs (= their group's head slot) from the active list
// bit i set: position i starts a word (i == 0 or the byte before it is a terminator)
__global__ __launch_bounds__(256) void word_start_bits_kernel(const uint8_t *__restrict__ s, uint64_t N, uint32_t *__restrict__ bits) {
  const uint64_t wi = (uint64_t)BID * 256 + threadIdx.x;      // 32 positions per thread
  const uint64_t p0 = wi * 32;
  if (p0 >= N) return;
  uint32_t m = 0;
  for (int k = 0; k < 32; k++) {
    const uint64_t i = p0 + k;
    if (i < N && (i == 0 || s[i - 1] == kEndOfWord)) m |= 1u << k;
  }
  bits[wi] = m;
}
template <class I>
__global__ void active_ranks_kernel(uint64_t m, const I *__restrict__ act_i, const I *__restrict__ act_grp, I *__restrict__ rank) {
  uint64_t a = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (a < m) rank[act_i[a]] = act_grp[a];
}
template <class I>
__global__ void mark_slots_kernel(uint64_t m, const I *__restrict__ aslot, uint8_t *__restrict__ flag) {
  uint64_t a = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (a < m) flag[aslot[a]] = 1;
}
template <class I>
__global__ void repair_ranks_kernel(uint64_t N, const uint8_t *__restrict__ refined, const uint8_t *__restrict__ active,
                                    const I *__restrict__ sa, const I *__restrict__ grp, I finbit,
                                    I *__restrict__ rank) {
  uint64_t t = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (t < N && refined[t] && !active[t]) rank[sa[t]] = grp[t] | finbit;
}

// The last few thousand suffixes.  A round costs a dozen launches and three host round trips however little is left,
// and the tail of a sort is many such rounds (-p 200: ten rounds for the last 40 K of 63 M suffixes).  Once at most
// kFinishMax suffixes are unresolved - in groups of at most kFinishGrp - every one of them is ranked inside its group by
// comparing the strings themselves (terminated words, or the integer string up to the first difference): lt = members
// that sort before it, eq = identical members before it in the list; nothing is written until no comparison ran past
// kFinishCmp bytes and no group was too long, so a refusal leaves the round machinery where it was.
constexpr uint64_t kFinishMax = 1u << 17;
constexpr int kFinishGrp = 64;
constexpr uint32_t kFinishCmp = 8192;
__device__ __forceinline__ int finish_cmp_dict(const uint8_t *__restrict__ s, uint64_t i, uint64_t j, uint32_t *ovf) {
  for (uint32_t k = 0; k < kFinishCmp; k += 8) {
    const uint64_t x = ld8u(s + i + k), y = ld8u(s + j + k);
    const uint64_t tb = (x - 0x0202020202020202ull) & ~x & 0x8080808080808080ull;   // bytes < 2 of x; lowest flag exact
    if (x == y) { if (tb) return 0; continue; }
    const int fd = __builtin_ctzll(x ^ y) >> 3;
    const int ft = tb ? (__builtin_ctzll(tb) >> 3) : 8;
    if (ft < fd) return 0;       // both end before they differ
    const uint32_t bx = (uint32_t)(x >> (8 * fd)) & 0xffu, by = (uint32_t)(y >> (8 * fd)) & 0xffu;
    return bx < by ? -1 : 1;
  }
  *ovf = 1;
  return 0;
}
__device__ __forceinline__ int finish_cmp_int(const uint32_t *__restrict__ sym, uint64_t i, uint64_t j, uint32_t *ovf) {
  for (uint32_t k = 0; k < kFinishCmp / 4; k++) {      // (the sentinel ends the shorter suffix with a difference)
    const uint32_t x = sym[i + k], y = sym[j + k];
    if (x != y) return x < y ? -1 : 1;
  }
  *ovf = 1;
  return 0;
}
template <class I>
__global__ void finish_rank_kernel(SufGeom g, const uint8_t *__restrict__ s, uint64_t m, const I *__restrict__ act_i,
                                   const I *__restrict__ act_grp, uint32_t *__restrict__ lt, uint32_t *__restrict__ eq,
                                   uint32_t *__restrict__ gstart, uint32_t *__restrict__ overflow) {
  // eight lanes per member, each comparing it with every eighth other member of its group (one lane per member was a chain of up
  // to 63 dependent string comparisons: 0.27 ms a launch for a few ten thousand members - the kernel's time is that chain's latency)
  const uint64_t t = (uint64_t)BID * blockDim.x + threadIdx.x;
  const uint64_t a = t >> 3;
  const int l8 = (int)(t & 7);
  if (a >= m) return;
  const I grp = act_grp[a];
  uint64_t gs = a, ge = a + 1;
  while (gs > 0 && a - gs < (uint64_t)kFinishGrp && act_grp[gs - 1] == grp) gs--;
  while (ge < m && ge - gs <= (uint64_t)kFinishGrp && act_grp[ge] == grp) ge++;
  if (ge - gs > (uint64_t)kFinishGrp || (gs > 0 && act_grp[gs - 1] == grp)) { if (l8 == 0) atomicOr(overflow, 1u); return; }
  const uint64_t ia = act_i[a];
  uint32_t nlt = 0, neq = 0, ovf = 0;
  for (uint64_t b = gs + (uint64_t)l8; b < ge; b += 8) {
    if (b == a) continue;
    if (*(volatile uint32_t *)overflow) break;      // somebody met a case for the rounds: no point in finishing the comparisons
    const uint64_t ib = act_i[b];
    const int cmp = g.mode == MODE_DICT ? finish_cmp_dict(s, ib, ia, &ovf) : finish_cmp_int(g.sym, ib, ia, &ovf);
    if (cmp < 0) nlt++;
    else if (cmp == 0 && b < a) neq++;
  }
  // (all eight lanes of a member arrive here together: everything above the loop is the same for them)
  uint32_t packed = nlt | (neq << 8) | (ovf << 16);      // <= 63 each
  packed += __shfl_xor(packed, 1, 64); packed += __shfl_xor(packed, 2, 64); packed += __shfl_xor(packed, 4, 64);
  if (l8 != 0) return;
  if (packed >> 16) atomicOr(overflow, 1u);
  lt[a] = packed & 0xffu; eq[a] = (packed >> 8) & 0xffu; gstart[a] = (uint32_t)gs;
}
template <class I>
__global__ void finish_write_kernel(SufGeom g, uint64_t m, const I *__restrict__ aslot, const I *__restrict__ act_i, const uint32_t *__restrict__ lt,
                                    const uint32_t *__restrict__ eq, const uint32_t *__restrict__ gstart, I finbit,
                                    I *__restrict__ sa, I *__restrict__ grp, I *__restrict__ rank, I *__restrict__ wordrank) {
  const uint64_t a = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (a >= m) return;
  const uint64_t gs = gstart[a];
  const I i = act_i[a], slot = aslot[gs + lt[a] + eq[a]], head = aslot[gs + lt[a]];
  sa[slot] = i;
  grp[slot] = head;            // identical strings share their group's first slot
  if (rank) rank[i] = head | finbit;
  else if (wordrank && ((uint64_t)i == 0 || g.wv.bytes[(uint64_t)i - 1] == kEndOfWord)) wordrank[word_of(g.wv, i)] = head | finbit;
}

// after the first round, pivot rounds are tried while the groups are families (average size up to
// kPivotAvg) and each one at least halves the unresolved set; at most kPivotCap bytes per comparison
constexpr uint32_t kPivotAvg = 1024u;
static const uint32_t kPivotCap = []() { const char *e = getenv("PFP_PIVOT_CAP"); uint32_t v = e ? (uint32_t)atoll(e) : 512u; return v > kPivCapMax ? kPivCapMax : v; }();

// the first round leaves more than N/kLazyRatio suffixes unresolved -> scatter all ranks after all
constexpr uint64_t kLazyRatio = 8ull;

template <class I>
RankViewT<I> rank_view(const SuffixOrderT<I> &so) {
  return RankViewT<I>{so.rank.p, so.skeys.p, so.tab.p, so.lut.p, so.bytes, so.N, so.kbits, so.shift, so.T, so.finbit, so.keymask, so.wordrank.p};
}
template RankViewT<uint32_t> rank_view(const SuffixOrderT<uint32_t> &);
template RankViewT<uint64_t> rank_view(const SuffixOrderT<uint64_t> &);

template <class I>
static void doubling(pfp_ctx *c, SufGeom g, DBuf<uint64_t> &key, DBuf<I> &val, uint64_t h0, SuffixOrderT<I> &out,
                     int key0_bits = 64, bool dict_keys = false, uint64_t n_elems = ~0ull, int idx_bits = 0) {
  // precondition: key/val hold the initial (prefix key, position) pairs of the N suffixes to sort: all
  // NP positions, or (range mode, n_elems given) the ones of this rank's key range
  using K = typename IdxTraits<I>::DKey;        // key of a doubling round: (group head, 1 + rank of the continuation)
  constexpr bool kWide = sizeof(I) == 8;
  const uint64_t NP = g.N;
  const bool range_mode = n_elems != ~0ull;
  const uint64_t N = range_mode ? n_elems : NP;
  const int TB = 256;
  out.N = N; out.NP = NP; out.range = range_mode; out.complete = true;
  out.rounds = 0;
  if (N == 0) { out.sa.alloc(c, 1); out.grp.alloc(c, 8); out.rank.alloc(c, 1); return; }
  const bool lazy = dict_keys;      // dictionary mode: out.lut/bytes/kbits are set by the caller
  // ---- first round: one device-wide sort of all N (key, position) pairs, ping-ponging between the two buffer
  //      pairs (no third copy inside the library); afterwards keyo/valo hold the sorted pairs
  DBuf<uint64_t> keyo(c, N);
  DBuf<I> valo(c, N);
  // the first round's sort: rocPRIM's onesweep, or (PFP_OWN_SORT=1) the hand-written MSD sort of radix.hip - bit-exact, and measured
  // slower on every workload of round 4 (configs[2] 5.0 vs 3.8 ms, configs[1] 14.0 vs 13.4): the alphabetic code's bits carry ~0.8 bit
  // of entropy each, so a partition by key BITS needs three passes where 16 uniform bits would do, and each pass reads its input
  // twice
  static const bool lib_sort = getenv("PFP_OWN_SORT") == nullptr;
  SortTag first_tag(g.mode == MODE_DICT ? "dictionary, first round" : "parse, first round");
  if (idx_bits) {
    // keys-only first round: the words are (key << idx_bits | position), 16 bytes per element and pass instead of 24;
    // the sort is stable on the key bits alone, so ties stay in position order; one streaming pass splits the result
    if (lib_sort) sort_keys_db(c, key, keyo, N, idx_bits, idx_bits + key0_bits);
    else msd_sort_keys_db(c, key, keyo, N, idx_bits, idx_bits + key0_bits);
    KScope ks(c, "pfp::split_keys_kernel", N * (16 + sizeof(I)));
    hipLaunchKernelGGL(split_keys_kernel<I>, gdim(cdiv(N, TB)), gdim(TB), 0, c->stream, key.p, N, idx_bits, keyo.p, valo.p);
  } else {
    if (lib_sort) sort_pairs_db(c, key, keyo, val, valo, N, 0, key0_bits);
    else msd_sort_pairs_db<I>(c, key, keyo, val, valo, N, 0, key0_bits);
    std::swap(key, keyo); std::swap(val, valo);
  }
  SortTag later_tag(g.mode == MODE_DICT ? "dictionary, later rounds" : "parse, later rounds");
  if (lazy) { key.release(); val.release(); }      // dictionary mode: later rounds sort the (much smaller) unresolved set
  // (allocated only now: while the first sort holds its four buffers - 32 bytes per suffix in the wide build - nothing else
  //  of that size is live: a 4.8 GB dictionary peaked at 206 GB with the group array next to them, 168 GB without)
  DBuf<uint8_t> hd(c, N + 1), keep(c, N);
  out.grp.alloc(c, N + 8);
  // active list (slot, suffix, group) of the unresolved suffixes and per-round scratch, `cap` elements each;
  // dictionary mode allocates them when the first round has told how many suffixes stay unresolved
  DBuf<I> aslot, aslot2, hv, newhead, act_i, act_grp;
  uint64_t list_cap = 0;
  auto alloc_lists = [&](uint64_t cap) {
    aslot.alloc(c, cap); aslot2.alloc(c, cap); hv.alloc(c, cap); newhead.alloc(c, cap); act_i.alloc(c, cap); act_grp.alloc(c, cap);
    list_cap = cap;
  };
  // a share of the suffix array (multi-GPU) never runs a doubling round - it stops where one would be needed - and finds
  // its whole words by looking at its own slots (gather_slots_range): no rank per dictionary position is kept
  const bool no_rank = range_mode;
  if (!lazy) {
    out.sa.alloc(c, N);
    alloc_lists(N);
    hipLaunchKernelGGL(iota_kernel<I>, gdim(cdiv(N, TB)), gdim(TB), 0, c->stream, aslot.p, N);
  }
  uint64_t m = N, h = h0;
  bool first = true;
  if (lazy) {
    const int tb = std::min(key0_bits, std::max(8, std::min(24, bits_for(N) - 5)));
    out.shift = key0_bits - tb;
    out.T = 1u << tb;
    out.tab.alloc(c, out.T);
    hipLaunchKernelGGL(fill_kernel<I>, gdim(cdiv(out.T, TB)), gdim(TB), 0, c->stream, out.tab.p, (uint64_t)out.T,
                       (I)(IdxTraits<I>::kNone - (I)N));
  }
  const int nb = bits_for(N);           // key of a later round = (group head << nb) | (1 + rank of the continuation)
  const int keybits = 2 * nb;
  static const bool no_finflag = getenv("PFP_NO_FINFLAG") != nullptr;      // tests: force the length-gather path
  // (the flag rides in the top bit of a POSITION: it needs positions below 2^31 in the 32-bit build - all NP of them, also when
  //  only a share of N < 2^31 suffixes is sorted here)
  out.finbit = (g.mode == MODE_DICT && (kWide || NP < (1ull << 31)) && !no_finflag) ? IdxTraits<I>::kTop : (I)0;
  constexpr bool use_segsort = true;
  DBuf<uint8_t> gs;
  DBuf<uint32_t> k32, k32o, segb, sege, nseg_d;
  DBuf<uint64_t> nsel_d;
  DBuf<K> dkey, dkeyo;          // wide build: 128-bit doubling keys (the 32-bit build keeps them in key/keyo)
  bool seg_round = false;       // the keys of this round live in k32o (segmented path) instead of keyo
  bool pivot_round = false;     // the keys of this round are pivot order keys (build_keys_pivot_kernel)
  bool dbl_round = false;       // the keys of this round are doubling keys (type K)
  bool pivot_ok = true;
  bool ipiv_round = false, ipiv_ok = true;      // parse: pivot rounds with the farthest-rare-symbol pivot (keys in keyo, sorted inside the groups)
  bool lazy_pending = false;    // dictionary mode: rank[] of the suffixes settled by the first round not scattered (yet)
  DBuf<uint8_t> veto, keep0;
  constexpr bool lazy_pivot_ranks = true;
  bool ranks_stale = false;     // pivot rounds skipped rank[] of settled suffixes that are not whole words
  bool active_stale = false;    // ... and the first round / pivot rounds skipped rank[] of the suffixes that stay unresolved
  DBuf<uint32_t> wstart_bits;   // pivot rounds: which positions start a word (their ranks order the dictionary)
  // lazy ranks are possible when pivot rounds (which never read rank[]) can follow the first round
  const bool lazy_active = lazy && lazy_pivot_ranks && out.finbit && g.mode == MODE_DICT && kPivotCap >= 16 && (uint64_t)nb + kPivBits <= 64;
  // rank[] (one entry per dictionary position: 4 / 8 bytes each, 38 GB for a 4.8 GB dictionary in the wide build) exists
  // from the start only where a round may read it.  With lazy ranks nothing reads it before the first doubling round - and a
  // dictionary of word families never gets there - so it is allocated when (if) that round comes (late_rank); until then the
  // whole words, whose ranks order the dictionary, report to wordrank[d].  (PFP_DEBUG validates every rank: eager.)
  const bool late_rank = !no_rank && lazy_active && !c->debug;
  bool rank_alloc = false;
  I *rank_p = nullptr;
  auto alloc_rank = [&]() {
    out.rank.alloc(c, NP);
    if (lazy) PFP_HIP(hipMemsetAsync(out.rank.p, 0xff, NP * sizeof(I), c->stream));
    rank_p = out.rank.p; rank_alloc = true;
  };
  if (!no_rank && !late_rank) alloc_rank();
  if (late_rank) {
    out.wordrank.alloc(c, (uint64_t)g.wv.d + 1);
    PFP_HIP(hipMemsetAsync(out.wordrank.p, 0xff, ((uint64_t)g.wv.d + 1) * sizeof(I), c->stream));
  }
  I *const wordrank_p = late_rank ? out.wordrank.p : (I *)nullptr;
  auto repair_ranks = [&](uint64_t m_active, const I *aslot_list) {
    if (!ranks_stale) return;
    DBuf<uint8_t> act(c, N);
    act.zero();
    if (m_active) hipLaunchKernelGGL(mark_slots_kernel<I>, gdim(cdiv(m_active, TB)), gdim(TB), 0, c->stream, m_active, aslot_list, act.p);
    KScope ks(c, "pfp::repair_ranks_kernel", N * 6);
    hipLaunchKernelGGL(repair_ranks_kernel<I>, gdim(cdiv(N, TB)), gdim(TB), 0, c->stream, N, keep0.p, act.p, out.sa.p, out.grp.p,
                       out.finbit, out.rank.p);
    ranks_stale = false;
  };
  DBuf<uint32_t> ovf_d;
  auto seg_bufs = [&]() {
    if (!gs.p) { gs.alloc(c, list_cap + 1); k32.alloc(c, list_cap); k32o.alloc(c, list_cap); nseg_d.alloc(c, 2); nsel_d.alloc(c, 1); ovf_d.alloc(c, 1); }
  };
  // k32/val -> k32o/valo, every group of the (grouped) active list sorted by its 32-bit key; false: some group has
  // more than kSmallSeg members (nothing usable was written)
  // 32-bit index build: the groups of more than kSmallSeg members are passed through in place, flagged, and finished
  // by ONE library sort of just their elements (a few long families no longer send the whole round to the library)
  constexpr bool big_side = true;
  DBuf<uint8_t> bigf;
  auto seg_small_sort = [&](uint64_t mm) -> bool {
    ovf_d.zero();
    const bool side = big_side && !kWide;
    if (side && (!bigf.p || bigf.n < mm + 16)) bigf.alloc(c, list_cap + 16);
    { KScope ks(c, "pfp::seg_small_sort_kernel", mm * (8 + 3 * sizeof(I) + 1));
      hipLaunchKernelGGL((seg_small_sort_kernel<I, uint32_t>), gdim(cdiv(mm, TB)), gdim(TB), 0, c->stream, mm, act_grp.p, k32.p, val.p, gs.p, k32o.p,
                         valo.p, ovf_d.p, side ? bigf.p : (uint8_t *)nullptr); }
    if (read_scalar(c, ovf_d.p) == 0) return true;
    if (!side) return false;
    PFP_HIP(hipMemsetAsync(bigf.p + mm, 0, 16, c->stream));      // (the flag reader looks at 16 bytes at a time)
    const uint64_t nbig = count_flags(c, bigf.p, mm);
    if (nbig * 2 > mm) return false;                // mostly long groups: the round is the library's after all
    DBuf<I> idx(c, nbig + 1), sv(c, nbig), sva(c, nbig);
    DBuf<uint64_t> sk(c, nbig), ska(c, nbig), cnt(c, 1);
    select_index<I>(c, bigf.p, idx.p, cnt.p, mm);
    hipLaunchKernelGGL(big_seg_gather_kernel<I>, gdim(cdiv(nbig, TB)), gdim(TB), 0, c->stream, nbig, idx.p, act_grp.p, k32o.p, valo.p, sk.p, sv.p);
    sort_pairs_db(c, sk, ska, sv, sva, nbig, 0, 32 + nb);
    hipLaunchKernelGGL(big_seg_scatter_kernel<I>, gdim(cdiv(nbig, TB)), gdim(TB), 0, c->stream, nbig, idx.p, sk.p, sv.p, k32o.p, valo.p);
    PFP_HIP(hipGetLastError());
    return true;
  };
  // (n_groups: the compaction that built the list counted its group heads - no read-back of the selection's count)
  auto seg_setup = [&](uint64_t mm, uint64_t n_groups, uint32_t &ng, uint32_t &maxlen) {      // segments = groups of the (grouped) active list
    seg_bufs();
    if (!segb.p) { segb.alloc(c, list_cap + 1); sege.alloc(c, list_cap + 1); }
    hipLaunchKernelGGL(group_starts_kernel<I>, gdim(cdiv(mm, TB)), gdim(TB), 0, c->stream, mm, act_grp.p, gs.p);
    select_index<uint32_t>(c, gs.p, segb.p, nsel_d.p, mm);
    PFP_HIP(hipMemsetAsync(nseg_d.p + 1, 0, 4, c->stream));
    ng = (uint32_t)n_groups;
    if (c->debug) PFP_REQUIRE(read_scalar(c, nsel_d.p) == n_groups, PFP_EHIP, "group count of the active list differs from the compaction's");
    hipLaunchKernelGGL(seg_end_kernel, gdim(cdiv(ng, TB)), gdim(TB), 0, c->stream, ng, (uint32_t)mm, segb.p, sege.p, nseg_d.p + 1);
    maxlen = read_scalar(c, nseg_d.p + 1);
  };
  DBuf<uint32_t> tile_keep, tile_heads;
  DBuf<I> tile_off, tile_hoff;
  uint32_t piv_cap = kPivotCap;  // bytes compared per member in the next pivot round
  bool small_failed = false;     // a direct-placement attempt met a group longer than its window
  bool finisher_ok = true;       // the comparison finisher has not refused yet
  for (;;) {
    DBuf<I> tile_last, tile_scan;      // first round of dictionary mode: last head per 256 slots, and its running maximum
    if (first && lazy) {
      tile_last.alloc(c, cdiv64(m, 256)); tile_scan.alloc(c, cdiv64(m, 256));
      { KScope ks(c, "pfp::heads0_kernel", m * 13);
        hipLaunchKernelGGL(heads0_kernel<I>, gdim(cdiv(m, TB)), gdim(TB), 0, c->stream, m, keyo.p, out.keymask, out.shift, out.T,
                           hd.p, tile_last.p, out.tab.p); }
      inclusive_max<I>(c, out.tab.p, out.tab.p, out.T);
      inclusive_max<I>(c, tile_last.p, tile_scan.p, cdiv64(m, 256));
    } else if (ipiv_round) {
      KScope ks(c, "pfp::heads64seg_kernel", m * 18);
      hipLaunchKernelGGL(heads64seg_kernel<I>, gdim(cdiv(m, TB)), gdim(TB), 0, c->stream, m, gs.p, keyo.p, aslot.p, hd.p, hv.p);
    } else if (seg_round) {
      KScope ks(c, "pfp::heads32_kernel", m * 14);
      hipLaunchKernelGGL(heads32_kernel<I>, gdim(cdiv(m, TB)), gdim(TB), 0, c->stream, m, gs.p, k32o.p, aslot.p, hd.p, hv.p);
    } else if (dbl_round && kWide) {
      KScope ks(c, "pfp::heads_kernel", m * 25);
      hipLaunchKernelGGL((heads_kernel<I, K>), gdim(cdiv(m, TB)), gdim(TB), 0, c->stream, m, dkeyo.p, aslot.p, hd.p, hv.p);
    } else {
      KScope ks(c, "pfp::heads_kernel", m * 17);
      hipLaunchKernelGGL((heads_kernel<I, uint64_t>), gdim(cdiv(m, TB)), gdim(TB), 0, c->stream, m, keyo.p, aslot.p, hd.p, hv.p);
    }
    const bool round0 = first && lazy;
    first = false;
    if (!round0) inclusive_max<I>(c, hv.p, newhead.p, m);
    if (round0) {
      { KScope ks(c, "pfp::write_back0_kernel", m * (4 + 4 + 1 + 8 + 4 + 1));
        hipLaunchKernelGGL(write_back0_kernel<I>, gdim(cdiv(m, TB)), gdim(TB), 0, c->stream, m, valo.p, tile_scan.p, hd.p, keyo.p,
                           rank_p, out.grp.p, keep.p, (lazy_active || no_rank) ? 0 : 1); }
      if (lazy_active) active_stale = true;
      // the sorted keys and the sorted positions stay with the result; later rounds sort the (smaller) active set elsewhere
      out.skeys = std::move(keyo);
      out.sa = std::move(valo);
    } else {
      if (pivot_round) {
        if (!veto.p) veto.alloc(c, N);
        // only the entries of this round's groups are looked at: clear those (m bytes at most), not all N
        hipLaunchKernelGGL(veto_clear_kernel<I>, gdim(cdiv(m, TB)), gdim(TB), 0, c->stream, m, newhead.p, veto.p);
        hipLaunchKernelGGL(pivot_veto_kernel<I>, gdim(cdiv(m, TB)), gdim(TB), 0, c->stream, m, keyo.p,
                           seg_round ? k32o.p : (const uint32_t *)nullptr, valo.p, newhead.p, out.finbit, veto.p);
      }
      // previous group head of the element now at a: the high part of its sort key, or (segmented
      // rounds keep every element inside its segment) the group of list position a.  The first round
      // of plain mode has neither: everything is written.
      const bool have_prev = out.rounds > 0;
      KScope ks(c, "pfp::write_back_kernel", m * (13 + 4 + 13));
      const uint8_t *vetop = pivot_round ? veto.p : (const uint8_t *)nullptr;
      if (pivot_round && lazy_pivot_ranks && !wstart_bits.p) {
        wstart_bits.alloc(c, cdiv64(NP, 32));
        hipLaunchKernelGGL(word_start_bits_kernel, gdim(cdiv(cdiv64(NP, 32), TB)), gdim(TB), 0, c->stream, out.bytes, NP, wstart_bits.p);
      }
      const uint32_t *lazyb = (pivot_round && lazy_pivot_ranks) ? wstart_bits.p : (const uint32_t *)nullptr;
      const I *prevgrp = (have_prev && seg_round) ? act_grp.p : (const I *)nullptr;
      if (dbl_round && kWide && !seg_round)
        hipLaunchKernelGGL((write_back_kernel<I, K>), gdim(cdiv(m, TB)), gdim(TB), 0, c->stream, g, m, h, aslot.p, valo.p, newhead.p, hd.p,
                           out.finbit, have_prev ? dkeyo.p : (const K *)nullptr, nb, prevgrp, vetop, lazyb, out.sa.p, rank_p,
                           wordrank_p, out.grp.p, keep.p);
      else
        hipLaunchKernelGGL((write_back_kernel<I, uint64_t>), gdim(cdiv(m, TB)), gdim(TB), 0, c->stream, g, m, h, aslot.p, valo.p, newhead.p,
                           hd.p, out.finbit, (have_prev && !seg_round) ? keyo.p : (const uint64_t *)nullptr, pivot_round ? kPivBits : nb,
                           prevgrp, vetop, lazyb, out.sa.p, rank_p, wordrank_p, out.grp.p, keep.p);
      if (pivot_round && lazy_pivot_ranks) { ranks_stale = true; active_stale = true; }
    }
    uint64_t m2 = 0, ngrp = 0;
    {
      // kept suffixes / kept group heads per tile -> offsets -> placement
      const uint64_t ntile = cdiv64(m, kTile);
      if (!tile_keep.p) { tile_keep.alloc(c, cdiv64(N, kTile) + 1); tile_heads.alloc(c, cdiv64(N, kTile) + 1);
                          tile_off.alloc(c, cdiv64(N, kTile) + 1); tile_hoff.alloc(c, cdiv64(N, kTile) + 1); }
      PFP_HIP(hipMemsetAsync(tile_keep.p + ntile, 0, 4, c->stream));
      PFP_HIP(hipMemsetAsync(tile_heads.p + ntile, 0, 4, c->stream));
      KScope ks(c, "pfp::active_place_kernel", m * 2 + 0);      // (+ active_count_kernel and the two tile scans)
      hipLaunchKernelGGL(active_count_kernel, gdim((unsigned)cdiv64(ntile, 16)), gdim(256), 0, c->stream, keep.p, hd.p, m,
                         tile_keep.p, tile_heads.p);
      exclusive_sum_u32_to<I>(c, tile_keep.p, tile_off.p, ntile + 1);
      exclusive_sum_u32_to<I>(c, tile_heads.p, tile_hoff.p, ntile + 1);
      PFP_HIP(hipMemcpyAsync(c->h_scalars, tile_off.p + ntile, sizeof(I), hipMemcpyDeviceToHost, c->stream));
      PFP_HIP(hipMemcpyAsync(c->h_scalars + 1, tile_hoff.p + ntile, sizeof(I), hipMemcpyDeviceToHost, c->stream));
      if (round0) {      // the lists are sized by what the first round left unresolved
        sync(c);
        I t0, t1;
        memcpy(&t0, c->h_scalars, sizeof(I)); memcpy(&t1, c->h_scalars + 1, sizeof(I));
        m2 = t0; ngrp = t1;
        if (m2) alloc_lists(m2);
      }
      if (!round0 || m2)
        hipLaunchKernelGGL(active_place_kernel<I>, gdim((unsigned)ntile), gdim(256), 0, c->stream, keep.p, m, tile_keep.p, tile_off.p,
                           round0 ? (const I *)nullptr : aslot.p, round0 ? out.sa.p : valo.p, round0 ? out.grp.p : newhead.p, out.finbit,
                           aslot2.p, act_i.p, act_grp.p);
      PFP_HIP(hipGetLastError());
    }
    if (!round0) {
      sync(c);
      I t0, t1;
      memcpy(&t0, c->h_scalars, sizeof(I)); memcpy(&t1, c->h_scalars + 1, sizeof(I));
      m2 = t0; ngrp = t1;
    }
    if (round0 && m2) {
      // whether the settled ranks get scattered after all is decided when (if) a doubling round
      // first needs them: pivot rounds read the strings, not rank[]
      lazy_pending = true;
      keep0.alloc(c, N);
      PFP_HIP(hipMemcpyAsync(keep0.p, keep.p, N, hipMemcpyDeviceToDevice, c->stream));
      out.n_refined = m2;
      if (out.paybits) {       // the merge wants to know which slots were re-ordered after this round
        out.refined.alloc(c, N + 16);
        PFP_HIP(hipMemcpyAsync(out.refined.p, keep.p, N, hipMemcpyDeviceToDevice, c->stream));
        PFP_HIP(hipMemsetAsync(out.refined.p + N, 0, 16, c->stream));
      }
    }

    static const bool trace_rounds = getenv("PFP_TRACE_ROUNDS") != nullptr;
    if (trace_rounds)
      fprintf(stderr, "[pfp] doubling N=%llu round=%llu h=%llu m=%llu -> %llu unresolved in %llu groups%s\n", (unsigned long long)N,
              (unsigned long long)out.rounds, (unsigned long long)h, (unsigned long long)m, (unsigned long long)m2,
              (unsigned long long)ngrp, seg_round ? " (seg)" : (pivot_round ? " (pivot)" : ""));
    // a pivot round that did not at least halve the unresolved set: what is left are members equal to
    // their pivot for the whole comparison window.  The window then grows fourfold (512 -> 2 K -> 8 K bytes) as long
    // as the bytes such a round may read - every member the full window - stay below 256 per dictionary byte; the
    // variants of long phrases settle there (-p 200: 3.3 M suffixes went through nine doubling rounds instead).
    // After the widest window the rest is doubling's business.
    if (ipiv_round && m2 * 4 > m) ipiv_ok = false;      // a parse pivot round that left more than a quarter: doubling from here
    if (pivot_round && m2 * 2 > m) {
      const uint32_t next_cap = std::min<uint32_t>(piv_cap * 4, kPivCapMax - 16);
      if (piv_cap < kPivCapMax - 16 && piv_cap >= 16 && m2 * (uint64_t)next_cap < N * 128) piv_cap = next_cap;
      else pivot_ok = false;
    }
    std::swap(aslot, aslot2);
    m = m2;
    if (m == 0) break;
    PFP_REQUIRE(h < 2 * NP, PFP_EHIP, "suffix sort failed to converge");
    if (!keyo.p || keyo.n < m) keyo.alloc(c, m);
    if (!key.p || key.n < m) key.alloc(c, m);
    if (!valo.p || valo.n < m) valo.alloc(c, m);
    if (!val.p || val.n < m) val.alloc(c, m);
    static const bool use_finisher = getenv("PFP_NO_FINISHER") == nullptr;
    if (use_finisher && finisher_ok && m <= kFinishMax && out.rounds >= 1 && (g.mode == MODE_DICT || (g.mode == MODE_PLAIN && g.sym))) {
      DBuf<uint32_t> flt(c, m), feq(c, m), fgs(c, m), fov(c, 1);
      fov.zero();
      KScope ks(c, "pfp::finish_rank_kernel", m * (sizeof(I) * 4 + 12));
      hipLaunchKernelGGL(finish_rank_kernel<I>, gdim(cdiv(m * 8, TB)), gdim(TB), 0, c->stream, g, out.bytes, m, act_i.p, act_grp.p, flt.p, feq.p,
                         fgs.p, fov.p);
      if (read_scalar(c, fov.p) == 0) {
        hipLaunchKernelGGL(finish_write_kernel<I>, gdim(cdiv(m, TB)), gdim(TB), 0, c->stream, g, m, aslot.p, act_i.p, flt.p, feq.p, fgs.p,
                           out.finbit, out.sa.p, out.grp.p, rank_p, wordrank_p);
        PFP_HIP(hipGetLastError());
        if (trace_rounds) fprintf(stderr, "[pfp] doubling N=%llu round=%llu: the last %llu suffixes ranked by comparison\n",
                                  (unsigned long long)N, (unsigned long long)out.rounds, (unsigned long long)m);
        out.rounds++;
        m = 0;
        break;
      }
      finisher_ok = false;      // a long group or a long common prefix: the rounds go on as they would have
    }
    // Rounds after the first: the unresolved suffixes are already grouped, only the 32-bit "next"
    // key has to be ordered inside every group.  When the groups are many and of moderate size (a
    // dictionary of near-identical variants) a segmented sort moves 16 B per suffix instead of the
    // 7 x 24 B of a global 53-bit radix sort (big: 240 -> 126 ms of sorting).  rocPRIM's segmented
    // sort serialises a giant segment on one workgroup (the 300 k run of one symbol in a parse cost
    // 68 ms), and for tiny groups its bookkeeping eats the gain, so the choice is per round.
    seg_round = false; dbl_round = false; ipiv_round = false;
    pivot_round = pivot_ok && g.mode == MODE_DICT && out.finbit && kPivotCap >= 16 && ngrp && m / ngrp <= kPivotAvg &&
                  (uint64_t)nb + kPivBits <= 64;
    if (pivot_round) {
      // few suffixes left: the widest window at once.  (A compare reads only as far as the strings agree; the window bounds
      // the worst case, m x window bytes.  What is left after two rounds are mostly members of phrases longer than the
      // 512-byte window: without this they cost one round that settles nothing and quadruples the window, then another.)
      if (piv_cap >= 16 && piv_cap < kPivCapMax - 16 && out.rounds >= 2 && m * (uint64_t)(kPivCapMax - 16) < N * 128) piv_cap = kPivCapMax - 16;
      // large families (a collection of hundreds of copies): the members are already grouped, a
      // segmented sort of the 23-bit order key moves 16 B per suffix instead of 7 x 24 B
      bool seg = false, small = false;
      uint32_t ng = 0;
      constexpr uint32_t seg_min_avg = 24u;
      static const bool use_small = getenv("PFP_NO_SMALLSEG") == nullptr;
      // families of a handful of members: placed directly (after a group proved too long, only once the average is tiny)
      // (32-bit build: long families go to the side sort, so the average may be larger)
      small = use_segsort && use_small && m / ngrp <= (kWide ? kSmallSeg / 4 : kSmallSeg / 2) && (!small_failed || m / ngrp <= 3);
      auto try_seg = [&]() {
        if (use_segsort && m >= (1u << 20) && m < 0xFFFFFFFFull && m / ngrp >= seg_min_avg) {
          uint32_t maxlen = 0;
          seg_setup(m, ngrp, ng, maxlen);
          seg = maxlen <= (1u << 15) && m / ng >= seg_min_avg;
        }
      };
      if (!small) try_seg();
      if (small) seg_bufs();
      // both key forms are written when the direct placement is tried: should a group prove too long, the
      // device-wide sort takes over with the 64-bit keys
      { KScope ks(c, "pfp::build_keys_pivot_kernel", m * (4 + 4 + 4 + 12 + 64));
        hipLaunchKernelGGL(build_keys_pivot_kernel<I>, gdim(cdiv(m, TB)), gdim(TB), 0, c->stream, out.bytes, m, h, piv_cap, act_i.p,
                           act_grp.p, out.sa.p, out.finbit, key.p, (seg || small) ? k32.p : (uint32_t *)nullptr, val.p, small ? 1 : 0); }
      if (small && seg_small_sort(m)) seg_round = true;
      else if (small && (small_failed = true, try_seg(), false)) {}
      else if (seg) { segsort_pairs_u32<I>(c, k32.p, k32o.p, val.p, valo.p, m, ng, segb.p, sege.p, 0, kPivBits); seg_round = true; }
      else { sort_pairs_db(c, key, keyo, val, valo, m, 0, nb + kPivBits); std::swap(key, keyo); std::swap(val, valo); }
      out.rounds++;
      continue;                 // the sorted prefix common to all groups is still h: no doubling of h
    }
    if constexpr (sizeof(I) == 4) {
      // the parse of a collection: a pivot round with the member whose next rare symbol is farthest (see ipivot_select_kernel)
      // (a share of the parse - range mode - has no doubling rounds to fall back on: it tries whatever the list's size)
      if (g.mode == MODE_PLAIN && g.dist && g.sym && ipiv_ok && ngrp && (range_mode || m >= parse_pivot_min()) && m < 0xFFFFFFFFull && m / ngrp >= 2) {
        {
          DBuf<unsigned long long> best(c, N);
          best.zero();
          KScope ks(c, "pfp::ipivot_keys_kernel", m * (4 + 4 + 4 + 8 + 16 + 12 + 64));
          hipLaunchKernelGGL(ipivot_select_kernel<I>, gdim(cdiv(m, TB)), gdim(TB), 0, c->stream, m, NP, h, act_i.p, act_grp.p, g.dist, best.p);
          hipLaunchKernelGGL(ipivot_keys_kernel<I>, gdim(cdiv(m, TB)), gdim(TB), 0, c->stream, m, NP, h, kIntPivCap, g.sym, act_i.p,
                             act_grp.p, best.p, key.p, val.p);
        }
        bool sorted = false;
        if (m / ngrp <= kSmallSeg / 2) {      // small groups (pairs of copies with a common SNP): placed by counting in LDS
          seg_bufs();
          ovf_d.zero();
          { KScope ks(c, "pfp::seg_small_sort_kernel", m * (16 + 3 * sizeof(I) + 1));
            hipLaunchKernelGGL((seg_small_sort_kernel<I, uint64_t>), gdim(cdiv(m, TB)), gdim(TB), 0, c->stream, m, act_grp.p, key.p, val.p, gs.p,
                               keyo.p, valo.p, ovf_d.p, (uint8_t *)nullptr); }
          sorted = read_scalar(c, ovf_d.p) == 0;
        }
        if (!sorted && (m / ngrp >= 4 || range_mode)) {      // (a share has nothing else to fall back on)
          uint32_t ng = 0, maxlen = 0;
          seg_setup(m, ngrp, ng, maxlen);
          if (maxlen <= (1u << 15)) {
            segsort_pairs_u64_u32(c, key.p, keyo.p, val.p, valo.p, m, ng, segb.p, sege.p, 0, 32 + bits_for(2 * kIntPivCap + 3));
            sorted = true;
          }
        }
        if (sorted) {
          seg_round = true; ipiv_round = true;
          out.rounds++;
          continue;               // the sorted prefix common to all groups is still h
        }
        ipiv_ok = false;
      }
    }
    if (range_mode) { out.complete = false; break; }     // doubling would read ranks of suffixes other ranks hold
    if (late_rank && !rank_alloc) {      // the first doubling round: rank[] is needed after all
      alloc_rank();
      ranks_stale = true;               // whatever pivot rounds settled (whole words included) is filled in from the slots
    }
    if (lazy_pending) {
      lazy_pending = false;
      if (m * kLazyRatio > N) {      // most lookups would need the search: scatter the settled ranks once
        KScope ks(c, "pfp::scatter_settled_kernel", N * (4 + 4 + 1 + 4));
        hipLaunchKernelGGL(scatter_settled_kernel<I>, gdim(cdiv(N, TB)), gdim(TB), 0, c->stream, N, out.sa.p, out.grp.p, keep0.p,
                           out.finbit, out.rank.p);
        if (!out.paybits) out.skeys.release();      // with payload the merge still reads the records from skeys
        out.tab.release();
      }
    }
    repair_ranks(m, aslot.p);                 // a doubling round reads rank[] of arbitrary positions
    if (active_stale) {
      hipLaunchKernelGGL(active_ranks_kernel<I>, gdim(cdiv(m, TB)), gdim(TB), 0, c->stream, m, act_i.p, act_grp.p, out.rank.p);
      active_stale = false;
    }
    const RankViewT<I> L = rank_view(out);
    dbl_round = true;
    if constexpr (!kWide) {
      static const bool use_small = getenv("PFP_NO_SMALLSEG") == nullptr;
      if (use_segsort && use_small && ngrp && m / ngrp <= (kWide ? kSmallSeg / 4 : kSmallSeg / 2) && (!small_failed || m / ngrp <= 3)) {
        seg_bufs();
        { KScope ks(c, "pfp::build_keys32_kernel", m * (4 + 4 + 8));
          hipLaunchKernelGGL(build_keys32_kernel, gdim(cdiv(m, TB)), gdim(TB), 0, c->stream, g, m, h, act_i.p, L, k32.p, val.p); }
        seg_round = seg_small_sort(m);
        if (!seg_round) small_failed = true;
      }
      if (seg_round) {
      } else if (use_segsort && m >= (1u << 20) && ngrp && m / ngrp >= 24) {
        uint32_t ng = 0, maxlen = 0;
        seg_setup(m, ngrp, ng, maxlen);
        if (maxlen <= (1u << 15) && m / ng >= 24) {
          { KScope ks(c, "pfp::build_keys32_kernel", m * (4 + 4 + 8));
            hipLaunchKernelGGL(build_keys32_kernel, gdim(cdiv(m, TB)), gdim(TB), 0, c->stream, g, m, h, act_i.p, L, k32.p, val.p); }
          segsort_pairs_u32<I>(c, k32.p, k32o.p, val.p, valo.p, m, ng, segb.p, sege.p, 0, bits_for(N));
          seg_round = true;
        }
      }
    }
    if (!seg_round) {
      if constexpr (kWide) {
        if (!dkey.p || dkey.n < m) { dkey.alloc(c, m); dkeyo.alloc(c, m); }
        { KScope ks(c, "pfp::build_keys_kernel", m * (8 + 8 + 8 + 24));
          hipLaunchKernelGGL(build_keys_kernel<I>, gdim(cdiv(m, TB)), gdim(TB), 0, c->stream, g, m, h, act_i.p, act_grp.p, L, nb, dkey.p, val.p); }
        sort_pairs_db(c, dkey, dkeyo, val, valo, m, 0, keybits);
        std::swap(dkey, dkeyo); std::swap(val, valo);
      } else {
        { KScope ks(c, "pfp::build_keys_kernel", m * (4 + 4 + 4 + 12));
          hipLaunchKernelGGL(build_keys_kernel<I>, gdim(cdiv(m, TB)), gdim(TB), 0, c->stream, g, m, h, act_i.p, act_grp.p, L, nb, key.p, val.p); }
        sort_pairs_db(c, key, keyo, val, valo, m, 0, keybits);
        std::swap(key, keyo); std::swap(val, valo);
      }
    }
    h *= 2;
    out.rounds++;
  }
  // PFP_DEBUG validates the rank of every position: fill in what the pivot rounds left out
  if (c->debug && out.complete && !no_rank) repair_ranks(0, nullptr);
  if (rank_alloc) out.wordrank.release();      // every rank is in rank[]
}

template <class I>
__global__ void gather_ranks_kernel(RankViewT<I> L, uint64_t count, const uint64_t *__restrict__ pos, I *__restrict__ out) {
  uint64_t a = (uint64_t)BID * blockDim.x + threadIdx.x;
  bool settled;
  if (a >= count) return;
  if (L.wordrank) {      // pos[] are the word starts, in word order: the whole words a later round settled reported here
    const I r = L.wordrank[a];
    if (r != IdxTraits<I>::kNone) { out[a] = r & ~L.finbit; return; }
  }
  out[a] = rank_at(L, pos[a], settled);
}
template <class I>
void gather_ranks(pfp_ctx *c, const SuffixOrderT<I> &so, const uint64_t *d_pos, uint64_t count, I *d_out) {
  if (!count) return;
  hipLaunchKernelGGL(gather_ranks_kernel<I>, gdim(cdiv(count, 256)), gdim(256), 0, c->stream, rank_view(so), count, d_pos, d_out);
  PFP_HIP(hipGetLastError());
}
template void gather_ranks<uint32_t>(pfp_ctx *, const SuffixOrderT<uint32_t> &, const uint64_t *, uint64_t, uint32_t *);
template void gather_ranks<uint64_t>(pfp_ctx *, const SuffixOrderT<uint64_t> &, const uint64_t *, uint64_t, uint64_t *);

static KeyCode dict_key_code(pfp_ctx *c, const uint8_t *bytes, uint64_t N, double rep_hint = 0) {
  DBuf<unsigned long long> hist(c, 256);
  hist.zero();
  hipLaunchKernelGGL(byte_histogram_kernel, gdim(std::min<uint64_t>(cdiv64(N, 4096), (uint64_t)c->n_cu * 8)), gdim(256), 0,
                     c->stream, bytes, N, hist.p);
  std::vector<uint64_t> hh(256);
  PFP_HIP(hipMemcpyAsync(hh.data(), hist.p, 2048, hipMemcpyDeviceToHost, c->stream));
  sync(c);
  return make_key_code(hh.data(), rep_hint);
}

// keys-only first round (see sort_dict_suffixes): index bits of the combined word, 0 = sort (key, position) pairs;
// shortens kc's key to what is left of 64 bits
template <class I>
static int keysonly_bits(uint64_t N, double rep_hint, KeyCode &kc) {
  const int keysonly_env = []() { const char *e = getenv("PFP_KEYSONLY"); return e ? atoi(e) : -1; }();      // (per call: tests switch it)
  if (!(sizeof(I) == 4 && N >= 2 && (keysonly_env == 1 || (keysonly_env != 0 && rep_hint >= 2.0 && N >= (1u << 20))))) return 0;
  const int ib = bits_for(N - 1);
  int width = 64 - ib;                      // key bits incl. the terminator flag
  if (width > 3 * kKeysDigitBits && width % kKeysDigitBits <= 2) width -= width % kKeysDigitBits;       // a radix pass for one or two bits is a whole pass
  // (measured: 36 key bits on 129 M suffixes - first sort 7.8 -> 3.9 ms, same rounds after it; 31 key bits on 1.7 G
  //  suffixes leave three times as many unresolved after the first pivot round - the pair sort stays there)
  if (!(width - 1 >= 35 || keysonly_env == 1)) return 0;
  if (kc.kbits > width - 1) { kc.kbits = std::max(8, width - 1); kc.hmin = std::max(1, std::min(32, kc.kbits / kc.maxlen)); }
  return ib;
}

template <class I>
void sort_dict_suffixes(pfp_ctx *c, const uint8_t *bytes, uint64_t N, const WordView &wv, SuffixOrderT<I> &out,
                        const SlotPayloadSrc *pay) {
  PFP_REQUIRE(N >= 1 && (sizeof(I) == 8 ? N < (1ull << 40) : N < 0xFFFFFFF0ull), PFP_ELIMIT,
              sizeof(I) == 8 ? "dictionary of 2^40 bytes or more" : "dictionary too large for 32-bit suffix indices");
  SufGeom g{MODE_DICT, N, wv};
  KeyCode kc = dict_key_code(c, bytes, N, out.rep_hint);
  // Keys-only first round.  A dictionary of a repetitive collection (text / dictionary >= 2) is mostly families of
  // variants that no first-round key separates, whatever its width: the first round only has to bring the families
  // together.  Then the position fits into the same 64-bit word as a (shorter) key, the sort moves 16 bytes per
  // element and pass instead of 24 and makes 4-5 passes instead of 7, and the merge records in the key's spare bits
  // (which such an input hardly uses: most slots are re-ordered later) are given up.
  const int idx_bits = keysonly_bits<I>(N, out.rep_hint, kc);
  DBuf<uint64_t> key(c, N);
  DBuf<I> val;
  if (!idx_bits) val.alloc(c, N);
  out.paybits = (pay && !idx_bits && kc.kbits + 1 <= 48) ? 16 : 0;
  out.keymask = kc.kbits + 1 >= 64 ? ~0ull : ((1ull << (kc.kbits + 1)) - 1);
  { KScope ks(c, "pfp::init_keys_packed_kernel", N * (9 + (idx_bits ? 0 : sizeof(I)) + (out.paybits ? 9 : 0)));
    hipLaunchKernelGGL(init_keys_packed_kernel<I>, gdim((unsigned)cdiv64(N, kKeyPos)), gdim(256), 0, c->stream, bytes, N, kc,
                       pay ? *pay : SlotPayloadSrc{}, out.paybits, key.p, val.p, idx_bits); }
  if (c->debug) {
    DBuf<unsigned long long> bad(c, 1);
    bad.zero();
    hipLaunchKernelGGL(check_keys_kernel, gdim(cdiv(N, 256)), gdim(256), 0, c->stream, bytes, N, kc, key.p, out.keymask, bad.p, idx_bits);
    PFP_HIP(hipMemcpyAsync(c->h_scalars, bad.p, 8, hipMemcpyDeviceToHost, c->stream));
    sync(c);
    PFP_REQUIRE(c->h_scalars[0] == 0, PFP_EHIP, "bit-stream keys differ from packed_key_at at " + std::to_string(c->h_scalars[0]) + " positions");
  }
  out.lut.alloc(c, 256);
  PFP_HIP(hipMemcpyAsync(out.lut.p, kc.lut, 1024, hipMemcpyHostToDevice, c->stream));
  sync(c);      // kc is a stack object
  out.bytes = bytes; out.kbits = kc.kbits;
  doubling<I>(c, g, key, val, (uint64_t)kc.hmin, out, kc.kbits + 1, true, ~0ull, idx_bits);
}
template void sort_dict_suffixes<uint32_t>(pfp_ctx *, const uint8_t *, uint64_t, const WordView &, SuffixOrderT<uint32_t> &, const SlotPayloadSrc *);
template void sort_dict_suffixes<uint64_t>(pfp_ctx *, const uint8_t *, uint64_t, const WordView &, SuffixOrderT<uint64_t> &, const SlotPayloadSrc *);

// ---- key-range sharded variant (multi-GPU)
__global__ void sample_keys_kernel(const uint8_t *__restrict__ s, uint64_t N, uint64_t stride, uint32_t ns, KeyCode kp,
                                   uint64_t *__restrict__ key, uint32_t *__restrict__ val) {
  __shared__ uint32_t lut[256];
  lut[threadIdx.x] = kp.lut[threadIdx.x];
  __syncthreads();
  const uint32_t k = BID * 256 + threadIdx.x;
  if (k >= ns) return;
  const uint64_t i = (uint64_t)k * stride;
  key[k] = i < N ? packed_key_at(s, i, kp.kbits, lut) : ~0ull;
  val[k] = k;
}
// The same selection without a key per position.  A share of the suffix array only has to be a contiguous range of the final
// order, and "is this suffix below that one" is a string comparison: the suffix at i against the suffix at a boundary position,
// 16 bytes at a time - decided at the first byte for three positions in four (DNA), never more than kRangeCmp bytes: a suffix that
// equals the boundary's that far goes to the upper side whichever way it would end (the suffixes that share 64 bytes with the
// boundary stand together in the order, so the cut just moves below them: still one cut, the same on both sides of it).
// Identical strings of different words (both reach their terminator) are a tie: upper side, never split.
// 8.7 -> ~2 ms for the 0.9 G positions of the 8-rank union dictionary (the key kernel codes every character of every position).
constexpr int kRangeCmp = 64;
__device__ __forceinline__ int cmp_suffix_boundary(const uint8_t *__restrict__ s, uint64_t i, uint64_t b) {
  if (i == b) return 0;
#pragma unroll 1
  for (int q = 0; q < kRangeCmp / 16; q++) {
    const uint4 xv = ld16u(s + i + 16 * q), yv = ld16u(s + b + 16 * q);
    const uint32_t x[4] = {xv.x, xv.y, xv.z, xv.w}, y[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t diff = x[k] ^ y[k];
      const uint32_t term = (x[k] - 0x02020202u) & ~x[k] & 0x80808080u;      // bytes < 2 of x (lowest flag exact)
      if (diff | term) {
        const int fd = diff ? (__builtin_ctz(diff) >> 3) : 4, ft = term ? (__builtin_ctz(term) >> 3) : 4;
        if (ft < fd) return 0;                                    // both end before they differ: the same string
        const uint32_t bx = (x[k] >> (8 * fd)) & 0xffu, by = (y[k] >> (8 * fd)) & 0xffu;
        return bx < by ? -1 : 1;
      }
    }
  }
  return 0;      // equal for kRangeCmp bytes: upper side
}
// the first 16 bytes of both suffixes in registers (two 64-bit words each); the full compare only where they do not decide
__device__ __forceinline__ int cmp_head16(const uint4 xv, const uint4 yv, const uint8_t *__restrict__ s, uint64_t i, uint64_t b) {
  const uint64_t x[2] = {(uint64_t)xv.x | ((uint64_t)xv.y << 32), (uint64_t)xv.z | ((uint64_t)xv.w << 32)};
  const uint64_t y[2] = {(uint64_t)yv.x | ((uint64_t)yv.y << 32), (uint64_t)yv.z | ((uint64_t)yv.w << 32)};
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const uint64_t diff = x[k] ^ y[k];
    const uint64_t term = (x[k] - 0x0202020202020202ull) & ~x[k] & 0x8080808080808080ull;      // bytes < 2 of x (lowest flag exact)
    if (diff | term) {
      const int fd = diff ? (__builtin_ctzll(diff) >> 3) : 8, ft = term ? (__builtin_ctzll(term) >> 3) : 8;
      if (ft < fd) return 0;
      const uint32_t bx = (uint32_t)(x[k] >> (8 * fd)) & 0xffu, by = (uint32_t)(y[k] >> (8 * fd)) & 0xffu;
      return bx < by ? -1 : 1;
    }
  }
  return cmp_suffix_boundary(s, i, b);      // (16 equal bytes: rare)
}
// the same with four consecutive positions per thread: their heads come out of five ALIGNED dwords (byte shifts) instead of four
// unaligned 16-byte loads and four byte loads, the four flags leave as one dword - the one-position form is bound by its byte-wide
// memory instructions, not by the compares.  1024 positions per workgroup.
__device__ __forceinline__ int cmp_words16(uint64_t x0, uint64_t x1, uint64_t y0, uint64_t y1, const uint8_t *__restrict__ s, uint64_t i, uint64_t b) {
  const uint64_t x[2] = {x0, x1}, y[2] = {y0, y1};
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const uint64_t diff = x[k] ^ y[k];
    const uint64_t term = (x[k] - 0x0202020202020202ull) & ~x[k] & 0x8080808080808080ull;
    if (diff | term) {
      const int fd = diff ? (__builtin_ctzll(diff) >> 3) : 8, ft = term ? (__builtin_ctzll(term) >> 3) : 8;
      if (ft < fd) return 0;
      const uint32_t bx = (uint32_t)(x[k] >> (8 * fd)) & 0xffu, by = (uint32_t)(y[k] >> (8 * fd)) & 0xffu;
      return bx < by ? -1 : 1;
    }
  }
  return i == b ? 0 : cmp_suffix_boundary(s, i, b);
}
__global__ __launch_bounds__(256) void range_flags_cmp4_kernel(const uint8_t *__restrict__ s, uint64_t N, uint64_t b_lo, int has_lo,
                                                               uint64_t b_hi, int has_hi, SlotPayloadSrc count, int want_count,
                                                               uint8_t *__restrict__ flag, unsigned long long *__restrict__ tile_below,
                                                               unsigned long long *__restrict__ tile_emits) {
  __shared__ unsigned long long wsum[2][4];
  __shared__ uint32_t wt[4];
  const uint64_t B0 = (uint64_t)BID * 1024;
  if (B0 >= N) return;      // a workgroup of the padded last grid row
  const uint64_t p0 = B0 + (uint64_t)threadIdx.x * 4;
  const int lane = threadIdx.x & 63, wvi = threadIdx.x >> 6;
  // bytes [p0, p0 + 20): the dictionary is padded with 64 zero bytes, so the loads of the last positions stay inside it
  uint32_t d[5] = {0u, 0u, 0u, 0u, 0u};
  if (p0 < N) {
    const uint4 v = ld16u(s + p0);
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    d[4] = *reinterpret_cast<const uint32_t *>(s + p0 + 16);
  }
  const uint4 lo4 = has_lo ? ld16u(s + b_lo) : make_uint4(0u, 0u, 0u, 0u), hi4 = has_hi ? ld16u(s + b_hi) : make_uint4(0u, 0u, 0u, 0u);
  const uint64_t lo0 = (uint64_t)lo4.x | ((uint64_t)lo4.y << 32), lo1 = (uint64_t)lo4.z | ((uint64_t)lo4.w << 32);
  const uint64_t hi0 = (uint64_t)hi4.x | ((uint64_t)hi4.y << 32), hi1 = (uint64_t)hi4.z | ((uint64_t)hi4.w << 32);
  uint32_t tmask = 0, fl = 0, below = 0, minemask = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint64_t i = p0 + k;
    if (i >= N) break;
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; j++) w[j] = k ? ((d[j] >> (8 * k)) | (d[j + 1] << (32 - 8 * k))) : d[j];
    const uint64_t x0 = (uint64_t)w[0] | ((uint64_t)w[1] << 32), x1 = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
    if ((w[0] & 0xffu) == (uint32_t)kEndOfWord) tmask |= 1u << k;
    const int cl = !has_lo ? 1 : cmp_words16(x0, x1, lo0, lo1, s, i, b_lo);
    if (cl < 0) below++;
    const bool mine = cl >= 0 && (!has_hi || cmp_words16(x0, x1, hi0, hi1, s, i, b_hi) < 0);
    if (mine) { fl |= 1u << (8 * k); minemask |= 1u << k; }
  }
  if (p0 + 4 <= N) *reinterpret_cast<uint32_t *>(flag + p0) = fl;
  else for (int k = 0; k < 4 && p0 + k < N; k++) flag[p0 + k] = (uint8_t)((fl >> (8 * k)) & 1u);
  // terminators before this thread's positions in the block (only the emit count of the share's own positions wants them)
  uint32_t tc = (uint32_t)__popc(tmask), inc = tc;
  for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
  if (lane == 63) wt[wvi] = inc;
  __syncthreads();
  unsigned long long emits = 0;
  if (minemask && want_count) {
    uint32_t before = inc - tc;
    for (int q = 0; q < wvi; q++) before += wt[q];
    const uint32_t wbase = count.wv.blk_word[B0 >> 6] + before;
#pragma unroll
    for (int k = 0; k < 4; k++)
      if ((minemask >> k) & 1u) {
        const uint32_t wd = wbase + (uint32_t)__popc(tmask & ((1u << k) - 1u));
        if (wd < count.wv.d && count.wv.wend[wd] - (p0 + k) > (uint64_t)count.w) emits += count.wocc[wd];
      }
  }
  unsigned long long cnt = below;
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
  if (__ballot(emits != 0)) for (int o = 32; o > 0; o >>= 1) emits += __shfl_down(emits, o, 64);
  if (lane == 0) { wsum[0][wvi] = cnt; wsum[1][wvi] = emits; }
  __syncthreads();
  if (threadIdx.x == 0) {
    tile_below[BID] = wsum[0][0] + wsum[0][1] + wsum[0][2] + wsum[0][3];
    tile_emits[BID] = wsum[1][0] + wsum[1][1] + wsum[1][2] + wsum[1][3];
  }
}
__global__ __launch_bounds__(256) void sum2_u64_kernel(const unsigned long long *__restrict__ a, const unsigned long long *__restrict__ b,
                                                       uint64_t n, unsigned long long *__restrict__ out) {
  __shared__ unsigned long long ws[2][4];
  unsigned long long x = 0, y = 0;
  for (uint64_t i = (uint64_t)BID * 256 + threadIdx.x; i < n; i += (uint64_t)GDIM * 256) { x += a[i]; y += b[i]; }
  for (int o = 32; o > 0; o >>= 1) { x += __shfl_down(x, o, 64); y += __shfl_down(y, o, 64); }
  if ((threadIdx.x & 63) == 0) { ws[0][threadIdx.x >> 6] = x; ws[1][threadIdx.x >> 6] = y; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(out, ws[0][0] + ws[0][1] + ws[0][2] + ws[0][3]);
    atomicAdd(out + 1, ws[1][0] + ws[1][1] + ws[1][2] + ws[1][3]);
  }
}
template <class I>
__global__ __launch_bounds__(256) void init_keys_list_kernel(const uint8_t *__restrict__ s, uint64_t n, KeyCode kp,
                                                             const I *__restrict__ idx,
                                                             uint64_t *__restrict__ key, I *__restrict__ val, int idx_bits) {
  __shared__ uint32_t lut[256];
  lut[threadIdx.x] = kp.lut[threadIdx.x];
  __syncthreads();
  uint64_t a = (uint64_t)BID * 256 + threadIdx.x;
  if (a >= n) return;
  const I i = idx[a];
  const uint64_t k = packed_key_at(s, i, kp.kbits, lut);
  if (idx_bits) { key[a] = (k << idx_bits) | (uint64_t)i; return; }      // keys-only sort (the list is in position order)
  key[a] = k; val[a] = i;
}

// the suffixes of a LIST of positions (ascending) sorted among themselves: first-round keys position by position, then the
// rounds of doubling() in its range mode (no rank per dictionary position; out.complete == false where only a doubling
// round - ranks of suffixes outside the list - could go on).  Releases idx.
template <class I>
static void sort_suffix_list(pfp_ctx *c, const SufGeom &g, const KeyCode &kc, int idx_bits, DBuf<I> &idx, uint64_t n, SuffixOrderT<I> &out) {
  DBuf<uint64_t> key(c, std::max<uint64_t>(n, 1));
  DBuf<I> val;
  if (!idx_bits) val.alloc(c, std::max<uint64_t>(n, 1));
  out.paybits = 0;
  out.keymask = kc.kbits + 1 >= 64 ? ~0ull : ((1ull << (kc.kbits + 1)) - 1);
  if (n) {
    KScope ks(c, "pfp::init_keys_list_kernel", (uint64_t)n * 17);
    hipLaunchKernelGGL(init_keys_list_kernel<I>, gdim(cdiv(n, 256)), gdim(256), 0, c->stream, g.wv.bytes, (uint64_t)n, kc, idx.p, key.p, val.p,
                       idx_bits);
  }
  idx.release();
  out.lut.alloc(c, 256);
  PFP_HIP(hipMemcpyAsync(out.lut.p, kc.lut, 1024, hipMemcpyHostToDevice, c->stream));
  sync(c);      // kc may be a stack object of the caller
  out.bytes = g.wv.bytes; out.kbits = kc.kbits;
  doubling<I>(c, g, key, val, (uint64_t)kc.hmin, out, kc.kbits + 1, true, (uint64_t)n, idx_bits);
}

template <class I>
void sort_dict_suffixes_range(pfp_ctx *c, const uint8_t *bytes, uint64_t N, const WordView &wv, uint32_t part,
                              uint32_t parts, SuffixOrderT<I> &out, const SlotPayloadSrc *pay, const SlotPayloadSrc *count) {
  PFP_REQUIRE(N >= 1 && (sizeof(I) == 8 ? N < (1ull << 40) : N < 0xFFFFFFF0ull), PFP_ELIMIT,
              sizeof(I) == 8 ? "dictionary of 2^40 bytes or more" : "dictionary too large for 32-bit suffix indices");
  PFP_REQUIRE(parts >= 1 && part < parts, PFP_EINVAL, "bad key-range share");
  SufGeom g{MODE_DICT, N, wv};
  KeyCode kc = dict_key_code(c, bytes, N, out.rep_hint);
  const int idx_bits = keysonly_bits<I>(N, out.rep_hint, kc);      // (before the splitters: every rank cuts the same keys)
  // splitters: every stride-th suffix's key, sorted; the same on every rank.  The share is cut by comparing every suffix with the
  // boundary SUFFIXES (range_flags_cmp4_kernel) - boundaries with distinct keys, so that they stand in the order they are used in
  // (the two earlier forms - a first-round key per position, one position per thread - went with round 4)
  uint64_t klo = 0, khi = ~0ull, b_lo = 0, b_hi = 0;
  bool has_lo = false, has_hi = false;
  if (parts > 1) {
    const uint32_t ns = (uint32_t)std::min<uint64_t>(N, 1u << 16);
    const uint64_t stride = N / ns;
    DBuf<uint64_t> sk(c, ns), sko(c, ns);
    DBuf<uint32_t> sv(c, ns), svo(c, ns);
    hipLaunchKernelGGL(sample_keys_kernel, gdim(cdiv(ns, 256)), gdim(256), 0, c->stream, bytes, N, stride, ns, kc, sk.p, sv.p);
    sort_pairs_u64_u32(c, sk.p, sko.p, sv.p, svo.p, ns, 0, 64);
    std::vector<uint64_t> hs(ns);
    std::vector<uint32_t> hv(ns);
    PFP_HIP(hipMemcpyAsync(hs.data(), sko.p, (size_t)ns * 8, hipMemcpyDeviceToHost, c->stream));
    PFP_HIP(hipMemcpyAsync(hv.data(), svo.p, (size_t)ns * 4, hipMemcpyDeviceToHost, c->stream));
    sync(c);
    if (part > 0) klo = hs[(uint64_t)part * ns / parts];
    if (part + 1 < parts) khi = hs[(uint64_t)(part + 1) * ns / parts];
    // boundary of part q (q >= 1): the sample at q ns / parts - or the boundary before it, if that sample's key is no larger
    // (equal keys do not say which suffix is the smaller: such a part holds nothing)
    auto boundary = [&](uint32_t q, uint64_t &pos) {
      uint64_t prev_key = 0; bool have = false;
      for (uint32_t t = 1; t <= q; t++) {
        const uint64_t at = (uint64_t)t * ns / parts;
        if (!have || hs[at] > prev_key) { pos = (uint64_t)hv[at] * stride; prev_key = hs[at]; have = true; }
      }
      return have;
    };
    if (part > 0) has_lo = boundary(part, b_lo);
    if (part + 1 < parts) has_hi = boundary(part + 1, b_hi);
  }
  const int khi_open = part + 1 == parts ? 1 : 0;
  DBuf<uint8_t> flag(c, N);
  DBuf<unsigned long long> below(c, 2);
  below.zero();
  {
    const uint64_t nblk = cdiv64(N, 1024);
    DBuf<unsigned long long> tb(c, nblk), te(c, nblk);
    KScope ks(c, "pfp::range_flags_cmp4_kernel", N * 2);
    hipLaunchKernelGGL(range_flags_cmp4_kernel, gdim((unsigned)nblk), gdim(256), 0, c->stream, bytes, N, b_lo, has_lo ? 1 : 0, b_hi,
                       has_hi ? 1 : 0, count ? *count : SlotPayloadSrc{}, count ? 1 : 0, flag.p, tb.p, te.p);
    hipLaunchKernelGGL(sum2_u64_kernel, gdim((int)std::min<uint64_t>(cdiv64(nblk, 256), 256)), gdim(256), 0, c->stream, tb.p, te.p, nblk,
                       below.p);
  }
  const uint64_t n_mine = count_flags(c, flag.p, N);      // the share's size first: its index list is then exactly that long
  PFP_REQUIRE(sizeof(I) == 8 || n_mine < 0xFFFFFFF0ull, PFP_ELIMIT, "a share of the suffix array too large for 32-bit slots");
  DBuf<I> idx(c, std::max<uint64_t>(n_mine, 1));
  DBuf<uint64_t> cnt_d(c, 1);
  select_index<I>(c, flag.p, idx.p, cnt_d.p, N);
  PFP_HIP(hipMemcpyAsync(c->h_scalars + 1, below.p, 16, hipMemcpyDeviceToHost, c->stream));
  sync(c);
  const uint64_t slot_base = c->h_scalars[1];
  const uint64_t range_emits = c->h_scalars[2];
  flag.release();
  sort_suffix_list<I>(c, g, kc, idx_bits, idx, n_mine, out);
  (void)pay;      // (no merge records in the keys of a share: its positions are a scattered list, every record would be a word lookup)
  out.slot_base = slot_base; out.klo = klo; out.khi = khi_open ? ~0ull : khi; out.range_emits = range_emits;
}
template void sort_dict_suffixes_range<uint32_t>(pfp_ctx *, const uint8_t *, uint64_t, const WordView &, uint32_t, uint32_t,
                                                 SuffixOrderT<uint32_t> &, const SlotPayloadSrc *, const SlotPayloadSrc *);
template void sort_dict_suffixes_range<uint64_t>(pfp_ctx *, const uint8_t *, uint64_t, const WordView &, uint32_t, uint32_t,
                                                 SuffixOrderT<uint64_t> &, const SlotPayloadSrc *, const SlotPayloadSrc *);

// range mode: out[word] = 1 + SA(D) slot of the word's whole-word suffix for the words whose suffix lies in this share (the others
// stay 0).  A share keeps no rank per position: it looks at its own slots - the suffix of slot t is a whole word iff the
// byte before it is a terminator - and asks the word lookup which word that is.
template <class I>
__global__ void word_slots_range_kernel(uint64_t N, const I *__restrict__ sa, WordView wv, uint64_t slot_base, uint64_t *__restrict__ out) {
  const uint64_t t = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (t >= N) return;
  const uint64_t i = sa[t];
  if (i != 0 && wv.bytes[i - 1] != kEndOfWord) return;
  const uint32_t wd = word_of(wv, i);
  if (wd < wv.d) out[wd] = slot_base + t + 1;
}
template <class I>
void gather_slots_range(pfp_ctx *c, const SuffixOrderT<I> &so, const WordView &wv, uint64_t count, uint64_t *d_out) {
  if (!count) return;
  PFP_HIP(hipMemsetAsync(d_out, 0, count * 8, c->stream));
  if (so.N) hipLaunchKernelGGL(word_slots_range_kernel<I>, gdim(cdiv(so.N, 256)), gdim(256), 0, c->stream, so.N, so.sa.p, wv, so.slot_base, d_out);
  PFP_HIP(hipGetLastError());
}
template void gather_slots_range<uint32_t>(pfp_ctx *, const SuffixOrderT<uint32_t> &, const WordView &, uint64_t, uint64_t *);
template void gather_slots_range<uint64_t>(pfp_ctx *, const SuffixOrderT<uint64_t> &, const WordView &, uint64_t, uint64_t *);

// (Tried in round 3 and removed: "grouped order without sorting duplicates".  Equal suffixes of a dictionary of variants can be
//  read off the WORDS: sort the d words by their reversed strings, keep the longest common suffix lcs[r] of neighbours; the
//  suffix of length L of the word at rank r duplicates its left neighbour's iff lcs[r] >= L, so only one representative per
//  distinct string (71 M of 129 M suffixes on the 64-copy workload) is suffix-sorted - as a list - and every sorted
//  representative is expanded into its run of words.  Bit-exact on all goldens, half the peak memory (12.6 GB input: 165 ->
//  90 GB) - and slower: the dictionary sort went 20.0 -> 22.2 ms (64 copies) and 354 -> 434 ms (1024 copies).  The first
//  round shrinks (radix 3.8 -> 2.2 ms, pivot 6.8 -> 4.9 ms), but with the identical strings gone every remaining tie is a real
//  difference: 7 rounds instead of 5 (23 pivot launches instead of 7 on 1024 copies), and finding the runs and laying out
//  the slots costs 4.5 ms / 87 ms.  It would pay if the MERGE worked on the 71 M groups instead of 129 M slots (sums over
//  a run from prefix sums, "all chars equal" from a range minimum over lcs[]): DESIGN.md 7b.)

template <class I>
void sort_byte_suffixes(pfp_ctx *c, const uint8_t *bytes, uint64_t N, SuffixOrderT<I> &out) {
  PFP_REQUIRE(N >= 1 && (sizeof(I) == 8 ? N < (1ull << 40) : N < 0xFFFFFFF0ull), PFP_ELIMIT, "text too large for the suffix index width");
  SufGeom g{MODE_PLAIN, N, WordView{}};
  DBuf<uint64_t> key(c, N);
  DBuf<I> val(c, N);
  hipLaunchKernelGGL(init_keys_bytes_kernel<I>, gdim(cdiv(N, 256)), gdim(256), 0, c->stream, bytes, N, key.p, val.p);
  doubling<I>(c, g, key, val, 8, out);
}
template void sort_byte_suffixes<uint32_t>(pfp_ctx *, const uint8_t *, uint64_t, SuffixOrderT<uint32_t> &);
template void sort_byte_suffixes<uint64_t>(pfp_ctx *, const uint8_t *, uint64_t, SuffixOrderT<uint64_t> &);

void sort_int_suffixes(pfp_ctx *c, const uint32_t *sym, uint64_t N, SuffixOrder &out, uint32_t max_sym, const uint32_t *occ,
                       uint32_t n_sym) {
  PFP_REQUIRE(N >= 1 && N < 0xFFFFFFF0ull, PFP_ELIMIT, "parse too large for 32-bit suffix indices");
  SufGeom g{MODE_PLAIN, N, WordView{}};
  DBuf<uint64_t> key(c, N);
  DBuf<uint32_t> val(c, N);
  // the symbol width is measured, not taken on trust (max_sym is only an upper bound for the check)
  DBuf<uint32_t> mx(c, 1);
  mx.zero();
  hipLaunchKernelGGL(max_u32_kernel, gdim((int)std::min<uint64_t>(cdiv64(N, 256), 1024)), gdim(256), 0, c->stream, sym, N, mx.p);
  const uint32_t real_max = read_scalar(c, mx.p);
  PFP_REQUIRE(real_max <= max_sym, PFP_EFORMAT, "integer string holds a symbol above its alphabet size");
  const int sb = bits_for(real_max);
  g.sym = sym;
  // occ[x - 1] = occurrences of symbol x (the parse: the words' counts): rare = fewer than the mean; dist[] for the pivot rounds
  DBuf<uint32_t> dist;
  if (occ && n_sym && N >= parse_pivot_min()) {
    DBuf<uint32_t> mk(c, N), pm(c, N);
    dist.alloc(c, N);
    const uint32_t below = (uint32_t)std::max<uint64_t>(2, N / n_sym);
    hipLaunchKernelGGL(rare_marks_kernel, gdim(cdiv(N, 256)), gdim(256), 0, c->stream, sym, (uint32_t)N, occ, n_sym, below, mk.p);
    inclusive_max_u32(c, mk.p, pm.p, N);
    hipLaunchKernelGGL(rare_dist_kernel, gdim(cdiv(N, 256)), gdim(256), 0, c->stream, (uint32_t)N, pm.p, dist.p);
    g.dist = dist.p;
  }
  if (64 - 2 * sb >= 6) {
    DBuf<uint32_t> v(c, N), pm(c, N);
    hipLaunchKernelGGL(run_marks_kernel, gdim(cdiv(N, 256)), gdim(256), 0, c->stream, sym, (uint32_t)N, v.p);
    inclusive_max_u32(c, v.p, pm.p, N);
    hipLaunchKernelGGL(init_keys_int_run_kernel, gdim(cdiv(N, 256)), gdim(256), 0, c->stream, sym, (uint32_t)N, pm.p, sb, key.p,
                       val.p);
  } else {
    hipLaunchKernelGGL(init_keys_int_kernel, gdim(cdiv(N, 256)), gdim(256), 0, c->stream, sym, N, key.p, val.p);
  }
  doubling<uint32_t>(c, g, key, val, 2, out);
}

// ---- one rank's share of the parse's suffix array (multi-GPU chain).  The parse is replicated, its suffix array was too; with the
// pivot round above a rank can sort the suffixes whose first-round key falls in its range without anybody's ranks, exactly as it
// sorts its share of the dictionary: the same keys for all N positions (a streaming pass), splitters from a deterministic sample
// (identical on every rank), the range's positions as a list, first sort + pivot round + comparison finisher on the list.
// out.complete = false where a doubling round would be needed (the caller then sorts the whole parse as before).
__global__ void sample_u64_kernel(const uint64_t *__restrict__ key, uint64_t stride, uint32_t ns, uint64_t *__restrict__ out, uint32_t *__restrict__ idx) {
  uint32_t j = BID * blockDim.x + threadIdx.x;
  if (j < ns) { out[j] = key[(uint64_t)j * stride]; idx[j] = j; }
}
__global__ __launch_bounds__(256) void range_flags_u64_kernel(const uint64_t *__restrict__ key, uint64_t N, uint64_t klo, uint64_t khi, int khi_open,
                                                              uint8_t *__restrict__ flag, unsigned long long *__restrict__ tile_below) {
  __shared__ unsigned long long ws[4];
  const uint64_t i = (uint64_t)BID * 256 + threadIdx.x;
  unsigned long long lt = 0;
  if (i < N) {
    const uint64_t k = key[i];
    flag[i] = (k >= klo && (khi_open || k < khi)) ? 1 : 0;
    lt = k < klo ? 1ull : 0ull;
  }
  for (int o = 32; o > 0; o >>= 1) lt += __shfl_down(lt, o, 64);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = lt;
  __syncthreads();
  // (per-workgroup sums: an atomic per wave on one address serialises a million of them - 12 ms for 63 M keys)
  if (threadIdx.x == 0 && (uint64_t)BID * 256 < N) tile_below[BID] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ void gather_list_keys_kernel(uint64_t n, const uint32_t *__restrict__ idx, const uint64_t *__restrict__ key,
                                        uint64_t *__restrict__ lkey, uint32_t *__restrict__ lval) {
  const uint64_t a = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (a < n) { const uint32_t i = idx[a]; lkey[a] = key[i]; lval[a] = i; }
}
void sort_int_suffixes_range(pfp_ctx *c, const uint32_t *sym, uint64_t N, uint32_t max_sym, const uint32_t *occ, uint32_t n_sym,
                             uint32_t part, uint32_t parts, SuffixOrder &out) {
  PFP_REQUIRE(N >= 1 && N < 0xFFFFFFF0ull, PFP_ELIMIT, "parse too large for 32-bit suffix indices");
  PFP_REQUIRE(parts >= 1 && part < parts && occ && n_sym, PFP_EINVAL, "bad share of the parse's suffix array");
  SufGeom g{MODE_PLAIN, N, WordView{}};
  DBuf<uint32_t> mx(c, 1);
  mx.zero();
  hipLaunchKernelGGL(max_u32_kernel, gdim((int)std::min<uint64_t>(cdiv64(N, 256), 1024)), gdim(256), 0, c->stream, sym, N, mx.p);
  const uint32_t real_max = read_scalar(c, mx.p);
  PFP_REQUIRE(real_max <= max_sym, PFP_EFORMAT, "integer string holds a symbol above its alphabet size");
  const int sb = bits_for(real_max);
  g.sym = sym;
  DBuf<uint32_t> dist(c, N);
  {
    KScope ks(c, "pfp::rare_dist_kernel", N * 24);      // (+ rare_marks_kernel and the running maximum between them)
    DBuf<uint32_t> mk(c, N), pm(c, N);
    const uint32_t below = (uint32_t)std::max<uint64_t>(2, N / n_sym);
    hipLaunchKernelGGL(rare_marks_kernel, gdim(cdiv(N, 256)), gdim(256), 0, c->stream, sym, (uint32_t)N, occ, n_sym, below, mk.p);
    inclusive_max_u32(c, mk.p, pm.p, N);
    hipLaunchKernelGGL(rare_dist_kernel, gdim(cdiv(N, 256)), gdim(256), 0, c->stream, (uint32_t)N, pm.p, dist.p);
    g.dist = dist.p;
  }
  DBuf<uint64_t> key(c, N);
  {
    KScope ks(c, "pfp::init_keys_int_run_kernel", N * 28);      // (+ run_marks_kernel and its running maximum)
    DBuf<uint32_t> val(c, N);
    if (64 - 2 * sb >= 6) {
      DBuf<uint32_t> v(c, N), pm(c, N);
      hipLaunchKernelGGL(run_marks_kernel, gdim(cdiv(N, 256)), gdim(256), 0, c->stream, sym, (uint32_t)N, v.p);
      inclusive_max_u32(c, v.p, pm.p, N);
      hipLaunchKernelGGL(init_keys_int_run_kernel, gdim(cdiv(N, 256)), gdim(256), 0, c->stream, sym, (uint32_t)N, pm.p, sb, key.p, val.p);
    } else {
      hipLaunchKernelGGL(init_keys_int_kernel, gdim(cdiv(N, 256)), gdim(256), 0, c->stream, sym, N, key.p, val.p);
    }
  }
  uint64_t klo = 0, khi = ~0ull;
  if (parts > 1) {
    const uint32_t ns = (uint32_t)std::min<uint64_t>(N, 1u << 16);
    const uint64_t stride = N / ns;
    DBuf<uint64_t> sk(c, ns), sko(c, ns);
    DBuf<uint32_t> sv(c, ns), svo(c, ns);
    hipLaunchKernelGGL(sample_u64_kernel, gdim(cdiv(ns, 256)), gdim(256), 0, c->stream, key.p, stride, ns, sk.p, sv.p);
    sort_pairs_u64_u32(c, sk.p, sko.p, sv.p, svo.p, ns, 0, 64);
    std::vector<uint64_t> hs(ns);
    PFP_HIP(hipMemcpyAsync(hs.data(), sko.p, (size_t)ns * 8, hipMemcpyDeviceToHost, c->stream));
    sync(c);
    if (part > 0) klo = hs[(uint64_t)part * ns / parts];
    if (part + 1 < parts) khi = hs[(uint64_t)(part + 1) * ns / parts];
  }
  const int khi_open = part + 1 == parts ? 1 : 0;
  std::optional<KScope> ks_sel;      // (ended before the list is sorted)
  ks_sel.emplace(c, "pfp::range_flags_u64_kernel", N * 10);      // (+ the flag count and the index selection)
  DBuf<uint8_t> flag(c, N + 16);
  PFP_HIP(hipMemsetAsync(flag.p + N, 0, 16, c->stream));
  const uint64_t nblk = cdiv64(N, 256);
  DBuf<unsigned long long> below(c, 2), tb(c, nblk);
  below.zero();
  hipLaunchKernelGGL(range_flags_u64_kernel, gdim(nblk), gdim(256), 0, c->stream, key.p, N, klo, khi, khi_open, flag.p, tb.p);
  hipLaunchKernelGGL(sum2_u64_kernel, gdim((int)std::min<uint64_t>(cdiv64(nblk, 256), 256)), gdim(256), 0, c->stream, tb.p, tb.p, nblk, below.p);
  const uint64_t n_mine = count_flags(c, flag.p, N);
  DBuf<uint32_t> idx(c, std::max<uint64_t>(n_mine, 1));
  DBuf<uint64_t> cnt_d(c, 1);
  select_index<uint32_t>(c, flag.p, idx.p, cnt_d.p, N);
  const uint64_t slot_base = read_scalar(c, (const uint64_t *)below.p);
  flag.release();
  DBuf<uint64_t> lkey(c, std::max<uint64_t>(n_mine, 1));
  DBuf<uint32_t> lval(c, std::max<uint64_t>(n_mine, 1));
  if (n_mine) hipLaunchKernelGGL(gather_list_keys_kernel, gdim(cdiv(n_mine, 256)), gdim(256), 0, c->stream, n_mine, idx.p, key.p, lkey.p, lval.p);
  PFP_HIP(hipGetLastError());
  key.release(); idx.release();
  ks_sel.reset();
  doubling<uint32_t>(c, g, lkey, lval, 2, out, 64, false, n_mine, 0);
  out.slot_base = slot_base; out.klo = klo; out.khi = khi_open ? ~0ull : khi;
}

}  // namespace pfp
