// radix.hip -- the first-round suffix-key sort (round 4): a hand-written most-significant-digit radix sort for gfx950.
//
// What it replaces: the reference orders the dictionary's suffixes by induced sorting (gsa/gsacak.c:1395-1524); here the first
// round of the suffix sorter (sufsort.hip) is ONE stable sort of all N (key, position) pairs by a packed prefix of 16-24
// characters.  Rounds 1-3 handed that sort to rocPRIM's onesweep (5-7 passes of 8 bits over every pair: 45 % of the step on
// BASELINE configs[1]).  This file does it in two (three) stable partition passes of 8-10 bits that cut the array into buckets of
// about 2 K elements, and one kernel that sorts every bucket by ALL remaining key bits inside LDS - each element crosses HBM three
// (four) times instead of six to eight.
//
//   rx_hist_kernel<BITS>      a workgroup counts the digits of its tile (<= 16 K consecutive elements of one segment) in LDS and
//                          writes one row of the count matrix, laid out [segment][digit][tile]: ONE exclusive scan of the whole
//                          matrix (rocPRIM, 1 / 64 of the data) then holds the output offset of every (tile, digit)
//   rx_scatter_kernel<BITS>   the same tile again: a wave takes 512 consecutive elements 64 at a time, lanes with equal digits find
//                          each other with BITS ballots (match-any), rank = the wave's counter for the digit + lanes below;
//                          wave counters -> tile offsets; the tile is laid out digit by digit in LDS and written with
//                          consecutive lanes on consecutive addresses of the same digit's output range.  Stable by
//                          construction (wave, step, lane = input order), so elements with equal keys stay in input order:
//                          the sorter's tie rule (position order = gsacak's separator order, gsacak.c:1559-1561)
//   rx_classify_kernel        the buckets of a pass: those of <= 4096 elements go on the list of the bucket sort, larger ones
//                          become the segments of the next pass (its digit width follows their mean size: a skewed key
//                          distribution - runs of one letter, the terminators - costs further passes over those elements only)
//   rx_bucket_sort_kernel     one workgroup per listed bucket: keys and values staged in LDS once, a 16-bit permutation sorted by
//                          ALL remaining bits 8 at a time with the same match-any ranking (LSD, stable), one write into the
//                          result buffer.  A bucket whose key bits are used up (equal keys: already in input order) is copied.
//
// Segments: the first pass has one segment (the array), every later pass the oversize buckets of the pass before; a tile
// never straddles two segments, so every pass is a plain partition by one digit.  Partition passes ping-pong between the two
// buffers; finished buckets are always written to the alternate buffer, which holds the whole result at the end.  No library
// sort anywhere (the scan of the count matrix is rocPRIM's).
#include "kernels.hpp"
#include "prims.hpp"
#include "devutil.hpp"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace pfp {

constexpr int kTh = 512, kWaves = 8, kItems = 8;
constexpr int kSub = kTh * kItems;            // 4096 elements ranked and staged in LDS at a time
constexpr int kSuper = 4;                     // sub-tiles per workgroup: one row of the count matrix per 16 K elements
constexpr uint32_t kTile = kSub * kSuper;
constexpr uint32_t kCap = 4096;               // elements a bucket sort holds in LDS

struct Tiles {                                // device view of a pass's segments and tiles
  const uint64_t *seg_begin, *seg_end;        // [nseg] disjoint ranges of the array
  const uint32_t *tile_first;                 // [nseg + 1] tiles before segment s; [nseg] = all tiles
  uint32_t nseg;
};

// tile -> (segment, tile inside the segment, tiles of the segment, first element, length); false past the last tile
__device__ __forceinline__ bool locate(const Tiles &T, uint32_t tile, uint32_t &s, uint32_t &j, uint32_t &nts, uint64_t &beg,
                                       uint32_t &len) {
  if (tile >= T.tile_first[T.nseg]) return false;
  uint32_t lo = 0, hi = T.nseg;               // tile_first[lo] <= tile < tile_first[hi]
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (T.tile_first[mid] <= tile) lo = mid; else hi = mid;
  }
  s = lo; j = tile - T.tile_first[s]; nts = T.tile_first[s + 1] - T.tile_first[s];
  const uint64_t se = T.seg_end[s];
  beg = T.seg_begin[s] + (uint64_t)j * kTile;
  const uint64_t rem = se - beg;
  len = rem < kTile ? (uint32_t)rem : kTile;
  return true;
}

__global__ void rx_tile_counts_kernel(const uint64_t *__restrict__ seg_begin, const uint64_t *__restrict__ seg_end, uint32_t nseg,
                                      uint32_t *__restrict__ cnt) {
  const uint32_t s = BID * blockDim.x + threadIdx.x;
  if (s < nseg) cnt[s] = (uint32_t)((seg_end[s] - seg_begin[s] + kTile - 1) / kTile);
  if (s == nseg) cnt[s] = 0;
}

// lanes of the wave whose digit equals this lane's (among the lanes with valid set): one ballot per digit bit; a lane keeps the
// lanes that voted like itself - peers &= ~(vote ^ mine), `mine` all ones or all zeros: one three-input logic op per half
__device__ __forceinline__ uint64_t match_any_rt(uint32_t d, bool valid, int nbits) {
  const uint64_t act = __ballot(valid);
  uint32_t plo = (uint32_t)act, phi = (uint32_t)(act >> 32);
  for (int b = 0; b < nbits; b++) {
    const uint32_t mine = 0u - ((d >> b) & 1u);
    const uint64_t m = __ballot(mine != 0u);
    plo &= ~((uint32_t)m ^ mine);
    phi &= ~((uint32_t)(m >> 32) ^ mine);
  }
  return ((uint64_t)phi << 32) | plo;
}
template <int BITS>
__device__ __forceinline__ uint64_t match_any(uint32_t d, bool valid) {
  const uint64_t act = __ballot(valid);
  uint32_t plo = (uint32_t)act, phi = (uint32_t)(act >> 32);
#pragma unroll
  for (int b = 0; b < BITS; b++) {
    const uint32_t mine = 0u - ((d >> b) & 1u);
    const uint64_t m = __ballot(mine != 0u);
    plo &= ~((uint32_t)m ^ mine);
    phi &= ~((uint32_t)(m >> 32) ^ mine);
  }
  return ((uint64_t)phi << 32) | plo;
}
__device__ __forceinline__ uint32_t lanes_below(uint64_t mask, int lane) {
  (void)lane;
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// exclusive prefix sums of a[0 .. nbins) in place, nbins <= 2 * kTh; every thread of the workgroup calls it; wtot: kWaves words of LDS
__device__ __forceinline__ void block_excl_scan(uint32_t *a, int nbins, uint32_t *wtot) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const uint32_t a0 = 2 * t < nbins ? a[2 * t] : 0u, a1 = 2 * t + 1 < nbins ? a[2 * t + 1] : 0u;
  const uint32_t s = a0 + a1;
  const uint32_t incl = wave_incl_sum(s);
  if (lane == 63) wtot[w] = incl;
  __syncthreads();
  uint32_t base = 0;
  for (int q = 0; q < w; q++) base += wtot[q];
  const uint32_t excl = base + incl - s;
  if (2 * t < nbins) a[2 * t] = excl;
  if (2 * t + 1 < nbins) a[2 * t + 1] = excl + a0;
  __syncthreads();
}

template <int BITS>
__global__ __launch_bounds__(kTh) void rx_hist_kernel(const uint64_t *__restrict__ key, Tiles T, int shift, uint32_t *__restrict__ mat) {
  constexpr int BINS = 1 << BITS;
  __shared__ uint32_t h[BINS];
  uint32_t s, j, nts, len; uint64_t beg;
  if (!locate(T, (uint32_t)BID, s, j, nts, beg, len)) return;
  for (int b = threadIdx.x; b < BINS; b += kTh) h[b] = 0;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < len; i += kTh) atomicAdd(&h[(uint32_t)(key[beg + i] >> shift) & (BINS - 1)], 1u);
  __syncthreads();
  const uint64_t row = (uint64_t)BINS * T.tile_first[s];
  for (int b = threadIdx.x; b < BINS; b += kTh) mat[row + (uint64_t)b * nts + j] = h[b];
}

// MO: type of the scanned matrix (uint32_t while the array is shorter than 2^32, else uint64_t)
template <int BITS, class V, class MO>
__global__ __launch_bounds__(kTh) void rx_scatter_kernel(const uint64_t *__restrict__ kin, const V *__restrict__ vin, Tiles T, int shift,
                                                      const MO *__restrict__ mat, uint64_t *__restrict__ kout, V *__restrict__ vout) {
  constexpr int BINS = 1 << BITS;
  constexpr bool HASV = sizeof(V) > 1;
  __shared__ uint16_t cnt[kWaves][BINS];
  __shared__ uint32_t lstart[BINS];
  __shared__ uint64_t gbase[BINS];
  __shared__ uint64_t skey[kSub];
  __shared__ V sval[HASV ? kSub : 1];
  __shared__ uint32_t wtot[kWaves];
  uint32_t s, j, nts, len; uint64_t beg;
  if (!locate(T, (uint32_t)BID, s, j, nts, beg, len)) return;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  {
    // (the scan ran over the whole matrix: an entry minus its segment's first entry = elements of the segment placed before)
    const uint64_t row = (uint64_t)BINS * T.tile_first[s];
    const MO row0 = mat[row];
    const uint64_t sb = T.seg_begin[s];
    for (int b = t; b < BINS; b += kTh) gbase[b] = sb + (uint64_t)(MO)(mat[row + (uint64_t)b * nts + j] - row0);
  }
  for (uint32_t sub0 = 0; sub0 < len; sub0 += kSub) {
    const uint32_t sublen = len - sub0 < (uint32_t)kSub ? len - sub0 : (uint32_t)kSub;
    for (int b = t; b < BINS; b += kTh) {
#pragma unroll
      for (int q = 0; q < kWaves; q++) cnt[q][b] = 0;
    }
    __syncthreads();
    uint64_t k[kItems]; V v[kItems]; uint32_t r[kItems];
#pragma unroll
    for (int st = 0; st < kItems; st++) {
      const uint32_t i = (uint32_t)w * (kSub / kWaves) + st * 64 + lane;
      const bool valid = i < sublen;
      k[st] = valid ? kin[beg + sub0 + i] : 0ull;
      if (HASV) v[st] = valid ? vin[beg + sub0 + i] : V(0);
    }
#pragma unroll
    for (int st = 0; st < kItems; st++) {
      const uint32_t i = (uint32_t)w * (kSub / kWaves) + st * 64 + lane;
      const bool valid = i < sublen;
      const uint32_t d = (uint32_t)(k[st] >> shift) & (BINS - 1);
      const uint64_t peers = match_any<BITS>(d, valid);
      const uint32_t prev = cnt[w][d];
      r[st] = prev + lanes_below(peers, lane);
      if (valid && (peers >> lane) == 1ull) cnt[w][d] = (uint16_t)(prev + (uint32_t)__popcll(peers));      // the highest lane of the group
    }
    __syncthreads();
    for (int b = t; b < BINS; b += kTh) {
      uint32_t run = 0;
#pragma unroll
      for (int q = 0; q < kWaves; q++) { const uint32_t c = cnt[q][b]; cnt[q][b] = (uint16_t)run; run += c; }
      lstart[b] = run;
    }
    __syncthreads();
    // (the totals are kept in registers across the scan: bins 2t, 2t + 1 of this thread)
    const uint32_t tot0 = 2 * t < BINS ? lstart[2 * t] : 0u, tot1 = 2 * t + 1 < BINS ? lstart[2 * t + 1] : 0u;
    block_excl_scan(lstart, BINS, wtot);
#pragma unroll
    for (int st = 0; st < kItems; st++) {
      const uint32_t i = (uint32_t)w * (kSub / kWaves) + st * 64 + lane;
      if (i < sublen) {
        const uint32_t d = (uint32_t)(k[st] >> shift) & (BINS - 1);
        const uint32_t lp = lstart[d] + cnt[w][d] + r[st];
        skey[lp] = k[st];
        if (HASV) sval[lp] = v[st];
      }
    }
    __syncthreads();
    for (uint32_t i = t; i < sublen; i += kTh) {
      const uint64_t kk = skey[i];
      const uint32_t d = (uint32_t)(kk >> shift) & (BINS - 1);
      const uint64_t o = gbase[d] + (i - lstart[d]);
      kout[o] = kk;
      if (HASV) vout[o] = sval[i];
    }
    __syncthreads();
    if (2 * t < BINS) gbase[2 * t] += tot0;
    if (2 * t + 1 < BINS) gbase[2 * t + 1] += tot1;
    __syncthreads();
  }
}

// the buckets (s, digit) of a finished pass: small ones to the bucket-sort list, large ones (while key bits remain) to the next
// pass's segments.  counts: [0] large buckets, [1] small non-empty buckets; big_elems: elements in the large ones
template <class MO>
__global__ void rx_classify_kernel(Tiles T, int bins, const MO *__restrict__ mat, int more_bits, uint64_t *__restrict__ next_begin,
                                   uint64_t *__restrict__ next_end, uint64_t *__restrict__ small_begin, uint32_t *__restrict__ small_len,
                                   uint32_t *__restrict__ counts, unsigned long long *__restrict__ big_elems) {
  const uint64_t q = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (q >= (uint64_t)T.nseg * bins) return;
  const uint32_t s = (uint32_t)(q / bins), b = (uint32_t)(q % bins);
  const uint32_t nts = T.tile_first[s + 1] - T.tile_first[s];
  if (!nts) return;
  const uint64_t row = (uint64_t)bins * T.tile_first[s];
  const MO row0 = mat[row];
  const uint64_t beg = T.seg_begin[s] + (uint64_t)(MO)(mat[row + (uint64_t)b * nts] - row0);
  const uint64_t end = b + 1 < (uint32_t)bins ? T.seg_begin[s] + (uint64_t)(MO)(mat[row + (uint64_t)(b + 1) * nts] - row0) : T.seg_end[s];
  const uint64_t m = end - beg;
  if (!m) return;
  if (m > kCap && more_bits) {
    const uint32_t at = atomicAdd(&counts[0], 1u);
    next_begin[at] = beg; next_end[at] = end;
    atomicAdd(big_elems, (unsigned long long)m);
  } else {
    const uint32_t at = atomicAdd(&counts[1], 1u);
    small_begin[at] = beg; small_len[at] = (uint32_t)(m > 0xFFFFFFFFull ? 0xFFFFFFFFull : m);
  }
}

// nbits == 0: the bucket's elements agree in every key bit - copied as they stand (any length)
template <class V>
__global__ __launch_bounds__(kTh) void rx_bucket_sort_kernel(const uint64_t *kin, const V *vin,      // (may be kout / vout: in place)
                                                          const uint64_t *__restrict__ bbegin, const uint32_t *__restrict__ blen,
                                                          uint64_t nb, int lo_bit, int nbits, uint64_t *kout, V *vout) {
  constexpr bool HASV = sizeof(V) > 1;
  __shared__ uint64_t skey[kCap];
  __shared__ V sval[HASV ? kCap : 1];
  __shared__ uint16_t perm[2][kCap];
  __shared__ uint16_t cnt[kWaves][256];
  __shared__ uint32_t lstart[256];
  __shared__ uint32_t wtot[kWaves];
  if (BID >= nb) return;      // (a workgroup of the padded last grid row)
  const uint64_t beg = bbegin[BID];
  const uint32_t m = blen[BID];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (nbits <= 0 || m == 1) {
    if (kin != kout) for (uint32_t i = t; i < m; i += kTh) { kout[beg + i] = kin[beg + i]; if (HASV) vout[beg + i] = vin[beg + i]; }
    return;
  }
  // stage the bucket; which key bits differ inside it at all?  (a bucket is a narrow key range: the bits above the first
  // difference - and any equal low bits - need no sorting)
  __shared__ unsigned long long sdiff[kWaves];
  uint64_t diff = 0;
  const uint64_t k0 = kin[beg];
  for (uint32_t i = t; i < m; i += kTh) {
    const uint64_t kk = kin[beg + i];
    skey[i] = kk; if (HASV) sval[i] = vin[beg + i]; perm[0][i] = (uint16_t)i;
    diff |= kk ^ k0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) diff |= __shfl_xor(diff, o, 64);
  if (lane == 0) sdiff[w] = diff;
  __syncthreads();
  diff = 0;
#pragma unroll
  for (int q = 0; q < kWaves; q++) diff |= sdiff[q];
  diff = (diff >> lo_bit) & (nbits >= 64 ? ~0ull : ((1ull << nbits) - 1ull));
  if (!diff) {      // equal keys: the input order stands
    if (kin != kout) for (uint32_t i = t; i < m; i += kTh) { kout[beg + i] = skey[i]; if (HASV) vout[beg + i] = sval[i]; }
    return;
  }
  const int b_lo = lo_bit + __builtin_ctzll(diff), b_hi = lo_bit + 64 - __builtin_clzll(diff);
  const int span = b_hi - b_lo, npass = (span + 7) / 8, wbits = (span + npass - 1) / npass;
  // a wave takes `chunk` consecutive places of the permutation, 64 at a time
  const uint32_t chunk = ((m + kWaves * 64 - 1) / (kWaves * 64)) * 64;
  const int nsteps = (int)(chunk / 64);
  int cur = 0;
  for (int pb = b_lo; pb < b_hi; pb += wbits) {
    const int wb = b_hi - pb < wbits ? b_hi - pb : wbits;
    const uint32_t mask = (1u << wb) - 1u;
    if (t < 256) {
#pragma unroll
      for (int q = 0; q < kWaves; q++) cnt[q][t] = 0;
    }
    __syncthreads();
    uint32_t pe[kItems], dd[kItems], r[kItems];
#pragma unroll
    for (int st = 0; st < kItems; st++) {
      if (st >= nsteps) break;
      const uint32_t i = (uint32_t)w * chunk + st * 64 + lane;
      const bool valid = i < m;
      pe[st] = valid ? perm[cur][i] : 0u;
      dd[st] = valid ? (uint32_t)(skey[pe[st]] >> pb) & mask : 0u;
      const uint64_t peers = match_any_rt(dd[st], valid, wb);
      const uint32_t prev = cnt[w][dd[st]];
      r[st] = prev + lanes_below(peers, lane);
      if (valid && (peers >> lane) == 1ull) cnt[w][dd[st]] = (uint16_t)(prev + (uint32_t)__popcll(peers));
    }
    __syncthreads();
    if (t < 256) {
      uint32_t run = 0;
#pragma unroll
      for (int q = 0; q < kWaves; q++) { const uint32_t c = cnt[q][t]; cnt[q][t] = (uint16_t)run; run += c; }
      lstart[t] = run;
    }
    __syncthreads();
    block_excl_scan(lstart, 256, wtot);
#pragma unroll
    for (int st = 0; st < kItems; st++) {
      if (st >= nsteps) break;
      const uint32_t i = (uint32_t)w * chunk + st * 64 + lane;
      if (i < m) perm[cur ^ 1][lstart[dd[st]] + cnt[w][dd[st]] + r[st]] = (uint16_t)pe[st];
    }
    __syncthreads();
    cur ^= 1;
  }
  for (uint32_t i = t; i < m; i += kTh) {
    const uint32_t p = perm[cur][i];
    kout[beg + i] = skey[p];
    if (HASV) vout[beg + i] = sval[p];
  }
}

template <int BITS>
static void launch_hist(pfp_ctx *c, uint32_t max_tiles, const uint64_t *key, const Tiles &T, int shift, uint32_t *mat) {
  hipLaunchKernelGGL(rx_hist_kernel<BITS>, gdim(max_tiles), gdim(kTh), 0, c->stream, key, T, shift, mat);
}
template <int BITS, class V, class MO>
static void launch_scatter(pfp_ctx *c, uint32_t max_tiles, const uint64_t *kin, const V *vin, const Tiles &T, int shift, const MO *mat,
                           uint64_t *kout, V *vout) {
  hipLaunchKernelGGL((rx_scatter_kernel<BITS, V, MO>), gdim(max_tiles), gdim(kTh), 0, c->stream, kin, vin, T, shift, mat, kout, vout);
}

struct NoVal { uint8_t x; NoVal() = default; __host__ __device__ explicit NoVal(int) : x(0) {} };
static_assert(sizeof(NoVal) == 1, "keys-only marker");

template <class MO> struct RxPass {      // scratch of the passes, reused from depth to depth
  DBuf<uint32_t> tile_first, mat32, small_len, counts;
  DBuf<MO> mat;
  DBuf<uint64_t> next_begin, next_end, small_begin;
  DBuf<unsigned long long> big_elems;
};

// V = NoVal: keys only.  The result is in kalt / valt.
template <class V, class MO>
static void msd_sort_impl(pfp_ctx *c, uint64_t *key, uint64_t *kalt, V *val, V *valt, uint64_t n, int lo, int hi) {
  constexpr bool HASV = sizeof(V) > 1;
  const int total = hi - lo;
  static const bool trace = getenv("PFP_TRACE_ROUNDS") != nullptr;
  uint64_t *kc = key, *ko = kalt; V *vc = val, *vo = valt;      // kc: where the unfinished elements are; ko: the other buffer
  DBuf<uint64_t> seg_begin(c, 1), seg_end(c, 1);
  DBuf<uint32_t> one_len(c, 1);
  {
    const uint64_t zero = 0; const uint32_t len32 = (uint32_t)std::min<uint64_t>(n, 0xFFFFFFFFull);
    PFP_HIP(hipMemcpyAsync(seg_begin.p, &zero, 8, hipMemcpyHostToDevice, c->stream));
    PFP_HIP(hipMemcpyAsync(seg_end.p, &n, 8, hipMemcpyHostToDevice, c->stream));
    PFP_HIP(hipMemcpyAsync(one_len.p, &len32, 4, hipMemcpyHostToDevice, c->stream));
    sync(c);      // (stack objects)
  }
  auto bucket_sort = [&](const uint64_t *kin, const V *vin, const uint64_t *bb, const uint32_t *bl, uint64_t nb, int nbits, uint64_t elems) {
    if (!nb) return;
    KScope ks(c, "pfp::rx_bucket_sort_kernel", elems * 2 * (8 + (HASV ? sizeof(V) : 0)));
    hipLaunchKernelGGL(rx_bucket_sort_kernel<V>, gdim(nb), gdim(kTh), 0, c->stream, kin, vin, bb, bl, nb, lo, nbits, kalt, valt);
    PFP_HIP(hipGetLastError());
  };
  if (n <= kCap) { bucket_sort(kc, vc, seg_begin.p, one_len.p, 1, total, n); return; }
  RxPass<MO> P;
  P.counts.alloc(c, 2); P.big_elems.alloc(c, 1);
  uint64_t nseg = 1, elems = n;
  int consumed = 0;
  for (int depth = 0;; depth++) {
    const int rem = total - consumed;
    // digit width: what brings the segments' mean size down to ~2 K elements, in passes of at most 10 bits
    int want = 0;
    while (((elems / nseg) >> want) > 2048) want++;
    const int np = (want + 9) / 10 > 0 ? (want + 9) / 10 : 1;
    int bits = std::max(1, std::min(10, (want + np - 1) / np));
    bits = std::min(bits, rem);
    const int shift = hi - consumed - bits;
    const int bins = 1 << bits;
    PFP_REQUIRE(nseg < (1ull << 31), PFP_ELIMIT, "radix sort: too many segments");
    const uint32_t max_tiles = (uint32_t)std::min<uint64_t>(elems / kTile + nseg + 1, 0x7FFFFFFFull);
    const uint64_t entries = (uint64_t)max_tiles << bits;
    if (P.tile_first.n < nseg + 2) P.tile_first.alloc(c, nseg + 2);
    hipLaunchKernelGGL(rx_tile_counts_kernel, gdim(cdiv(nseg + 1, 256)), gdim(256), 0, c->stream, seg_begin.p, seg_end.p, (uint32_t)nseg, P.tile_first.p);
    exclusive_sum_u32(c, P.tile_first.p, P.tile_first.p, nseg + 1);
    const Tiles T{seg_begin.p, seg_end.p, P.tile_first.p, (uint32_t)nseg};
    if (P.mat32.n < entries + 1) P.mat32.alloc(c, entries + 1);
    PFP_HIP(hipMemsetAsync(P.mat32.p, 0, (entries + 1) * 4, c->stream));
    {
      KScope ks(c, "pfp::rx_hist_kernel", elems * 8 + entries * 4);
      switch (bits) {
#define CASE(B) case B: launch_hist<B>(c, max_tiles, kc, T, shift, P.mat32.p); break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10)
#undef CASE
      }
    }
    const MO *scanned;
    if constexpr (sizeof(MO) == 4) { exclusive_sum_u32(c, P.mat32.p, P.mat32.p, entries + 1); scanned = (const MO *)P.mat32.p; }
    else { if (P.mat.n < entries + 1) P.mat.alloc(c, entries + 1); exclusive_sum_u32_u64(c, P.mat32.p, (uint64_t *)P.mat.p, entries + 1); scanned = P.mat.p; }
    {
      KScope ks(c, "pfp::rx_scatter_kernel", elems * 2 * (8 + (HASV ? sizeof(V) : 0)) + entries * sizeof(MO));
      switch (bits) {
#define CASE(B) case B: launch_scatter<B, V, MO>(c, max_tiles, kc, vc, T, shift, scanned, ko, vo); break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10)
#undef CASE
      }
    }
    PFP_HIP(hipGetLastError());
    // the buckets of this pass
    const uint64_t nb = nseg << bits;
    if (P.next_begin.n < nb) { P.next_begin.alloc(c, nb); P.next_end.alloc(c, nb); P.small_begin.alloc(c, nb); P.small_len.alloc(c, nb); }
    PFP_HIP(hipMemsetAsync(P.counts.p, 0, 8, c->stream));
    PFP_HIP(hipMemsetAsync(P.big_elems.p, 0, 8, c->stream));
    const int rem_after = rem - bits;
    hipLaunchKernelGGL(rx_classify_kernel<MO>, gdim(cdiv(nb, 256)), gdim(256), 0, c->stream, T, bins, scanned, rem_after > 0 ? 1 : 0,
                       P.next_begin.p, P.next_end.p, P.small_begin.p, P.small_len.p, P.counts.p, P.big_elems.p);
    PFP_HIP(hipMemcpyAsync(c->h_scalars, P.counts.p, 8, hipMemcpyDeviceToHost, c->stream));
    PFP_HIP(hipMemcpyAsync(c->h_scalars + 1, P.big_elems.p, 8, hipMemcpyDeviceToHost, c->stream));
    sync(c);
    uint32_t cnts[2]; memcpy(cnts, c->h_scalars, 8);
    const uint64_t big = c->h_scalars[1];
    if (trace)
      fprintf(stderr, "[pfp] msd sort n=%llu bits [%d,%d) depth %d: %llu elements in %llu segments, %d bits at %d -> %u small buckets, %u large (%llu elements)\n",
              (unsigned long long)n, lo, hi, depth, (unsigned long long)elems, (unsigned long long)nseg, bits, shift, cnts[1], cnts[0],
              (unsigned long long)big);
    // the partitioned elements are in ko now: small buckets are finished from there into the result buffer (kalt)
    bucket_sort(ko, vo, P.small_begin.p, P.small_len.p, cnts[1], rem_after, elems - big);
    if (!cnts[0]) break;
    consumed += bits;
    std::swap(kc, ko); std::swap(vc, vo);
    std::swap(seg_begin, P.next_begin); std::swap(seg_end, P.next_end);
    nseg = cnts[0]; elems = big;
  }
}

template <class V>
void msd_sort_pairs_db(pfp_ctx *c, DBuf<uint64_t> &k, DBuf<uint64_t> &kalt, DBuf<V> &v, DBuf<V> &valt, size_t n, int lo, int hi) {
  if (!n) return;
  PFP_REQUIRE(k.n >= n && kalt.n >= n && v.n >= n && valt.n >= n, PFP_EINVAL, "msd_sort_pairs_db: a buffer is shorter than n");
  PFP_REQUIRE(lo >= 0 && hi <= 64 && lo < hi, PFP_EINVAL, "msd_sort_pairs_db: bad bit range");
  if (n < (1ull << 32)) msd_sort_impl<V, uint32_t>(c, k.p, kalt.p, v.p, valt.p, n, lo, hi);
  else msd_sort_impl<V, uint64_t>(c, k.p, kalt.p, v.p, valt.p, n, lo, hi);
  std::swap(k, kalt); std::swap(v, valt);      // (the result is always left in the alternates)
}
void msd_sort_keys_db(pfp_ctx *c, DBuf<uint64_t> &k, DBuf<uint64_t> &kalt, size_t n, int lo, int hi) {
  if (!n) return;
  PFP_REQUIRE(k.n >= n && kalt.n >= n, PFP_EINVAL, "msd_sort_keys_db: a buffer is shorter than n");
  PFP_REQUIRE(lo >= 0 && hi <= 64 && lo < hi, PFP_EINVAL, "msd_sort_keys_db: bad bit range");
  NoVal *none = nullptr;
  if (n < (1ull << 32)) msd_sort_impl<NoVal, uint32_t>(c, k.p, kalt.p, none, none, n, lo, hi);
  else msd_sort_impl<NoVal, uint64_t>(c, k.p, kalt.p, none, none, n, lo, hi);
  std::swap(k, kalt);
}
template void msd_sort_pairs_db<uint32_t>(pfp_ctx *, DBuf<uint64_t> &, DBuf<uint64_t> &, DBuf<uint32_t> &, DBuf<uint32_t> &, size_t, int, int);
template void msd_sort_pairs_db<uint64_t>(pfp_ctx *, DBuf<uint64_t> &, DBuf<uint64_t> &, DBuf<uint64_t> &, DBuf<uint64_t> &, size_t, int, int);

}  // namespace pfp
