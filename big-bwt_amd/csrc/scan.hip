// scan.hip -- stage 1a: rolling Karp-Rabin window scan + phrase-boundary compaction.
//
// Replaces KR_window::addchar and the `hash % p == 0` test of process_file
// (reference newscan.cpp:168-202, 363-377; threaded twin pscan.hpp:44-108).
//
// The reference hash after >= w characters is  H_i = sum_j c[i-w+1+j] * 256^(w-1-j) mod q,
// q = 1999999973: a pure function of the last w bytes, so every position is independent.
// Layout: one thread owns 16 consecutive text positions (one aligned 16-byte load plus the
// 16 bytes before it as halo, all in registers), evaluates the first window with a 3-bytes-
// per-step Horner, then rolls 15 times.  All modular reductions are exact 32-bit Barrett
// steps (one v_mul_hi_u32 + one v_mul_lo_u32), no 64-bit division:
//     T < 2^(32+S):  qhat = mulhi(T >> S, floor(2^62/q)) >> (30-S) in {Q-1,Q};  r = T - qhat*q
// The `% p` test is the exact divisibility test  rotr(h * inv(p_odd), s) <= (2^32-1)/p.
// Kernel A writes a 16-bit trigger mask per thread (n/8 bytes) and a per-block count;
// kernel B turns the masks into the dense, ordered ends[] array (wave-shuffle prefix sums).
#include "kernels.hpp"
#include "prims.hpp"
#include "devutil.hpp"

namespace pfp {

constexpr uint32_t kBarrettM = 2305843040u;  // floor(2^62 / 1999999973)
static_assert((uint64_t)kBarrettM * kPrime <= (1ull << 62), "barrett");
static_assert((uint64_t)(kBarrettM + 1ull) * kPrime > (1ull << 62), "barrett");

// exact r = T mod q for T < 2^(32+S), S <= 28
template <int S>
__device__ __forceinline__ uint32_t kr_reduce(uint64_t T) {
  uint32_t xs = (uint32_t)(T >> S);
  uint32_t qhat = __umulhi(xs, kBarrettM) >> (30 - S);
  uint32_t r = (uint32_t)T - qhat * kPrime;  // true value in [0,2q) < 2^32
  uint32_t r2 = r - kPrime;
  return r < r2 ? r : r2;
}

// Rolling-step reduction: exact r = T mod q for T = hi:lo < 2^40, built from FULL-RATE 24-bit
// multiplies only (v_mul_hi_u32_u24 / v_mul_u32_u24; the 32-bit v_mul_hi/lo are quarter rate):
//   a = T >> 16 (< 2^24);  qhat = (a * floor(2^48/q)) >> 32  in {Q-1, Q}  (deficit < 2^-14.9 + 2^-8)
//   qhat < 2^10, so qhat*q mod 2^32 is two 24-bit products;  r = T - qhat*q in [0, 2q).
constexpr uint32_t kB48 = 140737u;   // floor(2^48 / 1999999973)
static_assert((uint64_t)kB48 * kPrime <= (1ull << 48) && (uint64_t)(kB48 + 1) * kPrime > (1ull << 48), "barrett48");
__device__ __forceinline__ uint32_t kr_reduce40(uint32_t lo, uint32_t hi) {
  const uint32_t a = __builtin_amdgcn_alignbit(hi, lo, 16) & 0xFFFFFFu;
  const uint32_t qhat = (uint32_t)(((uint64_t)a * (uint64_t)kB48) >> 32) & 0x3FFu;
  const uint32_t prod = __umul24(qhat, kPrime & 0xFFFFu) + (__umul24(qhat, kPrime >> 16) << 16);
  const uint32_t r = lo - prod;
  const uint32_t r2 = r - kPrime;
  return r < r2 ? r : r2;
}

// trigger test: `hash % p == 0` (newscan.cpp:367) or, in the fused chain only, membership in the
// small set of extra trigger hashes that splits giant phrases (see scan_text_adaptive)
__device__ __forceinline__ bool kr_divides(uint32_t h, const KRParams &kp) {
  uint32_t x = h * kp.pinv;
  x = __builtin_amdgcn_alignbit(x, x, kp.pshift);      // rotate right by the power of two in p
  bool t = x <= kp.plimit;
  if (kp.bloom && ((kp.bloom >> (h & 63)) & 1ull)) {
    for (uint32_t q = 0; q < kp.nextra; q++) t |= (h == kp.extra[q]);
  }
  return t;
}

// ---- the fused chain's own trigger (round 4).  The outputs of the chain do not depend on the parse, so
// pfp_bigbwt* is free to cut the text where ANY function of the last w bytes says so; the staged entry points (pfp_scan,
// pfp_parse, -k: the reference's files) keep Karp-Rabin above.  The exact `mod 1999999973, then mod p` costs ~20 vector
// instructions per text byte and bounds the scan pass at 0.2 of the HBM roofline; this one costs a third of that:
//     h(window) = sum_i c_i * m_i          m_i: fixed odd byte multipliers; evaluated as ceil(w / 4) v_dot4_u32_u8 on the unaligned
//                                          dwords of the window, which neighbouring positions share (one v_alignbyte per position)
//     trigger  <=>  (h + seed) * K mod 2^32  <  floor(2^32 / p)          (multiplicative hashing of the small integer h: density 1 / p)
// seed: the first value for which the text's FIRST window takes the decision the reference's hash takes for it (the reference
// writes 0x02 where the end-of-string byte belongs exactly when its first window triggers, SURVEY.md 2.2-Q1: reproduced), and
// for which no run of one DNA letter is cut into (w + 1)-byte crumbs.
__host__ __device__ constexpr uint32_t kFastMul(int i) {
  constexpr uint8_t m[20] = {0xB5, 0x6B, 0xD3, 0x97, 0xE9, 0x4F, 0xC7, 0x8D, 0xF1, 0x59, 0xA3, 0x3D, 0xDF, 0x75, 0xBB, 0x67, 0xCD, 0x9B, 0xE5, 0x53};
  return m[i];
}
constexpr uint32_t kFastK = 0x9E3779B1u;      // 2^32 / golden ratio, odd
// multipliers of dword q of a window of W bytes: dword 0 holds the newest four bytes (byte 3 = the window's last), bytes that lie
// before the window get 0
__host__ __device__ constexpr uint32_t fast_mvec(int W, int q) {
  uint32_t m = 0;
  for (int k = 0; k < 4; k++) { const int i = W - 4 - 4 * q + k; if (i >= 0) m |= kFastMul(i) << (8 * k); }
  return m;
}
// host / slow-path evaluation of the same hash: win = the w bytes of a window, oldest first
__host__ __device__ inline uint32_t fast_hash_bytes(const uint8_t *win, int w) {
  uint32_t h = 0;
  for (int i = 0; i < w; i++) h += (uint32_t)win[i] * kFastMul(i);
  return h;
}
// (everything on the device works with the SEEDED value hs = h + seed: the seed is the dot products' initial accumulator, and
//  the extra trigger hashes of a fast-mode KRParams are seeded values too)
__device__ __forceinline__ bool fast_extra(uint32_t hs, const KRParams &kp) {
  bool t = false;
  if ((kp.bloom >> (hs & 63)) & 1ull)
    for (uint32_t q = 0; q < kp.nextra; q++) t |= (hs == kp.extra[q]);
  return t;
}

__device__ __forceinline__ uint32_t byte_of(const uint32_t (&r)[8], int k) {
  return (r[k >> 2] >> (8 * (k & 3))) & 0xffu;
}

// any byte < 3 in x?  (exact "hasless" test)
__device__ __forceinline__ uint32_t bad_bytes(uint32_t x) {
  return (x - 0x03030303u) & ~x & 0x80808080u;
}
// ... in any of four words (the mask applied once)
__device__ __forceinline__ uint32_t bad_bytes4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  return (((a - 0x03030303u) & ~a) | ((b - 0x03030303u) & ~b) | ((c - 0x03030303u) & ~c) | ((d - 0x03030303u) & ~d)) & 0x80808080u;
}

// trigger mask of the 16 positions pos0 .. pos0 + 15 (bit j: position pos0 + j ends a phrase); reports the first byte <= 2
// interior (uniform): all 16 positions are known to be valid window ends inside the text (no clipping masks to compute)
template <int W, bool FAST = false>
__device__ __forceinline__ uint32_t kr_mask16(const uint8_t *__restrict__ tbase, uint64_t pos0, uint64_t n, const KRParams &kp,
                                              unsigned long long *__restrict__ first_bad, bool interior = false) {
  static_assert(W >= 2 && W <= 17, "register path needs w-1 <= 16");
  const uint4 pv = *reinterpret_cast<const uint4 *>(tbase + pos0 - 16);
  const uint4 cv = *reinterpret_cast<const uint4 *>(tbase + pos0);
  const uint32_t r[8] = {pv.x, pv.y, pv.z, pv.w, cv.x, cv.y, cv.z, cv.w};
  // bytes <= 2 stop the parse (newscan.cpp:364): report the first one
  uint32_t bad = bad_bytes4(cv.x, cv.y, cv.z, cv.w);
  if (bad) {
#pragma unroll
    for (int k = 0; k < 16; k++)
      if (byte_of(r, 16 + k) <= 2 && pos0 + k < n) { atomicMin(first_bad, (unsigned long long)(pos0 + k)); break; }
  }
  uint32_t mask = 0;
  if constexpr (FAST) {
    // dv[e] = the four bytes that END at byte e of r[] (bytes before r[] read as 0: their multipliers are 0)
    uint32_t dv[32];
#pragma unroll
    for (int e = 0; e < 32; e++) {
      const int hi = e >> 2, sh = (e & 3) + 1;
      dv[e] = sh == 4 ? r[hi] : __builtin_amdgcn_alignbyte(r[hi], hi ? r[hi - 1] : 0u, sh);
    }
    constexpr int nq = (W + 3) / 4;
    uint32_t hs[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      uint32_t h = kp.fseed;
#pragma unroll
      for (int q = 0; q < nq; q++) h = __builtin_amdgcn_udot4(dv[16 + j - 4 * q], fast_mvec(W, q), h, false);
      hs[j] = h;
      mask |= (h * kFastK < kp.fthr ? 1u : 0u) << j;
    }
    if (kp.bloom) {      // (uniform: a text whose giant phrases were split by extra triggers)
#pragma unroll
      for (int j = 0; j < 16; j++) mask |= (fast_extra(hs[j], kp) ? 1u : 0u) << j;
    }
  } else {
  // first window: bytes [16-(W-1) .. 16], Horner 3 bytes per step
  uint32_t h = 0;
  constexpr int first = 16 - (W - 1);
  int k = first;
#pragma unroll
  for (int s = 0; s < W / 3; s++) {
    uint64_t T = ((uint64_t)h << 24) | (byte_of(r, k) << 16) | (byte_of(r, k + 1) << 8) | byte_of(r, k + 2);
    h = kr_reduce<23>(T);
    k += 3;
  }
  if (W % 3 == 1) { h = kr_reduce<8>(((uint64_t)h << 8) | byte_of(r, k)); }
  if (W % 3 == 2) { h = kr_reduce<15>(((uint64_t)h << 16) | (byte_of(r, k) << 8) | byte_of(r, k + 1)); }
  mask = kr_divides(h, kp) ? 1u : 0u;
#pragma unroll
  for (int j = 1; j < 16; j++) {
    const uint32_t cin = byte_of(r, 16 + j), cout = byte_of(r, 16 + j - W);
    const uint64_t T = (uint64_t)cout * kp.negpw + (((uint64_t)h << 8) | cin);  // < 2^40
    h = kr_reduce40((uint32_t)T, (uint32_t)(T >> 32));
    mask |= (kr_divides(h, kp) ? 1u : 0u) << j;
  }
  }
  if (interior) return mask;
  // positions before w-1 (words shorter than w+1 are never saved, newscan.cpp:248) and past the text end
  const uint64_t lo_valid = (uint64_t)(W - 1);
  uint32_t valid = 0xFFFFu;
  if (pos0 < lo_valid) valid &= 0xFFFFu << (uint32_t)(lo_valid - pos0);
  if (n - pos0 < 16) valid &= (1u << (uint32_t)(n - pos0)) - 1u;
  return mask & valid;
}

template <int W>
__global__ __launch_bounds__(256) void kr_flag_kernel(const uint8_t *__restrict__ tbase, uint64_t n, KRParams kp,
                                                      uint16_t *__restrict__ flags16,
                                                      uint32_t *__restrict__ block_counts,
                                                      unsigned long long *__restrict__ first_bad) {
  const uint64_t c = (uint64_t)BID * 256 + threadIdx.x;  // 16-byte chunk index
  const uint64_t pos0 = c * 16;
  uint32_t mask = 0;
  if (pos0 < n) {
    mask = kr_mask16<W>(tbase, pos0, n, kp, first_bad);
    flags16[c] = (uint16_t)mask;
  }
  // block count of triggers
  uint32_t cnt = __popc(mask);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
  __shared__ uint32_t wsum[4];
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0 && (uint64_t)BID * 4096 < n) block_counts[BID] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// K1 + K2 in one pass (round 3): the trigger masks never reach HBM.  A workgroup takes the next tile of 4096 positions from a
// ticket counter (tiles are started in order, so the tiles before a running one are running or done), computes its masks,
// publishes its trigger count, and finds the number of phrase ends before its tile by a decoupled look-back over the tiles
// before it (Merrill & Garland's single-pass scan: every tile's state word holds "aggregate" or "inclusive prefix" with a
// 2-bit tag; value and tag travel in ONE 64-bit word, so relaxed device-scope atomics suffice - a release / acquire pair costs an
// L2 write-back / invalidate per tile on this multi-L2 part: 14 ms instead of 0.4); then every lane writes its ends at their final places.  ends[] has room for `cap` entries: what lies beyond is
// counted, not written (the caller repeats with the true size - a text whose windows trigger four times as often as 1 / p).
constexpr unsigned long long kTagAgg = 1ull << 62, kTagPre = 2ull << 62, kTagMask = 3ull << 62;      // tag in the two top bits
constexpr int kScanChunks = 16;                      // 4096-position chunks per tile: one look-back per 64 KB of text
template <int W, bool FAST>
__global__ __launch_bounds__(256) void kr_scan_kernel(const uint8_t *__restrict__ tbase, uint64_t n, KRParams kp, uint64_t ntiles,
                                                      unsigned int *__restrict__ ticket, unsigned long long *__restrict__ state,
                                                      uint64_t *__restrict__ ends, uint64_t cap,
                                                      unsigned long long *__restrict__ first_bad) {
  __shared__ uint32_t wsum[kScanChunks][4];
  __shared__ unsigned int s_tile;
  __shared__ unsigned long long s_excl;
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
  __syncthreads();
  const uint64_t tile = s_tile;
  if (tile >= ntiles) return;      // (a workgroup of the padded last grid row)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t mask[kScanChunks], incl[kScanChunks];
#pragma unroll 1
  for (int j = 0; j < kScanChunks; j++) {
    const uint64_t c0 = (tile * kScanChunks + j) * 4096;      // first position of the chunk (uniform)
    const uint64_t pos0 = c0 + (uint64_t)threadIdx.x * 16;
    const bool interior = c0 >= 16 && c0 + 4096 <= n;
    mask[j] = pos0 < n ? kr_mask16<W, FAST>(tbase, pos0, n, kp, first_bad, interior) : 0u;
  }
#pragma unroll
  for (int j = 0; j < kScanChunks; j++) {
    const uint32_t v = wave_incl_sum(__popc(mask[j]));
    incl[j] = v;
    if (lane == 63) wsum[j][wv] = v;
  }
  __syncthreads();
  if (wv == 0) {
    // the first wave looks back 64 tiles at a time: every lane waits for its tile's state, the lanes up to the nearest
    // tile that already knows its inclusive prefix contribute
    unsigned long long agg = 0;
#pragma unroll
    for (int j = 0; j < kScanChunks; j++) agg += (unsigned long long)wsum[j][0] + wsum[j][1] + wsum[j][2] + wsum[j][3];
    unsigned long long excl = 0;
    if (tile == 0) {
      if (lane == 0) __hip_atomic_store(&state[0], kTagPre | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (lane == 0) __hip_atomic_store(&state[tile], kTagAgg | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (long long base = (long long)tile - 1;; base -= 64) {
        const long long idx = base - lane;
        unsigned long long v = kTagPre;                      // before the first tile: prefix 0
        if (idx >= 0)
          while (((v = __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & kTagMask) == 0) __builtin_amdgcn_s_sleep(1);
        const unsigned long long pre = __ballot((v & kTagMask) == kTagPre);
        const int first = pre ? __ffsll((long long)pre) - 1 : 64;      // nearest tile with an inclusive prefix
        unsigned long long part = lane <= first ? (v & ~kTagMask) : 0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        excl += part;
        if (pre) break;
      }
      if (lane == 0) __hip_atomic_store(&state[tile], kTagPre | (excl + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) s_excl = excl;
  }
  __syncthreads();
  uint64_t chunk_base = s_excl;
#pragma unroll
  for (int j = 0; j < kScanChunks; j++) {
    uint32_t m = mask[j];
    if (m) {
      uint64_t o = chunk_base + (incl[j] - __popc(m));
      for (int q = 0; q < wv; q++) o += wsum[j][q];
      const uint64_t pos0 = ((tile * kScanChunks + j) * 256 + threadIdx.x) * 16;
      while (m) {
        const int b = __ffs(m) - 1;
        m &= m - 1;
        if (o < cap) ends[o] = pos0 + (uint64_t)b;
        o++;
      }
    }
    chunk_base += (uint64_t)wsum[j][0] + wsum[j][1] + wsum[j][2] + wsum[j][3];
  }
}

// generic window size (w > 17): same arithmetic, bytes fetched from memory
__global__ __launch_bounds__(256) void kr_flag_generic_kernel(const uint8_t *__restrict__ tbase, uint64_t n, int W,
                                                              KRParams kp, uint16_t *__restrict__ flags16,
                                                              uint32_t *__restrict__ block_counts,
                                                              unsigned long long *__restrict__ first_bad) {
  const uint64_t c = (uint64_t)BID * 256 + threadIdx.x;
  const uint64_t pos0 = c * 16;
  uint32_t mask = 0;
  if (pos0 < n) {
    for (int k = 0; k < 16; k++)
      if (pos0 + k < n && tbase[pos0 + k] <= 2) { atomicMin(first_bad, (unsigned long long)(pos0 + k)); break; }
    uint32_t h = 0;
    if (pos0 + 1 >= (uint64_t)W) {
      for (int k = 0; k < W; k++) h = kr_reduce<8>(((uint64_t)h << 8) | tbase[pos0 - (W - 1) + k]);
      if (kr_divides(h, kp)) mask |= 1u;
    }
    for (int j = 1; j < 16; j++) {
      uint64_t pos = pos0 + j;
      if (pos + 1 < (uint64_t)W) continue;
      if (pos + 1 == (uint64_t)W) {
        h = 0;
        for (int k = 0; k < W; k++) h = kr_reduce<8>(((uint64_t)h << 8) | tbase[k]);
      } else {
        uint32_t cin = tbase[pos], cout = tbase[pos - W];
        h = kr_reduce<8>((uint64_t)cout * kp.negpw + (((uint64_t)h << 8) | cin));
      }
      if (kr_divides(h, kp) && pos < n) mask |= 1u << j;
    }
    flags16[c] = (uint16_t)mask;
  }
  uint32_t cnt = __popc(mask);
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
  __shared__ uint32_t wsum[4];
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0 && (uint64_t)BID * 4096 < n) block_counts[BID] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// K2: masks -> dense ordered phrase ends (text positions)
__global__ __launch_bounds__(256) void kr_compact_kernel(const uint16_t *__restrict__ flags16, uint64_t nchunks,
                                                         const uint32_t *__restrict__ block_offsets,
                                                         uint64_t *__restrict__ ends) {
  if ((uint64_t)BID * 256 >= nchunks) return;      // a workgroup of the padded last grid row
  const uint64_t c = (uint64_t)BID * 256 + threadIdx.x;
  uint32_t mask = (c < nchunks) ? flags16[c] : 0u;
  uint32_t cnt = __popc(mask);
  // inclusive wave scan
  uint32_t incl = cnt;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  __shared__ uint32_t wsum[4];
  if (lane == 63) wsum[threadIdx.x >> 6] = incl;
  __syncthreads();
  uint32_t wbase = 0;
  for (int i = 0; i < (int)(threadIdx.x >> 6); i++) wbase += wsum[i];
  uint64_t o = (uint64_t)block_offsets[BID] + wbase + (incl - cnt);
  while (mask) {
    int b = __ffs(mask) - 1;
    mask &= mask - 1;
    ends[o++] = c * 16 + (uint64_t)b;
  }
}

// ------------------------------------------------------------------ host side

static uint32_t inv_mod_2_32(uint32_t odd) {
  uint32_t x = odd;  // Newton: correct to 3 bits initially
  for (int i = 0; i < 5; i++) x *= 2u - odd * x;
  return x;
}

KRParams make_kr_params(int w, uint64_t p) {
  KRParams kp{};
  uint64_t pw = 1;
  for (int i = 0; i < w; i++) pw = (pw * 256) % kPrime;  // 256^w mod q
  kp.negpw = (uint32_t)(kPrime - pw);
  if (p >= (1ull << 31)) {
    // h < 2^31 <= p: only h == 0 is divisible
    kp.pinv = 1; kp.pshift = 0; kp.plimit = 0;
  } else {
    uint32_t pp = (uint32_t)p; int s = 0;
    while ((pp & 1u) == 0) { pp >>= 1; s++; }
    kp.pinv = inv_mod_2_32(pp); kp.pshift = (uint32_t)s; kp.plimit = (uint32_t)(0xFFFFFFFFull / p);
  }
  return kp;
}

// the fused chain's trigger for a staged text (see fast_triggers): needs the text's first window for the seed
uint32_t window_hash_host(const uint8_t *win, int w, uint32_t seed) { return fast_hash_bytes(win, w) + seed; }
KRParams make_fast_params(pfp_ctx *c, const StagedText &tx, uint64_t n, int w, uint64_t p) {
  uint8_t fw[32] = {0};
  const bool have = n >= (uint64_t)w && w <= 32;
  if (have) {
    PFP_HIP(hipMemcpyAsync(fw, tx.tbase(), (size_t)w, hipMemcpyDeviceToHost, c->stream));
    sync(c);
  }
  return make_fast_params_host(have ? fw : nullptr, w, p, c->parse_density);
}
// the same from the window's bytes (host): what the multi-GPU hosts' rank 0 computes for everybody (pfp_dist_parse_plan)
KRParams make_fast_params_host(const uint8_t *first_window, int w, uint64_t p, double density_setting) {
  KRParams kp = make_kr_params(w, p);
  if (w < 4 || w > 17) return kp;          // (the register path of the scan kernel; wider windows keep Karp-Rabin)
  kp.fast = 1;
  // pfp_set_parse_density: 0 (default) = the chain chooses between the nominal density 1 / p and twice that (see
  // choose_parse_density below): the scan then cuts at 2 / p and remembers the nominal threshold; d > 0 = cut at d / p
  const bool auto_density = !(density_setting > 0);
  // (the dense candidate aims at phrases of ~48 bytes: measured optimum for collections at 10^-3 mutations per base, -p 100: 2-2.5
  //  times the nominal density, -p 200: 4 times)
  const double dens_auto = std::min(8.0, std::max(1.0, (double)p / 48.0));
  const double dens = auto_density ? dens_auto : density_setting;
  kp.fdens = (float)dens;
  const double thr_nom = 4294967296.0 / (double)p, thr = thr_nom * dens;
  kp.fthr = thr >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)thr;
  kp.fthr_nom = auto_density ? (thr_nom >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)thr_nom) : kp.fthr;
  kp.fauto = (auto_density && dens > 1.0) ? 1u : 0u;
  bool have_first = false, ref_first = false;
  uint8_t fw[32] = {0};
  if (first_window) {
    memcpy(fw, first_window, (size_t)w);
    uint64_t h = 0;
    for (int k = 0; k < w; k++) h = (h * 256 + fw[k]) % kPrime;      // newscan.cpp:168-202
    ref_first = h % p == 0;
    have_first = true;
  }
  const uint32_t h_first = fast_hash_bytes(fw, w);      // (unseeded here: the loop below is what chooses the seed)
  auto fires = [&](uint32_t h, uint32_t seed) { return (uint32_t)((h + seed) * kFastK) < kp.fthr; };
  // (the first window must decide like the reference's hash at whichever density is taken in the end: inside the nominal
  //  threshold if the reference cuts there, outside the scan's threshold if it does not)
  auto first_ok = [&](uint32_t seed) {
    const uint32_t x = (uint32_t)((h_first + seed) * kFastK);
    return ref_first ? x < kp.fthr_nom : x >= kp.fthr;
  };
  uint32_t best = 0;
  for (uint32_t seed = 0; seed < (1u << 20); seed++) {
    if (have_first && !first_ok(seed)) continue;
    best = seed;
    bool crumbs = false;
    if (p >= 32)
      for (const char ch : {'A', 'C', 'G', 'T', 'N', 'a', 'c', 'g', 't', 'n'}) {
        uint8_t run[32];
        memset(run, ch, sizeof run);
        const uint32_t hr = fast_hash_bytes(run, w);
        if (fires(hr, seed) && !(have_first && hr == h_first)) crumbs = true;
      }
    if (!crumbs) break;
  }
  kp.fseed = best;
  return kp;
}

template <int W>
static void launch_flag(pfp_ctx *c, int nblocks, const uint8_t *tbase, uint64_t n, const KRParams &kp,
                        uint16_t *flags16, uint32_t *bc, unsigned long long *fb) {
  hipLaunchKernelGGL(kr_flag_kernel<W>, gdim(nblocks), gdim(256), 0, c->stream, tbase, n, kp, flags16, bc, fb);
}

void scan_flags(pfp_ctx *c, const uint8_t *tbase, uint64_t n, int w, uint64_t p, uint16_t *flags16,
                uint32_t *block_counts, unsigned long long *first_bad, const KRParams *kp_override) {
  KRParams kp = kp_override ? *kp_override : make_kr_params(w, p);
  uint64_t nchunks = cdiv64(n, 16);
  int nblocks = (int)cdiv64(nchunks, 256);
  if (nblocks == 0) return;
  KScope ks(c, "pfp::kr_flag_kernel", n + n / 8);
  switch (w) {
#define CASE(W) case W: launch_flag<W>(c, nblocks, tbase, n, kp, flags16, block_counts, first_bad); break;
    CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16) CASE(17)
#undef CASE
    default:
      hipLaunchKernelGGL(kr_flag_generic_kernel, gdim(nblocks), gdim(256), 0, c->stream, tbase, n, w, kp, flags16,
                         block_counts, first_bad);
  }
  PFP_HIP(hipGetLastError());
}

template <int W>
static void launch_scan(pfp_ctx *c, uint64_t ntiles, const uint8_t *tbase, uint64_t n, const KRParams &kp, unsigned int *ticket,
                        unsigned long long *state, uint64_t *ends, uint64_t cap, unsigned long long *fb) {
  if (kp.fast) hipLaunchKernelGGL((kr_scan_kernel<W, true>), gdim(ntiles), gdim(256), 0, c->stream, tbase, n, kp, ntiles, ticket, state, ends, cap, fb);
  else hipLaunchKernelGGL((kr_scan_kernel<W, false>), gdim(ntiles), gdim(256), 0, c->stream, tbase, n, kp, ntiles, ticket, state, ends, cap, fb);
}

// Full stage 1a on a staged text.  Returns the number of trigger ends; d_ends receives them.
// If a byte <= 2 is found before n, *n_used is set to its position and the scan is redone on
// the prefix (the reference stops reading there: newscan.cpp:364).
uint64_t scan_text(pfp_ctx *c, const StagedText &tx, uint64_t n, int w, uint64_t p, DBuf<uint64_t> &d_ends,
                   uint64_t *n_used, const KRParams *kp_override) {
  PFP_REQUIRE(w >= 1 && w <= 4096, PFP_EINVAL, "window size out of range");
  uint64_t cur_n = n;
  uint64_t cap_hint = 0;      // a pass that found more ends than it had room for tells the next one how many
  for (int attempt = 0; attempt < 4; attempt++) {
    uint64_t nchunks = cdiv64(cur_n, 16);
    int nblocks = (int)cdiv64(nchunks, 256);
    *n_used = cur_n;
    if (nblocks == 0) { d_ends.alloc(c, 1); return 0; }
    DBuf<unsigned long long> fbad(c, 1);
    PFP_HIP(hipMemsetAsync(fbad.p, 0xff, 8, c->stream));
    if (w >= 4 && w <= 17) {
      // one pass: masks, counts, look-back and placement in one kernel (kr_scan_kernel)
      const KRParams kp = kp_override ? *kp_override : make_kr_params(w, p);
      const uint64_t ntiles = cdiv64((uint64_t)nblocks, kScanChunks);
      const uint64_t expect = (uint64_t)((double)(cur_n / (p ? p : 1)) * (kp.fast && kp.fdens > 1.0f ? (double)kp.fdens : 1.0));
      const uint64_t cap = cap_hint ? cap_hint : std::min<uint64_t>(cur_n, expect * 4 + 65536);
      d_ends.alloc(c, cap + 1);
      DBuf<unsigned long long> state(c, ntiles);
      DBuf<unsigned int> ticket(c, 1);
      state.zero(); ticket.zero();
      {
        KScope ks(c, "pfp::kr_scan_kernel", cur_n + 8 * std::min<uint64_t>(cap, expect));
        switch (w) {
#define CASE(W) case W: launch_scan<W>(c, ntiles, tx.tbase(), cur_n, kp, ticket.p, state.p, d_ends.p, cap, fbad.p); break;
          CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16) CASE(17)
#undef CASE
        }
        PFP_HIP(hipGetLastError());
      }
      PFP_HIP(hipMemcpyAsync(c->h_scalars, state.p + (ntiles - 1), 8, hipMemcpyDeviceToHost, c->stream));
      PFP_HIP(hipMemcpyAsync(c->h_scalars + 1, fbad.p, 8, hipMemcpyDeviceToHost, c->stream));
      sync(c);
      const uint64_t total = c->h_scalars[0] & ~kTagMask;
      const uint64_t fb = c->h_scalars[1];
      if (fb < cur_n) {
        PFP_REQUIRE(cur_n == n, PFP_EHIP, "scan: special byte after truncation");
        cur_n = fb;
        tx.restage_tail(c, cur_n, w);
        cap_hint = 0;
        continue;
      }
      if (total > cap) { cap_hint = total; continue; }      // more triggers than room: once more with the true size
      return total;
    }
    DBuf<uint16_t> flags16(c, nchunks);
    DBuf<uint32_t> bcnt(c, (size_t)nblocks + 1), boff(c, (size_t)nblocks + 1);
    PFP_HIP(hipMemsetAsync(bcnt.p + nblocks, 0, 4, c->stream));
    scan_flags(c, tx.tbase(), cur_n, w, p, flags16.p, bcnt.p, fbad.p, kp_override);
    exclusive_sum_u32(c, bcnt.p, boff.p, (size_t)nblocks + 1);
    // total + first_bad in one sync
    PFP_HIP(hipMemcpyAsync(c->h_scalars, boff.p + nblocks, 4, hipMemcpyDeviceToHost, c->stream));
    PFP_HIP(hipMemcpyAsync(c->h_scalars + 1, fbad.p, 8, hipMemcpyDeviceToHost, c->stream));
    sync(c);
    uint32_t total; memcpy(&total, c->h_scalars, 4);
    uint64_t fb = c->h_scalars[1];
    if (fb < cur_n) {
      // truncate at the first special byte and rescan the prefix with Dollar padding re-staged
      PFP_REQUIRE(cur_n == n, PFP_EHIP, "scan: special byte after truncation");
      cur_n = fb;
      tx.restage_tail(c, cur_n, w);
      continue;
    }
    d_ends.alloc(c, (size_t)total + 1);
    KScope ks(c, "pfp::kr_compact_kernel", nchunks * 2 + (uint64_t)total * 8);
    if (total)
      hipLaunchKernelGGL(kr_compact_kernel, gdim(nblocks), gdim(256), 0, c->stream, flags16.p, nchunks, boff.p, d_ends.p);
    PFP_HIP(hipGetLastError());
    return total;
  }
  PFP_REQUIRE(false, PFP_EHIP, "scan did not settle");
  return 0;
}

// ---------------------------------------------------------------- adaptive extra triggers
//
// A text with a multi-megabyte low-complexity stretch (the N runs of a chromosome) yields one
// giant phrase; its suffixes share prefixes as long as the stretch and prefix doubling over the
// dictionary needs log2(length) full rounds for it.  Prefix-free parsing is correct for ANY set
// of trigger windows, and the final BWT/SA are functions of the text alone,
// so the fused chain may enlarge the reference's trigger set {hash % p == 0} by a few specific
// window hashes taken from inside giant phrases: every occurrence of such a window then also
// ends a phrase, a periodic run collapses into many copies of one short phrase, and the
// dictionary keeps short words only.  The staged entry points (pfp_scan / pfp_parse), whose
// outputs are the reference's files, never do this.

__global__ void giant_phrases_kernel(const uint64_t *__restrict__ ends, uint64_t ne, uint64_t n, int w,
                                     uint64_t max_len, uint32_t *__restrict__ count, uint64_t *__restrict__ picks,
                                     uint32_t cap) {
  uint64_t k = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (k > ne) return;
  uint64_t e = k < ne ? ends[k] + 1 : n + (uint64_t)w;             // T' index of the last byte
  uint64_t s = k == 0 ? 0 : ends[k - 1] + 2 - (uint64_t)w;         // T' index of the first byte
  uint64_t len = e - s + 1;
  if (len <= max_len) return;
  uint32_t i = atomicAdd(count, 1u);
  if (i < cap) picks[i] = s + max_len / 2;                         // first of kCand candidate window ends, well inside
}
// hashes of kCand consecutive windows starting at every pick (one block per pick)
constexpr uint32_t kCand = 256;
__global__ void window_hash_kernel(const uint8_t *__restrict__ tbase, int w, const uint64_t *__restrict__ picks,
                                   uint32_t *__restrict__ out, int fast, uint32_t seed) {
  uint64_t e = picks[BID] + threadIdx.x;
  uint32_t h = 0;
  if (fast) { out[BID * kCand + threadIdx.x] = fast_hash_bytes(tbase + e - (uint64_t)(w - 1), w) + seed; return; }
  for (int k = 0; k < w; k++) h = kr_reduce<8>(((uint64_t)h << 8) | tbase[e - (uint64_t)(w - 1) + (uint64_t)k]);
  out[BID * kCand + threadIdx.x] = h;
}

// one proposal round: window hashes (at most 8) that would split the phrases of this scan result
// that are longer than max_phrase; appended to kp when new.  Returns how many were added.
uint32_t propose_extra_triggers(pfp_ctx *c, const StagedText &tx, uint64_t n_used, int w, uint64_t max_phrase,
                                const DBuf<uint64_t> &d_ends, uint64_t ne, KRParams &kp) {
  const uint32_t cap = 8;
  if (!max_phrase || max_phrase < 4 * (uint64_t)w + 2 * kCand + 64 || kp.nextra + cap > KRParams::kMaxExtra) return 0;
  DBuf<uint32_t> cnt(c, 1), hashes(c, cap * kCand);
  DBuf<uint64_t> picks(c, cap);
  std::vector<uint32_t> hv(cap * kCand);
  cnt.zero();
  hipLaunchKernelGGL(giant_phrases_kernel, gdim(cdiv(ne + 1, 256)), gdim(256), 0, c->stream, d_ends.p, ne, n_used, w,
                     max_phrase, cnt.p, picks.p, cap);
  uint32_t ng = read_scalar(c, cnt.p);
  if (!ng) return 0;
  uint32_t take = ng < cap ? ng : cap;
  hipLaunchKernelGGL(window_hash_kernel, gdim(take), gdim(kCand), 0, c->stream, tx.tbase(), w, picks.p, hashes.p, (int)kp.fast, kp.fseed);
  PFP_HIP(hipMemcpyAsync(hv.data(), hashes.p, (size_t)take * kCand * 4, hipMemcpyDeviceToHost, c->stream));
  sync(c);
  // The first window of the text must never become a trigger by this route: the reference writes 0x02
  // where the end-of-string 0x00 belongs when (and only when) ITS trigger set fires there (SURVEY.md
  // 2.2-Q1, reproduced), so an extra trigger on that window would change one output byte.
  uint32_t h_first = 0xFFFFFFFFu;
  if (n_used >= (uint64_t)w) {
    std::vector<uint8_t> fw((size_t)w);
    PFP_HIP(hipMemcpyAsync(fw.data(), tx.tbase(), (size_t)w, hipMemcpyDeviceToHost, c->stream));
    sync(c);
    uint64_t h = 0;
    for (int k = 0; k < w; k++) h = (h * 256 + fw[k]) % 1999999973ull;      // newscan.cpp:168-202
    h_first = kp.fast ? fast_hash_bytes(fw.data(), w) + kp.fseed : (uint32_t)h;
  }
  // per giant phrase: the candidate window that is rarest among kCand consecutive ones.  A
  // window seen more than 4 times there recurs every < 64 bytes (a run of one repeated char,
  // a short period): adding it would only trade the giant phrase for millions of tiny ones.
  uint32_t added = 0;
  for (uint32_t q = 0; q < take; q++) {
    const uint32_t *cand = hv.data() + (size_t)q * kCand;
    uint32_t best = 0, best_mult = kCand + 1;
    for (uint32_t x = 0; x < kCand; x++) {
      uint32_t mult = 0;
      if (cand[x] == h_first) continue;
      for (uint32_t y = 0; y < kCand; y++) mult += cand[y] == cand[x];
      if (mult < best_mult) { best_mult = mult; best = cand[x]; }
    }
    if (best_mult > 4) continue;
    bool dup = false;
    for (uint32_t z = 0; z < kp.nextra; z++) dup |= kp.extra[z] == best;
    if (!dup) { kp.extra[kp.nextra++] = best; kp.bloom |= 1ull << (best & 63); added++; }
  }
  return added;
}

// ---------------------------------------------------------------- phrase length by repetitiveness (round 4)
//
// For c copies of a genome at mutation rate r the dictionary grows with the phrase length L (about G (1 + c r L) bytes: every
// mutation makes a new word of length ~L) while the parse shrinks (n / L phrases), and the dictionary's suffix sort and merge are
// what the chain spends its time on: twice the trigger density takes configs[2] from 37.9 to 32.6 ms and halves the peak memory
// (12.6 GB: 141 -> 78 GB); a single genome, whose dictionary is the text whatever L is, only pays for the longer parse.  The
// outputs do not depend on the choice.  So the fused chain scans ONCE at the dense candidate (phrases of ~48
// bytes: p / 48 times the nominal density; `x < d thr` contains `x < thr`), looks at a content-defined sample of the cuts
// (x < thr / 16: the same loci in every copy) and sorts the hashes of the 64-byte contexts before them: contexts seen at least
// twice are loci, contexts seen once the variants around them; where the variants outweigh the loci all cuts are kept, else the
// cuts outside the nominal threshold are dropped again and the parse is the one -p asks for.
// pfp_set_parse_density(ctx, 1) / PFP_PARSE_DENSITY=1 pins the nominal density; pfp_stats.parse_density reports what was used.
constexpr uint64_t kSampleSlots = 1024;
// one pass over the cuts: is a cut inside the nominal threshold (nominal[k])?  is it one of the sample (x < thr_sample)? - then the
// hash of the 64 bytes that end with its window goes on the sample list (wave-aggregated append: the list is sorted afterwards,
// its order does not matter); sample_n counts every sampled cut, also those beyond the list's capacity
__global__ __launch_bounds__(256) void classify_ends_kernel(const uint8_t *__restrict__ tbase, const uint64_t *__restrict__ ends, uint64_t ne,
                                                            int w, uint32_t seed, uint32_t thr_nom, uint32_t thr_sample, uint64_t min_end,
                                                            uint8_t *__restrict__ nominal, uint64_t *__restrict__ sample_hash,
                                                            uint64_t slot_cap, unsigned long long *__restrict__ slot_n) {
  const uint64_t k = (uint64_t)BID * blockDim.x + threadIdx.x;
  bool smp = false;
  uint64_t e = 0;
  if (k < ne) {
    e = ends[k];
    const uint32_t x = (fast_hash_bytes(tbase + e - (uint64_t)(w - 1), w) + seed) * kFastK;
    nominal[k] = x < thr_nom ? 1 : 0;
    smp = x < thr_sample && e >= min_end;
  }
  const unsigned long long m = __ballot(smp);
  if (!m) return;
  const int lane = threadIdx.x & 63;
  // (kSampleSlots lists with a counter each: one counter for all waves was 2.3 ms of contended atomics on 16 M cuts)
  const uint64_t slot = BID % kSampleSlots;
  unsigned long long base = 0;
  if (lane == __ffsll((long long)m) - 1) base = atomicAdd(&slot_n[slot], (unsigned long long)__popcll(m));
  base = __shfl(base, __ffsll((long long)m) - 1, 64);
  if (!smp) return;
  uint64_t at = base + (uint64_t)__popcll(m & ((1ull << lane) - 1ull));
  if (at >= slot_cap) return;
  at += slot * slot_cap;
  const uint8_t *q = tbase + e - 63;      // (the staging buffer has 64 bytes of front padding)
  uint64_t h = 0x9E3779B97F4A7C15ull;
#pragma unroll
  for (int j = 0; j < 8; j++) h = fmix64(h ^ ld8u(q + 8 * j)) + 0x632BE59BD9B4E019ull * (uint64_t)(j + 1);
  sample_hash[at] = h;
}
// out[0] = distinct values, out[1] = values that occur exactly once
__global__ void count_distinct_kernel(const uint64_t *__restrict__ sorted, uint64_t ns, unsigned long long *__restrict__ out) {
  const uint64_t k = (uint64_t)BID * blockDim.x + threadIdx.x;
  const bool head = k < ns && (k == 0 || sorted[k] != sorted[k - 1]);
  const bool single = head && (k + 1 == ns || sorted[k + 1] != sorted[k]);
  const unsigned long long m = __ballot(head), m1 = __ballot(single);
  if ((threadIdx.x & 63) == 0 && m) { atomicAdd(out, (unsigned long long)__popcll(m)); if (m1) atomicAdd(out + 1, (unsigned long long)__popcll(m1)); }
}
__global__ void gather_ends_kernel(const uint64_t *__restrict__ ends, const uint32_t *__restrict__ idx, uint64_t cnt, uint64_t *__restrict__ out) {
  const uint64_t k = (uint64_t)BID * blockDim.x + threadIdx.x;
  if (k < cnt) out[k] = ends[idx[k]];
}
// the scan cut at kp.fthr = twice the nominal density: keep that (returns ne, *dense = true) or fall back to the nominal cuts
// (d_ends compacted, kp.fthr lowered to the nominal threshold for every later rescan)
// classify the cuts of a scan at the dense candidate: out.nominal[k] = cut k lies inside the nominal threshold; out.hashes = the
// context hashes of the sampled cuts at or after min_end, sorted, out.ns of them
void classify_cuts(pfp_ctx *c, const StagedText &tx, int w, const DBuf<uint64_t> &d_ends, uint64_t ne, const KRParams &kp, uint64_t min_end,
                   CutSample &out) {
  PFP_REQUIRE(ne < 0xFFFFFFFFull, PFP_ELIMIT, "more than 2^32 - 2 phrases (bwtparse.c:93)");
  out.nominal.alloc(c, ne + 16);
  DBuf<uint8_t> &nominal = out.nominal;
  PFP_HIP(hipMemsetAsync(nominal.p + ne, 0, 16, c->stream));
  // the sample (about 1 / (16 x density) of the cuts) is collected in kSampleSlots lists; unused places keep the all-ones filler,
  // which sorts behind every hash
  const uint64_t slot_cap = (ne / 8 + kSampleSlots - 1) / kSampleSlots + 64, scap = slot_cap * kSampleSlots;
  DBuf<uint64_t> h(c, scap);
  DBuf<unsigned long long> sn(c, kSampleSlots);
  PFP_HIP(hipMemsetAsync(h.p, 0xff, scap * 8, c->stream));
  sn.zero();
  hipLaunchKernelGGL(classify_ends_kernel, gdim(cdiv(ne, 256)), gdim(256), 0, c->stream, tx.tbase(), d_ends.p, ne, w, kp.fseed, kp.fthr_nom,
                     kp.fthr_nom / 16u, min_end, nominal.p, h.p, slot_cap, sn.p);
  std::vector<unsigned long long> hsn(kSampleSlots);
  PFP_HIP(hipMemcpyAsync(hsn.data(), sn.p, kSampleSlots * 8, hipMemcpyDeviceToHost, c->stream));
  sync(c);
  uint64_t ns = 0;
  for (unsigned long long v : hsn) ns += std::min<uint64_t>(v, slot_cap);
  out.ns = ns;
  out.hashes.alloc(c, scap);
  sort_keys_raw(c, h.p, out.hashes.p, scap, 0, 64);      // (the valid hashes first: the filler is all ones)
}
// a sorted sample of context hashes (a rank's own, or the ranks' samples gathered and sorted again): does it show a collection
// whose variants outweigh its loci?
bool sample_says_dense(pfp_ctx *c, const uint64_t *d_sorted, uint64_t ns, uint64_t p) {
  if (ns < 1024) return false;
  DBuf<unsigned long long> nd(c, 2);
  nd.zero();
  hipLaunchKernelGGL(count_distinct_kernel, gdim(cdiv(ns, 256)), gdim(256), 0, c->stream, d_sorted, ns, nd.p);
  PFP_HIP(hipMemcpyAsync(c->h_scalars, nd.p, 16, hipMemcpyDeviceToHost, c->stream));
  sync(c);
  const uint64_t distinct = c->h_scalars[0], singles = c->h_scalars[1], loci = distinct - singles;
  // contexts seen twice or more are the collection's loci, contexts seen once the variants around them: V / U = copies x 64 r, and
  // what shorter phrases save is the variants' share of the dictionary, c r L = (V / U) (p / 64): worth it from about 1
  // (16 copies at 10^-3: 1.5, measured -14 %; 64 copies at 10^-4: 0.6, left alone)
  // - in a COLLECTION, that is: where most sampled contexts repeat at all.  (A single genome with satellite arrays and repeat
  // families has loci too, and singles in plenty - its unique sequence: c2r lost 16 ms to shorter phrases before this condition.)
  return loci >= 64 && singles * 2 <= ns && singles * p >= loci * 64;
}
// drop the cuts outside the nominal threshold again; returns how many stay
uint64_t keep_nominal_cuts(pfp_ctx *c, DBuf<uint64_t> &d_ends, uint64_t ne, const DBuf<uint8_t> &nominal) {
  const uint64_t n1 = count_flags(c, nominal.p, ne);
  DBuf<uint64_t> keep(c, n1 + 1), cnt(c, 1);
  DBuf<uint32_t> idx(c, n1 + 1);
  select_index<uint32_t>(c, nominal.p, idx.p, cnt.p, ne);
  if (n1) hipLaunchKernelGGL(gather_ends_kernel, gdim(cdiv(n1, 256)), gdim(256), 0, c->stream, d_ends.p, idx.p, n1, keep.p);
  PFP_HIP(hipGetLastError());
  d_ends = std::move(keep);
  return n1;
}
// the scan cut at kp.fthr = the dense candidate: keep that (returns ne, *dense = true) or fall back to the nominal cuts
// (d_ends compacted, kp.fthr lowered to the nominal threshold for every later rescan)
static uint64_t choose_parse_density(pfp_ctx *c, const StagedText &tx, int w, uint64_t p, DBuf<uint64_t> &d_ends, uint64_t ne, KRParams &kp, bool *dense) {
  CutSample cs;
  classify_cuts(c, tx, w, d_ends, ne, kp, 0, cs);
  *dense = sample_says_dense(c, cs.hashes.p, cs.ns, p);
  if (*dense) return ne;
  kp.fthr = kp.fthr_nom;
  return keep_nominal_cuts(c, d_ends, ne, cs.nominal);
}

uint64_t scan_text_adaptive(pfp_ctx *c, const StagedText &tx, uint64_t n, int w, uint64_t p, uint64_t max_phrase,
                            DBuf<uint64_t> &d_ends, uint64_t *n_used, uint32_t *n_extra) {
  KRParams kp = c->fast_triggers ? make_fast_params(c, tx, n, w, p) : make_kr_params(w, p);
  *n_extra = 0;
  uint64_t ne = scan_text(c, tx, n, w, p, d_ends, n_used, &kp);
  c->stats.parse_density = kp.fast ? (c->parse_density > 0 ? c->parse_density : 1.0) : 1.0;
  if (kp.fast && kp.fauto && ne > 0) {
    bool dense = false;
    ne = choose_parse_density(c, tx, w, p, d_ends, ne, kp, &dense);
    c->stats.parse_density = dense ? (double)kp.fdens : 1.0;
  }
  if (kp.fast && ne == 0) {      // no cut at all: the reference's own hash decides whether this text has a parse (bwtparse.c:244)
    kp = make_kr_params(w, p);
    ne = scan_text(c, tx, n, w, p, d_ends, n_used, &kp);
    c->stats.parse_density = 1.0;
  }
  for (int iter = 0; iter < 4 && max_phrase; iter++) {
    if (!propose_extra_triggers(c, tx, *n_used, w, max_phrase, d_ends, ne, kp)) break;
    ne = scan_text(c, tx, *n_used, w, p, d_ends, n_used, &kp);
  }
  *n_extra = kp.nextra;
  return ne;
}

}  // namespace pfp
