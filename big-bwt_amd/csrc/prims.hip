// prims.hip -- rocPRIM instantiations (isolated in one translation unit: they compile slowly).
#include "prims.hpp"
#include "devutil.hpp"
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <mutex>
#include <set>
#include <string>
#include <utility>

namespace pfp {

#define PRIM2(call_with_temp)                                   \
  do {                                                          \
    size_t tb = 0; void *tmp = nullptr;                         \
    PFP_HIP(call_with_temp);                                    \
    DBuf<uint8_t> tbuf(c, tb ? tb : 1);                         \
    tmp = tbuf.p;                                               \
    PFP_HIP(call_with_temp);                                    \
  } while (0)

static thread_local const char *g_sort_tag = nullptr;
SortTag::SortTag(const char *what) : prev(g_sort_tag) { g_sort_tag = what; }
SortTag::~SortTag() { g_sort_tag = prev; }
const char *tagged_sort_name(const char *base) {
  if (!g_sort_tag) return base;
  static std::mutex mu;
  static std::set<std::string> names;      // (interned: the trace keeps the pointer)
  std::lock_guard<std::mutex> lk(mu);
  std::string nm = std::string(base) + " [" + g_sort_tag + "]";
  if (nm.size() > 62) nm.resize(62);
  return names.insert(nm).first->c_str();
}
// rocPRIM's onesweep with this chip's numbers (round 4, tools/microbench/sortcfg.hip, profiles/r04_sortcfg*.txt; 80 M and 1.6 G
// elements): u64 keys alone sort fastest with 9-bit digits in workgroups of 1024 x 8 keys (36 key bits: 4 passes instead of 5,
// 2.55 -> 1.99 ms / 50.8 -> 41.1 ms); (u64, u32) and (u32, u32) pairs with the default 8-bit digit but 1024 x 7 (3.38 -> 2.87 ms
// on 40 bits, 5.5 -> 4.85 on 64, 1.96 -> 1.72 for u32 keys; 1024 x 8 falls off a cliff: 4.1 ms); (u64, u64) pairs are best left alone.
// Inputs of up to 2^20 elements keep the library's merge / single-block paths; pairs below 6 M elements the library's own tile
// (7168-element tiles leave CUs idle there: 2 M pairs 0.285 -> 0.329 ms, 4 M even, 8 M 0.67 -> 0.59; profiles/r04_sortcfg_small.txt).
template <unsigned BS, unsigned IPT, unsigned BITS>
using OnesweepCfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                               rocprim::radix_sort_onesweep_config<rocprim::kernel_config<BS, IPT>, rocprim::kernel_config<BS, IPT>, BITS,
                                                                                   rocprim::block_radix_rank_algorithm::match>>;
using KeysCfg = OnesweepCfg<1024, 8, kKeysDigitBits>;
template <class K, class V> struct PairsCfgOf { using type = rocprim::default_config; };
template <> struct PairsCfgOf<uint64_t, uint32_t> { using type = OnesweepCfg<1024, 7, 8>; };
template <> struct PairsCfgOf<uint32_t, uint32_t> { using type = OnesweepCfg<1024, 7, 8>; };
template <class K, class V> using PairsCfg = typename PairsCfgOf<K, V>::type;
constexpr size_t kTunedPairsMin = size_t(6) << 20;
template <class K, class V> static const char *sort_name() {
  if (sizeof(K) == 8 && sizeof(V) == 4) return "rocprim::radix_sort_pairs<u64,u32>";
  if (sizeof(K) == 4 && sizeof(V) == 4) return "rocprim::radix_sort_pairs<u32,u32>";
  if (sizeof(K) == 8 && sizeof(V) == 8) return "rocprim::radix_sort_pairs<u64,u64>";
  return "rocprim::radix_sort_pairs<u128,u64>";
}
// Algorithmic bytes of a library sort: every element read once and written once.  (An LSD radix sort moves them once
// per 8-bit pass - plus one histogram read - and rounds 1-2 counted those passes as "algorithmic", which flattered the
// library: 0.37 of the roofline for a sort that is 0.06 by this definition.  The per-pass traffic is reported next to
// it by bench.py from the pass count: pfp::sort_passes(bb, eb).)
// rocPRIM's merge-sort path (small inputs) compares with a mask (1 << end_bit) - 1: undefined for end_bit == key width
// when begin_bit > 0.  No caller of the pair sorts passes that combination; refuse it should one appear.
#define PFP_SORT_GUARD(K, bb, eb) PFP_REQUIRE(!((bb) > 0 && (eb) == (int)(8 * sizeof(K))), PFP_EINVAL, "pair sort with begin_bit > 0 up to the key's last bit (rocPRIM merge-path mask)")
template <class K, class V>
void sort_pairs(pfp_ctx *c, const K *kin, K *kout, const V *vin, V *vout, size_t n, int bb, int eb) {
  if (!n) return;
  PFP_SORT_GUARD(K, bb, eb);
  KScope ks(c, tagged_sort_name(sort_name<K, V>()), n * 2 * (sizeof(K) + sizeof(V)));
  if (n >= kTunedPairsMin) PRIM2((rocprim::radix_sort_pairs<PairsCfg<K, V>>(tmp, tb, kin, kout, vin, vout, n, (unsigned)bb, (unsigned)eb, c->stream)));
  else PRIM2(rocprim::radix_sort_pairs(tmp, tb, kin, kout, vin, vout, n, (unsigned)bb, (unsigned)eb, c->stream));
}
template <class K, class V>
void sort_pairs_db(pfp_ctx *c, DBuf<K> &k, DBuf<K> &kalt, DBuf<V> &v, DBuf<V> &valt, size_t n, int bb, int eb) {
  if (!n) return;
  PFP_REQUIRE(k.n >= n && kalt.n >= n && v.n >= n && valt.n >= n, PFP_EINVAL, "sort_pairs_db: a buffer is shorter than n");
  PFP_SORT_GUARD(K, bb, eb);
  KScope ks(c, tagged_sort_name(sort_name<K, V>()), n * 2 * (sizeof(K) + sizeof(V)));
  rocprim::double_buffer<K> dk(k.p, kalt.p);
  rocprim::double_buffer<V> dv(v.p, valt.p);
  if (n >= kTunedPairsMin) PRIM2((rocprim::radix_sort_pairs<PairsCfg<K, V>>(tmp, tb, dk, dv, n, (unsigned)bb, (unsigned)eb, c->stream)));
  else PRIM2(rocprim::radix_sort_pairs(tmp, tb, dk, dv, n, (unsigned)bb, (unsigned)eb, c->stream));
  if (dk.current() != k.p) std::swap(k, kalt);
  if (dv.current() != v.p) std::swap(v, valt);
}
template <class K>
void sort_keys_db(pfp_ctx *c, DBuf<K> &k, DBuf<K> &kalt, size_t n, int bb, int eb) {
  if (!n) return;
  PFP_REQUIRE(k.n >= n && kalt.n >= n, PFP_EINVAL, "sort_keys_db: a buffer is shorter than n");
  // rocPRIM sorts up to 2^20 elements by merging, with a comparator whose mask is (1 << end_bit) - 1: a shift by the
  // key's whole width when end_bit == 64 - undefined, and in practice a mask of the bits BELOW begin_bit only
  // (found by fuzzing the keys-only suffix sort on small dictionaries).  The caller's low bits are a unique,
  // ascending index, so sorting the whole word gives the same order: do that where the merge path can be taken.
  // (the library's merge_sort_limit is 2^20 elements today; up to four times that the three extra passes cost well under
  //  a millisecond, beyond it the onesweep path is taken whatever a release sets the limit to - the full-size digests
  //  of the reference pin that path)
  if (bb > 0 && eb == (int)(8 * sizeof(K)) && n <= (size_t(1) << 22)) bb = 0;
  KScope ks(c, tagged_sort_name("rocprim::radix_sort_keys<u64>"), n * 2 * sizeof(K));
  rocprim::double_buffer<K> dk(k.p, kalt.p);
  PRIM2((rocprim::radix_sort_keys<KeysCfg>(tmp, tb, dk, n, (unsigned)bb, (unsigned)eb, c->stream)));
  if (dk.current() != k.p) std::swap(k, kalt);
}
void sort_keys_raw(pfp_ctx *c, const uint64_t *in, uint64_t *out, size_t n, int bb, int eb) {
  if (!n) return;
  if (bb > 0 && eb == 64 && n <= (size_t(1) << 22)) bb = 0;      // (the merge path's mask: see sort_keys_db)
  KScope ks(c, tagged_sort_name("rocprim::radix_sort_keys<u64>"), n * 16);
  PRIM2((rocprim::radix_sort_keys<KeysCfg>(tmp, tb, in, out, n, (unsigned)bb, (unsigned)eb, c->stream)));
}
template void sort_keys_db<uint64_t>(pfp_ctx *, DBuf<uint64_t> &, DBuf<uint64_t> &, size_t, int, int);
template void sort_pairs<uint64_t, uint32_t>(pfp_ctx *, const uint64_t *, uint64_t *, const uint32_t *, uint32_t *, size_t, int, int);
template void sort_pairs<uint32_t, uint32_t>(pfp_ctx *, const uint32_t *, uint32_t *, const uint32_t *, uint32_t *, size_t, int, int);
template void sort_pairs<uint64_t, uint64_t>(pfp_ctx *, const uint64_t *, uint64_t *, const uint64_t *, uint64_t *, size_t, int, int);
template void sort_pairs_db<uint64_t, uint32_t>(pfp_ctx *, DBuf<uint64_t> &, DBuf<uint64_t> &, DBuf<uint32_t> &, DBuf<uint32_t> &, size_t, int, int);
template void sort_pairs_db<uint64_t, uint64_t>(pfp_ctx *, DBuf<uint64_t> &, DBuf<uint64_t> &, DBuf<uint64_t> &, DBuf<uint64_t> &, size_t, int, int);
template void sort_pairs_db<u128, uint64_t>(pfp_ctx *, DBuf<u128> &, DBuf<u128> &, DBuf<uint64_t> &, DBuf<uint64_t> &, size_t, int, int);

// The segmented sort's tile (round 4; 12.6 GB -s, whose later rounds hand it groups of hundreds to a thousand members): the library's
// gfx942 default gives every segment above 128 elements a workgroup tile of 256 x 16 = 4096, mostly empty here.  256 x 4: 15.4 -> 11.6 ms
// (u32 keys) and 10.1 -> ~6 ms (u64 keys) per step; 256 x 8 / x 6 / x 3 / x 2 and 128 x 4 in between; wave-wide "medium" warp sorts
// (64 lanes x 8) three times slower (49 ms).  Outputs identical (full-size digests), configs[2] / c2r unchanged.
using SegCfg = rocprim::segmented_radix_sort_config<8, rocprim::kernel_config<256, 4>, rocprim::WarpSortConfig<8, 4, 256, 64, 16, 8, 256>, 1>;
template <class V>
void segsort_pairs_u32(pfp_ctx *c, const uint32_t *kin, uint32_t *kout, const V *vin, V *vout, size_t n,
                       size_t nseg, const uint32_t *seg_begin, const uint32_t *seg_end, int bb, int eb) {
  if (!n || !nseg) return;
  PFP_REQUIRE(n < 0xFFFFFFFFull, PFP_ELIMIT, "segmented sort of 2^32 or more elements");
  KScope ks(c, "rocprim::segmented_radix_sort_pairs<u32,u32>", n * (8 + 2 * sizeof(V)) + nseg * 8);
  PRIM2((rocprim::segmented_radix_sort_pairs<SegCfg>(tmp, tb, kin, kout, vin, vout, (unsigned)n, (unsigned)nseg, seg_begin, seg_end,
                                                     (unsigned)bb, (unsigned)eb, c->stream)));
}
template void segsort_pairs_u32<uint32_t>(pfp_ctx *, const uint32_t *, uint32_t *, const uint32_t *, uint32_t *, size_t, size_t,
                                          const uint32_t *, const uint32_t *, int, int);
template void segsort_pairs_u32<uint64_t>(pfp_ctx *, const uint32_t *, uint32_t *, const uint64_t *, uint64_t *, size_t, size_t,
                                          const uint32_t *, const uint32_t *, int, int);

void segsort_pairs_u64_u32(pfp_ctx *c, const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout, size_t n,
                           size_t nseg, const uint32_t *seg_begin, const uint32_t *seg_end, int bb, int eb) {
  if (!n || !nseg) return;
  PFP_REQUIRE(n < 0xFFFFFFFFull, PFP_ELIMIT, "segmented sort of 2^32 or more elements");
  KScope ks(c, "rocprim::segmented_radix_sort_pairs<u64,u32>", n * (16 + 8) + nseg * 8);
  PRIM2((rocprim::segmented_radix_sort_pairs<SegCfg>(tmp, tb, kin, kout, vin, vout, (unsigned)n, (unsigned)nseg, seg_begin, seg_end,
                                                     (unsigned)bb, (unsigned)eb, c->stream)));
}

void exclusive_sum_u32(pfp_ctx *c, const uint32_t *in, uint32_t *out, size_t n) {
  if (!n) return;
  KScope ks(c, "rocprim::scan<u32>", n * 8);
  PRIM2(rocprim::exclusive_scan(tmp, tb, in, out, 0u, n, rocprim::plus<uint32_t>(), c->stream));
}
struct U32ToU64 { __host__ __device__ uint64_t operator()(uint32_t x) const { return x; } };
void exclusive_sum_u32_u64(pfp_ctx *c, const uint32_t *in, uint64_t *out, size_t n) {
  if (!n) return;
  auto it = rocprim::make_transform_iterator(in, U32ToU64());
  KScope ks(c, "rocprim::scan<u32->u64>", n * 12);
  PRIM2(rocprim::exclusive_scan(tmp, tb, it, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), c->stream));
}
void exclusive_sum_u64(pfp_ctx *c, const uint64_t *in, uint64_t *out, size_t n) {
  if (!n) return;
  PRIM2(rocprim::exclusive_scan(tmp, tb, in, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), c->stream));
}
void inclusive_sum_u32(pfp_ctx *c, const uint32_t *in, uint32_t *out, size_t n) {
  if (!n) return;
  KScope ks(c, "rocprim::scan<u32>", n * 8);
  PRIM2(rocprim::inclusive_scan(tmp, tb, in, out, n, rocprim::plus<uint32_t>(), c->stream));
}
template <class T>
void inclusive_max(pfp_ctx *c, const T *in, T *out, size_t n) {
  if (!n) return;
  KScope ks(c, sizeof(T) == 4 ? "rocprim::scan<u32>" : "rocprim::scan<u64>", n * 2 * sizeof(T));
  PRIM2(rocprim::inclusive_scan(tmp, tb, in, out, n, rocprim::maximum<T>(), c->stream));
}
template void inclusive_max<uint32_t>(pfp_ctx *, const uint32_t *, uint32_t *, size_t);
template void inclusive_max<uint64_t>(pfp_ctx *, const uint64_t *, uint64_t *, size_t);
struct EqU8 { uint8_t v; __host__ __device__ uint32_t operator()(uint8_t x) const { return x == v ? 1u : 0u; } };
void inclusive_count_eq_u8(pfp_ctx *c, const uint8_t *bytes, uint8_t value, uint32_t *out, size_t n) {
  if (!n) return;
  auto it = rocprim::make_transform_iterator(bytes, EqU8{value});
  KScope ks(c, "rocprim::scan<u8->u32>", n * 5);
  PRIM2(rocprim::inclusive_scan(tmp, tb, it, out, n, rocprim::plus<uint32_t>(), c->stream));
}
void select_flagged_u32(pfp_ctx *c, const uint32_t *in, const uint8_t *flags, uint32_t *out, uint32_t *d_count, size_t n) {
  if (!n) { PFP_HIP(hipMemsetAsync(d_count, 0, 4, c->stream)); return; }
  KScope ks(c, "rocprim::select<u32>", n * 9);
  PRIM2(rocprim::select(tmp, tb, in, flags, out, d_count, n, c->stream));
}
// Indices of the non-zero flags, in order.  rocprim::select over (counting iterator, u8 flags) ran at
// ~95 G flags/s (2.8 ms for the 260 M hard-group flags of the 253 MB workload); the flags are bytes, so a
// workgroup can count and place 4096 of them from 16-byte loads: two streaming passes over n bytes.
constexpr int kSelTile = 4096;
// eq < 0: flag = byte != 0;  eq >= 0: flag = byte == eq.  Unaligned 16-byte loads (any base address).
__device__ __forceinline__ void load_sel16(const uint8_t *__restrict__ flags, uint64_t base, uint64_t n, int eq, uint32_t w[4]) {
  const uint32_t V = eq < 0 ? 0u : 0x01010101u * (uint32_t)eq, X = eq < 0 ? 0u : 0x01010101u;
  if (base + 16 <= n) {
    const uint4 v = ld16u(flags + base);
    w[0] = nonzero_bytes(v.x ^ V) ^ X; w[1] = nonzero_bytes(v.y ^ V) ^ X;
    w[2] = nonzero_bytes(v.z ^ V) ^ X; w[3] = nonzero_bytes(v.w ^ V) ^ X;
  } else {
    w[0] = w[1] = w[2] = w[3] = 0;
    for (int k = 0; k < 16; k++)
      if (base + k < n && (eq < 0 ? flags[base + k] != 0 : flags[base + k] == (uint8_t)eq)) w[k >> 2] |= 1u << (8 * (k & 3));
  }
}
__global__ __launch_bounds__(256) void flag_count_kernel(const uint8_t *__restrict__ flags, uint64_t n, int eq,
                                                         uint32_t *__restrict__ blocksum) {
  __shared__ uint32_t ws[4];
  const uint64_t base = (uint64_t)BID * kSelTile + (uint64_t)threadIdx.x * 16;
  uint32_t w[4];
  load_sel16(flags, base, n, eq, w);
  uint32_t cnt = __popc(w[0]) + __popc(w[1]) + __popc(w[2]) + __popc(w[3]);
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0 && (uint64_t)BID * kSelTile < n) blocksum[BID] = ws[0] + ws[1] + ws[2] + ws[3];
}
template <class I>
__global__ __launch_bounds__(256) void flag_place_kernel(const uint8_t *__restrict__ flags, uint64_t n, int eq,
                                                         const uint64_t *__restrict__ blockoff, I *__restrict__ out) {
  __shared__ uint32_t ws[4];
  if ((uint64_t)BID * kSelTile >= n) return;      // a workgroup of the padded last grid row
  const uint64_t base = (uint64_t)BID * kSelTile + (uint64_t)threadIdx.x * 16;
  uint32_t w[4];
  load_sel16(flags, base, n, eq, w);
  const uint32_t cnt = __popc(w[0]) + __popc(w[1]) + __popc(w[2]) + __popc(w[3]);
  uint32_t inc = cnt;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
  if (lane == 63) ws[wv] = inc;
  __syncthreads();
  uint64_t pos = blockoff[BID] + inc - cnt;
  for (int q = 0; q < wv; q++) pos += ws[q];
  if (!cnt) return;
#pragma unroll
  for (int k = 0; k < 16; k++)
    if ((w[k >> 2] >> (8 * (k & 3))) & 1u) out[pos++] = (I)(base + k);
}
template <class I>
static void select_bytes(pfp_ctx *c, const uint8_t *flags, int eq, I *out, uint64_t *d_count, size_t n) {
  if (!n) { PFP_HIP(hipMemsetAsync(d_count, 0, 8, c->stream)); return; }
  const size_t nblk = (n + kSelTile - 1) / kSelTile;
  DBuf<uint32_t> bsum(c, nblk + 1);
  DBuf<uint64_t> boff(c, nblk + 1);
  PFP_HIP(hipMemsetAsync(bsum.p + nblk, 0, 4, c->stream));
  KScope ks(c, "pfp::flag_place_kernel", n * 2);      // (+ flag_count_kernel and the scan of the tile counts)
  hipLaunchKernelGGL(flag_count_kernel, gdim((unsigned)nblk), gdim(256), 0, c->stream, flags, (uint64_t)n, eq, bsum.p);
  exclusive_sum_u32_u64(c, bsum.p, boff.p, nblk + 1);
  hipLaunchKernelGGL(flag_place_kernel<I>, gdim((unsigned)nblk), gdim(256), 0, c->stream, flags, (uint64_t)n, eq, boff.p, out);
  PFP_HIP(hipGetLastError());
  PFP_HIP(hipMemcpyAsync(d_count, boff.p + nblk, 8, hipMemcpyDeviceToDevice, c->stream));
}
template <class I> void select_index(pfp_ctx *c, const uint8_t *flags, I *out, uint64_t *d_count, size_t n) {
  select_bytes<I>(c, flags, -1, out, d_count, n);
}
template <class I> void select_byte_index(pfp_ctx *c, const uint8_t *bytes, uint8_t value, I *out, uint64_t *d_count, size_t n) {
  select_bytes<I>(c, bytes, (int)value, out, d_count, n);
}
template void select_index<uint32_t>(pfp_ctx *, const uint8_t *, uint32_t *, uint64_t *, size_t);
template void select_index<uint64_t>(pfp_ctx *, const uint8_t *, uint64_t *, uint64_t *, size_t);
template void select_byte_index<uint32_t>(pfp_ctx *, const uint8_t *, uint8_t, uint32_t *, uint64_t *, size_t);
template void select_byte_index<uint64_t>(pfp_ctx *, const uint8_t *, uint8_t, uint64_t *, uint64_t *, size_t);

__global__ __launch_bounds__(256) void sum_u32_kernel(const uint32_t *__restrict__ a, uint64_t n, unsigned long long *__restrict__ out) {
  __shared__ unsigned long long ws[4];
  unsigned long long x = 0;
  for (uint64_t i = (uint64_t)BID * 256 + threadIdx.x; i < n; i += (uint64_t)GDIM * 256) x += a[i];
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = x;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, ws[0] + ws[1] + ws[2] + ws[3]);
}
uint64_t count_flags(pfp_ctx *c, const uint8_t *flags, size_t n) {
  if (!n) return 0;
  const size_t nblk = (n + kSelTile - 1) / kSelTile;
  DBuf<uint32_t> bsum(c, nblk);
  DBuf<unsigned long long> tot(c, 1);
  tot.zero();
  hipLaunchKernelGGL(flag_count_kernel, gdim((unsigned)nblk), gdim(256), 0, c->stream, flags, (uint64_t)n, -1, bsum.p);
  hipLaunchKernelGGL(sum_u32_kernel, gdim((unsigned)std::min<size_t>((nblk + 255) / 256, 256)), gdim(256), 0, c->stream, bsum.p, (uint64_t)nblk, tot.p);
  PFP_HIP(hipGetLastError());
  return read_scalar(c, (const uint64_t *)tot.p);
}

}  // namespace pfp
