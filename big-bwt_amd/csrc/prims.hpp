// prims.hpp -- device-wide sort / scan / select building blocks (thin wrappers over rocPRIM,
// AMD's own header-only primitives library; all hot-path kernels are hand-written elsewhere).
#pragma once
#include "common.hpp"

namespace pfp {
using u128 = unsigned __int128;
// what a library sort is FOR, for the kernel trace: "rocprim::radix_sort_pairs<u64,u32> [phrase hashes]" instead of one row for
// every sort of that type (round 4: the roofline's dominant kernel should be one kind of launch).  Scoped; nests innermost-wins.
struct SortTag {
  const char *prev;
  explicit SortTag(const char *what);
  ~SortTag();
};
const char *tagged_sort_name(const char *base);      // base, or an interned "base [tag]"
// stable LSD radix sort of (key,value) pairs on key bits [begin_bit,end_bit); inputs are left intact
// (the library then keeps a temporary copy of both arrays: use sort_pairs_db for the big sorts)
template <class K, class V>
void sort_pairs(pfp_ctx *c, const K *kin, K *kout, const V *vin, V *vout, size_t n, int begin_bit, int end_bit);
inline void sort_pairs_u64_u32(pfp_ctx *c, const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout,
                               size_t n, int bb, int eb) { sort_pairs<uint64_t, uint32_t>(c, kin, kout, vin, vout, n, bb, eb); }
inline void sort_pairs_u32_u32(pfp_ctx *c, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout,
                               size_t n, int bb, int eb) { sort_pairs<uint32_t, uint32_t>(c, kin, kout, vin, vout, n, bb, eb); }
// the same ping-ponging between the two buffer pairs (no third copy): on return the sorted pairs are in
// (k, v) - the DBufs are swapped when the last pass ended in the alternates - and (kalt, valt) hold garbage.
// Both pairs must hold at least n elements.
template <class K, class V>
void sort_pairs_db(pfp_ctx *c, DBuf<K> &k, DBuf<K> &kalt, DBuf<V> &v, DBuf<V> &valt, size_t n, int begin_bit, int end_bit);
// digit width of the keys-only library sort (prims.hip: 9-bit onesweep digits for u64 keys): callers that choose their key width
// round it to whole passes
constexpr int kKeysDigitBits = 9;
// keys only, ping-ponging between k and kalt (swapped when the last pass ended in the alternate).  The bits below
// begin_bit must be a unique index that ascends with the input order (small inputs are sorted on the whole word)
template <class K>
void sort_keys_db(pfp_ctx *c, DBuf<K> &k, DBuf<K> &kalt, size_t n, int begin_bit, int end_bit);
// keys only, in -> out (the library keeps a temporary copy); same rule about the bits below begin_bit
void sort_keys_raw(pfp_ctx *c, const uint64_t *in, uint64_t *out, size_t n, int begin_bit, int end_bit);
// radix.hip (round 4): the hand-written first-round sort - stable on key bits [lo, hi), same buffer contract as sort_pairs_db /
// sort_keys_db (result in k / v, DBufs swapped where the last pass ended in the alternates)
template <class V>
void msd_sort_pairs_db(pfp_ctx *c, DBuf<uint64_t> &k, DBuf<uint64_t> &kalt, DBuf<V> &v, DBuf<V> &valt, size_t n, int lo, int hi);
void msd_sort_keys_db(pfp_ctx *c, DBuf<uint64_t> &k, DBuf<uint64_t> &kalt, size_t n, int lo, int hi);
// stable radix sort inside each segment [begin[k], end[k]) (one already-grouped array, n < 2^32)
template <class V>
void segsort_pairs_u32(pfp_ctx *c, const uint32_t *kin, uint32_t *kout, const V *vin, V *vout, size_t n,
                       size_t nseg, const uint32_t *seg_begin, const uint32_t *seg_end, int begin_bit, int end_bit);
// the same with 64-bit keys and 32-bit values
void segsort_pairs_u64_u32(pfp_ctx *c, const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout, size_t n,
                           size_t nseg, const uint32_t *seg_begin, const uint32_t *seg_end, int begin_bit, int end_bit);
// out[i] = sum_{j<i} in[j]
void exclusive_sum_u32(pfp_ctx *c, const uint32_t *in, uint32_t *out, size_t n);
void exclusive_sum_u32_u64(pfp_ctx *c, const uint32_t *in, uint64_t *out, size_t n);
void exclusive_sum_u64(pfp_ctx *c, const uint64_t *in, uint64_t *out, size_t n);
template <class O> inline void exclusive_sum_u32_to(pfp_ctx *c, const uint32_t *in, O *out, size_t n);
template <> inline void exclusive_sum_u32_to<uint32_t>(pfp_ctx *c, const uint32_t *in, uint32_t *out, size_t n) { exclusive_sum_u32(c, in, out, n); }
template <> inline void exclusive_sum_u32_to<uint64_t>(pfp_ctx *c, const uint32_t *in, uint64_t *out, size_t n) { exclusive_sum_u32_u64(c, in, out, n); }
// out[i] = sum_{j<=i} in[j]
void inclusive_sum_u32(pfp_ctx *c, const uint32_t *in, uint32_t *out, size_t n);
// out[i] = max_{j<=i} in[j]
template <class T> void inclusive_max(pfp_ctx *c, const T *in, T *out, size_t n);
inline void inclusive_max_u32(pfp_ctx *c, const uint32_t *in, uint32_t *out, size_t n) { inclusive_max<uint32_t>(c, in, out, n); }
// out[i] = #{ j <= i : bytes[j] == value }
void inclusive_count_eq_u8(pfp_ctx *c, const uint8_t *bytes, uint8_t value, uint32_t *out, size_t n);
// stream compaction: out = in[i] for flags[i]!=0, returns count through d_count (device u32)
void select_flagged_u32(pfp_ctx *c, const uint32_t *in, const uint8_t *flags, uint32_t *out, uint32_t *d_count, size_t n);
// out = i for flags[i]!=0 (16-byte aligned flags); d_count receives the number selected (u64)
template <class I> void select_index(pfp_ctx *c, const uint8_t *flags, I *out, uint64_t *d_count, size_t n);
// the same for bytes[i] == value
template <class I> void select_byte_index(pfp_ctx *c, const uint8_t *bytes, uint8_t value, I *out, uint64_t *d_count, size_t n);
// number of non-zero flags (syncs the stream)
uint64_t count_flags(pfp_ctx *c, const uint8_t *flags, size_t n);
}  // namespace pfp
