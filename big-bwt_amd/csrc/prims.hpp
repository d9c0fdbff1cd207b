// prims.hpp -- device-wide sort / scan / select building blocks (thin wrappers over rocPRIM,
// AMD's own header-only primitives library; all hot-path kernels are hand-written elsewhere).
#pragma once
#include "common.hpp"

namespace pfp {
// stable LSD radix sort of (key,value) pairs on key bits [begin_bit,end_bit)
void sort_pairs_u64_u32(pfp_ctx *c, const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout,
                        size_t n, int begin_bit, int end_bit);
void sort_pairs_u32_u32(pfp_ctx *c, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout,
                        size_t n, int begin_bit, int end_bit);
// stable radix sort inside each segment [begin[k], end[k]) (one already-grouped array)
void segsort_pairs_u32_u32(pfp_ctx *c, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, size_t n,
                           size_t nseg, const uint32_t *seg_begin, const uint32_t *seg_end, int begin_bit, int end_bit);
// out[i] = sum_{j<i} in[j]
void exclusive_sum_u32(pfp_ctx *c, const uint32_t *in, uint32_t *out, size_t n);
void exclusive_sum_u32_u64(pfp_ctx *c, const uint32_t *in, uint64_t *out, size_t n);
void exclusive_sum_u64(pfp_ctx *c, const uint64_t *in, uint64_t *out, size_t n);
// out[i] = sum_{j<=i} in[j]
void inclusive_sum_u32(pfp_ctx *c, const uint32_t *in, uint32_t *out, size_t n);
// out[i] = max_{j<=i} in[j]
void inclusive_max_u32(pfp_ctx *c, const uint32_t *in, uint32_t *out, size_t n);
// out[i] = #{ j <= i : bytes[j] == value }
void inclusive_count_eq_u8(pfp_ctx *c, const uint8_t *bytes, uint8_t value, uint32_t *out, size_t n);
// stream compaction: out = in[i] for flags[i]!=0, returns count through d_count (device u32)
void select_flagged_u32(pfp_ctx *c, const uint32_t *in, const uint8_t *flags, uint32_t *out, uint32_t *d_count, size_t n);
// out = i for flags[i]!=0
void select_index_u32(pfp_ctx *c, const uint8_t *flags, uint32_t *out, uint32_t *d_count, size_t n);
void select_byte_index_u32(pfp_ctx *c, const uint8_t *bytes, uint8_t value, uint32_t *out, uint32_t *d_count, size_t n);
}  // namespace pfp
