// prims.hip -- rocPRIM instantiations (isolated in one translation unit: they compile slowly).
#include "prims.hpp"
#include <rocprim/rocprim.hpp>

namespace pfp {

#define PRIM2(call_with_temp)                                   \
  do {                                                          \
    size_t tb = 0; void *tmp = nullptr;                         \
    PFP_HIP(call_with_temp);                                    \
    DBuf<uint8_t> tbuf(c, tb ? tb : 1);                         \
    tmp = tbuf.p;                                               \
    PFP_HIP(call_with_temp);                                    \
  } while (0)

void sort_pairs_u64_u32(pfp_ctx *c, const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout,
                        size_t n, int bb, int eb) {
  if (!n) return;
  const uint64_t passes = (uint64_t)(eb - bb + 7) / 8;
  KScope ks(c, "rocprim::radix_sort_pairs<u64,u32>", n * 8 + passes * n * 24);
  PRIM2(rocprim::radix_sort_pairs(tmp, tb, kin, kout, vin, vout, n, (unsigned)bb, (unsigned)eb, c->stream));
}
void sort_pairs_u32_u32(pfp_ctx *c, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout,
                        size_t n, int bb, int eb) {
  if (!n) return;
  const uint64_t passes = (uint64_t)(eb - bb + 7) / 8;
  KScope ks(c, "rocprim::radix_sort_pairs<u32,u32>", n * 4 + passes * n * 16);
  PRIM2(rocprim::radix_sort_pairs(tmp, tb, kin, kout, vin, vout, n, (unsigned)bb, (unsigned)eb, c->stream));
}
void segsort_pairs_u32_u32(pfp_ctx *c, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, size_t n,
                           size_t nseg, const uint32_t *seg_begin, const uint32_t *seg_end, int bb, int eb) {
  if (!n || !nseg) return;
  KScope ks(c, "rocprim::segmented_radix_sort_pairs<u32,u32>", n * 16 + nseg * 8);
  PRIM2(rocprim::segmented_radix_sort_pairs(tmp, tb, kin, kout, vin, vout, (unsigned)n, (unsigned)nseg, seg_begin, seg_end,
                                            (unsigned)bb, (unsigned)eb, c->stream));
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
void inclusive_max_u32(pfp_ctx *c, const uint32_t *in, uint32_t *out, size_t n) {
  if (!n) return;
  KScope ks(c, "rocprim::scan<u32>", n * 8);
  PRIM2(rocprim::inclusive_scan(tmp, tb, in, out, n, rocprim::maximum<uint32_t>(), c->stream));
}
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
void select_index_u32(pfp_ctx *c, const uint8_t *flags, uint32_t *out, uint32_t *d_count, size_t n) {
  if (!n) { PFP_HIP(hipMemsetAsync(d_count, 0, 4, c->stream)); return; }
  rocprim::counting_iterator<uint32_t> it(0);
  PRIM2(rocprim::select(tmp, tb, it, flags, out, d_count, n, c->stream));
}

}  // namespace pfp
