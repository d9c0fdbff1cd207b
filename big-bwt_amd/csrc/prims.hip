// prims.hip -- rocPRIM instantiations (isolated in one translation unit: they compile slowly).
#include "prims.hpp"
#include "devutil.hpp"
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
// Indices of the non-zero flags, in order.  rocprim::select over (counting iterator, u8 flags) ran at
// ~95 G flags/s (2.8 ms for the 260 M hard-group flags of the 253 MB workload); the flags are bytes, so a
// workgroup can count and place 4096 of them from 16-byte loads: two streaming passes over n bytes.
constexpr int kSelTile = 4096;
// eq < 0: flag = byte != 0;  eq >= 0: flag = byte == eq
__device__ __forceinline__ void load_sel16(const uint8_t *__restrict__ flags, uint64_t base, uint64_t n, int eq, uint32_t w[4]) {
  if (eq < 0) { load_flags16(flags, base, n, w); return; }
  const uint32_t V = 0x01010101u * (uint32_t)eq;
  if (base + 16 <= n) {
    const uint4 v = *reinterpret_cast<const uint4 *>(flags + base);
    w[0] = nonzero_bytes(v.x ^ V) ^ 0x01010101u; w[1] = nonzero_bytes(v.y ^ V) ^ 0x01010101u;
    w[2] = nonzero_bytes(v.z ^ V) ^ 0x01010101u; w[3] = nonzero_bytes(v.w ^ V) ^ 0x01010101u;
  } else {
    w[0] = w[1] = w[2] = w[3] = 0;
    for (int k = 0; k < 16; k++) if (base + k < n && flags[base + k] == (uint8_t)eq) w[k >> 2] |= 1u << (8 * (k & 3));
  }
}
__global__ __launch_bounds__(256) void flag_count_kernel(const uint8_t *__restrict__ flags, uint64_t n, int eq,
                                                         uint32_t *__restrict__ blocksum) {
  __shared__ uint32_t ws[4];
  const uint64_t base = (uint64_t)blockIdx.x * kSelTile + (uint64_t)threadIdx.x * 16;
  uint32_t w[4];
  load_sel16(flags, base, n, eq, w);
  uint32_t cnt = __popc(w[0]) + __popc(w[1]) + __popc(w[2]) + __popc(w[3]);
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) blocksum[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ __launch_bounds__(256) void flag_place_kernel(const uint8_t *__restrict__ flags, uint64_t n, int eq,
                                                         const uint32_t *__restrict__ blockoff, uint32_t *__restrict__ out) {
  __shared__ uint32_t ws[4];
  const uint64_t base = (uint64_t)blockIdx.x * kSelTile + (uint64_t)threadIdx.x * 16;
  uint32_t w[4];
  load_sel16(flags, base, n, eq, w);
  const uint32_t cnt = __popc(w[0]) + __popc(w[1]) + __popc(w[2]) + __popc(w[3]);
  uint32_t inc = cnt;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
  if (lane == 63) ws[wv] = inc;
  __syncthreads();
  uint32_t pos = blockoff[blockIdx.x] + inc - cnt;
  for (int q = 0; q < wv; q++) pos += ws[q];
  if (!cnt) return;
#pragma unroll
  for (int k = 0; k < 16; k++)
    if ((w[k >> 2] >> (8 * (k & 3))) & 1u) out[pos++] = (uint32_t)(base + k);
}
static void select_bytes(pfp_ctx *c, const uint8_t *flags, int eq, uint32_t *out, uint32_t *d_count, size_t n);
void select_index_u32(pfp_ctx *c, const uint8_t *flags, uint32_t *out, uint32_t *d_count, size_t n) {
  if (!n) { PFP_HIP(hipMemsetAsync(d_count, 0, 4, c->stream)); return; }
  if ((reinterpret_cast<uintptr_t>(flags) & 15) != 0) {      // unaligned view: library path
    rocprim::counting_iterator<uint32_t> it(0);
    PRIM2(rocprim::select(tmp, tb, it, flags, out, d_count, n, c->stream));
    return;
  }
  select_bytes(c, flags, -1, out, d_count, n);
}
// indices of the bytes equal to `value` (16-byte aligned buffer)
void select_byte_index_u32(pfp_ctx *c, const uint8_t *bytes, uint8_t value, uint32_t *out, uint32_t *d_count, size_t n) {
  if (!n) { PFP_HIP(hipMemsetAsync(d_count, 0, 4, c->stream)); return; }
  PFP_REQUIRE((reinterpret_cast<uintptr_t>(bytes) & 15) == 0, PFP_EINVAL, "select_byte_index_u32: unaligned buffer");
  select_bytes(c, bytes, (int)value, out, d_count, n);
}
static void select_bytes(pfp_ctx *c, const uint8_t *flags, int eq, uint32_t *out, uint32_t *d_count, size_t n) {
  const size_t nblk = (n + kSelTile - 1) / kSelTile;
  DBuf<uint32_t> bsum(c, nblk + 1), boff(c, nblk + 1);
  PFP_HIP(hipMemsetAsync(bsum.p + nblk, 0, 4, c->stream));
  KScope ks(c, "pfp::select_flags_kernel", n * 2);
  hipLaunchKernelGGL(flag_count_kernel, dim3((unsigned)nblk), dim3(256), 0, c->stream, flags, (uint64_t)n, eq, bsum.p);
  exclusive_sum_u32(c, bsum.p, boff.p, nblk + 1);
  hipLaunchKernelGGL(flag_place_kernel, dim3((unsigned)nblk), dim3(256), 0, c->stream, flags, (uint64_t)n, eq, boff.p, out);
  PFP_HIP(hipGetLastError());
  PFP_HIP(hipMemcpyAsync(d_count, boff.p + nblk, 4, hipMemcpyDeviceToDevice, c->stream));
}

}  // namespace pfp
