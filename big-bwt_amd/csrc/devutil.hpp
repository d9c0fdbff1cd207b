// devutil.hpp -- small device-side helpers (unaligned vector access, sub-wave reductions).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace pfp {

// gfx950 runs with unaligned-access mode on: a dwordx4 global load/store may start at any byte.
struct __attribute__((packed, aligned(1))) U128u { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(1))) U64u { uint64_t v; };

__device__ __forceinline__ uint4 ld16u(const uint8_t *p) {
  U128u v = *reinterpret_cast<const U128u *>(p);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st16u(uint8_t *p, uint4 v) {
  U128u u{v.x, v.y, v.z, v.w};
  *reinterpret_cast<U128u *>(p) = u;
}
__device__ __forceinline__ uint64_t ld8u(const uint8_t *p) { return reinterpret_cast<const U64u *>(p)->v; }

// zero the bytes at index >= keep (0..16) of a little-endian 16-byte register block
__device__ __forceinline__ uint4 keep_bytes16(uint4 v, int keep) {
  uint32_t r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int k = keep - 4 * i;
    r[i] = k >= 4 ? r[i] : (k <= 0 ? 0u : (r[i] & ((1u << (8 * k)) - 1u)));
  }
  return make_uint4(r[0], r[1], r[2], r[3]);
}

__device__ __forceinline__ uint64_t fmix64(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
  return k;
}

// sum / or over the 2^LG lanes of an aligned lane group (LG = 2 or 3)
template <int LG>
__device__ __forceinline__ uint64_t group_sum(uint64_t v) {
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64);
  if (LG > 2) v += __shfl_xor(v, 4, 64);
  return v;
}
template <int LG>
__device__ __forceinline__ uint32_t group_or(uint32_t v) {
  v |= __shfl_xor(v, 1, 64); v |= __shfl_xor(v, 2, 64);
  if (LG > 2) v |= __shfl_xor(v, 4, 64);
  return v;
}

// inclusive prefix sum over the 64 lanes of a wave with DPP moves only (row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes, then
// the two row broadcasts gfx9 keeps for exactly this): six v_add with a DPP operand instead of six ds_bpermute round trips
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);      // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);      // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);      // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);      // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
  return v;
}

// 16 flag bytes -> four words with bit 0 of every byte set where the flag is non-zero
__device__ __forceinline__ uint32_t nonzero_bytes(uint32_t x) {
  x |= x >> 4; x |= x >> 2; x |= x >> 1;
  return x & 0x01010101u;
}
__device__ __forceinline__ void load_flags16(const uint8_t *__restrict__ flags, uint64_t base, uint64_t n, uint32_t w[4]) {
  if (base + 16 <= n) {
    const uint4 v = *reinterpret_cast<const uint4 *>(flags + base);
    w[0] = nonzero_bytes(v.x); w[1] = nonzero_bytes(v.y); w[2] = nonzero_bytes(v.z); w[3] = nonzero_bytes(v.w);
  } else {
    w[0] = w[1] = w[2] = w[3] = 0;
    for (int k = 0; k < 16; k++) if (base + k < n && flags[base + k]) w[k >> 2] |= 1u << (8 * (k & 3));
  }
}
}  // namespace pfp
