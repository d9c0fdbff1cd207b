// wordview.hpp -- "which word holds dictionary position i, and how far is its terminator?" without per-position arrays.
//
// Rounds 1-2 kept pos_word[i] and slen[i] for every position of the dictionary: 8 bytes per dictionary byte, written
// once (1 GB on the 0.79 GB workload) and - the real cost - REPLICATED on every rank of the multi-GPU chain, which is
// what kept BASELINE configs[3] / configs[4] (|D| = 8 / 30 GB) from fitting.  The word of a position is the number of
// terminators (0x01) before it: one 32-bit count per 64-byte line of the dictionary (|D| / 16 bytes) plus the bytes of
// that one line answer it, and the terminator's position comes from the word table (8 bytes per word).  Streaming
// kernels do not even need the line: they count the terminators of their block with ballots.
// (reference: getlen / binsearch, pfbwt.cpp:449-473, answer the same question by bisection over the word ends)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace pfp {

struct WordView {
  const uint8_t *bytes;        // the dictionary (64-byte aligned, padded)
  const uint32_t *blk_word;    // [NP / 64 + 1] word holding position 64 b = terminators in [0, 64 b)
  const uint64_t *wend;        // [d + 1] position of word j's terminator (wend[d] = NP - 1: the final 0x00 is its own word)
  uint32_t d;
  uint64_t NP;
};

// number of bytes of x equal to 0x01 (exact: no borrow between bytes)
__device__ __forceinline__ uint32_t count_term_bytes(uint32_t x) {
  const uint32_t t = x ^ 0x01010101u;
  return (uint32_t)__popc(~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu));
}
__device__ __forceinline__ uint32_t count_term_bytes16(uint4 x, uint32_t keep /* leading bytes that count, 0..16 */) {
  uint32_t r[4] = {x.x, x.y, x.z, x.w}, n = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int k = (int)keep - 4 * q;
    if (k <= 0) break;
    n += count_term_bytes(k >= 4 ? r[q] : (r[q] & ((1u << (8 * k)) - 1u)));      // (masked bytes become 0x00: not a terminator)
  }
  return n;
}
// word containing position i (d for the final 0x00)
__device__ __forceinline__ uint32_t word_of(const WordView &v, uint64_t i) {
  const uint64_t b = i & ~63ull;
  uint32_t wd = v.blk_word[i >> 6];
  const uint32_t nb = (uint32_t)(i - b);
  const uint4 *line = reinterpret_cast<const uint4 *>(v.bytes + b);
#pragma unroll
  for (int q = 0; q < 4; q++)
    if (nb > 16u * q) wd += count_term_bytes16(line[q], nb - 16u * q < 16u ? nb - 16u * q : 16u);
  return wd;
}
// distance from i to the terminator of its word (the terminator not counted; 0 for the final 0x00)
__device__ __forceinline__ uint64_t slen_of(const WordView &v, uint64_t i) { return v.wend[word_of(v, i)] - i; }
__device__ __forceinline__ uint64_t slen_of(const WordView &v, uint64_t i, uint32_t word) { return v.wend[word] - i; }

}  // namespace pfp
