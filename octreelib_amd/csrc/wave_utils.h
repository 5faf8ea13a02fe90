// wave64 helpers shared by the sort / partition kernels (gfx950: a wavefront is 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Lanes of the wave holding the same BITS-bit digit as this lane (among the valid lanes).
// BITS ballots; every lane of the wave must call it (no divergence around the call).
template <int BITS>
__device__ __forceinline__ uint64_t wave_match(uint32_t digit, bool valid) {
  uint64_t peers = __ballot(valid);
#pragma unroll
  for (int b = 0; b < BITS; ++b) {
    const bool bit = (digit >> b) & 1u;
    const uint64_t m = __ballot(bit);
    peers &= bit ? m : ~m;
  }
  return peers;
}

// inclusive prefix sum over the wavefront
__device__ __forceinline__ uint32_t wave_inclusive_add(uint32_t v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(v, off);
    if (lane >= off) v += t;
  }
  return v;
}

__device__ __forceinline__ uint64_t lanemask_lt() {
  const unsigned lane = threadIdx.x & 63u;
  return (lane == 0) ? 0ull : (~0ull >> (64u - lane));
}

// Stable rank of each item inside one wave-sized round plus the running per-digit counter of
// this wave kept in LDS (`cnt`, one counter per digit, touched only by this wave).  Returns the
// number of items with the same digit that precede this one in the wave's stream so far.
template <int BITS>
__device__ __forceinline__ uint32_t wave_stable_rank(uint32_t digit, bool valid, uint32_t* cnt) {
  const uint64_t peers = wave_match<BITS>(digit, valid);
  const uint32_t rank_in_round = __popcll(peers & lanemask_lt());
  const int leader = __ffsll((unsigned long long)peers) - 1;
  uint32_t old = 0;
  if (valid && (int)(threadIdx.x & 63u) == leader)
    old = atomicAdd(&cnt[digit], (uint32_t)__popcll(peers));
  old = __shfl(old, leader < 0 ? 0 : leader);
  return old + rank_in_round;
}
