// Decoupled look-back over a chain of workgroups ("tiles"), wave64: the piece of scan.hip's single-pass scan that the
// fused table kernels share.  A tile's status is ONE 64-bit word {epoch : 30, flag : 2, value : 32} written and read
// with single 8-byte relaxed atomics at agent scope; the epoch makes words of earlier chains read as "not yet
// published" (octl_scan_status_acquire hands out the array and a fresh epoch per launch).  The spin cannot deadlock
// as long as tiles are taken in blockIdx order: the lowest unfinished tile is always resident and only waits for
// finished ones (see scan.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr uint32_t ST_AGGREGATE = 1, ST_PREFIX = 2;

__device__ __forceinline__ uint64_t st_pack(uint32_t epoch, uint32_t flag, uint32_t value) {
  return ((uint64_t)((epoch << 2) | flag) << 32) | value;
}

// Publishes this tile's aggregate `total` and returns the sum of the aggregates of all tiles in front of it.
// Every thread of the workgroup calls it (it contains a barrier); `s_excl` is one LDS word.
__device__ __forceinline__ uint32_t lookback_exclusive(uint64_t* __restrict__ status, uint32_t epoch, uint32_t tile,
                                                       uint32_t total, uint32_t* s_excl) {
  if (threadIdx.x == 0)
    __hip_atomic_store(&status[tile], st_pack(epoch, tile == 0 ? ST_PREFIX : ST_AGGREGATE, total),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint32_t excl = 0;
  if (tile > 0) {
    if (threadIdx.x < 64) {
      const int lane = threadIdx.x;
      int64_t look = (int64_t)tile - 1;  // highest predecessor not yet accounted for
      for (;;) {
        const int64_t t = look - lane;   // lanes past tile 0 see "prefix 0"
        uint32_t flag = ST_PREFIX, value = 0;
        if (t >= 0) {
          for (;;) {
            const uint64_t w = __hip_atomic_load(&status[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t hi = (uint32_t)(w >> 32);
            flag = hi & 3u;
            value = (uint32_t)w;
            if ((hi >> 2) == epoch && flag != 0) break;
            __builtin_amdgcn_s_sleep(1);
          }
        }
        const unsigned long long has_prefix = __ballot(flag == ST_PREFIX);
        const int first = __ffsll((long long)has_prefix) - 1;  // >= 0: lanes past tile 0 report a prefix
        uint32_t c = (has_prefix == 0 || lane <= first) ? value : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
        excl += c;
        if (has_prefix) break;
        look -= 64;
      }
      if (lane == 0) {
        *s_excl = excl;
        __hip_atomic_store(&status[tile], st_pack(epoch, ST_PREFIX, excl + total), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
    excl = *s_excl;
  }
  return excl;
}
