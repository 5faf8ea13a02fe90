// Device restatements of the reference's coordinate arithmetic shared by build.hip / route.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// np.floor_divide(a, L) for L > 0 (grid/grid.py:72-76): NumPy evaluates the floor of the TRUE
// quotient (fmod-based npy_divmod).  Here: floor of the rounded quotient, then an exact
// correction from the sign of the fused remainder a - q*L.
__host__ __device__ __forceinline__ double floor_div_exact(double a, double L) {
  double q = floor(a / L);
  const double r = fma(-q, L, a);
  if (r < 0.0) {
    q -= 1.0;
  } else if (fma(-(q + 1.0), L, a) >= 0.0) {
    q += 1.0;
  }
  return q;
}

// Owner rank of a top-level voxel: splitmix64 finaliser over the three integer voxel indices.
// A hash, not x-slabs: scene clouds are spatially skewed.
__host__ __device__ __forceinline__ int32_t voxel_owner_hash(int64_t qx, int64_t qy, int64_t qz,
                                                             int32_t n_ranks) {
  uint64_t h = (uint64_t)qx * 0x9E3779B97F4A7C15ull;
  h ^= (uint64_t)qy * 0xC2B2AE3D27D4EB4Full + 0x165667B19E3779F9ull + (h << 6) + (h >> 2);
  h ^= (uint64_t)qz * 0xD6E8FEB86659FD93ull + 0x27D4EB2F165667C5ull + (h << 6) + (h >> 2);
  h ^= h >> 30;
  h *= 0xBF58476D1CE4E5B9ull;
  h ^= h >> 27;
  h *= 0x94D049BB133111EBull;
  h ^= h >> 31;
  return (int32_t)(h % (uint64_t)(n_ranks > 0 ? n_ranks : 1));
}
