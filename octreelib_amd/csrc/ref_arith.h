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

// ---- child digits of six levels at once ----------------------------------------------------------------------
// Reference per level j below a cube (corner c_j, edge e_j), octree/octree.py:73-75,94-97,181-191:
//     idx = floor(fl(p - c_j) / (e_j / 2)) in {0, 1} per axis,  c_{j+1} = c_j + idx * (e_j / 2),  e_{j+1} = e_j / 2
// i.e. one ROUNDED subtraction from the point per level and axis.  For a point with p >= c_0 >= 0 (per axis) whose
// cube has an integer-valued corner and edge, every one of those subtractions is EXACT:
//   * c_j = c_0 + k e_0 2^-j is a multiple of 2^-6 for j <= 6, and so is a multiple of ulp(p) = 2^(floor(log2 p) - 52)
//     as long as p < 2^46; p is a multiple of ulp(p) by definition;
//   * 0 <= p - c_j <= p, so the difference - a multiple of ulp(p) no larger than p - has at most 53 significant
//     bits at that ulp: representable, fl(p - c_j) = p - c_j.
// (A negative coordinate breaks the second point: -0.3 - (-1) = 0.7 needs a coarser ulp than -0.3 has and is
//  rounded; such points keep the level-by-level evaluation.)  With exact differences the digits are the binary
// expansion of t = p - c_0 against e_0: t_{j+1} = t_j - bit_j e_0 2^-(j+1), again exact, so
//     bit_j = floor(t / e_0 * 2^(j+1)) mod 2
// and for a power-of-two edge the six bits of an axis are (uint)(t * 64 / e_0) - one multiply and one conversion
// instead of six dependent rounds of subtract / compare / select per axis.  Points outside the cube (t < 0 or
// t >= e_0) are "bad" at level 0 with an empty path, exactly as the level-by-level form reports them.
// tests/test_gpu_primitives.py compares both forms bit for bit (random, face-adjacent and ulp-adjacent points).
__host__ __device__ __forceinline__ uint32_t spread6_by3(uint32_t v) {  // bit i of a 6-bit value -> bit 3 i
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

// integer valued and in [0, 2^45)?  (exponent test on the bit pattern; false for NaN / inf / negative / -0.0)
__host__ __device__ __forceinline__ bool nonneg_integer_below_2p45(double v) {
  return v >= 0.0 && v < 0x1p45 && v == (double)(long long)v;
}

// Precondition of digits18_exact for one axis: p finite, 0 <= p < 2^45 (the cube's corner and edge are checked once
// per kernel with nonneg_integer_below_2p45).
__host__ __device__ __forceinline__ bool coord_takes_exact_digits(double p) { return p >= 0.0 && p < 0x1p45; }

// The 18 bits of the first six child digits (level 0 in bits 17..15, x the most significant bit of a digit) of a
// point that meets the precondition; *bad: outside the cube.  edge_pow2: the edge is a power of two and
// inv64 = 64 / edge (exact); otherwise the six levels are walked on the exact difference.
__host__ __device__ __forceinline__ uint32_t digits18_exact(double px, double py, double pz, double cx, double cy,
                                                           double cz, double e, bool edge_pow2, double inv64,
                                                           bool* bad) {
  double tx = px - cx, ty = py - cy, tz = pz - cz;  // exact
  if (!((tx >= 0.0) && (tx < e) && (ty >= 0.0) && (ty < e) && (tz >= 0.0) && (tz < e))) {
    *bad = true;
    return 0u;
  }
  if (edge_pow2) {
    const uint32_t mx = (uint32_t)(tx * inv64), my = (uint32_t)(ty * inv64), mz = (uint32_t)(tz * inv64);
    return (spread6_by3(mx) << 2) | (spread6_by3(my) << 1) | spread6_by3(mz);
  }
  uint32_t path = 0;
  double h = e * 0.5;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const bool bx = tx >= h, by = ty >= h, bz = tz >= h;
    path = (path << 3) | (bx ? 4u : 0u) | (by ? 2u : 0u) | (bz ? 1u : 0u);
    tx -= bx ? h : 0.0;
    ty -= by ? h : 0.0;
    tz -= bz ? h : 0.0;
    h *= 0.5;
  }
  return path;
}
