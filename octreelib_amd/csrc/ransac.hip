// Per-leaf RANSAC plane fit on the device.
//
// Replaces (paths relative to /root/reference):
//   kernel                 ransac/cuda_ransac.py:85-155   one CUDA block per leaf, one thread per
//                                                         hypothesis
//   get_plane_from_points  ransac/util.py:27-84
//   measure_distance       ransac/util.py:12-24
//   CudaRansac.evaluate    ransac/cuda_ransac.py:43-81
//
// Arithmetic contract (parity mode, the only mode): IEEE f64 in the reference's operation
// order, NO fused multiply-add (this file is compiled with -ffp-contract=off; the reference's CI
// path is numba's CUDA simulator = NumPy scalar arithmetic), the plane rounded to f32 before
// scoring (cuda_ransac.py:110-113), strict comparisons.  The only non-determinism of the
// reference - which of several hypotheses tied at the maximal inlier count wins the CAS race
// (cuda_ransac.py:140-145) - is resolved to the LOWEST hypothesis index.
//
// Mapping on CDNA4: one workgroup of W waves per (leaf, pose) block, hypotheses on the lanes
// (HPL per lane), the block's points broadcast to all lanes through wave-uniform LDS reads.  The
// kernel is VALU bound.  What the reference asks for is 1024 plane fits per block (~260
// f64-dominated instructions each) and H x n distance tests; what runs (round 6):
//   * the first THREADS hypotheses exactly - the fit in the reference's f64 order, the distance
//     tests on an f32 screen wherever a rigorous error bound allows, recounted in the reference's
//     f64 sequence otherwise (exact counts either way);
//   * for every later hypothesis an APPROXIMATE f32 plane with a rigorous bound on its distances
//     and the count inside the widened threshold: an upper bound of its exact count.  Only the
//     hypotheses whose bound exceeds the best exact count so far - a handful per block - are
//     fitted and counted exactly (see "the prescreen of the hypotheses" below).
// Results are the reference's, bit for bit.
// MFMA is not used: the f64 evaluation order (and the f32-rounded plane) must be reproduced
// exactly, and the product is 4 deep; an f32 MFMA screen was measured slower.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "forest.h"
#include "lookback.h"
#include "wave_utils.h"

namespace {

// c / k, correctly rounded, for an integer 1 <= k <= 16 (see plane_from_samples).
// zh = RN(1/k), zl = RN(1/k - zh); denormal / overflow ranges fall back to the true division.
__device__ __forceinline__ double div_by_small_int(double c, int k) {
  const double kd = (double)k;
  const double zh = 1.0 / kd;                    // compile-time constant when k is
  const double zl = fma(-kd, zh, 1.0) / kd;      // (1 - k*zh) is exact; zl to 2^-53 relative
  const double a = fabs(c);
  if (!((int(a > 1e-290) | int(c == 0.0)) & int(a < 1e290))) return c / kd;  // also NaN / inf / tiny
  return copysign(fma(c, zh, c * zl), c);         // (the sign: zl may be negative, c may be -0)
}

// ---- wave-uniform range certificate ----------------------------------------------------------------------------
// The three shortcuts below (division of the centroid by k, in-range square root, shared-reciprocal
// division by the norm) are exact only while no operand or intermediate leaves the normal range, and each
// used to test that per lane and per fit: ~20 f64 compares out of ~250 instructions.  Almost all of that
// follows from ONE property of the block, tested once per point when the block is staged:
//
//   every coordinate of every point of the block (and of its spill point) is +0.0 or has 2^-30 <= |v| < 2^31
//
// With it (the "grid" argument: a value that is a multiple of g stays a multiple of g under +, - and
// round-to-nearest, so a non-zero result is at least g in magnitude):
//   * coordinates are multiples of 2^-82; a non-zero centroid sum is >= 2^-82 and < 2^35: in the range of
//     div3_by_small_int, and never -0 (x + y is -0 only for x = y = -0), so its copysign is the identity;
//     `0.0 + x` is x (util.py:35-40 starts the sums at 0.0);
//   * the centroid (|c| >= 2^-86 or 0) is a multiple of 2^-138, so are the residuals; products and the six
//     covariance sums are multiples of 2^-276 below 2^68; cofactors are multiples of 2^-552 below 2^137;
//     s = |cofactor row|^2 < 2^276 is never NaN / inf;
//   * per fit ONE test remains: s >= 2^-400 (an integer compare on the high word).  Then norm is in
//     [2^-200, 2^138] - inside sqrt_rn_guarded's and div3_by_norm's ranges and non-zero (util.py:77-78 is the
//     slow path's business) - and every numerator is 0 or >= 2^-552: quotients >= 2^-690 stay normal, and
//     the residual fma(-norm, q0, a), a multiple of ulp(norm) ulp(q0) >= 2^-252 2^-743, is exact.
// Blocks that fail the property (denormal-scale or astronomically large coordinates, -0.0, NaN, inf) keep the
// per-lane guards.  tests/test_gpu_primitives.py checks the unguarded forms bit-for-bit against IEEE division
// and square root over this wider domain.
__device__ __forceinline__ bool coord_in_fast_range(double v) {
  const uint32_t hi = (uint32_t)__double2hiint(v), lo = (uint32_t)__double2loint(v);
  const uint32_t e = (hi >> 20) & 0x7FFu;
  return (e - (1023u - 30u)) <= 60u || (hi | lo) == 0u;  // (-0.0: hi = 0x80000000 - not accepted)
}

// the three centroid coordinates at once: one flat predicate, one (never taken) slow branch
__device__ __forceinline__ void div3_by_small_int(double& cx, double& cy, double& cz, int k, bool fast = false) {
  const double kd = (double)k;
  const double zh = 1.0 / kd;
  const double zl = fma(-kd, zh, 1.0) / kd;
  if (fast) {  // wave-uniform: sums are 0 or in [2^-82, 2^35), never -0
    cx = fma(cx, zh, cx * zl);
    cy = fma(cy, zh, cy * zl);
    cz = fma(cz, zh, cz * zl);
    return;
  }
  const double ax = fabs(cx), ay = fabs(cy), az = fabs(cz);
  // (ints on purpose: one flat predicate, no short-circuit branches)
  const int ok = (int(ax > 1e-290) | int(cx == 0.0)) & int(ax < 1e290) &
                 (int(ay > 1e-290) | int(cy == 0.0)) & int(ay < 1e290) &
                 (int(az > 1e-290) | int(cz == 0.0)) & int(az < 1e290);
  if (!ok) {
    cx /= kd;
    cy /= kd;
    cz /= kd;
    return;
  }
  cx = copysign(fma(cx, zh, cx * zl), cx);  // (the sign: zl may be negative, c may be -0)
  cy = copysign(fma(cy, zh, cy * zl), cy);
  cz = copysign(fma(cz, zh, cz * zl), cz);
}

// sqrt(s), correctly rounded (util.py:76, `** 0.5`).  In range this IS the compiler's IEEE f64 square
// root - v_rsq_f64, one coupled Goldschmidt step, two residual corrections - without the operand
// rescaling it applies below 2^-767 and the patch for 0 / inf; everything else takes the full one.
__device__ __forceinline__ double sqrt_rn_inrange(double s) {
  const double y = __builtin_amdgcn_rsq(s);
  double g = s * y;
  double h = y * 0.5;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  double d = fma(-g, g, s);
  g = fma(d, h, g);
  d = fma(-g, g, s);
  return fma(d, h, g);
}
__device__ __forceinline__ double sqrt_rn_guarded(double s) {
  if (!(int(s >= 0x1p-700) & int(s < 0x1p1000))) return __dsqrt_rn(s);  // also NaN, 0, inf, negative
  return sqrt_rn_inrange(s);
}

// (ax, ay, az) / norm, correctly rounded (util.py:80-82), norm > 0.  In range this IS the
// compiler's IEEE f64 division - v_rcp_f64, two Newton steps on the reciprocal, q0 = a*r,
// q = fma(fma(-b, q0, a), r, q0) - with the reciprocal computed once for the three numerators and
// without v_div_scale / v_div_fmas / v_div_fixup, which only act outside the guarded exponent range
// (they rescale operands whose quotient or intermediates could leave the normal range, and patch
// zeros / infinities / NaNs).  The sign is taken from the numerator: a (-0) numerator would
// otherwise come out as +0 from the final fma.  Everything else takes the true division.
// Domain of the shortcut: |a_i| <= 2^60 norm (in the plane fit |a_i| <= norm (1 + 2^-52) always).
__device__ __forceinline__ void div3_by_norm_inrange(double& ax, double& ay, double& az, double norm) {
  double r = __builtin_amdgcn_rcp(norm);
  double e = fma(-norm, r, 1.0);
  r = fma(r, e, r);
  e = fma(-norm, r, 1.0);
  r = fma(r, e, r);
  const double qx = ax * r, qy = ay * r, qz = az * r;
  ax = copysign(fma(fma(-norm, qx, ax), r, qx), ax);
  ay = copysign(fma(fma(-norm, qy, ay), r, qy), ay);
  az = copysign(fma(fma(-norm, qz, az), r, qz), az);
}
__device__ __forceinline__ void div3_by_norm(double& ax, double& ay, double& az, double norm) {
  const double lo = 0x1p-400, hi = 0x1p500;
  // (ints and bitwise on purpose: one flat predicate instead of a chain of short-circuit
  //  branches; all comparisons are false for NaN)
  const int ok = int(norm >= lo) & int(norm < hi) & (int(fabs(ax) >= lo) | int(ax == 0.0)) &
                 (int(fabs(ay) >= lo) | int(ay == 0.0)) & (int(fabs(az) >= lo) | int(az == 0.0));
  if (!ok) {
    ax /= norm;
    ay /= norm;
    az /= norm;
    return;
  }
  div3_by_norm_inrange(ax, ay, az, norm);
}

// util.py:59-84: the plane of the centroid and the six covariance sums.  `fast`: the block holds the range
// certificate above (wave-uniform).
__device__ __forceinline__ void plane_from_moments(double cx, double cy, double cz, double xx, double xy,
                                                   double xz, double yy, double yz, double zz,
                                                   float (&plane)[4], bool fast = false) {
  const double det_x = yy * zz - yz * yz;  // util.py:59-61
  const double det_y = xx * zz - xz * xz;
  const double det_z = xx * yy - xy * xy;
  // util.py:63-74: the three cofactors that the branches share, then selects instead of a three-way
  // branch (lanes of a wave take all three, so every branch body was executed anyway):
  //   x: (det_x, A, B)   y: (A, det_y, C)   z: (B, C, det_z)
  const double cA = xz * yz - xy * zz;
  const double cB = xy * yz - xz * yy;
  const double cC = xy * xz - yz * xx;
  const bool is_x = det_x > det_y && det_x > det_z;
  const bool is_y = !is_x && det_y > det_z;
  double ax = is_x ? det_x : (is_y ? cA : cB);
  double ay = is_x ? cA : (is_y ? det_y : cC);
  double az = is_x ? cB : (is_y ? cC : det_z);
  const double s = ax * ax + ay * ay + az * az;  // util.py:76
  // one integer compare on the high word: s >= 2^-400 (s is finite and non-negative here)
  if (fast && (uint32_t)__double2hiint(s) >= ((1023u - 400u) << 20)) {
    const double norm = sqrt_rn_inrange(s);
    div3_by_norm_inrange(ax, ay, az, norm);
  } else {
    const double norm = sqrt_rn_guarded(s);
    if (norm == 0.0) {  // util.py:77-78
      plane[0] = plane[1] = plane[2] = plane[3] = 0.0f;
      return;
    }
    div3_by_norm(ax, ay, az, norm);
  }
  const double d = -(ax * cx + ay * cy + az * cz);  // util.py:83
  plane[0] = (float)ax;
  plane[1] = (float)ay;
  plane[2] = (float)az;
  plane[3] = (float)d;
}

// util.py:27-84 on k sampled points; returns the plane already rounded to f32
// (cuda_ransac.py:110-113).  KT > 0: compile-time k (arrays stay in registers).
template <int KT, int KMAX>
__device__ __forceinline__ void plane_from_samples(const double (&sx)[KMAX],
                                                   const double (&sy)[KMAX],
                                                   const double (&sz)[KMAX], int k_rt,
                                                   float (&plane)[4], bool fast = false) {
  const int k = KT > 0 ? KT : k_rt;
  double cx = 0.0, cy = 0.0, cz = 0.0;
  if (fast) {  // (wave-uniform) no -0.0 among the coordinates: 0.0 + x == x
    cx = sx[0];
    cy = sy[0];
    cz = sz[0];
  } else {
    asm volatile("" ::: "memory");  // (a real branch: if-converted this is three adds AND six selects)
    cx += sx[0];
    cy += sy[0];
    cz += sz[0];
  }
#pragma unroll
  for (int i = 1; i < (KT > 0 ? KT : KMAX); ++i) {  // util.py:37-40
    if (i < k) {
      cx += sx[i];
      cy += sy[i];
      cz += sz[i];
    }
  }
  // util.py:42-44: centroid / k with a true (correctly rounded) division.  For the integer
  // divisor k <= 16 the quotient c/k is never closer than 1/(2k) ulp to a rounding midpoint, so
  // RN(c*zh + RN(c*zl)) with zh + zl = 1/k to ~2^-106 IS the correctly rounded quotient:
  // one multiply + one FMA instead of the 11-instruction IEEE division sequence.
  div3_by_small_int(cx, cy, cz, k, fast);
  // util.py:48-57.  The reference starts every sum at 0.0: for the squares 0.0 + r*r == r*r exactly
  // (a square is never -0), so their first add is dropped; a cross product CAN be -0 and 0.0 + (-0) is
  // +0, so the cross terms keep it (axis-aligned samples reach the sign of a zero plane coefficient).
  double xx = 0.0, xy = 0.0, xz = 0.0, yy = 0.0, yz = 0.0, zz = 0.0;
#pragma unroll
  for (int i = 0; i < (KT > 0 ? KT : KMAX); ++i) {
    if (i < k) {
      const double rx = sx[i] - cx;
      const double ry = sy[i] - cy;
      const double rz = sz[i] - cz;
      if (i == 0) {
        xx = rx * rx;
        yy = ry * ry;
        zz = rz * rz;
      } else {
        xx += rx * rx;
        yy += ry * ry;
        zz += rz * rz;
      }
      xy += rx * ry;
      xz += rx * rz;
      yz += ry * rz;
    }
  }
  plane_from_moments(cx, cy, cz, xx, xy, xz, yy, yz, zz, plane, fast);
}

// util.py:22-24 with the f32 plane promoted to f64: ((a*x + b*y) + c*z) + d
__device__ __forceinline__ double plane_distance(double a, double b, double c, double d, double x,
                                                 double y, double z) {
  return fabs(((a * x + b * y) + c * z) + d);
}

// launch shapes (A/B history of every number: HISTORY.md 4 and 8)
constexpr int RS_MINWAVES = 4;         // waves per SIMD asked of the compiler
constexpr int RS_BIG_PER_CU = 16;      // workgroups per CU striding over the list of large blocks
constexpr int RS_BIG_THREADS = 256;    // H > 256: lanes per workgroup of the instance for blocks of 128 .. 255 points
constexpr int RS_BIG_HPL = 1024 / RS_BIG_THREADS;
constexpr int RS_SMALL_THREADS = 128;  // ... blocks with fewer points than this get two waves x 8 hypotheses per lane
constexpr int RS_TINY_THREADS = 64;    // ... and blocks with fewer points than this ONE wave x 16
constexpr int RS_PER_CU = 64, RS_SMALL_PER_CU = 128, RS_TINY_PER_CU = 256;   // workgroups per CU of the three
constexpr int RS_KMAX = 16;   // initial_points_number supported by the register path

// everything the kernel needs to know about one batch entry, written by k_block_desc so that a
// workgroup reads ONE 32-byte record instead of chasing order -> size -> start -> vstart
struct __attribute__((aligned(32))) BlockDesc {
  uint32_t pstart;   // physical start of the block in the point array
  int32_t n;         // points in the block
  int64_t vstart;    // start of the block in the reference's concatenated batch cloud
  uint32_t pspill;   // physical index of the point one past the block in that cloud (the first
                     // point of the next block; the last block of a batch clamps to its own last
                     // point): reachable only through f64 rounding of R*n + s
  uint32_t pad[3];
};

struct RansacOut {
  uint8_t* mask;
  float* plane;
  int32_t* count;
  int32_t* index;
};

// Position (inside its block) of sample i of a hypothesis: nb.int32(R * block_size + block_start)
// - block_start (cuda_ransac.py:103-107): f64 multiply, f64 add, truncation.  May be == n: the
// f64 rounding of R*n + s can reach the first point of the next block of the batch.
__device__ __forceinline__ int sample_index_exact(double r, int n, int64_t vstart) {
  const double v = r * (double)n + (double)vstart;
  const int g = (int)((int64_t)(int)v - vstart);
  return g < n ? g : n;
}

// The same position WITHOUT the block's start: with x = fl(R*n), g0 = trunc(x), the reference's
// index is g0 for EVERY start s in [0, 2^31) unless frac(x) > 1 - 2^-20: fl(x + s) differs from
// x + s by at most ulp/2 <= 2^-22, which cannot reach the next integer.  Positions are therefore
// a function of (hypothesis, n) alone - the kernel caches them while consecutive blocks have the
// same size - and the rare "risky" draws (probability 2^-20) take the exact path above.
__device__ __forceinline__ int sample_index_cached(double r, int n, bool* risky) {
  const double x = r * (double)n;
  const int g0 = (int)x;
  *risky = (x - (double)g0) > (1.0 - 0x1p-20);
  return g0 < n ? g0 : n;
}

// The same fit for ANY k (the reference puts no bound on initial_points_number): the k sampled points
// are streamed from global memory twice instead of being held in registers, and the centroid is
// divided by k with the true division (div_by_small_int is only proven for k <= 16).
__device__ __forceinline__ void plane_streamed(const double* __restrict__ hyp_row, int k, const BlockDesc& d,
                                               const double* __restrict__ xyz, float (&plane)[4]) {
  auto point = [&](int i) {
    const int g = sample_index_exact(hyp_row[i], d.n, d.vstart);
    return (g < d.n) ? (int64_t)d.pstart + g : (int64_t)d.pspill;
  };
  double cx = 0.0, cy = 0.0, cz = 0.0;
  for (int i = 0; i < k; ++i) {  // util.py:37-40
    const int64_t p = point(i);
    cx += xyz[3 * p];
    cy += xyz[3 * p + 1];
    cz += xyz[3 * p + 2];
  }
  const double kd = (double)k;
  cx = cx / kd;  // util.py:42-44
  cy = cy / kd;
  cz = cz / kd;
  double xx = 0.0, xy = 0.0, xz = 0.0, yy = 0.0, yz = 0.0, zz = 0.0;
  for (int i = 0; i < k; ++i) {  // util.py:48-57
    const int64_t p = point(i);
    const double rx = xyz[3 * p] - cx, ry = xyz[3 * p + 1] - cy, rz = xyz[3 * p + 2] - cz;
    xx += rx * rx;
    xy += rx * ry;
    xz += rx * rz;
    yy += ry * ry;
    yz += ry * rz;
    zz += rz * rz;
  }
  plane_from_moments(cx, cy, cz, xx, xy, xz, yy, yz, zz, plane);
}

// The exact (reference order) evaluation of one block whose points are in global memory.
// THREADS lanes, HPL hypotheses per lane (lane t owns hypotheses t, t+THREADS, ...).
template <int THREADS, int HPL, int KT>
__device__ __forceinline__ void ransac_block_global(const BlockDesc& d, int be,
                                                    const double* __restrict__ xyz,
                                                    const double* __restrict__ hyp, int H, int k_rt,
                                                    double thr, const RansacOut& out,
                                                    unsigned long long* s_best, float* s_plane) {
  constexpr int KS = KT > 0 ? KT : RS_KMAX;
  const int k = KT > 0 ? KT : k_rt;
  const int n = d.n;
  const int64_t pstart = d.pstart;
  const double* __restrict__ pts = xyz + 3 * pstart;
  double pa[HPL], pb[HPL], pc[HPL], pd[HPL];
  int cnt[HPL];
#pragma unroll
  for (int q = 0; q < HPL; ++q) {
    const int t = threadIdx.x + q * THREADS;
    cnt[q] = -1;
    pa[q] = pb[q] = pc[q] = pd[q] = 0.0;
    if (t < H) {
      float pf[4];
      if constexpr (KT < 0) {
        plane_streamed(hyp + (int64_t)t * k, k, d, xyz, pf);
      } else {
        double sx[KS], sy[KS], sz[KS];
#pragma unroll
        for (int i = 0; i < KS; ++i) {
          sx[i] = sy[i] = sz[i] = 0.0;
          if (i < k) {
            const int g = sample_index_exact(hyp[(int64_t)t * k + i], n, d.vstart);
            const int64_t p = (g < n) ? pstart + g : (int64_t)d.pspill;
            sx[i] = xyz[3 * p];
            sy[i] = xyz[3 * p + 1];
            sz[i] = xyz[3 * p + 2];
          }
        }
        plane_from_samples<KT, KS>(sx, sy, sz, k, pf);
      }
      pa[q] = (double)pf[0];
      pb[q] = (double)pf[1];
      pc[q] = (double)pf[2];
      pd[q] = (double)pf[3];
      cnt[q] = 0;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int i = 0; i < n; ++i) {  // wave-uniform scalar loads
    const double x = pts[3 * (int64_t)i], y = pts[3 * (int64_t)i + 1], z = pts[3 * (int64_t)i + 2];
#pragma unroll
    for (int q = 0; q < HPL; ++q)
      cnt[q] += (plane_distance(pa[q], pb[q], pc[q], pd[q], x, y, z) < thr) ? 1 : 0;
  }
  unsigned long long best = 0;
#pragma unroll
  for (int q = 0; q < HPL; ++q) {
    const int t = threadIdx.x + q * THREADS;
    if (t < H) {
      const unsigned long long key =
          ((unsigned long long)(unsigned)cnt[q] << 32) | (unsigned)(0x7FFFFFFF - t);
      best = key > best ? key : best;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(best, off);
    best = o > best ? o : best;
  }
  if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = best;
  __syncthreads();
  best = s_best[0];
#pragma unroll
  for (int w = 1; w < THREADS / 64; ++w) best = s_best[w] > best ? s_best[w] : best;
  const int win = 0x7FFFFFFF - (int)(unsigned)(best & 0xFFFFFFFFu);
#pragma unroll
  for (int q = 0; q < HPL; ++q) {
    if ((int)threadIdx.x + q * THREADS == win) {
      const float f0 = (float)pa[q], f1 = (float)pb[q], f2 = (float)pc[q], f3 = (float)pd[q];
      s_plane[0] = f0; s_plane[1] = f1; s_plane[2] = f2; s_plane[3] = f3;
      if (out.plane) {
        out.plane[4 * (int64_t)be + 0] = f0; out.plane[4 * (int64_t)be + 1] = f1;
        out.plane[4 * (int64_t)be + 2] = f2; out.plane[4 * (int64_t)be + 3] = f3;
      }
      if (out.count) out.count[be] = cnt[q];
      if (out.index) out.index[be] = win;
    }
  }
  __syncthreads();
  const double a = (double)s_plane[0], bb = (double)s_plane[1], c = (double)s_plane[2],
               dd = (double)s_plane[3];
  for (int i = threadIdx.x; i < n; i += THREADS)
    out.mask[pstart + i] = (plane_distance(a, bb, c, dd, pts[3 * (int64_t)i], pts[3 * (int64_t)i + 1],
                                           pts[3 * (int64_t)i + 2]) < thr) ? 1 : 0;
}

// Helpers of the screened scoring loop, as inline asm on purpose: the loop's instruction mix is
// the whole point.  Plain v_fma_f32: left to itself hipcc SLP-packs neighbouring hypotheses into
// v_pk_fma_f32, which measured SLOWER than two scalar FMAs on gfx950 (both as compiler output
// and hand-packed with op_sel broadcasts: 5.56 vs 5.47 ms in round 2; again in round 5 on the
// VALU-issue-bound kernel: 120 v_pk_fma_f32 for 240 v_fma_f32, same moves, 119 VGPRs - 3.60 vs
// 3.52 ms, although tools/probes/pkfma_probe.hip has the packed form at 5.2 cycles against
// 2 x 3.5 in isolation), and adds canonicalising v_max around fminf.
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float fma32(float a, float b, float c) {
  float r;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// min(m, |a|, |b|)
__device__ __forceinline__ float min3abs(float m, float a, float b) {
  float r;
  asm("v_min3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(m), "v"(a), "v"(b));
  return r;
}

// Maximum over the wavefront as six DPP-modified v_max (quad swaps, half-row / row mirrors, the two row
// broadcasts of gfx9) and one v_readlane - instead of six ds_bpermute round trips through the LDS crossbar per
// 32-bit word (the block's winner key and the block's extent are reduced once per block by every wave: with
// 64-bit keys that was 18 bpermutes + 6 64-bit compare / select pairs per wave and block).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
  // (lanes the row mask leaves out keep `old` = their own value: max(v, v) = v)
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  v = max(v, dpp_u32<0xB1, 0xF>(v));   // quad_perm [1,0,3,2]
  v = max(v, dpp_u32<0x4E, 0xF>(v));   // quad_perm [2,3,0,1]
  v = max(v, dpp_u32<0x141, 0xF>(v));  // row_half_mirror
  v = max(v, dpp_u32<0x140, 0xF>(v));  // row_mirror: every lane of a row holds the row's maximum
  v = max(v, dpp_u32<0x142, 0xA>(v));  // row_bcast:15 into rows 1 and 3
  v = max(v, dpp_u32<0x143, 0xC>(v));  // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// (non-negative floats order like their bit patterns; +inf included, NaN excluded by the caller)
__device__ __forceinline__ float wave_max_nonneg_f32(float m) {
  return __uint_as_float(wave_max_u32(__float_as_uint(m)));
}

// block-local f32 coordinates of the point a lane is staging (relative to the block's first
// point, loaded wave-uniformly by the caller well ahead of this call) and the per-wave maximum of their magnitudes
template <int THREADS>
__device__ __forceinline__ void stage_local(double ox, double oy, double oz, const BlockDesc& d,
                                            double px, double py, double pz, f4* loc,
                                            float* wext, uint32_t* wfast, float* ext_out = nullptr,
                                            uint32_t* fast_out = nullptr) {
  float m = 0.f;
  if ((int)threadIdx.x < d.n) {
    const float u = (float)(px - ox), v = (float)(py - oy), w = (float)(pz - oz);
    loc[threadIdx.x] = f4{u, v, w, 0.f};
    m = fmaxf(fabsf(u), fmaxf(fabsf(v), fabsf(w)));
    // a NaN coordinate would be dropped by fmaxf: force "everything is ambiguous" instead
    if (!(m == m) || u != u || v != v || w != w) m = __int_as_float(0x7f800000);
  }
  m = wave_max_nonneg_f32(m);
  // the range certificate of the plane fit's shortcuts (see coord_in_fast_range): lanes that stage nothing hold +0.0
  const bool inr = coord_in_fast_range(px) && coord_in_fast_range(py) && coord_in_fast_range(pz);
  const bool wave_fast = __all(inr);
  if (ext_out) {  // (one wave per block: the two words stay in the wave's registers)
    *ext_out = m;
    *fast_out = wave_fast ? 1u : 0u;
  } else if ((threadIdx.x & 63) == 0) {
    wext[threadIdx.x >> 6] = m;
    wfast[threadIdx.x >> 6] = wave_fast ? 1u : 0u;
  }
}

// The screened scoring of the block's n staged points (block-local f32 coordinates in LDS, read
// as wave-uniform broadcasts) against NH hypotheses of the lane; see "Screening" in k_ransac.
// Per point and hypothesis: 4 v_fma_f32, 1 v_alignbit (inlier bit into a 32-point history word,
// popcounted per 32 points) and half a v_min3_f32 (one margin for the group: the smallest |e|).
template <int NH>
__device__ __forceinline__ void screen_group(const f4* __restrict__ loc, int n, const float* fa,
                                             const float* fb, const float* fc, const float* sto,
                                             float nthr2, int* cnt, float* margin) {
  // the smallest |e| of every hypothesis (two running minima each: even / odd point pairs)
  float mg[2][NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) mg[0][h] = mg[1][h] = __int_as_float(0x7f800000);
  for (int base = 0; base < n; base += 32) {
    const int m = __builtin_amdgcn_readfirstlane(min(32, n - base));
    uint32_t hist[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) hist[h] = 0;
    auto score = [&](const f4 L, float (&e)[NH]) {
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const float sv = fma32(fa[h], L.x, fma32(fb[h], L.y, fma32(fc[h], L.z, sto[h])));
        e[h] = fma32(sv, sv, nthr2);
        hist[h] = __builtin_amdgcn_alignbit(hist[h], __float_as_uint(e[h]), 31);
      }
    };
    // unrolled by hand (the inline asm keeps the loop unroller away): 4 LDS reads in flight
    // (issuing the next iteration's reads ahead of the scoring gained nothing and cost 16 VGPRs)
    int i = 0;
    for (; i + 4 <= m; i += 4) {
      const f4 L0 = loc[base + i], L1 = loc[base + i + 1], L2 = loc[base + i + 2],
               L3 = loc[base + i + 3];
      float e0[NH], e1[NH], e2[NH], e3[NH];
      score(L0, e0);
      score(L1, e1);
      score(L2, e2);
      score(L3, e3);
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        mg[0][h] = min3abs(mg[0][h], e0[h], e1[h]);
        mg[1][h] = min3abs(mg[1][h], e2[h], e3[h]);
      }
    }
    for (; i < m; ++i) {
      float e[NH];
      score(loc[base + i], e);
#pragma unroll
      for (int h = 0; h < NH; ++h) mg[0][h] = min3abs(mg[0][h], e[h], e[h]);
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) cnt[h] += __popc(hist[h]);
  }
#pragma unroll
  for (int h = 0; h < NH; ++h) margin[h] = fminf(mg[0][h], mg[1][h]);
}

// ---- the prescreen of the hypotheses (round 6) ----------------------------------------------------------------
// The reference fits 1024 planes per leaf and keeps the one with the most inliers, the lowest index among the tied
// (cuda_ransac.py:100-146).  After the first 64 hypotheses have been evaluated exactly, a later one matters only if
// its inlier count EXCEEDS the best count L so far - and on a planar leaf hardly any does (2-3 of the other 960 on the
// benchmark scene, 8-9 on a uniform cloud).  The prescreen decides that without the reference's arithmetic: an
// APPROXIMATE plane of the same six sample points in f32 on the block's local coordinates (FMAs, no division, no
// square root), a rigorous bound eps on how far its distances can be from the reference's, and the count of the
// block's points inside the threshold WIDENED by eps - an upper bound of the reference's count.  A hypothesis whose
// upper bound does not exceed L cannot win (a tie goes to the lower index, which L belongs to); the others - and
// every hypothesis the bound cannot vouch for - are fitted and scored exactly as before.  Results are bit-identical.
//
// The bound (u = 2^-24; E = the block's extent in local coordinates, G >= |coordinate| of its points; "ideal" =
// exact arithmetic on the exact coordinates; T = trace of the six-sample covariance sums = sum of |residual|^2).
// The covariance about ANY point c~ near the centroid c differs from the one about c by 6 (c - c~)_a (c - c~)_b, so the
// f32 centroid (error 8 u E) only has to be subtracted accurately:
//   residual component r of a sample against c~      |err| <= u E + u |r|     (input rounding of the local coordinate,
//       one f32 subtraction; the reference: f64 operations on global coordinates, util.py:35-52, 2^-49 G)
//   each of the six covariance sums (sum over samples of |r_a| <= sqrt(6 T), sqrt(T) <= (T / E + E) / 2,
//       FMA chain 7 u T, the shifted centre 384 u^2 E^2)
//                                                     |err| <= sigma = T (12 u + 2^-46 G / E) + 3 u E^2 + 2^-46 G E
//   each cofactor (|S_ab| <= T)                       |err| <= mu = T (4 sigma + 4 u T) + 4 sigma^2 (+ 2^-100)
//   the reference branches on its three diagonal cofactors (util.py:63-74): the approximate ones pick the same row
//       when the largest exceeds the other two by more than 4 mu;
//   normal = row / |row|: |n_approx - n_ideal|, |n_ref - n_ideal| <= 2 sqrt3 mu / |row| (+ 4 u), valid while
//       |row| > 4 mu (the reference's zero-norm plane, util.py:77-78, has |row| <= 2 sqrt3 mu: never vouched for);
//   distance of a block point p (|p - centroid| <= 2 sqrt3 E): normal error x lever, the f32 evaluation of the
//       approximate plane in local coordinates and through c~ instead of c (36 u E), the f32 rounding of the
//       reference's GLOBAL plane (3.5 u G, cuda_ransac.py:110-113):
//       |s_approx - t_ref| <= eps = 3.5 E (4 mu / |row|) + 64 u E + 2^-21 G.
// Blocks outside 2^-7 <= E <= 2^7 (f32 under- / overflow of the fourth-order terms), without the range certificate
// (coord_in_fast_range), or whose constant part of eps exceeds thr / 8 take every hypothesis through the exact path.
struct PreConst {     // block-uniform constants of the bound, every one rounded UP
  float qa, qb, qc;       // 4 mu (1 + 2^-10) = (qa T + qb) T + qc
  float e35;              // 3.5 E (1 + 2^-10)
  float thrblk;           // (thr + 64 u E + 2^-21 G) (1 + 2^-10)
};
__device__ __forceinline__ bool prescreen_constants(float extent, double ox, double oy, double oz, double thr,
                                                    PreConst& pc) {
  const float up = 0x1.004p0f;                    // 1 + 2^-10
  const float u = 0x1p-24f;
  const float E = extent;
  const float G = ((float)(fabs(ox) + fabs(oy) + fabs(oz)) + E) * 0x1.0002p0f;
  // 16 sigma = k1 T + k0
  const float k1 = 16.f * (12.f * u + 0x1p-46f * (G / E) * 0x1.0002p0f) * 0x1.0002p0f;
  const float k0 = 16.f * (3.f * u * (E * E) + 0x1p-46f * G * E) * 0x1.0002p0f;
  // 4 mu = T (16 sigma + 16 u T) + 16 sigma^2 = T^2 (k1 + 16 u + k1^2 / 16) + T (k0 + k1 k0 / 8) + k0^2 / 16 (+ 2^-98)
  pc.qa = (k1 + 0x1p-20f + k1 * k1 * 0x1p-4f) * up;
  pc.qb = (k0 + k1 * k0 * 0x1p-3f) * up;
  pc.qc = (k0 * k0 * 0x1p-4f) * up + 0x1p-98f;
  pc.e35 = 3.5f * E * up;
  const float eblk = (64.f * u * E + 0x1p-21f * G) * 0x1.0002p0f;
  const float thr_f = (float)thr * 0x1.0002p0f;
  pc.thrblk = (thr_f + eblk) * up;
  // (all comparisons are false for NaN)
  return E >= 0x1p-7f && E <= 0x1p7f && thr >= 0x1p-40 && thr <= 0x1p40 && eblk * 8.f <= thr_f;
}

// count of the block's points inside each hypothesis' WIDENED threshold: e = fma(s, s, -T2) < 0, s as in
// screen_group.  No margins: the widening is the bound.
template <int NH>
__device__ __forceinline__ void screen_ub(const f4* __restrict__ loc, int n, const float* fa, const float* fb,
                                          const float* fc, const float* sto, const float* T2, int* cnt) {
  for (int base = 0; base < n; base += 32) {
    const int m = __builtin_amdgcn_readfirstlane(min(32, n - base));
    uint32_t hist[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) hist[h] = 0;
    auto score = [&](const f4 L) {
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const float sv = fma32(fa[h], L.x, fma32(fb[h], L.y, fma32(fc[h], L.z, sto[h])));
        float e;
        asm("v_fma_f32 %0, %1, %1, -%2" : "=v"(e) : "v"(sv), "v"(T2[h]));
        hist[h] = __builtin_amdgcn_alignbit(hist[h], __float_as_uint(e), 31);
      }
    };
    int i = 0;
    for (; i + 4 <= m; i += 4) {
      const f4 L0 = loc[base + i], L1 = loc[base + i + 1], L2 = loc[base + i + 2], L3 = loc[base + i + 3];
      score(L0);
      score(L1);
      score(L2);
      score(L3);
    }
    for (; i < m; ++i) score(loc[base + i]);
#pragma unroll
    for (int h = 0; h < NH; ++h) cnt[h] += __popc(hist[h]);
  }
}

// -DRS_COUNTS (experiments only: tools/build_variant.sh): what the kernel EXECUTES of what the algorithm asks for.
// Wave 0 of every workgroup keeps the counts in registers and adds them to a device array at the end of the kernel;
// tools/rs_counts.py reads it through octl_debug_rs_stamps (-> profiles/rNN_ransac_counts.json):
//   [0] shader-clock ticks / [1] 100 MHz ticks over the workgroups' lives (s_memtime / s_memrealtime: the clock the chip
//       held under this kernel's load, MI355X_MICROARCH.md "DVFS give-back")
//   [2] survivors of the prescreen   [3] batches of hypotheses fitted exactly behind group 0   [4] blocks prescreened
//   [5] blocks whose survivors overflowed the queue   [8] blocks   [9] blocks that ended with group 0
//   [10] workgroups   [11] sum of block sizes   [12] hypothesis groups prescreened   [13] hypotheses recounted exactly
//   [14] hypothesis groups fitted and scored exactly   [15] sum over those groups of the block size
#ifdef RS_COUNTS
__device__ unsigned long long g_rs_stamps[16];
#define RS_COUNT_INIT unsigned long long _rs_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; \
                      const unsigned long long _rs_m0 = __builtin_amdgcn_s_memtime(),                      \
                                               _rs_r0 = __builtin_amdgcn_s_memrealtime()
#define RS_COUNT(k, v) _rs_acc[k] += (unsigned long long)(v)
#define RS_COUNT_FLUSH                                                                  \
  do {                                                                                  \
    if (threadIdx.x == 0) {                                                             \
      _rs_acc[0] = __builtin_amdgcn_s_memtime() - _rs_m0;                               \
      _rs_acc[1] = __builtin_amdgcn_s_memrealtime() - _rs_r0;                           \
      _rs_acc[10] += 1;                                                                 \
      for (int _k = 0; _k < 16; ++_k) atomicAdd(&g_rs_stamps[_k], _rs_acc[_k]);         \
    }                                                                                   \
  } while (0)
#else
#define RS_COUNT_INIT do {} while (0)
#define RS_COUNT(k, v) do {} while (0)
#define RS_COUNT_FLUSH do {} while (0)
#endif

// Persistent workgroups over the descriptors of all blocks with k <= n <= THREADS-1 points,
// SORTED BY SIZE (largest first): workgroup w handles a contiguous chunk of the list
//   * the grid is oversubscribed (64 ... 256 workgroups per CU, 4 ... 16 resident): the chunks with the largest
//     blocks are dispatched first and the dispatcher evens out the rest (no tail);
//   * while entry e is computed out of one LDS buffer the points of entry e+1 are already in
//     flight into registers and the descriptor of entry e+2 is being fetched, so the per-block
//     latency chain (descriptor -> points) is off the critical path;
//   * the block's points are three f64 LDS arrays: the sampled points of a hypothesis are
//     per-lane LDS gathers (conflict free up to 32 points), the scoring loop reads each point as
//     a wave-uniform broadcast;
//   * ONE barrier per block (three rotating point buffers, reduction slots by parity).
// A lane fits ONE hypothesis at a time: first its hypothesis of group 0 (index lane < THREADS), then - HPL > 1 -
// whichever later ones the prescreen could not rule out (or all of them, where there is no prescreen).
// FULLH: the table has exactly THREADS x HPL hypotheses (the default 1024): the `index < H` guards fold away.
// PT: the sample positions - a function of (block size, hypothesis) alone, see sample_index_cached - come from the
// launch's position table (pos_table_part: k <= 6); otherwise they are computed from the table row in front of a fit.
template <int THREADS, int HPL, int KT, bool FULLH, bool PT>
__global__ __launch_bounds__(THREADS, RS_MINWAVES) void k_ransac(
    const double* __restrict__ xyz, const BlockDesc* __restrict__ sdesc,
    const uint32_t* __restrict__ lo_ptr, const uint32_t* __restrict__ hi_ptr, const double* __restrict__ hyp, int H,
    int k_rt, double thr, RansacOut out, const uint2* __restrict__ pos_tab, int no_prescreen) {
  static_assert(!PT || (KT > 0 && KT <= 6), "the position table packs up to six positions and their risk bits");
  static_assert(THREADS * HPL <= 1024, "hypothesis index in 10 bits");
  // PRE: the instances with the prescreen of the hypotheses (see prescreen_constants): three to six sample points,
  // positions from the table, more than one hypothesis per lane - one, two or four waves per block.  Every WAVE
  // prescreens its own hypotheses against the best count of its OWN first group (indices below THREADS, lower than
  // every later one): no exchange between the waves before the block's one barrier.
  constexpr bool PRE = PT && KT >= 3 && HPL >= 2;
  constexpr int W = THREADS / 64;
  constexpr int KS = KT > 0 ? KT : RS_KMAX;
  constexpr int GW = (KS + 3) / 4;  // packed sample positions: one byte each
  // survivors a wave can queue (more: every hypothesis takes the exact path); an entry is the hypothesis index, and
  // with one wave per block (under 64 points) its count bound above it
  constexpr int PRE_LIST = THREADS == 64 ? 512 : 256;
  __shared__ uint16_t s_surv_all[PRE ? W : 1][PRE ? PRE_LIST : 1];
  uint16_t* const s_surv = s_surv_all[PRE ? (threadIdx.x >> 6) : 0];
  const int k = KT > 0 ? KT : k_rt;
  __shared__ double s_pts[3][3][THREADS];        // rotated: block t lives in buffer t % 3
  __shared__ uint32_t s_wbest[2][W];             // by block parity
  __shared__ float s_wplane[2][W][4];
  // f32 screening of the scoring loop (see "screening" below): block-local f32 coordinates
  // relative to the block's first point and, per wave, the largest |coordinate|
  __shared__ f4 s_loc[3][THREADS];
  __shared__ float s_wext[3][W];
  __shared__ uint32_t s_wfast[3][W];
  // this launch's part [lo, hi) of the size-sorted list (device-side counts; lo_ptr == nullptr: from the front)
  const int lo = lo_ptr ? (int)*lo_ptr : 0;
  const int hi = (int)*hi_ptr;
  const int nbs = hi - lo;
  // workgroup w owns the CONTIGUOUS chunk [w*C, (w+1)*C) of the size-sorted list: the chunks with
  // the largest blocks are dispatched first
  const int C = (nbs + (int)gridDim.x - 1) / (int)gridDim.x;
  int j = lo + (int)blockIdx.x * C;
  const int j_end = min(hi, j + C);
  if (j >= j_end) return;
  float ext_carry = 0.f;       // (W == 1) extent and range certificate of the staged block
  uint32_t fast_carry = 0;
  BlockDesc cur = sdesc[j];
  BlockDesc nxt = cur;
  if (j + 1 < j_end) nxt = sdesc[j + 1];
  {
    double px = 0.0, py = 0.0, pz = 0.0;
    if ((int)threadIdx.x <= cur.n) {
      const int64_t p = ((int)threadIdx.x < cur.n) ? (int64_t)cur.pstart + threadIdx.x : (int64_t)cur.pspill;
      px = xyz[3 * p];
      py = xyz[3 * p + 1];
      pz = xyz[3 * p + 2];
      s_pts[0][0][threadIdx.x] = px;
      s_pts[0][1][threadIdx.x] = py;
      s_pts[0][2][threadIdx.x] = pz;
    }
    stage_local<THREADS>(xyz[3 * (int64_t)cur.pstart], xyz[3 * (int64_t)cur.pstart + 1],
                         xyz[3 * (int64_t)cur.pstart + 2], cur, px, py, pz, s_loc[0], s_wext[0], s_wfast[0],
                         W == 1 ? &ext_carry : nullptr, W == 1 ? &fast_carry : nullptr);
  }
  __syncthreads();
  int buf = 0, par = 0;
  RS_COUNT_INIT;
  for (;;) {
    // lane-constant addresses (LDS slots, output offsets) are cheap to recompute; derived from
    // threadIdx.x directly the compiler hoists them out of this loop and then SPILLS them (48 B of
    // scratch per lane, written by every workgroup: 200 MB of HBM writes per launch - measured).
    // An opaque copy of the lane index per iteration keeps them in-loop.
    unsigned tx = threadIdx.x;
    asm volatile("" : "+v"(tx));
    const bool has_next = j + 1 < j_end;
    BlockDesc nxt2 = nxt;
    if (j + 2 < j_end) nxt2 = sdesc[j + 2];
    // points of the next block -> registers (in flight during the compute below)
    double rx = 0.0, ry = 0.0, rz = 0.0;
    const bool pre = has_next && (int)tx <= nxt.n;
    if (pre) {
      const int64_t p = ((int)tx < nxt.n) ? (int64_t)nxt.pstart + tx : (int64_t)nxt.pspill;
      rx = xyz[3 * p];
      ry = xyz[3 * p + 1];
      rz = xyz[3 * p + 2];
    }
    // ... and its first point, the origin of its block-local coordinates: a wave-uniform load whose latency
    // used to sit in front of the staging at the END of the iteration
    double nox = 0.0, noy = 0.0, noz = 0.0;
    if (has_next) {
      nox = xyz[3 * (int64_t)nxt.pstart];
      noy = xyz[3 * (int64_t)nxt.pstart + 1];
      noz = xyz[3 * (int64_t)nxt.pstart + 2];
    }
    const int n = cur.n;
    const int be = (int)cur.pad[1];
    const double* __restrict__ lx = s_pts[buf][0];
    const double* __restrict__ ly = s_pts[buf][1];
    const double* __restrict__ lz = s_pts[buf][2];

    RS_COUNT(8, 1);
    RS_COUNT(11, n);
    // ---- the block's hypotheses -------------------------------------------------------------
    // Screening.  The H x n distance tests are evaluated in f32 on block-local coordinates; a
    // pair is decided there only when its f32 distance is farther from the threshold than a
    // rigorous bound of |s - t_ref|, everything else is re-evaluated with the reference's exact
    // f64 sequence, so the inlier COUNTS are exactly the reference's:
    //   reference   t_ref = fl(fl(fl(A x + B y) + C z) + D)           (util.py:22-24, f64)
    //   identity    A x + B y + C z + D = T_o + A (x-ox) + B (y-oy) + C (z-oz),   o = first point
    //   screen      s = fma32(a, u, fma32(b, v, fma32(c, w, to)))     u = fl32(x-ox).., to = fl32(T_o)
    //   |s - t_ref| <= 2^-24 (4 |T_o| + 9 E) + 2^-50 (|o|_1 + |D| + E),   E = max |u|,|v|,|w|
    //     (three f32 FMA roundings on partial sums <= |T_o| + 3E, the f32 roundings of u,v,w and
    //      of to, the f64 roundings of T_o and of t_ref itself; |a|,|b|,|c| <= 1)
    //   delta = twice that bound
    // The test itself is done on squares:
    //   e = fma32(s, s, -fl32(thr^2))     sign(e) = inlier bit, |e| = distance from the threshold
    //   |e| >= dprime = 2 thr delta + delta^2 + 2^-21 thr^2   =>   | |s| - thr | >= delta and the
    //   sign of e is the sign of s^2 - thr^2 (the last term absorbs the roundings of e and thr^2)
    // so a hypothesis whose smallest |e| over the block stays above its dprime has the
    // reference's count; otherwise (rare) it is recounted with the exact f64 sequence
    // (screen_group: 5.5 instructions per point and hypothesis against 8 f64 ones).
    //
    // Early exit (exact).  The winner is the LOWEST index among the hypotheses with the maximal
    // count, and no count exceeds n.  Hypotheses 0 .. THREADS-1 (group 0) go first; when one of them holds
    // all n points, no later hypothesis can win - on the benchmark scene this is one leaf in five.
    const double ox = lx[0], oy = ly[0], oz = lz[0];
    float extent = W == 1 ? ext_carry : s_wext[buf][0];
#pragma unroll
    for (int w = 1; w < W; ++w) extent = fmaxf(extent, s_wext[buf][w]);
    // every staged coordinate (block + spill point) is +0.0 or in [2^-30, 2^31): the plane fits run without
    // their per-lane range guards (see coord_in_fast_range).  Wave-uniform, kept in an SGPR.
    uint32_t fastw = W == 1 ? fast_carry : s_wfast[buf][0];
#pragma unroll
    for (int w = 1; w < W; ++w) fastw &= s_wfast[buf][w];
    const bool blk_fast = __builtin_amdgcn_readfirstlane((int)fastw) != 0;
    // block-uniform parts of the bound.  Thresholds or extents outside the sane range (nobody's
    // plane tolerance) are always recounted exactly.
    const double delta_blk = fma(0x1p-23 * 9.0, (double)extent,
                                 0x1p-49 * (fabs(ox) + fabs(oy) + fabs(oz) + (double)extent + 1.0));
    const double dprime_thr = 0x1p-21 * (thr * thr);
    const bool blk_sane = thr >= 0x1p-40 && thr <= 0x1p40 && extent < 0x1p60f;
    const float nthr2 = -(float)(thr * thr);
    const f4* __restrict__ loc = s_loc[buf];
    // f32 forms of the bound's block-uniform parts, rounded up
    const bool fast_screen = blk_fast && thr >= 0x1p-40 && thr <= 0x1p40;
    const float delta_blk_f = (float)delta_blk * 0x1.0002p0f;
    const float thr2_f = (float)(thr + thr) * 0x1.0002p0f;
    const float dprime_thr_f = (float)dprime_thr * 0x1.0002p0f;

    // ---- one hypothesis per lane, exactly: positions, plane fit, count --------------------------------------------
    uint32_t gpk[GW], rk = 0;     // packed sample positions of the lane's hypothesis and their risky-draw bits
    float fa = 0.f, fb = 0.f, fc = 0.f, fd = 0.f, sto = 0.f, sdl = 0.f;
    int cnt = 0;
    auto load_pos = [&](const int t, const bool act) {
#pragma unroll
      for (int w = 0; w < GW; ++w) gpk[w] = 0;
      rk = 0;
      if constexpr (PT) {
        // (8 bytes per lane out of the launch's table)
        const uint2 e = pos_tab[(size_t)n * (size_t)H + (size_t)(act ? t : 0)];
        gpk[0] = e.x;
        if (GW > 1) gpk[GW - 1] = e.y & 0xFFFFu;
        rk = e.y >> 16;
      } else if (act) {
        const double* __restrict__ row = hyp + (int64_t)t * k;
#pragma unroll
        for (int i = 0; i < KS; ++i) {
          if (i < k) {
            bool risky;
            const int g = sample_index_cached(row[i], n, &risky);
            gpk[i >> 2] |= (uint32_t)g << (8 * (i & 3));
            rk |= (risky || g == n) ? (1u << i) : 0u;
          }
        }
      }
    };
    // plane of hypothesis t (f32, as the reference stores it) + its screening constants
    auto fit = [&](const int t, const bool act) {
      cnt = 0;
      fa = fb = fc = fd = sto = sdl = 0.f;
      if (FULLH || act) {
        double sx[KS], sy[KS], sz[KS];
        // does any lane of this wave hold a draw that needs the block's start (probability 2^-20
        // per draw)?  Otherwise the gathers skip the per-sample check.
        if (!__any(rk != 0u)) {  // wave-uniform: practically always
#pragma unroll
          for (int i = 0; i < KS; ++i) {
            sx[i] = sy[i] = sz[i] = 0.0;
            if (i < k) {
              const int g = (int)((gpk[i >> 2] >> (8 * (i & 3))) & 0xFFu);
              sx[i] = lx[g];
              sy[i] = ly[g];
              sz[i] = lz[g];
            }
          }
        } else {
#pragma unroll
          for (int i = 0; i < KS; ++i) {
            sx[i] = sy[i] = sz[i] = 0.0;
            if (i < k) {
              int g = (int)((gpk[i >> 2] >> (8 * (i & 3))) & 0xFFu);
              if (rk & (1u << i)) g = sample_index_exact(hyp[(int64_t)t * k + i], n, cur.vstart);
              sx[i] = lx[g];
              sy[i] = ly[g];
              sz[i] = lz[g];
            }
          }
        }
        float pf[4];
        plane_from_samples<KT, KS>(sx, sy, sz, k, pf, blk_fast);
        // (explicit fma: error BOUNDS and the screen's own inputs, not parity arithmetic)
        const double A = (double)pf[0], B = (double)pf[1], Cc = (double)pf[2], D = (double)pf[3];
        const double to = fma(A, ox, fma(B, oy, fma(Cc, oz, D)));
        fa = pf[0]; fb = pf[1]; fc = pf[2]; fd = pf[3];
        sto = (float)to;
        if (fast_screen) {
          // (wave-uniform) the same bound evaluated in f32, every constant rounded UP and the result
          // inflated by 2^-15 - the f32 roundings and the rounding of `to` lose at most 2^-21 of it;
          // |to| < 2^35 and no NaN under the block's range certificate, thr is in range: always "sane"
          const float dl = fma32(fabsf(sto), 0x1.0002p-21f, fma32(fabsf(pf[3]), 0x1.0002p-49f, delta_blk_f));
          sdl = fma32(dl, thr2_f + dl, dprime_thr_f) * 0x1.0002p0f;
        } else {
          const double delta = fma(0x1p-21, fabs(to), fma(0x1p-49, fabs(D), delta_blk));
          const double dprime = fma(delta, thr + thr + delta, dprime_thr) * 1.000001;
          const bool sane = blk_sane && (fabs(to) < 0x1p60);  // false for NaN
          sdl = sane ? (float)dprime : __int_as_float(0x7f800000);
        }
      }
    };
    // its count: the screen, then the exact recount where it cannot decide
    auto score = [&](const bool act) {
      float margin[1];
      int c1[1] = {0};
      RS_COUNT(14, 1);
      RS_COUNT(15, n);
      screen_group<1>(loc, n, &fa, &fb, &fc, &sto, nthr2, c1, margin);
      cnt = c1[0];
      // a hypothesis whose smallest |e| does not clear its bound is recounted exactly, by the whole wavefront
      // for one flagged lane at a time (almost always exactly one): its plane is broadcast, every lane tests one
      // point per round and a ballot counts.  Left to the flagged lane alone it is a serial loop over the block
      // that holds up the lane's wave - and behind it the workgroup's barrier - for ~n LDS round trips.
      const bool redo = act && !(margin[0] > sdl);  // (NaN: recount)
      unsigned long long todo = __ballot(redo);
      RS_COUNT(13, __popcll(todo));
      while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const double A = (double)__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(fa), src));
        const double B = (double)__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(fb), src));
        const double Cc = (double)__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(fc), src));
        const double D = (double)__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(fd), src));
        int c = 0;
        for (int base = 0; base < n; base += 64) {
          const int ii = base + (int)(tx & 63u);
          const bool in = ii < n && plane_distance(A, B, Cc, D, lx[ii < n ? ii : 0], ly[ii < n ? ii : 0],
                                                   lz[ii < n ? ii : 0]) < thr;
          c += __popcll(__ballot(in));
        }
        if ((int)(tx & 63u) == src) cnt = c;
      }
    };
    // The lane's best hypothesis so far - lowest hypothesis index among the tied (cuda_ransac.py:125-146)
    // (one 32-bit key: (count + 1) << 10 | 1023 - index; count <= 255, index < 1024 <= H_max; 0 = no hypothesis)
    uint32_t best = 0;
    float wa = 0.f, wb = 0.f, wc = 0.f, wd = 0.f;
    auto take = [&](const int t, const bool act) {
      if (act) {
        const uint32_t key = ((uint32_t)(cnt + 1) << 10) | (uint32_t)(1023 - t);
        if (key > best) {
          best = key;
          wa = fa; wb = fb; wc = fc; wd = fd;
        }
      }
    };
    // ---- group 0 (hypotheses 0 .. THREADS-1) ----------------------------------------------------------------------
    {
      const int t = (int)tx;
      const bool act = FULLH || t < H;
      load_pos(t, act);
      fit(t, act);
      score(act);
      take(t, act);
    }
    if constexpr (HPL > 1) {
      // the best exact count so far (wave-uniform); a hypothesis that holds every point ends the block for this wave:
      // every later hypothesis of the wave has a higher index (decided per wavefront, no barrier)
      int Lcur = (int)wave_max_u32((FULLH || (int)tx < H) ? (uint32_t)cnt : 0u);
      if (Lcur != n) {
        // ---- prescreen of the wave's later hypotheses ---------------------------------------------------------------
        int S = 0;   // survivors queued (wave-uniform)
        bool use_list = false;
        if constexpr (PRE) {
          PreConst pc;
          const bool elig = prescreen_constants(extent, ox, oy, oz, thr, pc) && blk_fast && !no_prescreen;
          if (elig) {
            RS_COUNT(4, 1);
            // NH hypothesis groups at a time (three; the last one or two of a lane's HPL - 1 on their own): every point
            // read from LDS serves NH hypotheses of the lane
            auto prescreen = [&](auto nh_tag, const int qb) {
              constexpr int NH = decltype(nh_tag)::value;
              uint2 pe[NH];
              auto hyp_a = [&](const int h) { return (int)tx + (qb + h) * THREADS; };
#pragma unroll
              for (int h = 0; h < NH; ++h)
                pe[h] = pos_tab[(size_t)n * (size_t)H + (size_t)((FULLH || hyp_a(h) < H) ? hyp_a(h) : 0)];
              float qa[NH], qb_[NH], qc[NH], qto[NH], qT2[NH];
              bool qok[NH];
              int qcnt[NH];
              uint32_t risk = 0;   // a draw of these hypotheses that the position table cannot vouch for (bits 16..21)
#pragma unroll
              for (int h = 0; h < NH; ++h) {
                const uint32_t w0 = pe[h].x, w1 = pe[h].y;
                f4 P[KS];
#pragma unroll
                for (int i = 0; i < KS; ++i) P[i] = loc[((i < 4 ? w0 : w1) >> (8 * (i & 3))) & 0xFFu];
                // centroid (only a centre near it is needed), residuals, covariance sums: util.py:35-57 in f32
                const float inv_k = 1.0f / (float)KS;
                float cx = P[0].x, cy = P[0].y, cz = P[0].z;
#pragma unroll
                for (int i = 1; i < KS; ++i) {
                  cx += P[i].x;
                  cy += P[i].y;
                  cz += P[i].z;
                }
                cx *= inv_k;
                cy *= inv_k;
                cz *= inv_k;
                float xx, xy, xz, yy, yz, zz;
                {
                  const float ax = P[0].x - cx, ay = P[0].y - cy, az = P[0].z - cz;
                  xx = ax * ax; xy = ax * ay; xz = ax * az; yy = ay * ay; yz = ay * az; zz = az * az;
                }
#pragma unroll
                for (int i = 1; i < KS; ++i) {
                  const float ax = P[i].x - cx, ay = P[i].y - cy, az = P[i].z - cz;
                  xx = fma32(ax, ax, xx); xy = fma32(ax, ay, xy); xz = fma32(ax, az, xz);
                  yy = fma32(ay, ay, yy); yz = fma32(ay, az, yz); zz = fma32(az, az, zz);
                }
                // the three diagonal cofactors and the row the reference's branch takes (util.py:59-74)
                const float dx = fma32(yy, zz, -(yz * yz)), dy = fma32(xx, zz, -(xz * xz)), dz = fma32(xx, yy, -(xy * xy));
                const float cA = fma32(xz, yz, -(xy * zz)), cB = fma32(xy, yz, -(xz * yy)), cC = fma32(xy, xz, -(yz * xx));
                const bool is_x = dx > dy && dx > dz;
                const bool is_y = !is_x && dy > dz;
                const float m1 = is_x ? dx : (is_y ? cA : cB);
                const float m2 = is_x ? cA : (is_y ? dy : cC);
                const float m3 = is_x ? cB : (is_y ? cC : dz);
                const float ss = fma32(m3, m3, fma32(m2, m2, m1 * m1));
                const float r = __builtin_amdgcn_rsqf(ss);
                const float a = m1 * r, b = m2 * r, c = m3 * r;
                qa[h] = a; qb_[h] = b; qc[h] = c;
                float to = -fma32(c, cz, fma32(b, cy, a * cx));
                asm volatile("" : "+v"(to));   // (materialised here: left symbolic, the negation is redone per point of the screen)
                qto[h] = to;
                // the bound
                const float T = (xx + yy) + zz;
                const float mu4 = fma32(fma32(pc.qa, T, pc.qb), T, pc.qc);
                const float gap = __builtin_fmaxf(__builtin_fmaxf(dx, dy), dz) - __builtin_amdgcn_fmed3f(dx, dy, dz);
                // (false for NaN.  f32 range: with E >= 2^-7, 4 mu >= 2^-65; a row that passes with |row|^2 in the
                //  denormal range - its normalisation good to 2^-18 only - has 4 mu / |row| > 2^-3, an eps of half the
                //  block's extent: nothing is ruled out on such a plane.  |row|^2 flushed to zero fails the test.)
                qok[h] = (ss * r > mu4) && (gap > mu4);
                risk |= w1;
                const float thr_h = fma32(pc.e35, mu4 * r, pc.thrblk);
                qT2[h] = thr_h * thr_h;
                qcnt[h] = 0;
              }
              if (__any((risk >> 16) != 0u)) {   // (2^-20 per draw: such a hypothesis takes the exact path)
#pragma unroll
                for (int h = 0; h < NH; ++h) qok[h] = qok[h] && (pe[h].y >> 16) == 0u;
              }
              screen_ub<NH>(loc, n, qa, qb_, qc, qto, qT2, qcnt);
              RS_COUNT(12, NH);
#pragma unroll
              for (int h = 0; h < NH; ++h) {
                const bool sv = !(qok[h] && qcnt[h] <= Lcur) && (FULLH || hyp_a(h) < H);
                const unsigned long long mk = __ballot(sv);
                if (sv) {
                  const int pos = S + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                  const uint32_t ub = (THREADS == 64 && qok[h]) ? (uint32_t)qcnt[h] : 63u;
                  if (pos < PRE_LIST) s_surv[pos] = (uint16_t)((ub << 10) | (uint32_t)hyp_a(h));
                }
                S += __popcll(mk);
              }
            };
            constexpr int PNH = 3;   // (four or five at a time: within the box-to-box noise, HISTORY 9)
            constexpr int TRIOS = (HPL - 1) / PNH, REST = (HPL - 1) % PNH;
#pragma unroll 1
            for (int i = 0; i < TRIOS; ++i) prescreen(std::integral_constant<int, PNH>{}, 1 + PNH * i);
            if constexpr (REST > 0) prescreen(std::integral_constant<int, (REST > 0 ? REST : 1)>{}, 1 + PNH * TRIOS);
            RS_COUNT(2, S);
            use_list = S <= PRE_LIST;
            if (!use_list) RS_COUNT(5, 1);
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        // ---- the survivors (in index order) - or, without a prescreen, groups 1 .. HPL-1 - exactly ------------------
        const int nbatch = use_list ? ((S + 63) >> 6) : HPL - 1;
#pragma unroll 1
        for (int bt = 0; bt < nbatch; ++bt) {
          int t;
          bool act;
          if (use_list) {
            const int idx = bt * 64 + (int)(tx & 63u);
            const uint32_t ent = s_surv[idx < S ? idx : S - 1];
            t = (int)(ent & 1023u);
            // (one wave per block: the best count may have grown since the hypothesis was queued - by a hypothesis of
            //  lower index - beyond what its bound allows)
            act = idx < S && (THREADS != 64 || (int)(ent >> 10) > Lcur);
          } else {
            t = (int)tx + (bt + 1) * THREADS;
            act = FULLH || t < H;
          }
          if (!__any(act)) continue;
          RS_COUNT(3, 1);
          load_pos(t, act);
          fit(t, act);
          score(act);
          take(t, act);
          Lcur = max(Lcur, (int)wave_max_u32(act ? (uint32_t)cnt : 0u));
          if (Lcur == n) break;   // (every later hypothesis of the wave has a higher index)
        }
      } else {
        RS_COUNT(9, 1);
      }
    }
    const uint32_t wbest = wave_max_u32(best);
    float f0, f1, f2, f3;
    uint32_t gbest;
    if constexpr (W == 1) {
      // one wave per block: the winner's plane comes out of its lane's registers, nothing goes through LDS
      const unsigned long long wl = __ballot(best == wbest && best != 0);  // exactly one lane: keys are unique
      const int src = wl ? __ffsll((long long)wl) - 1 : 0;
      gbest = wbest;
      f0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(wa), src));
      f1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(wb), src));
      f2 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(wc), src));
      f3 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(wd), src));
      if (!wl) f0 = f1 = f2 = f3 = 0.f;
    } else if (best == wbest && best != 0) {  // exactly one lane: keys are unique
      const int w = tx >> 6;
      s_wbest[par][w] = best;
      s_wplane[par][w][0] = wa;
      s_wplane[par][w][1] = wb;
      s_wplane[par][w][2] = wc;
      s_wplane[par][w][3] = wd;
    } else if (wbest == 0 && (tx & 63) == 0) {
      s_wbest[par][tx >> 6] = 0;  // a wave without hypotheses (H < THREADS)
    }
    // stage the next block.  Its buffer, (t+1) % 3, was last read for block t-2: every wave has
    // passed the barrier of block t-1 since.  The reduction slots alternate by parity for the
    // same reason, so ONE barrier per block publishes both the reduction and the staged points.
    const int nbuf = buf == 2 ? 0 : buf + 1;
    if (pre) {
      s_pts[nbuf][0][tx] = rx;
      s_pts[nbuf][1][tx] = ry;
      s_pts[nbuf][2][tx] = rz;
    }
    if (has_next)
      stage_local<THREADS>(nox, noy, noz, nxt, rx, ry, rz, s_loc[nbuf], s_wext[nbuf], s_wfast[nbuf],
                           W == 1 ? &ext_carry : nullptr, W == 1 ? &fast_carry : nullptr);
    __syncthreads();
    if constexpr (W > 1) {
      gbest = s_wbest[par][0];
      int gw = 0;
#pragma unroll
      for (int w = 1; w < W; ++w) {
        if (s_wbest[par][w] > gbest) {
          gbest = s_wbest[par][w];
          gw = w;
        }
      }
      f0 = s_wplane[par][gw][0]; f1 = s_wplane[par][gw][1]; f2 = s_wplane[par][gw][2]; f3 = s_wplane[par][gw][3];
    }
    if (tx == 0) {
      if (out.plane) {
        out.plane[4 * (int64_t)be + 0] = f0; out.plane[4 * (int64_t)be + 1] = f1;
        out.plane[4 * (int64_t)be + 2] = f2; out.plane[4 * (int64_t)be + 3] = f3;
      }
      if (out.count) out.count[be] = (int32_t)(gbest >> 10) - 1;
      if (out.index) out.index[be] = 1023 - (int)(gbest & 1023u);
    }
    // final mask with the winning f32 plane (cuda_ransac.py:149-155); n <= THREADS - 1
    if ((int)tx < n) {
      const double dist = plane_distance((double)f0, (double)f1, (double)f2, (double)f3,
                                         lx[tx], ly[tx], lz[tx]);
      out.mask[(int64_t)cur.pstart + tx] = (dist < thr) ? 1 : 0;
    }
    if (!has_next) break;
    cur = nxt;
    nxt = nxt2;
    j += 1;
    buf = nbuf;
    par ^= 1;
  }
  RS_COUNT_FLUSH;
}

// A block with more than THREADS-1 points (unsplit voxels - what Grid.map_leaf_points_cuda_ransac sees
// when nobody called subdivide, as in the reference's own test - poses outside the scheme, large K):
// the same screened scoring as k_ransac, the block taken in TILES of THREADS points staged in LDS as
// block-local f32 coordinates.  The bound of the screen needs the extent of the whole block first (one
// extra pass over its points); a hypothesis the screen cannot decide is recounted exactly by its whole
// wavefront over all n points (the per-hypothesis recount is what makes the screen pay here: per group it
// was slower than the plain f64 loop, 4.9 vs 3.0 ms per 10 M points of unsplit voxels).
template <int THREADS, int HPL, int KT>
__device__ __forceinline__ void ransac_block_tiled(const BlockDesc& d, int be,
                                                   const double* __restrict__ xyz,
                                                   const double* __restrict__ hyp, int H, int k_rt,
                                                   double thr, const RansacOut& out, f4* s_loc,
                                                   float* s_wext, unsigned long long* s_best,
                                                   float* s_plane) {
  constexpr int KS = KT > 0 ? KT : RS_KMAX;
  constexpr int W = THREADS / 64;
  const int k = KT > 0 ? KT : k_rt;
  const int n = d.n;
  const int64_t pstart = d.pstart;
  const double* __restrict__ pts = xyz + 3 * pstart;
  const double ox = pts[0], oy = pts[1], oz = pts[2];
  // ---- extent of the block in local coordinates ----------------------------------------------------------
  float m = 0.f;
  for (int i = threadIdx.x; i < n; i += THREADS) {
    const float u = (float)(pts[3 * (int64_t)i] - ox), v = (float)(pts[3 * (int64_t)i + 1] - oy),
                w = (float)(pts[3 * (int64_t)i + 2] - oz);
    float mm = fmaxf(fabsf(u), fmaxf(fabsf(v), fabsf(w)));
    if (!(mm == mm) || u != u || v != v || w != w) mm = __int_as_float(0x7f800000);  // NaN: all ambiguous
    m = fmaxf(m, mm);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0) s_wext[threadIdx.x >> 6] = m;
  __syncthreads();
  float extent = s_wext[0];
#pragma unroll
  for (int w = 1; w < W; ++w) extent = fmaxf(extent, s_wext[w]);
  const double delta_blk = fma(0x1p-23 * 9.0, (double)extent,
                               0x1p-49 * (fabs(ox) + fabs(oy) + fabs(oz) + (double)extent + 1.0));
  const double dprime_thr = 0x1p-21 * (thr * thr);
  const bool blk_sane = thr >= 0x1p-40 && thr <= 0x1p40 && extent < 0x1p60f;
  const float nthr2 = -(float)(thr * thr);
  // ---- the hypotheses' planes (samples gathered from global memory) + screening constants -----------------
  float fa[HPL], fb[HPL], fc[HPL], fd[HPL], sto[HPL], sdl[HPL], margin[HPL];
  int cnt[HPL];
#pragma unroll
  for (int q = 0; q < HPL; ++q) {
    const int t = threadIdx.x + q * THREADS;
    cnt[q] = 0;
    fa[q] = fb[q] = fc[q] = fd[q] = sto[q] = 0.f;
    sdl[q] = 0.f;
    margin[q] = __int_as_float(0x7f800000);
    if (t < H) {
      float pf[4];
      if constexpr (KT < 0) {
        plane_streamed(hyp + (int64_t)t * k, k, d, xyz, pf);
      } else {
        double sx[KS], sy[KS], sz[KS];
#pragma unroll
        for (int i = 0; i < KS; ++i) {
          sx[i] = sy[i] = sz[i] = 0.0;
          if (i < k) {
            const int g = sample_index_exact(hyp[(int64_t)t * k + i], n, d.vstart);
            const int64_t p = (g < n) ? pstart + g : (int64_t)d.pspill;
            sx[i] = xyz[3 * p];
            sy[i] = xyz[3 * p + 1];
            sz[i] = xyz[3 * p + 2];
          }
        }
        plane_from_samples<KT, KS>(sx, sy, sz, k, pf);
      }
      // (explicit fma: error BOUNDS and the screen's own inputs, not parity arithmetic; see k_ransac)
      const double A = (double)pf[0], B = (double)pf[1], Cc = (double)pf[2], D = (double)pf[3];
      const double to = fma(A, ox, fma(B, oy, fma(Cc, oz, D)));
      const double delta = fma(0x1p-21, fabs(to), fma(0x1p-49, fabs(D), delta_blk));
      const double dprime = fma(delta, thr + thr + delta, dprime_thr) * 1.000001;
      const bool sane = blk_sane && (fabs(to) < 0x1p60);
      fa[q] = pf[0]; fb[q] = pf[1]; fc[q] = pf[2]; fd[q] = pf[3];
      sto[q] = (float)to;
      sdl[q] = sane ? (float)dprime : __int_as_float(0x7f800000);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- screened scoring, tile by tile -----------------------------------------------------------------------
  for (int t0 = 0; t0 < n; t0 += THREADS) {
    const int nt = min(THREADS, n - t0);
    if ((int)threadIdx.x < nt) {
      const int64_t i = t0 + threadIdx.x;
      s_loc[threadIdx.x] = f4{(float)(pts[3 * i] - ox), (float)(pts[3 * i + 1] - oy), (float)(pts[3 * i + 2] - oz), 0.f};
    }
    __syncthreads();
    float mt[HPL];
    screen_group<HPL>(s_loc, nt, fa, fb, fc, sto, nthr2, cnt, mt);
#pragma unroll
    for (int q = 0; q < HPL; ++q) margin[q] = fminf(margin[q], mt[q]);
    __syncthreads();
  }
  // ---- exact recount of the hypotheses the screen could not decide (whole wave, one point per lane) -------
#pragma unroll
  for (int q = 0; q < HPL; ++q) {
    const bool redo = (int)threadIdx.x + q * THREADS < H && !(margin[q] > sdl[q]);  // (NaN: recount)
    unsigned long long todo = __ballot(redo);
    while (todo) {
      const int src = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const double A = (double)__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(fa[q]), src));
      const double B = (double)__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(fb[q]), src));
      const double Cc = (double)__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(fc[q]), src));
      const double D = (double)__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(fd[q]), src));
      int c = 0;
      for (int base = 0; base < n; base += 64) {
        const int ii = base + (int)(threadIdx.x & 63u);
        const int64_t jj = ii < n ? ii : 0;
        const bool in = ii < n && plane_distance(A, B, Cc, D, pts[3 * jj], pts[3 * jj + 1], pts[3 * jj + 2]) < thr;
        c += __popcll(__ballot(in));
      }
      if ((int)(threadIdx.x & 63u) == src) cnt[q] = c;
    }
  }
  // ---- maximum, lowest index among the tied (cuda_ransac.py:125-146), final mask -------------------------------
  unsigned long long best = 0;
#pragma unroll
  for (int q = 0; q < HPL; ++q) {
    const int t = threadIdx.x + q * THREADS;
    if (t < H) {
      const unsigned long long key = ((unsigned long long)(unsigned)cnt[q] << 32) | (unsigned)(0x7FFFFFFF - t);
      best = key > best ? key : best;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(best, off);
    best = o > best ? o : best;
  }
  if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = best;
  __syncthreads();
  best = s_best[0];
#pragma unroll
  for (int w = 1; w < W; ++w) best = s_best[w] > best ? s_best[w] : best;
  const int win = 0x7FFFFFFF - (int)(unsigned)(best & 0xFFFFFFFFu);
#pragma unroll
  for (int q = 0; q < HPL; ++q) {
    if ((int)threadIdx.x + q * THREADS == win) {
      s_plane[0] = fa[q]; s_plane[1] = fb[q]; s_plane[2] = fc[q]; s_plane[3] = fd[q];
      if (out.plane) {
        out.plane[4 * (int64_t)be + 0] = fa[q]; out.plane[4 * (int64_t)be + 1] = fb[q];
        out.plane[4 * (int64_t)be + 2] = fc[q]; out.plane[4 * (int64_t)be + 3] = fd[q];
      }
      if (out.count) out.count[be] = cnt[q];
      if (out.index) out.index[be] = win;
    }
  }
  __syncthreads();
  const double a = (double)s_plane[0], bb = (double)s_plane[1], c = (double)s_plane[2], dd = (double)s_plane[3];
  for (int i = threadIdx.x; i < n; i += THREADS)
    out.mask[pstart + i] = (plane_distance(a, bb, c, dd, pts[3 * (int64_t)i], pts[3 * (int64_t)i + 1],
                                           pts[3 * (int64_t)i + 2]) < thr) ? 1 : 0;
}

// Blocks with more than THREADS-1 points (unsplit voxels, poses outside the scheme, large K):
// points stay in global memory (wave-uniform scalar loads in the scoring loop).  The list of such
// batch entries is appended by k_block_desc; its length lives in device memory, so the grid is
// fixed and every workgroup strides over the list.
template <int THREADS, int HPL, int KT>
__global__ __launch_bounds__(THREADS) void k_ransac_big(const double* __restrict__ xyz,
                                                        const BlockDesc* __restrict__ desc,
                                                        const uint32_t* __restrict__ big_list,
                                                        const uint32_t* __restrict__ big_count,
                                                        const double* __restrict__ hyp, int H,
                                                        int k, double thr, RansacOut out) {
  __shared__ unsigned long long s_best[THREADS / 64];
  __shared__ float s_plane[4];
  __shared__ f4 s_loc[THREADS];
  __shared__ float s_wext[THREADS / 64];
  const uint32_t count = *big_count;
  for (uint32_t j = blockIdx.x; j < count; j += gridDim.x) {
    const int be = (int)big_list[j];
    const BlockDesc d = desc[be];
    if constexpr (KT < 0)
      ransac_block_global<THREADS, HPL, KT>(d, be, xyz, hyp, H, k, thr, out, s_best, s_plane);
    else
      ransac_block_tiled<THREADS, HPL, KT>(d, be, xyz, hyp, H, k, thr, out, s_loc, s_wext, s_best, s_plane);
    __syncthreads();
  }
}

// ---- batch preparation ------------------------------------------------------------------------
// scratch counters of one launch (device): [0] big blocks, [1] sorted small blocks,
// [8..8+256) blocks per size, [264..264+256) start of every size in the sorted list,
// [520..520+256) fill cursor per size
enum { RC_BIG = 0, RC_SORTED = 1, RC_BINS = 8, RC_START = 264, RC_FILL = 520, RC_WORDS = 776 };

// sizes in batch order (scanned into virtual starts, cuda_ransac.py:64-66)
__global__ __launch_bounds__(256) void k_block_sizes_in_order(const int32_t* __restrict__ order,
                                                              const int32_t* __restrict__ size,
                                                              int64_t nb,
                                                              uint32_t* __restrict__ tmp_sizes,
                                                              uint32_t* __restrict__ counters) {
  // (the launch counters are first touched by the NEXT kernel: this one clears them on the way)
  if (blockIdx.x == 0)
    for (int w = threadIdx.x; w < RC_WORDS; w += blockDim.x) counters[w] = 0;
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  tmp_sizes[b] = (uint32_t)size[order ? order[b] : b];
}


// batch entry -> descriptor; blocks with n < k are finished right here (the reference's kernel
// returns at once and their mask stays False, cuda_ransac.py:96-97)
constexpr int BD_PER_THREAD = 8;
constexpr int BS_PER_THREAD = 16;  // k_block_scatter: entries per thread (same reason)
__global__ __launch_bounds__(256) void k_block_desc(const int32_t* __restrict__ order,
                                                    const uint32_t* __restrict__ start,
                                                    const int32_t* __restrict__ size,
                                                    const uint32_t* __restrict__ scanned,
                                                    int64_t nb, int64_t n_points, int cap, int k,
                                                    BlockDesc* __restrict__ desc,
                                                    uint32_t* __restrict__ big_list,
                                                    uint32_t* __restrict__ counters, RansacOut out) {
  __shared__ uint32_t bins[256];
  bins[threadIdx.x] = 0;
  __syncthreads();
  // (BD_PER_THREAD entries per thread: the per-size counts leave the workgroup as ONE global atomic per
  //  non-empty size, and same-address atomics serialise - with one entry per thread they were half the kernel)
#pragma unroll 1
  for (int r = 0; r < BD_PER_THREAD; ++r) {
    const int64_t b = ((int64_t)blockIdx.x * BD_PER_THREAD + r) * blockDim.x + threadIdx.x;
    if (b >= nb) break;
    const int32_t phys = order ? order[b] : (int32_t)b;
    BlockDesc d;
    d.pstart = start[phys];
    d.n = size[phys];
    d.vstart = (int64_t)scanned[b];
    uint32_t sp = d.n > 0 ? d.pstart + (uint32_t)d.n - 1u : d.pstart;
    if (b + 1 < nb) {
      sp = start[order ? order[b + 1] : b + 1];
    } else if (!order) {
      // stand-alone operator: the cloud may continue past the last block (cuda_ransac.py:43-81)
      const int64_t e = (int64_t)d.pstart + d.n;
      if (e < n_points) sp = (uint32_t)e;
    }
    d.pspill = sp;
    d.pad[0] = 0;
    d.pad[1] = (uint32_t)b;  // batch entry: index of the per-block outputs
    d.pad[2] = 0;
    desc[b] = d;
    if (d.n < k) {
      for (int i = 0; i < d.n; ++i) out.mask[(int64_t)d.pstart + i] = 0;
      if (out.plane) {
        out.plane[4 * b + 0] = 0.f; out.plane[4 * b + 1] = 0.f;
        out.plane[4 * b + 2] = 0.f; out.plane[4 * b + 3] = 0.f;
      }
      if (out.count) out.count[b] = 0;
      if (out.index) out.index[b] = -1;
    } else if (d.n > cap) {
      big_list[atomicAdd(&counters[RC_BIG], 1u)] = (uint32_t)b;
    } else {
      atomicAdd(&bins[d.n], 1u);
    }
  }
  __syncthreads();
  if (bins[threadIdx.x]) atomicAdd(&counters[RC_BINS + threadIdx.x], bins[threadIdx.x]);
}


// Sizes in batch order, their prefix sums (the blocks' virtual starts, cuda_ransac.py:64-66) and the descriptors in
// ONE launch (round 5; before: k_block_sizes_in_order, a scan, k_block_desc): a workgroup takes 2048 consecutive batch
// entries, eight per thread, and the workgroups' totals are chained by decoupled look-back.  The launch counters it
// adds to were zeroed by the PREVIOUS launch's k_block_scatter (two sets per context, used alternately).
constexpr int BP_PER_THREAD = 8;
__global__ __launch_bounds__(256) void k_block_prepare(const int32_t* __restrict__ order,
                                                       const uint32_t* __restrict__ start,
                                                       const int32_t* __restrict__ size, int64_t nb, int64_t n_points,
                                                       int cap, int k, BlockDesc* __restrict__ desc,
                                                       uint32_t* __restrict__ big_list, uint32_t* __restrict__ counters,
                                                       RansacOut out, uint64_t* __restrict__ status, uint32_t epoch,
                                                       int64_t max_block, uint32_t* __restrict__ mirror) {
  __shared__ uint32_t bins[256];
  __shared__ uint32_t s_wave[4];
  __shared__ uint32_t s_excl;
  bins[threadIdx.x] = 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t b0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * BP_PER_THREAD;
  int32_t phys[BP_PER_THREAD + 1];
  int32_t nn[BP_PER_THREAD];
#pragma unroll
  for (int r = 0; r <= BP_PER_THREAD; ++r) {
    const int64_t b = b0 + r;
    phys[r] = b < nb ? (order ? order[b] : (int32_t)b) : -1;
  }
  uint32_t sum = 0;
#pragma unroll
  for (int r = 0; r < BP_PER_THREAD; ++r) {
    nn[r] = phys[r] >= 0 ? size[phys[r]] : 0;
    sum += (uint32_t)nn[r];
  }
  uint32_t inc = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(inc, off);
    if (lane >= off) inc += t;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  uint32_t pre = inc - sum, total = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (w < wave) pre += s_wave[w];
    total += s_wave[w];
  }
  pre += lookback_exclusive(status, epoch, blockIdx.x, total, &s_excl);
#pragma unroll
  for (int r = 0; r < BP_PER_THREAD; ++r) {
    const int64_t b = b0 + r;
    if (phys[r] < 0) break;
    BlockDesc d;
    d.pstart = start[phys[r]];
    d.n = nn[r];
    d.vstart = (int64_t)pre;
    pre += (uint32_t)nn[r];
    uint32_t sp = d.n > 0 ? d.pstart + (uint32_t)d.n - 1u : d.pstart;
    if (phys[r + 1] >= 0) {
      sp = start[phys[r + 1]];
    } else if (!order) {
      // stand-alone operator: the cloud may continue past the last block (cuda_ransac.py:43-81)
      const int64_t e = (int64_t)d.pstart + d.n;
      if (e < n_points) sp = (uint32_t)e;
    }
    d.pspill = sp;
    d.pad[0] = 0;
    d.pad[1] = (uint32_t)b;  // batch entry: index of the per-block outputs
    d.pad[2] = 0;
    desc[b] = d;
    if (d.n < k) {
      // finished right here: the reference's kernel returns at once, the mask stays False (cuda_ransac.py:96-97)
      for (int i = 0; i < d.n; ++i) out.mask[(int64_t)d.pstart + i] = 0;
      if (out.plane) {
        out.plane[4 * b + 0] = 0.f; out.plane[4 * b + 1] = 0.f;
        out.plane[4 * b + 2] = 0.f; out.plane[4 * b + 3] = 0.f;
      }
      if (out.count) out.count[b] = 0;
      if (out.index) out.index[b] = -1;
    } else if (d.n > cap) {
      big_list[atomicAdd(&counters[RC_BIG], 1u)] = (uint32_t)b;
    } else {
      atomicAdd(&bins[d.n], 1u);
    }
    // the launch was told that no block holds more than max_block points and left out the instances for larger
    // ones: a block that breaks the promise would keep a stale mask - the next host wait reports it instead
    if (d.n >= k && (int64_t)d.n > max_block)
      __hip_atomic_store(&mirror[MIRROR_RS_VIOLATION], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();
  if (bins[threadIdx.x]) atomicAdd(&counters[RC_BINS + threadIdx.x], bins[threadIdx.x]);
}

// Sample positions of every hypothesis for every block size that occurs in the launch (k <= 6, sizes up to 255):
// one 8-byte entry per (size, hypothesis) - positions in bytes 0..5, the risky-draw bits (sample_index_cached) in
// bits 16..21 of the second word.  k_ransac<..., PT = true> reads its entries a batch ahead of the plane fits.
// (Run by the extra workgroups of k_block_scatter: part `part` of the hypotheses, block size n.)
__device__ __forceinline__ void pos_table_entry(const double* __restrict__ hyp, int H, int k,
                                                uint2* __restrict__ tab, int n, int t) {
  const double* __restrict__ row = hyp + (int64_t)t * k;
  uint32_t x = 0, y = 0;
  for (int i = 0; i < k; ++i) {
    bool risky;
    const uint32_t g = (uint32_t)sample_index_cached(row[i], n, &risky);
    if (i < 4) x |= g << (8 * i); else y |= g << (8 * (i - 4));
    // (g == n: fl(R n) rounded up to n - the spill point, outside the block; the exact path re-derives the position
    //  of a flagged draw from the block's start, the prescreen does not vouch for its hypothesis)
    y |= (risky || g == (uint32_t)n) ? (0x10000u << i) : 0u;
  }
  tab[(size_t)n * (size_t)H + t] = uint2{x, y};
}
__device__ __forceinline__ void pos_table_part(const double* __restrict__ hyp, int H, int k,
                                               const uint32_t* __restrict__ counters, uint2* __restrict__ tab,
                                               int n, int part) {
  const int t = part * 256 + (int)threadIdx.x;
  if (t >= H || n < k || counters[RC_BINS + n] == 0u) return;
  pos_table_entry(hyp, H, k, tab, n, t);
}

// descriptors -> size-sorted list.  Position inside a size class: rank inside the workgroup (LDS
// atomics) + ONE global atomic per (workgroup, size) - same-address global atomics serialise.
// The order inside a size class is arbitrary; blocks are independent, results do not depend on it.
// The start of every size class in the list (largest size first) is formed by every workgroup for itself out of
// the per-size counts - 256 numbers - and published by workgroup 0 for the k_ransac instances (RC_START, RC_SORTED;
// RC_FILL was zeroed with the other counters by k_block_sizes_in_order): no kernel of its own.  Workgroups
// [n_scatter, gridDim.x) write the position table (tab != nullptr): no kernel of its own either.
__global__ __launch_bounds__(256) void k_block_scatter(const BlockDesc* __restrict__ desc, int64_t nb,
                                                       int cap, int k,
                                                       uint32_t* __restrict__ counters,
                                                       BlockDesc* __restrict__ sdesc, unsigned n_scatter,
                                                       const double* __restrict__ hyp, int H,
                                                       uint2* __restrict__ tab, uint32_t* __restrict__ next_counters) {
  // (the counter set of the context's NEXT launch: its previous users finished before this launch started)
  if (next_counters && blockIdx.x == gridDim.x - 1)
    for (int w = threadIdx.x; w < RC_WORDS; w += blockDim.x) next_counters[w] = 0;
  if (blockIdx.x >= n_scatter) {
    const unsigned parts = (unsigned)(H + 255) / 256u;
    const unsigned id = blockIdx.x - n_scatter;
    pos_table_part(hyp, H, k, counters, tab, (int)(id / parts), (int)(id % parts));
    return;
  }
  __shared__ uint32_t cnt[256];
  __shared__ uint32_t base[256];
  __shared__ uint32_t c[256];
  cnt[threadIdx.x] = 0;
  c[threadIdx.x] = counters[RC_BINS + threadIdx.x];
  __syncthreads();
  uint32_t first = 0;  // start of size class threadIdx.x
  for (int m = 255; m > (int)threadIdx.x; --m) first += c[m];
  if (blockIdx.x == 0) {
    counters[RC_START + threadIdx.x] = first;
    if (threadIdx.x == 0) counters[RC_SORTED] = first + c[0];
  }
  constexpr int PER_THREAD = BS_PER_THREAD;
  const int64_t b0 = (int64_t)blockIdx.x * (256 * PER_THREAD);
  int nn[PER_THREAD];
  uint32_t rank[PER_THREAD];
#pragma unroll
  for (int r = 0; r < PER_THREAD; ++r) {
    const int64_t b = b0 + r * 256 + threadIdx.x;
    nn[r] = -1;
    rank[r] = 0;
    if (b < nb) {
      const int n = desc[b].n;
      if (n >= k && n <= cap) {
        nn[r] = n;
        rank[r] = atomicAdd(&cnt[n], 1u);
      }
    }
  }
  __syncthreads();
  if (cnt[threadIdx.x]) base[threadIdx.x] = first + atomicAdd(&counters[RC_FILL + threadIdx.x], cnt[threadIdx.x]);
  __syncthreads();
#pragma unroll
  for (int r = 0; r < PER_THREAD; ++r) {
    if (nn[r] >= 0) {
      const int64_t b = b0 + r * 256 + threadIdx.x;
      sdesc[base[nn[r]] + rank[r]] = desc[b];
    }
  }
}

// k_block_prepare and k_block_scatter in ONE launch for a launch of up to BPS_MAX blocks (round 6: a 100 k-point scan
// has ~5 000 leaves and its two preparation kernels were 21 of its 165 us - each a chain of latencies, not work).
// Workgroup 0: sizes in batch order, their prefix sums, the descriptors (registers), the per-size counts and the
// ranks inside a size class (LDS atomics), the starts of the size classes, the sorted list - no look-back, no
// global atomics, no unsorted list in between (only the oversize blocks' descriptors are written there, for
// k_ransac_big).  Workgroups [1, ...): the position table for EVERY size k .. cap - which sizes occur is only known
// to workgroup 0, and a size costs one workgroup.
// (BPS_PER_THREAD blocks per thread, 2 / 4 / 6 / 8 by the size of the launch: the workgroup's time is its threads'
//  serial work - sixteen per thread for launches of up to 16 384 blocks was slower than the two kernels)
constexpr int BPS_THREADS = 1024, BPS_MAX = BPS_THREADS * 8;
template <int BPS_PER_THREAD>
__global__ __launch_bounds__(BPS_THREADS) void k_block_prepare_small(
    const int32_t* __restrict__ order, const uint32_t* __restrict__ start, const int32_t* __restrict__ size,
    int64_t nb, int64_t n_points, int cap, int k, BlockDesc* __restrict__ desc, BlockDesc* __restrict__ sdesc,
    uint32_t* __restrict__ big_list, uint32_t* __restrict__ counters, RansacOut out, int64_t max_block,
    uint32_t* __restrict__ mirror, const double* __restrict__ hyp, int H, uint2* __restrict__ tab,
    uint32_t* __restrict__ next_counters) {
  if (blockIdx.x > 0) {
    const int parts = (H + BPS_THREADS - 1) / BPS_THREADS;
    const int id = (int)blockIdx.x - 1;
    const int n = k + id / parts, t = (id % parts) * BPS_THREADS + (int)threadIdx.x;
    if (t < H && n <= cap) pos_table_entry(hyp, H, k, tab, n, t);
    return;
  }
  __shared__ uint32_t bins[256];
  __shared__ uint32_t first[256];
  __shared__ uint32_t s_wave[BPS_THREADS / 64];
  // (the counter set of the context's NEXT launch: its previous users finished before this launch started)
  if (next_counters)
    for (int w = threadIdx.x; w < RC_WORDS; w += BPS_THREADS) next_counters[w] = 0;
  if (threadIdx.x < 256) bins[threadIdx.x] = 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t b0 = (int64_t)threadIdx.x * BPS_PER_THREAD;
  int32_t phys[BPS_PER_THREAD + 1];
  int32_t nn[BPS_PER_THREAD];
#pragma unroll
  for (int r = 0; r <= BPS_PER_THREAD; ++r) {
    const int64_t b = b0 + r;
    phys[r] = b < nb ? (order ? order[b] : (int32_t)b) : -1;
  }
  uint32_t sum = 0;
#pragma unroll
  for (int r = 0; r < BPS_PER_THREAD; ++r) {
    nn[r] = phys[r] >= 0 ? size[phys[r]] : 0;
    sum += (uint32_t)nn[r];
  }
  uint32_t pst[BPS_PER_THREAD + 1];
#pragma unroll
  for (int r = 0; r <= BPS_PER_THREAD; ++r) pst[r] = phys[r] >= 0 ? start[phys[r]] : 0u;
  uint32_t inc = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(inc, off);
    if (lane >= off) inc += t;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  uint32_t pre = inc - sum;
  for (int w = 0; w < wave; ++w) pre += s_wave[w];
  BlockDesc dd[BPS_PER_THREAD];
  uint32_t rank[BPS_PER_THREAD];
#pragma unroll
  for (int r = 0; r < BPS_PER_THREAD; ++r) {
    const int64_t b = b0 + r;
    BlockDesc& d = dd[r];
    d.n = -1;
    rank[r] = 0;
    if (phys[r] < 0) continue;
    d.pstart = pst[r];
    d.n = nn[r];
    d.vstart = (int64_t)pre;
    pre += (uint32_t)nn[r];
    uint32_t sp = d.n > 0 ? d.pstart + (uint32_t)d.n - 1u : d.pstart;
    if (phys[r + 1] >= 0) {
      sp = pst[r + 1];
    } else if (!order) {
      // stand-alone operator: the cloud may continue past the last block (cuda_ransac.py:43-81)
      const int64_t e = (int64_t)d.pstart + d.n;
      if (e < n_points) sp = (uint32_t)e;
    }
    d.pspill = sp;
    d.pad[0] = 0;
    d.pad[1] = (uint32_t)b;  // batch entry: index of the per-block outputs
    d.pad[2] = 0;
    const int n = d.n;
    d.n = -1;                // (-1 from here on: not a member of the sorted list)
    if (n < k) {
      // finished right here: the reference's kernel returns at once, the mask stays False (cuda_ransac.py:96-97)
      for (int i = 0; i < n; ++i) out.mask[(int64_t)d.pstart + i] = 0;
      if (out.plane) {
        out.plane[4 * b + 0] = 0.f; out.plane[4 * b + 1] = 0.f;
        out.plane[4 * b + 2] = 0.f; out.plane[4 * b + 3] = 0.f;
      }
      if (out.count) out.count[b] = 0;
      if (out.index) out.index[b] = -1;
    } else if (n > cap) {
      d.n = n;
      desc[b] = d;
      d.n = -1;
      big_list[atomicAdd(&counters[RC_BIG], 1u)] = (uint32_t)b;
    } else {
      d.n = n;
      rank[r] = atomicAdd(&bins[n], 1u);
    }
    if (n >= k && (int64_t)n > max_block)
      __hip_atomic_store(&mirror[MIRROR_RS_VIOLATION], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();
  // start of size class m in the list, largest size first: an exclusive scan over the sizes in descending order
  // (thread t < 256 holds size 255 - t)
  uint32_t cm = 0, incm = 0;
  if (threadIdx.x < 256) {
    cm = bins[255 - threadIdx.x];
    incm = wave_inclusive_add(cm);
    if (lane == 63) s_wave[wave] = incm;
  }
  __syncthreads();
  if (threadIdx.x < 256) {
    uint32_t f = incm - cm;
    for (int w = 0; w < wave; ++w) f += s_wave[w];
    const int m = 255 - (int)threadIdx.x;
    first[m] = f;
    counters[RC_BINS + m] = cm;
    counters[RC_START + m] = f;
    if (m == 0) counters[RC_SORTED] = f + cm;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < BPS_PER_THREAD; ++r)
    if (dd[r].n >= 0) sdesc[first[dd[r].n] + rank[r]] = dd[r];
}

}  // namespace

int ransac_check_table(octl_ctx* ctx, const double* hyp, int32_t H, int32_t k) {
  const int64_t n = (int64_t)H * k;
  for (int64_t i = 0; i < n; ++i)
    if (!(hyp[i] >= 0.0 && hyp[i] < 1.0))  // also false for NaN
      return octl_set_error(ctx, OCTL_E_INVALID,
                            "hypothesis table entry %lld = %g is outside [0, 1): the table must hold "
                            "np.random.random draws (cuda_ransac.py:39-41)", (long long)i, hyp[i]);
  return OCTL_OK;
}

// Launch the kernels over nb batch entries.  `order` (device, nullable) maps batch entry ->
// physical block.  Descriptors (virtual start, spill point) are derived on the device from the
// sizes in batch order; blocks are then processed largest first.
int ransac_launch(octl_ctx* ctx, const double* xyz_dev, int64_t n_points,
                  const uint32_t* blk_start, const int32_t* blk_size, const int32_t* order_dev,
                  int64_t nb, const double* hyp_dev, int32_t H, int32_t k, double thr,
                  uint8_t* mask_dev, float* plane_dev, int32_t* count_dev, int32_t* index_dev,
                  uint8_t* evaluated_dev, DevBuf& scratch, int64_t max_block) {
  (void)evaluated_dev;
  if (nb <= 0) return OCTL_OK;
  if (H < 1 || H > 1024) return octl_set_error(ctx, OCTL_E_INVALID, "H must be in [1, 1024]");
  if (k < 1) return octl_set_error(ctx, OCTL_E_INVALID, "initial_points_number must be positive");
  // more sample points than the register path holds: every block takes the exact path with a streamed fit
  const bool any_k = k > RS_KMAX;
  if (nb >= ((int64_t)1 << 31)) return octl_set_error(ctx, OCTL_E_INVALID, "too many blocks");
  hipStream_t st = ctx->stream;
  // scratch: [sizes/scanned u32 nb+8 | desc 32 B x nb | sorted desc 32 B x nb | big list u32 nb |
  //           counters]
  const size_t off_d = (((size_t)nb + 8) * 4 + 31) & ~(size_t)31;
  const size_t off_s = off_d + (size_t)nb * sizeof(BlockDesc);
  const size_t off_b = off_s + (size_t)nb * sizeof(BlockDesc);
  const size_t off_c = (off_b + (size_t)nb * 4 + 31) & ~(size_t)31;
  // (position table of the instances with compile-time k = 3 .. 6: 256 sizes x H entries of 8 bytes)
  const bool use_tab = !any_k && k >= 3 && k <= 6 && (H > 256 || k == 6);
  const size_t off_t = (off_c + RC_WORDS * 4 + 31) & ~(size_t)31;
  OCTL_TRY(devbuf_reserve(ctx, scratch, off_t + (use_tab ? (size_t)256 * (size_t)H * sizeof(uint2) : 0)));
  char* base = static_cast<char*>(scratch.p);
  uint32_t* tmp = scratch.as<uint32_t>();
  BlockDesc* desc = reinterpret_cast<BlockDesc*>(base + off_d);
  BlockDesc* sdesc = reinterpret_cast<BlockDesc*>(base + off_s);
  uint32_t* big_list = reinterpret_cast<uint32_t*>(base + off_b);
  uint32_t* counters = reinterpret_cast<uint32_t*>(base + off_c);
  // fused preparation: the launch counters live in the CONTEXT, two sets used alternately - a launch's
  // k_block_scatter zeroes the set of the next one, so that no launch starts with a clearing kernel
  const bool fused = !ctx->opt.no_fused_tables;
  uint32_t* next_counters = nullptr;
  if (fused) {
    if (!ctx->rs_counters.p || ctx->rs_counters_dirty) {
      // (first launch of the context, or a launch that failed between its two preparation kernels)
      OCTL_TRY(devbuf_reserve(ctx, ctx->rs_counters, (size_t)2 * RC_WORDS * 4));
      HIP_TRY(ctx, hipMemsetAsync(ctx->rs_counters.p, 0, (size_t)2 * RC_WORDS * 4, st));
      ctx->rs_parity = 0;
    }
    counters = ctx->rs_counters.as<uint32_t>() + (size_t)ctx->rs_parity * RC_WORDS;
    next_counters = ctx->rs_counters.as<uint32_t>() + (size_t)(ctx->rs_parity ^ 1) * RC_WORDS;
    ctx->rs_parity ^= 1;
    ctx->rs_counters_dirty = true;   // until this launch's k_block_scatter (which zeroes the other set) is enqueued
  }
  uint2* pos_tab = use_tab ? reinterpret_cast<uint2*>(base + off_t) : nullptr;
  const int threads = (H <= 64) ? 64 : (H <= 256 ? 256 : RS_BIG_THREADS);
  RansacOut out{mask_dev, plane_dev, count_dev, index_dev};
  if (fused && nb <= BPS_MAX) {
    KTimer t(ctx, "ransac_prepare");
    const int cap = any_k ? 0 : threads - 1;
    const unsigned n_table = use_tab && cap >= k ? (unsigned)(cap - k + 1) * (unsigned)ceil_div(H, BPS_THREADS) : 0u;
    auto kp = nb <= 2 * BPS_THREADS ? k_block_prepare_small<2>
              : (nb <= 4 * BPS_THREADS ? k_block_prepare_small<4>
                                       : (nb <= 6 * BPS_THREADS ? k_block_prepare_small<6> : k_block_prepare_small<8>));
    OCTL_LAUNCH(kp, dim3(1 + n_table), dim3(BPS_THREADS), 0, st, order_dev, blk_start, blk_size, nb,
                       n_points, cap, (int)k, desc, sdesc, big_list, counters, out, max_block,
                       static_cast<uint32_t*>(ctx->small_host), hyp_dev, (int)H, pos_tab, next_counters);
    HIP_TRY(ctx, hipGetLastError());
    ctx->rs_counters_dirty = false;
  } else {
    KTimer t(ctx, "ransac_prepare");
    if (fused) {
      const unsigned g = (unsigned)ceil_div(nb, 256 * BP_PER_THREAD);
      uint64_t* status = nullptr;
      uint32_t epoch = 0;
      OCTL_TRY(octl_scan_status_acquire(ctx, g, &status, &epoch));
      OCTL_LAUNCH(k_block_prepare, dim3(g), dim3(256), 0, st, order_dev, blk_start, blk_size, nb, n_points,
                         any_k ? 0 : threads - 1, (int)k, desc, big_list, counters, out, status, epoch, max_block,
                         static_cast<uint32_t*>(ctx->small_host));
      HIP_TRY(ctx, hipGetLastError());
    } else {
      const unsigned g = (unsigned)ceil_div(nb, 256);
      OCTL_LAUNCH(k_block_sizes_in_order, dim3(g), dim3(256), 0, st, order_dev, blk_size, nb, tmp, counters);
      HIP_TRY(ctx, hipGetLastError());
      OCTL_TRY(octl_exclusive_scan_u32(ctx, tmp, tmp, nb, nullptr));
      OCTL_LAUNCH(k_block_desc, dim3((unsigned)ceil_div(nb, 256 * BD_PER_THREAD)), dim3(256), 0, st, order_dev,
                         blk_start, blk_size,
                         (const uint32_t*)tmp, nb, n_points, any_k ? 0 : threads - 1, (int)k, desc, big_list,
                         counters, out);
      HIP_TRY(ctx, hipGetLastError());
    }
    // (size-class starts and the position table are made by k_block_scatter's own workgroups)
    const unsigned n_scatter = (unsigned)ceil_div(nb, 256 * BS_PER_THREAD);
    const unsigned n_table = use_tab ? 256u * (unsigned)ceil_div(H, 256) : 0u;
    OCTL_LAUNCH(k_block_scatter, dim3(n_scatter + n_table), dim3(256), 0, st,
                       (const BlockDesc*)desc, nb, any_k ? 0 : threads - 1, (int)k, counters, sdesc, n_scatter, hyp_dev,
                       (int)H, pos_tab, next_counters);
    HIP_TRY(ctx, hipGetLastError());
    ctx->rs_counters_dirty = false;
  }
  const int cus = octl_ctx_cus(ctx);
  KTimer t(ctx, "ransac");
  // The instances of the split launch work on disjoint parts of the list and write disjoint outputs: the two for
  // the larger blocks go to the context's side stream, NEXT TO the one-wave instance on the context's stream, which
  // waits for them at the end (on the benchmark scene the 128-lane instance has the few leaves of exactly 64 points:
  // 76 us one after the other, nothing beside the main kernel).
  hipStream_t side = st;
  bool on_side = false;
  // waves per block of the H > 256 instances: 1 / 2 / 4 by size class (0), or every block under 128 points on two
  // waves (2), every block under 256 on four (4) - see the launch policy below.  OCTL_RANSAC_WAVES forces one.
  int waves_per_block = (int)ctx->opt.ransac_waves;
  if (waves_per_block == 0 && !any_k && H > 256) {
    // Launch policy.  One wave per block is the cheapest form per block (no cross-wave reduction, no barrier) and wins
    // when there are many blocks per wave slot; a launch with few blocks is bound by the LATENCY of a block and by
    // how evenly the blocks fall on the SIMDs: two or four waves per block shorten both.  Measured on dense planar
    // scans (one cloud, whole step): 1 140 blocks 0.160 / 0.144 / 0.139 ms with 1 / 2 / 4 waves, 5 269 blocks 0.191 /
    // 0.182 / 0.182, 16 644 blocks 0.266 / 0.262 / 0.272, 54 432 blocks 0.544 / 0.556 / 0.575.
    const int64_t slots = (int64_t)cus * 16;   // resident one-wave workgroups
    if (nb * 2 <= slots) waves_per_block = 4;
    else if (nb <= 6 * slots) waves_per_block = 2;
  }
  // (max_block: no block of the launch holds more points - the instances for larger blocks are not launched at all)
  const bool need_small = max_block >= RS_TINY_THREADS, need_mid = max_block >= RS_SMALL_THREADS,
             need_big = any_k || max_block >= threads;
  // (the side stream is gated in and out with events: only when one of the instances below will go there)
  const bool side_work = waves_per_block == 4 ? need_big
                         : (waves_per_block == 2 ? (need_mid || need_big) : (need_small || need_mid || need_big));
  if (!any_k && H > 256 && side_work && octl_ctx_side_stream(ctx) &&
      hipEventRecord(ctx->self_gate, st) == hipSuccess &&
      hipStreamWaitEvent(ctx->self_stream, ctx->self_gate, 0) == hipSuccess) {
    side = ctx->self_stream;
    on_side = true;
  }
  const int no_prescreen = ctx->opt.no_ransac_prescreen ? 1 : 0;
  // (LO, HI: device words holding the launch's part [lo, hi) of the size-sorted list; LO nullptr = from the front)
#define OCTL_RANSAC_RANGE(THREADS, HPL, KT, PER_CU, LO, HI, PT, ST)                              \
  do {                                                                                           \
    const unsigned g = (unsigned)std::min<int64_t>(nb, (int64_t)cus * (PER_CU));                 \
    if (H == (THREADS) * (HPL))                                                                  \
      OCTL_LAUNCH((k_ransac<THREADS, HPL, KT, true, PT>), dim3(g), dim3(THREADS), 0, ST,         \
                         xyz_dev, (const BlockDesc*)sdesc, (const uint32_t*)(LO), (const uint32_t*)(HI), \
                         hyp_dev, H, k, thr, out, (const uint2*)pos_tab, no_prescreen);           \
    else                                                                                         \
      OCTL_LAUNCH((k_ransac<THREADS, HPL, KT, false, PT>), dim3(g), dim3(THREADS), 0, ST,        \
                         xyz_dev, (const BlockDesc*)sdesc, (const uint32_t*)(LO), (const uint32_t*)(HI), \
                         hyp_dev, H, k, thr, out, (const uint2*)pos_tab, no_prescreen);           \
  } while (0)
  // one instance over the whole sorted list (H <= 256: one hypothesis per lane; other sample sizes)
#define OCTL_RANSAC_LAUNCH(THREADS, HPL, KT, PER_CU, PT) \
  OCTL_RANSAC_RANGE(THREADS, HPL, KT, PER_CU, nullptr, counters + RC_SORTED, PT, st)
  // H > 256: a block of n points is worked on by one workgroup, and every wave of it pays the per-block work
  // (staging, reduction, barrier, the winner's mask) whatever n is.  Three instances therefore share the size-sorted
  // list, split at the device-side starts of size classes RS_SMALL_THREADS - 1 and RS_TINY_THREADS - 1:
  //   n < RS_TINY_THREADS (64)     ONE wave x 16 hypotheses per lane - the bulk: a leaf has at most K points
  //   n < RS_SMALL_THREADS (128)   two waves x 8
  //   n < RS_BIG_THREADS (256)     four waves x 4
  // (four -> two waves per block: -8 % on the benchmark scene; one wave: -3 % more; a launch with few blocks puts
  //  every block on two or four waves: waves_per_block above.)
#define OCTL_RANSAC_SPLIT(KT)                                                                                    \
  do {                                                                                                           \
    if (waves_per_block == 4) {                                                                                  \
      OCTL_RANSAC_RANGE(RS_BIG_THREADS, RS_BIG_HPL, KT, RS_PER_CU, nullptr, counters + RC_SORTED, true, st);     \
    } else if (waves_per_block == 2) {                                                                           \
      if (need_mid)                                                                                              \
        OCTL_RANSAC_RANGE(RS_BIG_THREADS, RS_BIG_HPL, KT, RS_PER_CU, nullptr,                                    \
                          counters + RC_START + RS_SMALL_THREADS - 1, true, side);                               \
      HIP_TRY(ctx, hipGetLastError());                                                                           \
      OCTL_RANSAC_RANGE(RS_SMALL_THREADS, (1024 / RS_SMALL_THREADS), KT, RS_SMALL_PER_CU,                        \
                        counters + RC_START + RS_SMALL_THREADS - 1, counters + RC_SORTED, true, st);             \
    } else {                                                                                                     \
      if (need_mid)                                                                                              \
        OCTL_RANSAC_RANGE(RS_BIG_THREADS, RS_BIG_HPL, KT, RS_PER_CU, nullptr,                                    \
                          counters + RC_START + RS_SMALL_THREADS - 1, true, side);                               \
      HIP_TRY(ctx, hipGetLastError());                                                                           \
      if (need_small)                                                                                            \
        OCTL_RANSAC_RANGE(RS_SMALL_THREADS, (1024 / RS_SMALL_THREADS), KT, RS_SMALL_PER_CU,                      \
                          counters + RC_START + RS_SMALL_THREADS - 1, counters + RC_START + RS_TINY_THREADS - 1, \
                          true, side);                                                                           \
      HIP_TRY(ctx, hipGetLastError());                                                                           \
      OCTL_RANSAC_RANGE(RS_TINY_THREADS, (1024 / RS_TINY_THREADS), KT, RS_TINY_PER_CU,                           \
                        counters + RC_START + RS_TINY_THREADS - 1, counters + RC_SORTED, true, st);              \
    }                                                                                                            \
  } while (0)
  if (any_k) {
    // (nothing was put on the sorted list)
  } else if (H <= 64) {
    if (k == 6) OCTL_RANSAC_LAUNCH(64, 1, 6, 16, true); else OCTL_RANSAC_LAUNCH(64, 1, 0, 8, false);
  } else if (H <= 256) {
    if (k == 6) OCTL_RANSAC_LAUNCH(256, 1, 6, 8, true); else OCTL_RANSAC_LAUNCH(256, 1, 0, 4, false);
  } else {
    // the small sample sizes get compile-time k (sample arrays in registers; the generic
    // instantiation indexes them at run time)
    if (k == 6) OCTL_RANSAC_SPLIT(6);
    else if (k == 5) OCTL_RANSAC_SPLIT(5);
    else if (k == 4) OCTL_RANSAC_SPLIT(4);
    else if (k == 3) OCTL_RANSAC_SPLIT(3);
    else OCTL_RANSAC_LAUNCH(RS_BIG_THREADS, RS_BIG_HPL, 0, 2, false);
  }
#undef OCTL_RANSAC_SPLIT
#undef OCTL_RANSAC_LAUNCH
#undef OCTL_RANSAC_RANGE
  HIP_TRY(ctx, hipGetLastError());
  // the (rare) blocks that do not fit the LDS staging; the grid is fixed, the count is on the device
#define OCTL_RANSAC_BIG(THREADS, HPL, KT)                                                       \
  OCTL_LAUNCH((k_ransac_big<THREADS, HPL, KT>), dim3((unsigned)std::min<int64_t>(nb, (int64_t)RS_BIG_PER_CU * cus)), \
                     dim3(THREADS), 0, side, xyz_dev, (const BlockDesc*)desc,                   \
                     (const uint32_t*)big_list, (const uint32_t*)(counters + RC_BIG), hyp_dev,  \
                     H, k, thr, out)
  if (!need_big) {
    // (no block of the launch can be that large)
  } else if (any_k) {
    if (H <= 64) OCTL_RANSAC_BIG(64, 1, -1);
    else if (H <= 256) OCTL_RANSAC_BIG(256, 1, -1);
    else OCTL_RANSAC_BIG(RS_BIG_THREADS, RS_BIG_HPL, -1);
  } else if (H <= 64) {
    if (k == 6) OCTL_RANSAC_BIG(64, 1, 6); else OCTL_RANSAC_BIG(64, 1, 0);
  } else if (H <= 256) {
    if (k == 6) OCTL_RANSAC_BIG(256, 1, 6); else OCTL_RANSAC_BIG(256, 1, 0);
  } else {
    if (k == 6) OCTL_RANSAC_BIG(RS_BIG_THREADS, RS_BIG_HPL, 6); else OCTL_RANSAC_BIG(RS_BIG_THREADS, RS_BIG_HPL, 0);
  }
#undef OCTL_RANSAC_BIG
  HIP_TRY(ctx, hipGetLastError());
  if (on_side) {
    HIP_TRY(ctx, hipEventRecord(ctx->self_done, side));
    HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->self_done, 0));
  }
  return OCTL_OK;
}

// ---- test hook: the two division shortcuts of the plane fit against the host's IEEE division --
namespace {
__global__ __launch_bounds__(256) void k_debug_plane_arith(const double* __restrict__ num3,
                                                            const double* __restrict__ den,
                                                            const double* __restrict__ c, int kdiv,
                                                            int64_t n, double* __restrict__ q3,
                                                            double* __restrict__ ck,
                                                            double* __restrict__ sq, int certified) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a = num3[3 * i], b = num3[3 * i + 1], cc = num3[3 * i + 2];
  double c0 = c[i], c1 = c[i + 1 < n ? i + 1 : 0], c2 = c[i + 2 < n ? i + 2 : 0];
  if (certified) {
    // the guard-free forms a block with the range certificate (coord_in_fast_range) runs
    div3_by_norm_inrange(a, b, cc, den[i]);
    div3_by_small_int(c0, c1, c2, kdiv, true);
    ck[i] = c0;
    sq[i] = sqrt_rn_inrange(c[i]);
  } else {
    div3_by_norm(a, b, cc, den[i]);
    // the 3-wide form the plane fit uses, with neighbours as the other two lanes of the predicate
    div3_by_small_int(c0, c1, c2, kdiv);
    ck[i] = (i & 1) ? c0 : div_by_small_int(c[i], kdiv);
    sq[i] = sqrt_rn_guarded(c[i]);
  }
  q3[3 * i] = a;
  q3[3 * i + 1] = b;
  q3[3 * i + 2] = cc;
}
}  // namespace

// q3[i] = num3[i] / den[i] (three numerators per divisor, den > 0), ck[i] = c[i] / kdiv and
// sq[i] = sqrt(c[i]) as the plane fit computes them; host arrays in, host arrays out
// (tests/test_gpu_primitives.py)
static int debug_plane_arith(octl_ctx* ctx, const double* num3, const double* den, const double* c,
                             int32_t kdiv, int64_t n, double* q3, double* ck, double* sq, int certified);
extern "C" int octl_debug_plane_arith(octl_ctx* ctx, const double* num3, const double* den,
                                      const double* c, int32_t kdiv, int64_t n, double* q3,
                                      double* ck, double* sq) {
  return debug_plane_arith(ctx, num3, den, c, kdiv, n, q3, ck, sq, 0);
}
// the same three operations WITHOUT their range guards, as the plane fit runs them on a block that holds the
// range certificate (coord_in_fast_range): the caller keeps the operands inside the certified ranges
// (den in [2^-200, 2^138], num3 zero or >= 2^-552 and <= den, c zero (+0.0) or in [2^-82, 2^35) for the
// division by kdiv, c in [2^-400, 2^276] for the square root)
extern "C" int octl_debug_plane_arith_certified(octl_ctx* ctx, const double* num3, const double* den,
                                                const double* c, int32_t kdiv, int64_t n, double* q3,
                                                double* ck, double* sq) {
  return debug_plane_arith(ctx, num3, den, c, kdiv, n, q3, ck, sq, 1);
}
static int debug_plane_arith(octl_ctx* ctx, const double* num3, const double* den, const double* c,
                             int32_t kdiv, int64_t n, double* q3, double* ck, double* sq, int certified) {
  if (!ctx || n < 0 || kdiv < 1 || kdiv > 16 ||
      (n > 0 && (!num3 || !den || !c || !q3 || !ck || !sq)))
    return OCTL_E_INVALID;
  if (n == 0) return OCTL_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  DevBuf buf;
  OCTL_TRY(devbuf_reserve(ctx, buf, (size_t)n * 10 * 8));
  double* d_num = buf.as<double>();
  double* d_den = d_num + 3 * n;
  double* d_c = d_den + n;
  double* d_q = d_c + n;
  double* d_ck = d_q + 3 * n;
  double* d_sq = d_ck + n;
  int rc = OCTL_OK;
  if (hipMemcpyAsync(d_num, num3, (size_t)n * 24, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(d_den, den, (size_t)n * 8, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(d_c, c, (size_t)n * 8, hipMemcpyHostToDevice, st) != hipSuccess)
    rc = OCTL_E_HIP;
  if (rc == OCTL_OK) {
    OCTL_LAUNCH(k_debug_plane_arith, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st,
                       (const double*)d_num, (const double*)d_den, (const double*)d_c, (int)kdiv, n,
                       d_q, d_ck, d_sq, certified);
    if (hipGetLastError() != hipSuccess ||
        hipMemcpyAsync(q3, d_q, (size_t)n * 24, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(ck, d_ck, (size_t)n * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(sq, d_sq, (size_t)n * 8, hipMemcpyDeviceToHost, st) != hipSuccess)
      rc = OCTL_E_HIP;
  }
  if (hipStreamSynchronize(st) != hipSuccess) rc = OCTL_E_HIP;
  devbuf_free(buf);
  return rc == OCTL_OK ? OCTL_OK : octl_set_error(ctx, rc, "octl_debug_plane_arith failed");
}

#ifdef RS_COUNTS
extern "C" int octl_debug_rs_stamps(octl_ctx* ctx, unsigned long long out[16], int reset) {
  if (!ctx || !out) return OCTL_E_INVALID;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rs_stamps), sizeof(unsigned long long) * 16));
  if (reset) {
    unsigned long long z[16] = {0};
    HIP_TRY(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_rs_stamps), z, sizeof(z)));
  }
  return OCTL_OK;
}
#endif
