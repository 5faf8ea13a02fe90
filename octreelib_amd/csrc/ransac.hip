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
// (HPL per lane), the block's points broadcast to all lanes through wave-uniform loads.  The
// scoring loop is FP64-VALU bound (3 mul + 3 add + compare + count per point x hypothesis);
// MFMA is not used: the f64 evaluation order (and the f32-rounded plane) must be reproduced
// exactly, and the product is 4 deep.
#include <algorithm>
#include <cstdlib>

#include "forest.h"

namespace {

// c / k, correctly rounded, for an integer 1 <= k <= 16 (see plane_from_samples).
// zh = RN(1/k), zl = RN(1/k - zh); denormal / overflow ranges fall back to the true division.
__device__ __forceinline__ double div_by_small_int(double c, int k) {
  const double kd = (double)k;
  const double zh = 1.0 / kd;                    // compile-time constant when k is
  const double zl = fma(-kd, zh, 1.0) / kd;      // (1 - k*zh) is exact; zl to 2^-53 relative
  const double a = fabs(c);
  if (!(a > 1e-290 && a < 1e290)) return c / kd;  // also NaN / inf / 0
  return fma(c, zh, c * zl);
}

// util.py:27-84 on k sampled points; returns the plane already rounded to f32
// (cuda_ransac.py:110-113).  KT > 0: compile-time k (arrays stay in registers).
template <int KT, int KMAX>
__device__ __forceinline__ void plane_from_samples(const double (&sx)[KMAX],
                                                   const double (&sy)[KMAX],
                                                   const double (&sz)[KMAX], int k_rt,
                                                   float (&plane)[4]) {
  const int k = KT > 0 ? KT : k_rt;
  double cx = 0.0, cy = 0.0, cz = 0.0;
#pragma unroll
  for (int i = 0; i < (KT > 0 ? KT : KMAX); ++i) {  // util.py:37-40
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
  cx = div_by_small_int(cx, k);
  cy = div_by_small_int(cy, k);
  cz = div_by_small_int(cz, k);
  double xx = 0.0, xy = 0.0, xz = 0.0, yy = 0.0, yz = 0.0, zz = 0.0;
#pragma unroll
  for (int i = 0; i < (KT > 0 ? KT : KMAX); ++i) {  // util.py:48-57
    if (i < k) {
      const double rx = sx[i] - cx;
      const double ry = sy[i] - cy;
      const double rz = sz[i] - cz;
      xx += rx * rx;
      xy += rx * ry;
      xz += rx * rz;
      yy += ry * ry;
      yz += ry * rz;
      zz += rz * rz;
    }
  }
  const double det_x = yy * zz - yz * yz;  // util.py:59-61
  const double det_y = xx * zz - xz * xz;
  const double det_z = xx * yy - xy * xy;
  double ax, ay, az;
  if (det_x > det_y && det_x > det_z) {  // util.py:63-74
    ax = det_x;
    ay = xz * yz - xy * zz;
    az = xy * yz - xz * yy;
  } else if (det_y > det_z) {
    ax = xz * yz - xy * zz;
    ay = det_y;
    az = xy * xz - yz * xx;
  } else {
    ax = xy * yz - xz * yy;
    ay = xy * xz - yz * xx;
    az = det_z;
  }
  const double norm = __dsqrt_rn(ax * ax + ay * ay + az * az);  // util.py:76
  if (norm == 0.0) {                                            // util.py:77-78
    plane[0] = plane[1] = plane[2] = plane[3] = 0.0f;
    return;
  }
  ax /= norm;
  ay /= norm;
  az /= norm;
  const double d = -(ax * cx + ay * cy + az * cz);  // util.py:83
  plane[0] = (float)ax;
  plane[1] = (float)ay;
  plane[2] = (float)az;
  plane[3] = (float)d;
}

// util.py:22-24 with the f32 plane promoted to f64: ((a*x + b*y) + c*z) + d
__device__ __forceinline__ double plane_distance(double a, double b, double c, double d, double x,
                                                 double y, double z) {
  return fabs(((a * x + b * y) + c * z) + d);
}

#ifndef RS_MINWAVES
#define RS_MINWAVES 4
#endif
#ifndef RS_SCHED_BARRIER
#define RS_SCHED_BARRIER 1
#endif
#ifndef RS_SCREEN
#define RS_SCREEN 0  // measured: exact but not faster yet (register pressure); see DESIGN.md
#endif
#ifndef RS_SCORE_UNROLL
#define RS_SCORE_UNROLL 4
#endif
#define RS_PRAGMA_(x) _Pragma(#x)
#define RS_PRAGMA(x) RS_PRAGMA_(x)
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

// The k sampled points of hypothesis t of a block -> f32 plane (cuda_ransac.py:100-113)
template <int KT, bool IN_LDS>
__device__ __forceinline__ void fit_hypothesis(const double* __restrict__ hyp, uint32_t row_off,
                                               int k_rt, int n,
                                               int64_t vstart, const double* __restrict__ lx,
                                               const double* __restrict__ ly,
                                               const double* __restrict__ lz,
                                               const double* __restrict__ xyz, int64_t pstart,
                                               int64_t pspill, float (&pf)[4]) {
  constexpr int KS = KT > 0 ? KT : RS_KMAX;
  const int k = KT > 0 ? KT : k_rt;
  double r[KS];
  // uniform base + 32-bit per-lane byte offset: one VGPR of addressing instead of a 64-bit
  // pointer per hypothesis (which the register allocator spilled)
  const double* __restrict__ hyp_row =
      reinterpret_cast<const double*>(reinterpret_cast<const char*>(hyp) + row_off);
  if (KT > 0 && (KT % 2) == 0) {
    // rows of an even number of doubles are 16-byte aligned: dwordx4 loads
    const double2* __restrict__ h2 = reinterpret_cast<const double2*>(hyp_row);
#pragma unroll
    for (int i = 0; i < KS / 2; ++i) {
      const double2 v = h2[i];
      r[2 * i] = v.x;
      r[2 * i + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int i = 0; i < KS; ++i) r[i] = (i < k) ? hyp_row[i] : 0.0;
  }
  int g[KS];
  const double dn = (double)n, dv = (double)vstart;
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    // initial_point_indices[i] = nb.int32(random_hypotheses[t][i] * block_size + block_start)
    // (cuda_ransac.py:103-107): f64 multiply, f64 add, truncation
    const double v = r[i] * dn + dv;
    const int gi = (int)((int64_t)(int)v - vstart);  // position inside the block; may be == n
    g[i] = gi < n ? gi : n;
  }
  double sx[KS], sy[KS], sz[KS];
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    sx[i] = sy[i] = sz[i] = 0.0;
    if (i < k) {
      if (IN_LDS) {
        sx[i] = lx[g[i]];
        sy[i] = ly[g[i]];
        sz[i] = lz[g[i]];
      } else {
        const int64_t p = (g[i] < n) ? pstart + g[i] : pspill;
        sx[i] = xyz[3 * p];
        sy[i] = xyz[3 * p + 1];
        sz[i] = xyz[3 * p + 2];
      }
    }
  }
  plane_from_samples<KT, KS>(sx, sy, sz, k, pf);
}

// One block (leaf x pose) on one workgroup: THREADS lanes, HPL hypotheses per lane (lane t owns
// hypotheses t, t+THREADS, ...).  IN_LDS: the block's points (and the spill point at index n)
// are in the LDS arrays lx/ly/lz; otherwise they are read from global memory.
//
// SCREEN (LDS path only): the scoring loop - H x n distance tests, the FP64-VALU hot spot - is
// first evaluated in f32 on coordinates local to the block and only the (rare) pairs whose f32
// distance lands within a rigorous error bound of the threshold are re-evaluated with the
// reference's exact f64 sequence.  The inlier COUNTS are therefore exactly the reference's:
//   reference      t_ref = fl(fl(fl(A x + B y) + C z) + D)            (util.py:22-24, f64)
//   identity       A x + B y + C z + D = T_o + A (x-ox) + B (y-oy) + C (z-oz),  o = first point
//   screen         s = fma32(a, u, fma32(b, v, fma32(c, w, to)))      u = fl32(x-ox) ..., to = fl32(T_o)
//   |s - t_ref| <= 2^-24 (4 |T_o| + 9 E) + 2^-50 (|o|_1 + |D| + E)     E = max |u|,|v|,|w| of the block
//     (three f32 FMA roundings on partial sums <= |T_o| + 3E, the f32 roundings of u,v,w and to,
//      and the f64 roundings of T_o and of t_ref itself; |a|,|b|,|c| <= 1)
// delta is taken as TWICE that bound plus the f32 rounding of the thresholds themselves:
//   |s| <  thr - delta  =>  |t_ref| < thr   (inlier)      |s| > thr + delta  =>  |t_ref| > thr
// everything in between is decided by the exact sequence.
template <int THREADS, int HPL, int KT, bool IN_LDS, int ABL, bool SCREEN = false>
__device__ __forceinline__ void ransac_block(const BlockDesc& d, int be,
                                             const double* __restrict__ xyz,
                                             const double* __restrict__ lx,
                                             const double* __restrict__ ly,
                                             const double* __restrict__ lz,
                                             const double* __restrict__ hyp, int H, int k,
                                             double thr, const RansacOut& out,
                                             unsigned long long* s_best, float* s_plane,
                                             float4* s_loc = nullptr, int* s_extent = nullptr) {
  const int n = d.n;
  const int64_t pstart = d.pstart;
  const double* __restrict__ pts = xyz + 3 * pstart;
  double ox = 0.0, oy = 0.0, oz = 0.0;
  float extent = 0.f;
  if (SCREEN) {
    // block-local f32 coordinates and their extent (s_extent was zeroed two barriers ago)
    ox = lx[0]; oy = ly[0]; oz = lz[0];
    float m = 0.f;
    if ((int)threadIdx.x < n) {
      const float u = (float)(lx[threadIdx.x] - ox), v = (float)(ly[threadIdx.x] - oy),
                  w = (float)(lz[threadIdx.x] - oz);
      s_loc[threadIdx.x] = make_float4(u, v, w, 0.f);
      m = fmaxf(fabsf(u), fmaxf(fabsf(v), fabsf(w)));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    // non-negative floats order like their bit patterns; NaN/inf extents make everything ambiguous
    if ((threadIdx.x & 63) == 0) atomicMax(s_extent, __float_as_int(m));
    __syncthreads();
    extent = __int_as_float(*s_extent);
  }
  // SCREEN keeps only the f32 plane (its f64 promotion is exact and is redone on demand)
  double pa[HPL], pb[HPL], pc[HPL], pd[HPL];
  float fa[HPL], fb[HPL], fc[HPL], fd[HPL], sto[HPL], slo[HPL], shi[HPL];
  int cnt[HPL];
#pragma unroll
  for (int q = 0; q < HPL; ++q) {
    const int t = threadIdx.x + q * THREADS;
    cnt[q] = -1;
    if (!SCREEN) pa[q] = pb[q] = pc[q] = pd[q] = 0.0;
    if (t < H) {
      float pf[4];
      if (ABL == 2) {  // ablation: no plane fit (timing only, results meaningless)
        pf[0] = (float)hyp[(int64_t)t * k]; pf[1] = pf[0]; pf[2] = pf[0]; pf[3] = pf[0];
      } else {
        // d.pad[0] is always 0; adding it keeps the (loop invariant) row addresses from being
        // hoisted out of the persistent block loop and spilled to scratch
        fit_hypothesis<KT, IN_LDS>(hyp, (uint32_t)t * (uint32_t)(KT > 0 ? KT : k) * 8u + d.pad[0], k, n,
                                   d.vstart, lx, ly, lz, xyz, pstart, (int64_t)d.pspill, pf);
      }
      // the f32-rounded plane, promoted back to f64 for scoring (cuda_ransac.py:110-121)
      cnt[q] = 0;
      if (SCREEN) {
        const double A = (double)pf[0], B = (double)pf[1], Cc = (double)pf[2], D = (double)pf[3];
        const double to = ((A * ox + B * oy) + Cc * oz) + D;
        const double delta = 0x1p-23 * (4.0 * fabs(to) + 9.0 * (double)extent) +
                             0x1p-49 * (fabs(ox) + fabs(oy) + fabs(oz) + fabs(D) + (double)extent + 1.0) +
                             0x1p-23 * fabs(thr);
        fa[q] = pf[0]; fb[q] = pf[1]; fc[q] = pf[2]; fd[q] = pf[3];
        sto[q] = (float)to;
        slo[q] = (float)(thr - delta);
        shi[q] = (float)(thr + delta);
      } else {
        pa[q] = (double)pf[0];
        pb[q] = (double)pf[1];
        pc[q] = (double)pf[2];
        pd[q] = (double)pf[3];
      }
    } else if (SCREEN) {
      fa[q] = fb[q] = fc[q] = fd[q] = 0.f;
      sto[q] = 0.f; slo[q] = 0.f; shi[q] = 0.f;
    }
    // keep the plane fits of the lane's hypotheses apart: interleaving them (all sample loads
    // hoisted to the top) needs > 200 VGPRs and halves the resident waves
#if RS_SCHED_BARRIER
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
  // scoring: every point of the block against every hypothesis of the lane
  // (cuda_ransac.py:116-121); the point is wave uniform (LDS broadcast / scalar load)
  if (ABL != 1 && SCREEN) {
    // two points per iteration, the next pair's LDS reads issued before the current pair is
    // evaluated (the loop is otherwise bound by the LDS round trip, not by the VALU)
    auto screen = [&](const float4& L, int q) -> int {
      const float sv = fabsf(fmaf(fa[q], L.x, fmaf(fb[q], L.y, fmaf(fc[q], L.z, sto[q]))));
      const bool in = sv < slo[q];
      cnt[q] += in ? 1 : 0;
      return (!in && (sv <= shi[q])) ? 1 : 0;
    };
    auto exact = [&](int i, int bits) {
      const double x = lx[i], y = ly[i], z = lz[i];
#pragma unroll
      for (int q = 0; q < HPL; ++q) {
        if (bits & (1 << q)) {
          const double dist = plane_distance((double)fa[q], (double)fb[q], (double)fc[q],
                                             (double)fd[q], x, y, z);
          cnt[q] += (dist < thr) ? 1 : 0;
        }
      }
    };
    float4 L0 = s_loc[0], L1 = s_loc[1];  // s_loc has THREADS >= n + 1 entries
    int i = 0;
    for (; i + 1 < n; i += 2) {
      const float4 N0 = s_loc[i + 2 < THREADS ? i + 2 : 0], N1 = s_loc[i + 3 < THREADS ? i + 3 : 0];
      int b0 = 0, b1 = 0;
#pragma unroll
      for (int q = 0; q < HPL; ++q) {
        b0 |= screen(L0, q) << q;
        b1 |= screen(L1, q) << q;
      }
      if (__any((b0 | b1) != 0)) {  // rare: borderline pairs get the reference's f64 sequence
        if (b0) exact(i, b0);
        if (b1) exact(i + 1, b1);
      }
      L0 = N0;
      L1 = N1;
    }
    if (i < n) {
      int b0 = 0;
#pragma unroll
      for (int q = 0; q < HPL; ++q) b0 |= screen(L0, q) << q;
      if (__any(b0 != 0)) {
        if (b0) exact(i, b0);
      }
    }
  } else if (ABL != 1) {
RS_PRAGMA(unroll RS_SCORE_UNROLL)
    for (int i = 0; i < n; ++i) {
      double x, y, z;
      if (IN_LDS) {
        x = lx[i]; y = ly[i]; z = lz[i];
      } else {
        x = pts[3 * (int64_t)i]; y = pts[3 * (int64_t)i + 1]; z = pts[3 * (int64_t)i + 2];
      }
#pragma unroll
      for (int q = 0; q < HPL; ++q) {
        const double dist = plane_distance(pa[q], pb[q], pc[q], pd[q], x, y, z);
        cnt[q] += (dist < thr) ? 1 : 0;
      }
    }
  }
  // block-wide maximum, lowest hypothesis index among the tied (cuda_ransac.py:125-146)
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
      // (float)pa is exact: pa was promoted from the f32 plane
      const float f0 = SCREEN ? fa[q] : (float)pa[q], f1 = SCREEN ? fb[q] : (float)pb[q],
                  f2 = SCREEN ? fc[q] : (float)pc[q], f3 = SCREEN ? fd[q] : (float)pd[q];
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
  // final mask with the winning f32 plane (cuda_ransac.py:149-155)
  const double a = (double)s_plane[0], bb = (double)s_plane[1], c = (double)s_plane[2],
               dd = (double)s_plane[3];
  for (int i = threadIdx.x; i < n; i += THREADS) {
    double x, y, z;
    if (IN_LDS) {
      x = lx[i]; y = ly[i]; z = lz[i];
    } else {
      x = pts[3 * (int64_t)i]; y = pts[3 * (int64_t)i + 1]; z = pts[3 * (int64_t)i + 2];
    }
    out.mask[pstart + i] = (plane_distance(a, bb, c, dd, x, y, z) < thr) ? 1 : 0;
  }
}

// Persistent workgroups: workgroup w handles batch entries w, w+G, w+2G, ...  While entry e is
// being computed out of one LDS buffer, the points of entry e+G are already in flight into
// registers (one point per lane) and the descriptor of entry e+2G is being fetched, so the
// per-block HBM/L2 latency chain (descriptor -> points) is off the critical path.  The first
// version (one workgroup per block, points fetched at its start) spent more time waiting for
// that chain than computing: a block holds ~17 points.
template <int THREADS, int HPL, int KT, int ABL>
__global__ __launch_bounds__(THREADS, RS_MINWAVES) void k_ransac(const double* __restrict__ xyz,
                                                    const BlockDesc* __restrict__ desc, int nb,
                                                    const double* __restrict__ hyp, int H, int k,
                                                    double thr, RansacOut out) {
  constexpr int CAP = THREADS - 1;
  __shared__ double s_pts[2][3][THREADS];
  __shared__ float4 s_loc[THREADS];
  __shared__ unsigned long long s_best[THREADS / 64];
  __shared__ float s_plane[4];
  __shared__ int s_extent;
  const int G = gridDim.x;
  int be = blockIdx.x;
  if (be >= nb) return;
  if (threadIdx.x == 0) s_extent = 0;
  BlockDesc cur = desc[be];
  BlockDesc nxt = cur;
  if (be + G < nb) nxt = desc[be + G];
  int buf = 0;
  // stage the first block
  if (cur.n >= k && cur.n <= CAP && (int)threadIdx.x <= cur.n) {
    const int64_t p = ((int)threadIdx.x < cur.n) ? (int64_t)cur.pstart + threadIdx.x : (int64_t)cur.pspill;
    s_pts[0][0][threadIdx.x] = xyz[3 * p];
    s_pts[0][1][threadIdx.x] = xyz[3 * p + 1];
    s_pts[0][2][threadIdx.x] = xyz[3 * p + 2];
  }
  __syncthreads();
  while (true) {
    const bool has_next = be + G < nb;
    const bool has_next2 = be + 2 * G < nb;
    BlockDesc nxt2 = nxt;
    if (has_next2) nxt2 = desc[be + 2 * G];
    // points of the next block -> registers (in flight during the compute below)
    double rx = 0.0, ry = 0.0, rz = 0.0;
    const bool pre = has_next && nxt.n >= k && nxt.n <= CAP && (int)threadIdx.x <= nxt.n;
    if (pre) {
      const int64_t p = ((int)threadIdx.x < nxt.n) ? (int64_t)nxt.pstart + threadIdx.x : (int64_t)nxt.pspill;
      rx = xyz[3 * p];
      ry = xyz[3 * p + 1];
      rz = xyz[3 * p + 2];
    }
    // the current block
    if (cur.n < k) {  // cuda_ransac.py:96-97: the whole block returns, mask stays False
      for (int i = threadIdx.x; i < cur.n; i += THREADS) out.mask[(int64_t)cur.pstart + i] = 0;
      if (threadIdx.x == 0) {
        if (out.plane) {
          out.plane[4 * (int64_t)be + 0] = 0.f; out.plane[4 * (int64_t)be + 1] = 0.f;
          out.plane[4 * (int64_t)be + 2] = 0.f; out.plane[4 * (int64_t)be + 3] = 0.f;
        }
        if (out.count) out.count[be] = 0;
        if (out.index) out.index[be] = -1;
      }
    } else if (cur.n <= CAP) {
      ransac_block<THREADS, HPL, KT, true, ABL, RS_SCREEN != 0>(
          cur, be, xyz, s_pts[buf][0], s_pts[buf][1], s_pts[buf][2], hyp, H, k, thr, out, s_best,
          s_plane, s_loc, &s_extent);
    }  // larger blocks: k_ransac_big
    if (!has_next) break;
    if (threadIdx.x == 0) s_extent = 0;  // read only before the plane fits; re-armed for the next block
    __syncthreads();  // every wave is done with s_pts[buf ^ 1]'s previous contents (and s_best)
    if (pre) {
      s_pts[buf ^ 1][0][threadIdx.x] = rx;
      s_pts[buf ^ 1][1][threadIdx.x] = ry;
      s_pts[buf ^ 1][2][threadIdx.x] = rz;
    }
    __syncthreads();
    cur = nxt;
    nxt = nxt2;
    be += G;
    buf ^= 1;
  }
}

// Blocks with more than THREADS-1 points (unsplit voxels, poses outside the scheme, large K):
// points stay in global memory (wave-uniform scalar loads in the scoring loop).  The list of
// such batch entries is appended by k_block_desc; its length lives in device memory, so the
// grid is fixed and every workgroup strides over the list.
template <int THREADS, int HPL, int KT>
__global__ __launch_bounds__(THREADS) void k_ransac_big(const double* __restrict__ xyz,
                                                        const BlockDesc* __restrict__ desc,
                                                        const uint32_t* __restrict__ big_list,
                                                        const uint32_t* __restrict__ big_count,
                                                        const double* __restrict__ hyp, int H,
                                                        int k, double thr, RansacOut out) {
  __shared__ unsigned long long s_best[THREADS / 64];
  __shared__ float s_plane[4];
  const uint32_t count = *big_count;
  for (uint32_t j = blockIdx.x; j < count; j += gridDim.x) {
    const int be = (int)big_list[j];
    const BlockDesc d = desc[be];
    ransac_block<THREADS, HPL, KT, false, 0>(d, be, xyz, nullptr, nullptr, nullptr, hyp, H, k, thr,
                                             out, s_best, s_plane);
    __syncthreads();
  }
}

// batch entry -> descriptor (sizes in batch order were scanned into `scanned`)
__global__ __launch_bounds__(256) void k_block_sizes_in_order(const int32_t* __restrict__ order,
                                                              const int32_t* __restrict__ size,
                                                              int64_t nb,
                                                              uint32_t* __restrict__ tmp_sizes) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  tmp_sizes[b] = (uint32_t)size[order ? order[b] : b];
}

__global__ __launch_bounds__(256) void k_block_desc(const int32_t* __restrict__ order,
                                                    const uint32_t* __restrict__ start,
                                                    const int32_t* __restrict__ size,
                                                    const uint32_t* __restrict__ scanned,
                                                    int64_t nb, int64_t n_points, int cap,
                                                    int k, BlockDesc* __restrict__ desc,
                                                    uint32_t* __restrict__ big_list,
                                                    uint32_t* __restrict__ big_count) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  const int32_t phys = order ? order[b] : (int32_t)b;
  BlockDesc d;
  d.pstart = start[phys];
  d.n = size[phys];
  d.vstart = (int64_t)scanned[b];  // np.cumsum([0] + sizes[:-1]) in batch order (cuda_ransac.py:64-66)
  uint32_t sp = d.n > 0 ? d.pstart + (uint32_t)d.n - 1u : d.pstart;
  if (b + 1 < nb) {
    sp = start[order ? order[b + 1] : b + 1];
  } else if (!order) {
    // stand-alone operator: the cloud may continue past the last block (cuda_ransac.py:43-81)
    const int64_t e = (int64_t)d.pstart + d.n;
    if (e < n_points) sp = (uint32_t)e;
  }
  d.pspill = sp;
  d.pad[0] = d.pad[1] = d.pad[2] = 0;
  desc[b] = d;
  if (d.n > cap && d.n >= k) big_list[atomicAdd(big_count, 1u)] = (uint32_t)b;
}

}  // namespace

// Launch the kernel over nb batch entries.  `order` (device, nullable) maps batch entry ->
// physical block.  Descriptors (virtual start, spill point) are derived on the device from the
// sizes in batch order.
int ransac_launch(octl_ctx* ctx, const double* xyz_dev, int64_t n_points,
                  const uint32_t* blk_start, const int32_t* blk_size, const int32_t* order_dev,
                  int64_t nb, const double* hyp_dev, int32_t H, int32_t k, double thr,
                  uint8_t* mask_dev, float* plane_dev, int32_t* count_dev, int32_t* index_dev,
                  uint8_t* evaluated_dev, DevBuf& scratch) {
  (void)evaluated_dev;
  if (nb <= 0) return OCTL_OK;
  if (H < 1 || H > 1024) return octl_set_error(ctx, OCTL_E_INVALID, "H must be in [1, 1024]");
  if (k < 1 || k > RS_KMAX)
    return octl_set_error(ctx, OCTL_E_INVALID, "initial_points_number must be in [1, %d]", RS_KMAX);
  if (nb >= ((int64_t)1 << 31)) return octl_set_error(ctx, OCTL_E_INVALID, "too many blocks");
  hipStream_t st = ctx->stream;
  // scratch: [sizes/scanned u32 nb+8 | descriptors 32 B x nb | big list u32 nb | big count]
  const size_t off_d = (((size_t)nb + 8) * 4 + 31) & ~(size_t)31;
  const size_t off_b = off_d + (size_t)nb * sizeof(BlockDesc);
  OCTL_TRY(devbuf_reserve(ctx, scratch, off_b + ((size_t)nb + 8) * 4));
  uint32_t* tmp = scratch.as<uint32_t>();
  BlockDesc* desc = reinterpret_cast<BlockDesc*>(static_cast<char*>(scratch.p) + off_d);
  uint32_t* big_list = reinterpret_cast<uint32_t*>(static_cast<char*>(scratch.p) + off_b);
  uint32_t* big_count = big_list + nb;
  const int threads = (H <= 64) ? 64 : 256;
  {
    KTimer t(ctx, "ransac_prepare");
    const unsigned g = (unsigned)ceil_div(nb, 256);
    hipLaunchKernelGGL(k_block_sizes_in_order, dim3(g), dim3(256), 0, st, order_dev, blk_size, nb, tmp);
    HIP_TRY(ctx, hipGetLastError());
    OCTL_TRY(octl_exclusive_scan_u32(ctx, tmp, tmp, nb, nullptr));
    HIP_TRY(ctx, hipMemsetAsync(big_count, 0, 4, st));
    hipLaunchKernelGGL(k_block_desc, dim3(g), dim3(256), 0, st, order_dev, blk_start, blk_size,
                       (const uint32_t*)tmp, nb, n_points, threads - 1, (int)k, desc, big_list,
                       big_count);
    HIP_TRY(ctx, hipGetLastError());
  }
  RansacOut out{mask_dev, plane_dev, count_dev, index_dev};
  // persistent grid: 4 workgroups of 256 threads per CU (VGPR-limited residency)
  int cus = 256;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
  }
  KTimer t(ctx, "ransac");
#define OCTL_RANSAC_LAUNCH(THREADS, HPL, KT, ABL, PER_CU)                                       \
  do {                                                                                          \
    const unsigned g = (unsigned)std::min<int64_t>(nb, (int64_t)cus * (PER_CU));                \
    hipLaunchKernelGGL((k_ransac<THREADS, HPL, KT, ABL>), dim3(g), dim3(THREADS), 0, st,         \
                       xyz_dev, (const BlockDesc*)desc, (int)nb, hyp_dev, H, k, thr, out);       \
  } while (0)
  if (H <= 64) {
    if (k == 6) OCTL_RANSAC_LAUNCH(64, 1, 6, 0, 16); else OCTL_RANSAC_LAUNCH(64, 1, 0, 0, 8);
  } else if (H <= 256) {
    if (k == 6) OCTL_RANSAC_LAUNCH(256, 1, 6, 0, 8); else OCTL_RANSAC_LAUNCH(256, 1, 0, 0, 4);
  } else {
    const char* abl = getenv("OCTL_RANSAC_ABLATE");  // timing experiments only
    if (k == 6 && abl && abl[0] == '1') OCTL_RANSAC_LAUNCH(256, 4, 6, 1, 4);
    else if (k == 6 && abl && abl[0] == '2') OCTL_RANSAC_LAUNCH(256, 4, 6, 2, 4);
    else if (k == 6) OCTL_RANSAC_LAUNCH(256, 4, 6, 0, 4);
    else OCTL_RANSAC_LAUNCH(256, 4, 0, 0, 2);
  }
#undef OCTL_RANSAC_LAUNCH
  HIP_TRY(ctx, hipGetLastError());
  // the (rare) blocks that do not fit the LDS staging; the grid is fixed, the count is on the device
#define OCTL_RANSAC_BIG(THREADS, HPL, KT)                                                       \
  hipLaunchKernelGGL((k_ransac_big<THREADS, HPL, KT>), dim3((unsigned)std::min<int64_t>(nb, 2 * cus)), \
                     dim3(THREADS), 0, st, xyz_dev, (const BlockDesc*)desc,                     \
                     (const uint32_t*)big_list, (const uint32_t*)big_count, hyp_dev, H, k, thr, out)
  if (H <= 64) {
    if (k == 6) OCTL_RANSAC_BIG(64, 1, 6); else OCTL_RANSAC_BIG(64, 1, 0);
  } else if (H <= 256) {
    if (k == 6) OCTL_RANSAC_BIG(256, 1, 6); else OCTL_RANSAC_BIG(256, 1, 0);
  } else {
    if (k == 6) OCTL_RANSAC_BIG(256, 4, 6); else OCTL_RANSAC_BIG(256, 4, 0);
  }
#undef OCTL_RANSAC_BIG
  HIP_TRY(ctx, hipGetLastError());
  return OCTL_OK;
}
