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
#include "forest.h"

namespace {

struct RansacBlocks {
  const uint32_t* start;   // physical start of the block in the point array
  const int32_t* size;     // points in the block
  const int64_t* vstart;   // start of the block in the reference's concatenated batch cloud
  const uint32_t* spill;   // physical index of the point that follows the block in the
                           // reference's cloud (first point of the next block), or 0xFFFFFFFF
  const int32_t* order;    // optional indirection: physical block id of batch entry b
};

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
  const double kd = (double)k;
  cx /= kd;  // util.py:42-44
  cy /= kd;
  cz /= kd;
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

constexpr int RS_KMAX = 16;  // initial_points_number supported by the register path

// One workgroup per block; THREADS = 64*W lanes each owning HPL hypotheses: lane t handles
// hypotheses t, t+THREADS, ...  (H <= THREADS*HPL)
template <int THREADS, int HPL, int KT>
__global__ __launch_bounds__(THREADS) void k_ransac(
    const double* __restrict__ xyz, int64_t n_points, RansacBlocks blk,
    const double* __restrict__ hyp, int H, int k, double thr, uint8_t* __restrict__ mask,
    float* __restrict__ plane_out, int32_t* __restrict__ count_out,
    int32_t* __restrict__ index_out, uint8_t* __restrict__ evaluated) {
  __shared__ unsigned long long s_best[THREADS / 64];
  __shared__ float s_plane[4];
  const int be = blockIdx.x;
  const int b = blk.order ? blk.order[be] : be;
  const int n = blk.size[b];
  if (evaluated) {
    if (threadIdx.x == 0) evaluated[b] = 1;
  }
  if (n < k) {  // cuda_ransac.py:96-97: the whole block returns, mask stays False
    for (int i = threadIdx.x; i < n; i += THREADS) mask[(int64_t)blk.start[b] + i] = 0;
    if (threadIdx.x == 0) {
      if (plane_out) {
        plane_out[4 * (int64_t)be + 0] = 0.f; plane_out[4 * (int64_t)be + 1] = 0.f;
        plane_out[4 * (int64_t)be + 2] = 0.f; plane_out[4 * (int64_t)be + 3] = 0.f;
      }
      if (count_out) count_out[be] = 0;
      if (index_out) index_out[be] = -1;
    }
    return;
  }
  const int64_t pstart = blk.start[b];
  const int64_t vstart = blk.vstart ? blk.vstart[be] : pstart;
  const uint32_t spill = blk.spill ? blk.spill[be] : 0xFFFFFFFFu;
  const double* __restrict__ pts = xyz + 3 * pstart;

  double pa[HPL], pb[HPL], pc[HPL], pd[HPL];
  float pf[HPL][4];
  int cnt[HPL];
#pragma unroll
  for (int q = 0; q < HPL; ++q) {
    const int t = threadIdx.x + q * THREADS;
    cnt[q] = -1;
    pa[q] = pb[q] = pc[q] = pd[q] = 0.0;
    pf[q][0] = pf[q][1] = pf[q][2] = pf[q][3] = 0.f;
    if (t < H) {
      constexpr int KS = KT > 0 ? KT : RS_KMAX;
      double sx[KS], sy[KS], sz[KS];
#pragma unroll
      for (int i = 0; i < KS; ++i) {
        sx[i] = sy[i] = sz[i] = 0.0;
        if (i < k) {
          // initial_point_indices[i] = nb.int32(random_hypotheses[t][i] * block_size +
          // block_start) (cuda_ransac.py:103-107): f64 multiply, f64 add, truncation
          const double v = hyp[(int64_t)t * k + i] * (double)n + (double)vstart;
          const int64_t g = (int64_t)(int)v - vstart;  // position inside the block; may be == n
          int64_t p;
          if (g < n) {
            p = pstart + g;
          } else {
            // rounding of R*n + s reached the first point of the next block of the batch
            p = (spill != 0xFFFFFFFFu) ? (int64_t)spill : pstart + n - 1;
          }
          sx[i] = xyz[3 * p];
          sy[i] = xyz[3 * p + 1];
          sz[i] = xyz[3 * p + 2];
        }
      }
      plane_from_samples<KT, KS>(sx, sy, sz, k, pf[q]);
      pa[q] = (double)pf[q][0];
      pb[q] = (double)pf[q][1];
      pc[q] = (double)pf[q][2];
      pd[q] = (double)pf[q][3];
      cnt[q] = 0;
    }
  }
  // scoring: for each point of the block (wave-uniform address -> scalar loads), every
  // hypothesis of the lane (cuda_ransac.py:116-121)
  for (int i = 0; i < n; ++i) {
    const double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
#pragma unroll
    for (int q = 0; q < HPL; ++q) {
      const double dist = plane_distance(pa[q], pb[q], pc[q], pd[q], x, y, z);
      cnt[q] += (dist < thr) ? 1 : 0;
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
      s_plane[0] = pf[q][0]; s_plane[1] = pf[q][1]; s_plane[2] = pf[q][2]; s_plane[3] = pf[q][3];
      if (plane_out) {
        plane_out[4 * (int64_t)be + 0] = pf[q][0]; plane_out[4 * (int64_t)be + 1] = pf[q][1];
        plane_out[4 * (int64_t)be + 2] = pf[q][2]; plane_out[4 * (int64_t)be + 3] = pf[q][3];
      }
      if (count_out) count_out[be] = cnt[q];
      if (index_out) index_out[be] = win;
    }
  }
  __syncthreads();
  // final mask with the winning f32 plane (cuda_ransac.py:149-155)
  const double a = (double)s_plane[0], bb = (double)s_plane[1], c = (double)s_plane[2],
               d = (double)s_plane[3];
  for (int i = threadIdx.x; i < n; i += THREADS) {
    const double dist = plane_distance(a, bb, c, d, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    mask[pstart + i] = (dist < thr) ? 1 : 0;
  }
}

__global__ __launch_bounds__(256) void k_vstart_from_sizes(const int32_t* __restrict__ order,
                                                           const int32_t* __restrict__ size,
                                                           int64_t nb,
                                                           uint32_t* __restrict__ tmp_sizes) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  tmp_sizes[b] = (uint32_t)size[order ? order[b] : b];
}

__global__ __launch_bounds__(256) void k_vstart_finish(const int32_t* __restrict__ order,
                                                       const uint32_t* __restrict__ start,
                                                       const int32_t* __restrict__ size,
                                                       const uint32_t* __restrict__ scanned,
                                                       int64_t nb, int64_t n_points,
                                                       int64_t* __restrict__ vstart,
                                                       uint32_t* __restrict__ spill) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  vstart[b] = (int64_t)scanned[b];
  uint32_t sp = 0xFFFFFFFFu;
  if (b + 1 < nb) {
    sp = start[order ? order[b + 1] : b + 1];
  } else if (!order) {
    // stand-alone operator: the cloud may continue past the last block (cuda_ransac.py:43-81)
    const int64_t e = (int64_t)start[b] + size[b];
    if (e < n_points) sp = (uint32_t)e;
  }
  spill[b] = sp;
}

}  // namespace

// Launch the kernel over nb batch entries.  `order` (device, nullable) maps batch entry ->
// physical block.  vstart/spill are derived on the device from the sizes in batch order.
int ransac_launch(octl_ctx* ctx, const double* xyz_dev, int64_t n_points,
                  const uint32_t* blk_start, const int32_t* blk_size, const int32_t* order_dev,
                  int64_t nb, const double* hyp_dev, int32_t H, int32_t k, double thr,
                  uint8_t* mask_dev, float* plane_dev, int32_t* count_dev, int32_t* index_dev,
                  uint8_t* evaluated_dev, DevBuf& scratch) {
  if (nb <= 0) return OCTL_OK;
  if (H < 1 || H > 1024) return octl_set_error(ctx, OCTL_E_INVALID, "H must be in [1, 1024]");
  if (k < 1 || k > RS_KMAX)
    return octl_set_error(ctx, OCTL_E_INVALID, "initial_points_number must be in [1, %d]", RS_KMAX);
  if (nb >= ((int64_t)1 << 31)) return octl_set_error(ctx, OCTL_E_INVALID, "too many blocks");
  hipStream_t st = ctx->stream;
  // scratch: [sizes/scanned u32 nb+8 | vstart i64 nb | spill u32 nb]
  const size_t off_v = (((size_t)nb + 8) * 4 + 15) & ~(size_t)15;
  const size_t off_s = off_v + (size_t)nb * 8;
  OCTL_TRY(devbuf_reserve(ctx, scratch, off_s + (size_t)nb * 4 + 16));
  uint32_t* tmp = scratch.as<uint32_t>();
  int64_t* vstart = reinterpret_cast<int64_t*>(static_cast<char*>(scratch.p) + off_v);
  uint32_t* spill = reinterpret_cast<uint32_t*>(static_cast<char*>(scratch.p) + off_s);
  {
    KTimer t(ctx, "ransac_prepare");
    const unsigned g = (unsigned)ceil_div(nb, 256);
    hipLaunchKernelGGL(k_vstart_from_sizes, dim3(g), dim3(256), 0, st, order_dev, blk_size, nb, tmp);
    HIP_TRY(ctx, hipGetLastError());
    OCTL_TRY(octl_exclusive_scan_u32(ctx, tmp, tmp, nb, nullptr));
    hipLaunchKernelGGL(k_vstart_finish, dim3(g), dim3(256), 0, st, order_dev, blk_start, blk_size,
                       (const uint32_t*)tmp, nb, n_points, vstart, spill);
    HIP_TRY(ctx, hipGetLastError());
  }
  RansacBlocks blk;
  blk.start = blk_start;
  blk.size = blk_size;
  blk.vstart = vstart;
  blk.spill = spill;
  blk.order = order_dev;
  KTimer t(ctx, "ransac");
#define OCTL_RANSAC_LAUNCH(THREADS, HPL, KT)                                                    \
  hipLaunchKernelGGL((k_ransac<THREADS, HPL, KT>), dim3((unsigned)nb), dim3(THREADS), 0, st,      \
                     xyz_dev, n_points, blk, hyp_dev, H, k, thr, mask_dev, plane_dev, count_dev,  \
                     index_dev, evaluated_dev)
  if (H <= 64) {
    if (k == 6) OCTL_RANSAC_LAUNCH(64, 1, 6); else OCTL_RANSAC_LAUNCH(64, 1, 0);
  } else if (H <= 256) {
    if (k == 6) OCTL_RANSAC_LAUNCH(256, 1, 6); else OCTL_RANSAC_LAUNCH(256, 1, 0);
  } else {
    if (k == 6) OCTL_RANSAC_LAUNCH(256, 4, 6); else OCTL_RANSAC_LAUNCH(256, 4, 0);
  }
#undef OCTL_RANSAC_LAUNCH
  HIP_TRY(ctx, hipGetLastError());
  return OCTL_OK;
}
