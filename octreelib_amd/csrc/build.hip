// insert + subdivide on the device.
//
// Replaces, for the count criterion len(points) > K (paths relative to /root/reference):
//   Grid.insert_points          grid/grid.py:58-109      top-level voxel bucketing
//   OctreeManager.subdivide     octree_manager.py:36-66  scheme from the union of poses
//   OctreeNode.subdivide/_as    octree/octree.py:20-53   recursive 8-way split
//   OctreeNode.insert_points    octree/octree.py:67-100  child index arithmetic
//   OctreeNode._generate_children octree/octree.py:177-191 child corners / edges
//
// This is the GENERAL path: schemes with history, keep_scheme re-placement, voxels with more points than
// the bucket build of bucket_build.hip sorts in LDS (a bare Octree / OctreeManager with millions of
// points in one cube), trees deeper than 7 levels, points outside their cube.  A fresh forest is
// built by bucket_build.hip (step 0 of forest_build).
// Pipeline (all kernels HBM-bound integer/compare work; nothing here is a contraction):
//   k_keygen      xyz -> packed voxel key + 21-level child-digit path (reference arithmetic
//                 restated as exact f64 comparisons), voxel bounding box
//   k_linkey      voxel key -> compact linear key, point index (+ scheme-pose bit)
//   radix sort    stable LSD sort by linear voxel key (radix_sort.hip)
//   roots         one scheme node per top-level voxel
//   level loop    level-synchronous recursive subdivide: every node whose scheme-pose count
//                 exceeds K is split 8 ways by a stable tile partition (ballot ranks in LDS)
//   k_finalize    leaf-ordered point permutation + coordinates, (leaf, pose) block table
#include <algorithm>
#include <chrono>
#include <cstdlib>

#include "build_common.h"
#include "forest.h"
#include "ref_arith.h"
#include "wave_utils.h"

// OCTL_TRACE_BUILD=1: host wall time between the phases of forest_build on stderr (diagnostics)
struct BuildTrace {
  bool on = false;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void mark(const char* what) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[build] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
    t = now;
  }
};

namespace {

constexpr int LV_THREADS = 256;
constexpr int LV_WAVES = LV_THREADS / 64;
#ifndef OCTL_LV_IPT
#define OCTL_LV_IPT 4
#endif
constexpr int LV_IPT = OCTL_LV_IPT;  // (positions per thread of a level tile; -D for experiments)
constexpr int LV_TILE = LV_THREADS * LV_IPT;  // 1024 positions per tile
constexpr int LV_WAVE_ITEMS = 64 * LV_IPT;
// child digits precomputed per path word.  The word has room for 21; trees are rarely deeper
// than a few levels, so only the first PATH_LEVELS are computed up front (k_keygen is VALU bound on
// exactly this loop) and k_lv_rekey extends the paths of the points that do go deeper.
#ifndef OCTL_PATH_LEVELS
#define OCTL_PATH_LEVELS 6
#endif
constexpr int PATH_LEVELS = OCTL_PATH_LEVELS;

// ---------------------------------------------------------------------------------------------
// reference arithmetic
// ---------------------------------------------------------------------------------------------

// Child digits of up to 21 consecutive levels below the cube (c, e).
// Reference per level (octree/octree.py:73-75,94-97,181-191):
//     idx = floor((p - corner) / (edge / 2))  per axis, must be 0 or 1
//     child_id = 4*ix + 2*iy + iz ; child corner = corner + idx * (edge / 2) ; child edge = edge / 2
// floor((p-c)/h) in {0,1} <=> 0 <= fl(p-c) < 2h, and it is 1 <=> fl(p-c) >= h: the division is
// replaced by exact comparisons on the same rounded difference the reference forms.
// Layout (32 bits: the level loop streams this word once per level, 8 bytes were a quarter of its traffic):
// digit of level j at bits [29-3j, 31-3j], j < PATH_LEVELS = 6; bit 0 = "bad" (some level had idx outside
// {0,1}, or a non-finite coordinate) - such a point takes the slow path and raises a domain
// error only if a node containing it is actually split, as in the reference.
static_assert(OCTL_PATH_LEVELS >= 1 && OCTL_PATH_LEVELS <= 10, "the path word holds at most 10 digits above the bad bit");
__device__ __forceinline__ uint32_t compute_path(double px, double py, double pz, double cx,
                                                 double cy, double cz, double e) {
  uint32_t path = 0;
  double h = e / 2.0;
#pragma unroll 1
  for (int j = 0; j < PATH_LEVELS; ++j) {
    const double ax = px - cx, ay = py - cy, az = pz - cz;
    const bool ok = (ax >= 0.0) && (ax < e) && (ay >= 0.0) && (ay < e) && (az >= 0.0) && (az < e);
    if (!ok) return path | 1u;
    const bool bx = ax >= h, by = ay >= h, bz = az >= h;
    const uint32_t digit = (bx ? 4u : 0u) | (by ? 2u : 0u) | (bz ? 1u : 0u);
    path |= digit << (29 - 3 * j);
    cx = cx + (bx ? h : 0.0);
    cy = cy + (by ? h : 0.0);
    cz = cz + (bz ? h : 0.0);
    e = h;
    h = e / 2.0;
  }
  return path;
}

// compute_path for the cube of a top-level voxel / a whole single cube: non-negative coordinates below an
// integer-valued cube take the exact short form (ref_arith.h: digits18_exact), everything else the levels
__device__ __forceinline__ uint32_t compute_path_root(double px, double py, double pz, double cx, double cy,
                                                      double cz, double e, bool exact_cube, bool edge_pow2,
                                                      double inv64) {
  if (PATH_LEVELS == 6 && exact_cube && coord_takes_exact_digits(px) && coord_takes_exact_digits(py) &&
      coord_takes_exact_digits(pz)) {
    bool bad = false;
    const uint32_t d18 = digits18_exact(px, py, pz, cx, cy, cz, e, edge_pow2, inv64, &bad);
    return (d18 << 14) | (bad ? 1u : 0u);
  }
  return compute_path(px, py, pz, cx, cy, cz, e);
}

__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
  return v;
}

// ---------------------------------------------------------------------------------------------
// key generation
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_keygen(const double* __restrict__ xyz,
                                                const uint8_t* __restrict__ alive, int64_t n,
                                                int mode, double L, double c0x, double c0y,
                                                double c0z, VoxOrg org, int exact_digits,
                                                uint64_t* __restrict__ vkey, uint32_t* __restrict__ path,
                                                uint32_t* __restrict__ small) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool live = (i < n) && alive[i];
  int qx = 0, qy = 0, qz = 0;
  // (kernel-uniform: see compute_path_root)
  const bool exact_cube = exact_digits && nonneg_integer_below_2p45(L) && L >= 1.0 &&
                          (mode == 0 || (nonneg_integer_below_2p45(c0x) && nonneg_integer_below_2p45(c0y) &&
                                         nonneg_integer_below_2p45(c0z)));
  const bool edge_pow2 = (__double_as_longlong(L) & 0xFFFFFFFFFFFFFll) == 0;
  const double inv64 = 64.0 / L;
  if (live) {
    const double px = xyz[3 * i + 0], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
    double cx = c0x, cy = c0y, cz = c0z;
    if (mode == 0) {
      // voxel_indices = ((points - corner) // L * L).astype(int)   (grid.py:72-76, corner = 0)
      const double fx = floor_div_exact(px, L), fy = floor_div_exact(py, L),
                   fz = floor_div_exact(pz, L);
      const double lim = (double)OCTL_VOX_ABS_LIMIT;
      // (false for NaN; the second test: inside the window of the packed keys around the forest's origin)
      const bool in_range = (fabs(fx) < lim) && (fabs(fy) < lim) && (fabs(fz) < lim) &&
                            vkey_in_window((int64_t)fx, (int64_t)fy, (int64_t)fz, org);
      if (!in_range) {
        atomicExch(&small[SM_ERR], (uint32_t)(-OCTL_E_DOMAIN));
        live = false;
      } else {
        qx = (int)fx;
        qy = (int)fy;
        qz = (int)fz;
        // the manager's corner is np.array(voxel_coords): int64(q*L) (grid.py:96-105)
        // (truncation towards zero, a zero comes back as +0.0: see lin_corner_of in bucket_build.hip)
        cx = trunc(fx * L) + 0.0;
        cy = trunc(fy * L) + 0.0;
        cz = trunc(fz * L) + 0.0;
      }
    }
    if (live) {
      vkey[i] = vkey_pack(qx, qy, qz, org);
      path[i] = compute_path_root(px, py, pz, cx, cy, cz, L, exact_cube, edge_pow2, inv64);
    }
  }
  if (i < n && !live) {
    vkey[i] = OCTL_VOX_DEAD;
    path[i] = 0;
  }
  // voxel bounding box: wave + block reduction, then at most six atomics per BLOCK, and only
  // when the block actually widens the box (same-address atomics serialise at ~10 ns each)
  __shared__ int s_bb[4][6];
  const int big = 1 << 30;
  const int mnx = wave_min_i32(live ? qx : big), mny = wave_min_i32(live ? qy : big),
            mnz = wave_min_i32(live ? qz : big);
  const int mxx = wave_max_i32(live ? qx : -big), mxy = wave_max_i32(live ? qy : -big),
            mxz = wave_max_i32(live ? qz : -big);
  if ((threadIdx.x & 63) == 0) {
    int* w = s_bb[threadIdx.x >> 6];
    w[0] = mnx; w[1] = mny; w[2] = mnz; w[3] = mxx; w[4] = mxy; w[5] = mxz;
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int a = threadIdx.x;
    int v = s_bb[0][a];
    for (int w = 1; w < 4; ++w) v = (a < 3) ? min(v, s_bb[w][a]) : max(v, s_bb[w][a]);
    int* bb = reinterpret_cast<int*>(small + SM_BBOX);
    if (a < 3) {
      if (v != big && v < __hip_atomic_load(&bb[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMin(&bb[a], v);
    } else {
      if (v != -big && v > __hip_atomic_load(&bb[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(&bb[a], v);
    }
  }
}

__device__ __forceinline__ int find_slot(const int64_t* __restrict__ pose_off, int n_poses,
                                         int64_t idx) {
  int lo = 0, hi = n_poses;  // pose_off[lo] <= idx < pose_off[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (pose_off[mid] <= idx) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void k_linkey(const uint64_t* __restrict__ vkey, int64_t n,
                                                int minx, int miny, int minz, uint64_t ny,
                                                uint64_t nz, uint64_t dead_lin, VoxOrg org,
                                                const int64_t* __restrict__ pose_off, int n_poses,
                                                const uint8_t* __restrict__ scheme,
                                                uint64_t* __restrict__ lin,
                                                uint32_t* __restrict__ val) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t k = vkey[i];
  uint64_t l = dead_lin;
  if (k != OCTL_VOX_DEAD) {
    int64_t qd[3];
    vkey_decode(k, org, qd);
    l = ((uint64_t)(qd[0] - minx) * ny + (uint64_t)(qd[1] - miny)) * nz + (uint64_t)(qd[2] - minz);
  }
  lin[i] = l;
  uint32_t v = (uint32_t)i;
  if (scheme) {
    if (scheme[find_slot(pose_off, n_poses, i)]) v |= 0x80000000u;
  } else {
    v |= 0x80000000u;
  }
  val[i] = v;
}

// ---------------------------------------------------------------------------------------------
// roots
// ---------------------------------------------------------------------------------------------
// Voxel heads of the sorted keys, per TILE of 2048 positions (eight per thread): k_root_tiles<false> counts them,
// a scan over the tiles (not over the positions) gives every tile its first voxel ordinal, k_root_tiles<true>
// writes the voxel list and k_init_level0 derives every position's voxel from the same tile bases.  (Round 2
// wrote a flag per position, scanned n flags and read them back twice: ~20 B per point of traffic for a table
// of a few thousand voxels.)
constexpr int RT_IPT = 8;
constexpr int RT_TILE = 256 * RT_IPT;

// bit q of the result: position first + q starts a voxel (lin differs from the position in front of it);
// positions behind n_alive never do
__device__ __forceinline__ uint32_t tile_voxel_heads(const uint64_t* __restrict__ lin, int64_t first,
                                                     int64_t n_alive) {
  uint64_t k[RT_IPT];
  if (first + RT_IPT <= n_alive) {  // (16-byte aligned: first is a multiple of 8)
    const ulonglong2* p = reinterpret_cast<const ulonglong2*>(lin + first);
#pragma unroll
    for (int q = 0; q < RT_IPT / 2; ++q) {
      const ulonglong2 v = p[q];
      k[2 * q] = v.x;
      k[2 * q + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int q = 0; q < RT_IPT; ++q) k[q] = first + q < n_alive ? lin[first + q] : 0ull;
  }
  uint64_t prev = (first > 0 && first < n_alive) ? lin[first - 1] : 0ull;
  uint32_t heads = 0;
#pragma unroll
  for (int q = 0; q < RT_IPT; ++q) {
    if (first + q < n_alive && (first + q == 0 || k[q] != prev)) heads |= 1u << q;
    prev = k[q];
  }
  return heads;
}

// exclusive prefix of `mine` over the 256 threads of the block (one barrier; s_w: 4 words of LDS)
__device__ __forceinline__ uint32_t tile_excl_prefix(uint32_t mine, uint32_t* s_w, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t inc = wave_inclusive_add(mine);
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  uint32_t b = inc - mine;
  for (int w = 0; w < wave; ++w) b += s_w[w];
  *total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  return b;
}

template <bool FILL>
__global__ __launch_bounds__(256) void k_root_tiles(const uint64_t* __restrict__ lin, int64_t n_alive,
                                                    uint32_t* __restrict__ tile_cnt,
                                                    uint64_t* __restrict__ vlin,
                                                    uint32_t* __restrict__ vstart) {
  __shared__ uint32_t s_w[4];
  const int64_t first = (int64_t)blockIdx.x * RT_TILE + (int64_t)threadIdx.x * RT_IPT;
  const uint32_t heads = tile_voxel_heads(lin, first, n_alive);
  uint32_t total;
  uint32_t r = tile_excl_prefix((uint32_t)__popc(heads), s_w, &total);
  if (!FILL) {
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = total;
    return;
  }
  // the tile's voxels through LDS: written out in runs (a scene of nearly one point per voxel has a head at
  // almost every position; from the threads' own registers that was eight strided stores each)
  __shared__ uint64_t s_lin[RT_TILE];
  __shared__ uint32_t s_pos[RT_TILE];
#pragma unroll
  for (int q = 0; q < RT_IPT; ++q) {
    if ((heads >> q) & 1u) {
      s_lin[r] = lin[first + q];
      s_pos[r] = (uint32_t)(first + q);
      ++r;
    }
  }
  __syncthreads();
  const uint32_t base = tile_cnt[blockIdx.x];  // (scanned, exclusive)
  for (uint32_t j = threadIdx.x; j < total; j += 256) {
    vlin[base + j] = s_lin[j];
    vstart[base + j] = s_pos[j];
  }
}

// level-0 buffers: position -> root node, point index (+scheme bit), path word
__global__ __launch_bounds__(256) void k_init_level0(
    const uint64_t* __restrict__ lin, const uint32_t* __restrict__ tile_first,
    const uint32_t* __restrict__ val_sorted, const uint32_t* __restrict__ path, int64_t n_alive,
    const int32_t* __restrict__ local2root, int32_t* __restrict__ pos_node,
    uint32_t* __restrict__ idx0, uint32_t* __restrict__ path0) {
  __shared__ uint32_t s_w[4];
  const int64_t first = (int64_t)blockIdx.x * RT_TILE + (int64_t)threadIdx.x * RT_IPT;
  const uint32_t heads = tile_voxel_heads(lin, first, n_alive);
  uint32_t total;
  const uint32_t before = tile_first[blockIdx.x] + tile_excl_prefix((uint32_t)__popc(heads), s_w, &total);
  uint32_t v[RT_IPT];
  if (first + RT_IPT <= n_alive) {
    const uint4* p = reinterpret_cast<const uint4*>(val_sorted + first);
    const uint4 a = p[0], b = p[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
#pragma unroll
    for (int q = 0; q < RT_IPT; ++q) v[q] = first + q < n_alive ? val_sorted[first + q] : 0u;
  }
  uint32_t pw[RT_IPT];
#pragma unroll
  for (int q = 0; q < RT_IPT; ++q) pw[q] = first + q < n_alive ? path[v[q] & IDX_MASK] : 0u;
  int32_t pn[RT_IPT];
#pragma unroll
  for (int q = 0; q < RT_IPT; ++q) {
    // voxels that start at or in front of this position, minus one
    const uint32_t local = before + (uint32_t)__popc(heads & ((2u << q) - 1u)) - 1u;
    pn[q] = first + q < n_alive ? (local2root ? local2root[local] : (int32_t)local) : 0;
  }
  if (first + RT_IPT <= n_alive) {  // whole run: 16-byte stores (a thread's eight positions are 32 / 64 bytes)
    int4* o_n = reinterpret_cast<int4*>(pos_node + first);
    o_n[0] = int4{pn[0], pn[1], pn[2], pn[3]};
    o_n[1] = int4{pn[4], pn[5], pn[6], pn[7]};
    uint4* o_i = reinterpret_cast<uint4*>(idx0 + first);
    o_i[0] = uint4{v[0], v[1], v[2], v[3]};
    o_i[1] = uint4{v[4], v[5], v[6], v[7]};
    uint4* o_p = reinterpret_cast<uint4*>(path0 + first);
    o_p[0] = uint4{pw[0], pw[1], pw[2], pw[3]};
    o_p[1] = uint4{pw[4], pw[5], pw[6], pw[7]};
  } else {
#pragma unroll
    for (int q = 0; q < RT_IPT; ++q) {
      const int64_t i = first + q;
      if (i < n_alive) {
        pos_node[i] = pn[q];
        idx0[i] = v[q];
        path0[i] = pw[q];
      }
    }
  }
}

// A fresh single cube (bare Octree / OctreeManager, mode 1) whose points are all alive needs none of the voxel
// machinery: there is ONE root, the store order is the level-0 order.  One pass replaces k_keygen + k_linkey +
// k_root_tiles x 2 + k_init_level0 (BASELINE config 4, 64 M points: 1.66 ms of its 7.2 ms -> 0.6 ms).
__global__ __launch_bounds__(256) void k_cube_level0(const double* __restrict__ xyz, int64_t n, double L, double c0x,
                                                     double c0y, double c0z, const int64_t* __restrict__ pose_off,
                                                     int n_poses, const uint8_t* __restrict__ scheme,
                                                     int exact_digits, int32_t* __restrict__ pos_node,
                                                     uint32_t* __restrict__ idx0, uint32_t* __restrict__ path0) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool exact_cube = exact_digits && nonneg_integer_below_2p45(L) && L >= 1.0 && nonneg_integer_below_2p45(c0x) &&
                          nonneg_integer_below_2p45(c0y) && nonneg_integer_below_2p45(c0z);
  const bool edge_pow2 = (__double_as_longlong(L) & 0xFFFFFFFFFFFFFll) == 0;
  const double px = xyz[3 * i + 0], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
  path0[i] = compute_path_root(px, py, pz, c0x, c0y, c0z, L, exact_cube, edge_pow2, 64.0 / L);
  uint32_t v = (uint32_t)i;
  if (!scheme || scheme[find_slot(pose_off, n_poses, i)]) v |= 0x80000000u;
  idx0[i] = v;
  pos_node[i] = 0;
}

// ---- a big single cube whose store was partitioned by its first pm levels (bucket_build.hip:
//      forest_prefix_partition): the complete top of the tree and the level buffers of level pm ------------------
// Depth d of the top tree has all 8^d nodes, node (d, path p) at id (8^d - 1) / 7 + p: exactly the numbering the
// level loop produces when every node above depth pm splits (children of the k-th internal node of a level are
// consecutive, levels follow each other) - which is what this path requires (*flag otherwise: the caller builds
// the plain way).  Ranges come from the bucket starts of the partition, corners and edges from the reference's
// own descent (octree.py:181-191).
__host__ __device__ __forceinline__ int64_t top_base(int d) { return ((int64_t(1) << (3 * d)) - 1) / 7; }

__global__ __launch_bounds__(256) void k_top_tree(const uint32_t* __restrict__ bstart, uint32_t bstride, int pm,
                                                  int64_t n_alive, int64_t K, double L, double c0x, double c0y,
                                                  double c0z, int cur_epoch, NodePtrs nd,
                                                  uint32_t* __restrict__ flag) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= top_base(pm + 1)) return;
  int d = 0;
  while (g >= top_base(d + 1)) ++d;
  const uint32_t p = (uint32_t)(g - top_base(d));
  const uint32_t nb = 1u << (3 * pm), span = 1u << (3 * (pm - d)), first = p * span;
  const uint32_t s = bstart[(size_t)first * bstride];
  const uint32_t e = (first + span < nb) ? bstart[(size_t)(first + span) * bstride] : (uint32_t)n_alive;
  nd.start[g] = s;
  nd.count[g] = e - s;
  nd.scount[g] = e - s;
  nd.depth[g] = d;
  nd.voxel[g] = 0;
  nd.parent[g] = d > 0 ? (int32_t)(top_base(d - 1) + (p >> 3)) : -1;
  nd.first_child[g] = d < pm ? (int32_t)(top_base(d + 1) + 8 * (int64_t)p) : -1;
  nd.old_id[g] = -1;
  nd.epoch[g] = d < pm ? cur_epoch : 0;
  double cx = c0x, cy = c0y, cz = c0z, ed = L;
  for (int t = 0; t < d; ++t) {
    const uint32_t dg = (p >> (3 * (d - 1 - t))) & 7u;
    const double h = ed / 2.0;
    cx = cx + ((dg & 4u) ? h : 0.0);
    cy = cy + ((dg & 2u) ? h : 0.0);
    cz = cz + ((dg & 1u) ? h : 0.0);
    ed = h;
  }
  nd.edge[g] = ed;
  nd.corner[3 * g + 0] = cx;
  nd.corner[3 * g + 1] = cy;
  nd.corner[3 * g + 2] = cz;
  // a node above depth pm that the count criterion would NOT split (octree.py:26) is a leaf in the reference:
  // the complete top tree does not apply
  if (d < pm && !((int64_t)(e - s) > K)) atomicExch(flag, 1u);
}

// level-pm buffers from the partition records: position = record, "index" = the record's own position (the level
// loop and the final gather then read coordinates out of the records, i.e. out of the 15 000-point neighbourhood
// of their depth-pm node instead of the whole store), path word = the digits of levels pm..5 the record carries
__global__ __launch_bounds__(256) void k_pre_level0(const uint4* __restrict__ recs, int64_t n, int pm,
                                                    int32_t* __restrict__ pos_node, uint32_t* __restrict__ idx0,
                                                    uint32_t* __restrict__ path0) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t vp = recs[2 * i + 1].z;
  const uint32_t d18 = (vp >> 1) & 0x3FFFFu;
  idx0[i] = (uint32_t)i | 0x80000000u;
  path0[i] = ((d18 & ((1u << (3 * (6 - pm))) - 1u)) << 14) | (vp & 1u);
  pos_node[i] = (int32_t)(top_base(pm) + (d18 >> (18 - 3 * pm)));
}

// k_finalize over the records: the leaf-ordered permutation holds the records' ORIGINAL store indices
__global__ __launch_bounds__(256) void k_finalize_rec(const int32_t* __restrict__ pos_node,
                                                      const int32_t* __restrict__ depth,
                                                      const uint32_t* __restrict__ idx_a,
                                                      const uint32_t* __restrict__ idx_b,
                                                      const uint4* __restrict__ recs, int64_t n_alive,
                                                      uint32_t* __restrict__ ord_idx,
                                                      double* __restrict__ xyz_ord) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_alive) return;
  const int d = depth[pos_node[i]];
  const uint32_t v = ((d & 1) ? idx_b[i] : idx_a[i]) & IDX_MASK;
  const uint4 a = recs[2 * (size_t)v], b = recs[2 * (size_t)v + 1];
  ord_idx[i] = b.w & IDX_MASK;
  uint2* o = reinterpret_cast<uint2*>(xyz_ord + 3 * i);
  o[0] = uint2{a.x, a.y};
  o[1] = uint2{a.z, a.w};
  o[2] = uint2{b.x, b.y};
}

// scheme-pose point count of every root (only when a pose subset drives the scheme)
__global__ __launch_bounds__(256) void k_count_scheme(const int32_t* __restrict__ pos_node,
                                                      const uint32_t* __restrict__ idx0,
                                                      int64_t n_alive,
                                                      uint32_t* __restrict__ scount) {
  __shared__ uint32_t part[4];
  const int64_t base = (int64_t)blockIdx.x * 2048;
  const int64_t last = min(base + 2048, n_alive) - 1;
  const bool uniform = pos_node[base] == pos_node[last];  // positions are sorted by node
  uint32_t c = 0;
  for (int r = 0; r < 8; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    if (i < n_alive && (idx0[i] >> 31)) {
      if (uniform) ++c; else atomicAdd(&scount[pos_node[i]], 1u);
    }
  }
  if (uniform) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t t = part[0] + part[1] + part[2] + part[3];
      if (t) atomicAdd(&scount[pos_node[base]], t);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// level loop
// ---------------------------------------------------------------------------------------------
// roots of a forest without a previous scheme, built on the device (no PCIe round trip)
__global__ __launch_bounds__(256) void k_make_roots(const uint64_t* __restrict__ vlin,
                                                    const uint32_t* __restrict__ vstart, int64_t V,
                                                    int64_t n_alive, int mode, double L, double c0x,
                                                    double c0y, double c0z, int minx, int miny,
                                                    int minz, uint64_t ny, uint64_t nz,
                                                    int scount_is_count, NodePtrs nd) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= V) return;
  const uint32_t s = vstart[r];
  const uint32_t e = (r + 1 < V) ? vstart[r + 1] : (uint32_t)n_alive;
  nd.start[r] = s;
  nd.count[r] = e - s;
  nd.scount[r] = scount_is_count ? e - s : 0u;
  nd.depth[r] = 0;
  nd.voxel[r] = (int32_t)r;
  nd.parent[r] = -1;
  nd.first_child[r] = -1;
  nd.old_id[r] = -1;
  nd.epoch[r] = 0;
  nd.edge[r] = L;
  if (mode == 0) {
    const uint64_t l = vlin[r];
    const int64_t qz = (int64_t)(l % nz) + minz, qy = (int64_t)((l / nz) % ny) + miny,
                  qx = (int64_t)(l / (nz * ny)) + minx;
    // np.array(voxel_coordinates): int64(q * L), L integer valued (grid.py:72-76,104)
    nd.corner[3 * r + 0] = (double)(long long)((double)qx * L);
    nd.corner[3 * r + 1] = (double)(long long)((double)qy * L);
    nd.corner[3 * r + 2] = (double)(long long)((double)qz * L);
  } else {
    nd.corner[3 * r + 0] = c0x;
    nd.corner[3 * r + 1] = c0y;
    nd.corner[3 * r + 2] = c0z;
  }
}

// split predicate for the freshly created nodes [first, first+n_new)
//   K mode   : scheme-pose count > K                       (octree.py:26, octree_manager.py:53-61)
//   keep mode: the node is internal in the previous scheme (octree.py:39-47, subdivide_as)
__global__ __launch_bounds__(256) void k_split_flags(NodePtrs nd, int64_t first, int64_t n_new,
                                                     int keep_mode, int64_t K,
                                                     const int32_t* __restrict__ old_fc, int leaves_only,
                                                     uint32_t* __restrict__ flags) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_new) return;
  const int64_t c = first + j;
  bool split;
  if (keep_mode) {
    const int32_t o = nd.old_id[c];
    split = (o >= 0) && (old_fc[o] >= 0);
  } else {
    split = (K >= 0) && ((int64_t)nd.scount[c] > K) && !(leaves_only && nd.first_child[c] >= 0);
  }
  flags[j] = split ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_compact_split(NodePtrs nd, int64_t first, int64_t n_new,
                                                       const uint32_t* __restrict__ flags_scanned,
                                                       int keep_mode, int64_t K,
                                                       const int32_t* __restrict__ old_fc, int leaves_only,
                                                       int32_t* __restrict__ split_nodes,
                                                       uint32_t* __restrict__ split_tiles) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_new) return;
  const int64_t c = first + j;
  bool split;
  if (keep_mode) {
    const int32_t o = nd.old_id[c];
    split = (o >= 0) && (old_fc[o] >= 0);
  } else {
    split = (K >= 0) && ((int64_t)nd.scount[c] > K) && !(leaves_only && nd.first_child[c] >= 0);
  }
  if (split) {
    const uint32_t pos = flags_scanned[j];
    split_nodes[pos] = (int32_t)c;
    split_tiles[pos] = (nd.count[c] + LV_TILE - 1) / LV_TILE;
  }
}

// tile -> (split slot, tile inside the node, tiles of the node)
struct TileRef {
  int s;
  uint32_t tl, nt, tile_base;
};
__device__ __forceinline__ TileRef locate_tile(const uint32_t* __restrict__ tile_base, int ns,
                                               uint32_t n_tiles_total, uint32_t t) {
  int lo = 0, hi = ns;  // tile_base[lo] <= t < tile_base[hi] (tile_base[ns] = total)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tile_base[mid] <= t) lo = mid; else hi = mid;
  }
  // nodes without points own zero tiles: several slots may share one tile_base; the owner of
  // tile t is the LAST slot whose base is <= t, which the search above returns
  TileRef r;
  r.s = lo;
  r.tile_base = tile_base[lo];
  r.tl = t - r.tile_base;
  r.nt = ((lo + 1 < ns) ? tile_base[lo + 1] : n_tiles_total) - r.tile_base;
  return r;
}

// digit of a point whose precomputed path is unusable: redo the reference arithmetic against
// the node's own cube; a point outside the cube of a node that is being split is the
// reference's IndexError / wrong-child case (octree.py:94-98) -> domain error
// (xs: doubles per point of the coordinate source - 3 for the store, 4 for the 32-byte partition records)
__device__ __forceinline__ uint32_t slow_digit(const double* __restrict__ xyz, int xs, uint32_t v,
                                               const double* __restrict__ corner, double e,
                                               uint32_t* __restrict__ small) {
  const int64_t i = (int64_t)(v & IDX_MASK);
  const double ax = xyz[xs * i] - corner[0], ay = xyz[xs * i + 1] - corner[1],
               az = xyz[xs * i + 2] - corner[2];
  const double h = e / 2.0;
  const bool ok = (ax >= 0.0) && (ax < e) && (ay >= 0.0) && (ay < e) && (az >= 0.0) && (az < e);
  if (!ok) {
    atomicExch(&small[SM_ERR], (uint32_t)(-OCTL_E_DOMAIN));
    return 0;
  }
  return ((ax >= h) ? 4u : 0u) | ((ay >= h) ? 2u : 0u) | ((az >= h) ? 1u : 0u);
}

// recompute the path words of the points of the nodes about to be split, relative to those
// nodes (every 21 levels)
__global__ __launch_bounds__(LV_THREADS) void k_lv_rekey(
    const int32_t* __restrict__ split_nodes, const uint32_t* __restrict__ tile_base, int ns,
    uint32_t n_tiles, NodePtrs nd, const uint32_t* __restrict__ idx_in,
    uint32_t* __restrict__ path_io, const double* __restrict__ xyz, int xs) {
  const TileRef tr = locate_tile(tile_base, ns, n_tiles, blockIdx.x);
  const int32_t node = split_nodes[tr.s];
  const uint32_t nstart = nd.start[node], nend = nstart + nd.count[node];
  const double e = nd.edge[node];
  const double cx = nd.corner[3 * (int64_t)node], cy = nd.corner[3 * (int64_t)node + 1],
               cz = nd.corner[3 * (int64_t)node + 2];
  for (int r = 0; r < LV_IPT; ++r) {
    const uint32_t i = nstart + tr.tl * LV_TILE + r * LV_THREADS + threadIdx.x;
    if (i < nend) {
      const int64_t p = (int64_t)(idx_in[i] & IDX_MASK);
      path_io[i] = compute_path(xyz[xs * p], xyz[xs * p + 1], xyz[xs * p + 2], cx, cy, cz, e);
    }
  }
}

// level-0 buffer of the roots a resumed build subdivides: store index | scheme bit of the points in
// the root's range of the leaf-ordered permutation
__global__ __launch_bounds__(LV_THREADS) void k_lv_init_idx(
    const int32_t* __restrict__ split_nodes, const uint32_t* __restrict__ tile_base, int ns,
    uint32_t n_tiles, NodePtrs nd, const uint32_t* __restrict__ ord_idx,
    const int64_t* __restrict__ pose_off, int n_poses, const uint8_t* __restrict__ scheme,
    uint32_t* __restrict__ idx_out) {
  const TileRef tr = locate_tile(tile_base, ns, n_tiles, blockIdx.x);
  const int32_t node = split_nodes[tr.s];
  const uint32_t nstart = nd.start[node], nend = nstart + nd.count[node];
  for (int r = 0; r < LV_IPT; ++r) {
    const uint32_t i = nstart + tr.tl * LV_TILE + r * LV_THREADS + threadIdx.x;
    if (i < nend) {
      uint32_t v = ord_idx[i];
      if (!scheme || scheme[find_slot(pose_off, n_poses, (int64_t)v)]) v |= 0x80000000u;
      idx_out[i] = v;
    }
  }
}

__global__ __launch_bounds__(256) void k_mark_nodes(const int32_t* __restrict__ nodes, int n,
                                                    uint8_t* __restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[nodes[i]] = 1;
}

template <bool SCHEME_SUBSET>
__global__ __launch_bounds__(LV_THREADS) void k_lv_hist(
    const int32_t* __restrict__ split_nodes, const uint32_t* __restrict__ tile_base, int ns,
    uint32_t n_tiles, NodePtrs nd, const uint32_t* __restrict__ idx_in,
    const uint32_t* __restrict__ path_in, const double* __restrict__ xyz, int xs, int shift,
    uint32_t* __restrict__ entries, uint32_t* __restrict__ child_sc,
    uint32_t* __restrict__ small) {
  __shared__ uint32_t wc[LV_WAVES][8], wsc[LV_WAVES][8];
  const TileRef tr = locate_tile(tile_base, ns, n_tiles, blockIdx.x);
  const int32_t node = split_nodes[tr.s];
  const uint32_t nstart = nd.start[node], nend = nstart + nd.count[node];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t wbase = nstart + tr.tl * LV_TILE + wave * LV_WAVE_ITEMS;
  uint32_t c[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < LV_IPT; ++r) {
    const uint32_t i = wbase + r * 64 + lane;
    const bool valid = i < nend;
    uint32_t d = 0, v = 0;
    if (valid) {
      const uint32_t pw = path_in[i];
      v = idx_in[i];
      d = (pw & 1u) ? slow_digit(xyz, xs, v, nd.corner + 3 * (int64_t)node, nd.edge[node], small)
                    : (pw >> shift) & 7u;
    }
    const uint64_t sm = SCHEME_SUBSET ? __ballot(valid && (v >> 31)) : 0ull;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const uint64_t m = __ballot(valid && d == (uint32_t)b);
      c[b] += __popcll(m);
      if (SCHEME_SUBSET) sc[b] += __popcll(m & sm);
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      wc[wave][b] = c[b];
      if (SCHEME_SUBSET) wsc[wave][b] = sc[b];
    }
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    const int b = threadIdx.x;
    uint32_t t = 0, ts = 0;
#pragma unroll
    for (int w = 0; w < LV_WAVES; ++w) {
      t += wc[w][b];
      if (SCHEME_SUBSET) ts += wsc[w][b];
    }
    entries[(size_t)8 * tr.tile_base + (size_t)b * tr.nt + tr.tl] = t;
    if (SCHEME_SUBSET && ts) atomicAdd(&child_sc[(size_t)8 * tr.s + b], ts);
  }
}

__global__ __launch_bounds__(LV_THREADS) void k_lv_scatter(
    const int32_t* __restrict__ split_nodes, const uint32_t* __restrict__ tile_base, int ns,
    uint32_t n_tiles, NodePtrs nd, const uint32_t* __restrict__ idx_in,
    const uint32_t* __restrict__ path_in, const double* __restrict__ xyz, int xs, int shift,
    const uint32_t* __restrict__ entries_scanned, int32_t child_base,
    uint32_t* __restrict__ idx_out, uint32_t* __restrict__ path_out,
    int32_t* __restrict__ pos_node, uint32_t* __restrict__ small, int64_t K_leaf) {
  __shared__ uint32_t cnt[LV_WAVES][8];
  // position -> node is only READ behind the level loop (k_finalize, the block table): a point that moves into a
  // child which splits again gets its entry from that deeper level, so only children that stay LEAVES write it
  // here.  K_leaf >= 0: a child stays a leaf iff its count is at most K_leaf (count-driven split over all poses:
  // scount == count); K_leaf < 0: every child writes (scheme from a pose subset, keep_scheme, resumed builds).
  // BASELINE config 4 (64 M points, 5 levels): 4 bytes per point and level less, 1 GB of 9 GB of the loop.
  __shared__ uint32_t leaf_child[8];
  const TileRef tr = locate_tile(tile_base, ns, n_tiles, blockIdx.x);
  const int32_t node = split_nodes[tr.s];
  const uint32_t nstart = nd.start[node], nend = nstart + nd.count[node];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x < LV_WAVES * 8) cnt[threadIdx.x >> 3][threadIdx.x & 7] = 0;
  __syncthreads();
  const uint32_t wbase = nstart + tr.tl * LV_TILE + wave * LV_WAVE_ITEMS;
  uint32_t v[LV_IPT], d[LV_IPT], rank[LV_IPT];
  uint32_t pw[LV_IPT];
#pragma unroll
  for (int r = 0; r < LV_IPT; ++r) {
    const uint32_t i = wbase + r * 64 + lane;
    const bool valid = i < nend;
    v[r] = 0; d[r] = 0; pw[r] = 0;
    if (valid) {
      pw[r] = path_in[i];
      v[r] = idx_in[i];
      d[r] = (pw[r] & 1u)
                 ? slow_digit(xyz, xs, v[r], nd.corner + 3 * (int64_t)node, nd.edge[node], small)
                 : (pw[r] >> shift) & 7u;
    }
    rank[r] = wave_stable_rank<3>(d[r], valid, cnt[wave]);
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    const int b = threadIdx.x;
    const size_t ebase = (size_t)8 * tr.tile_base;
    {
      // points of child b of the node (as k_make_children counts them)
      const uint32_t lo = entries_scanned[ebase + (size_t)b * tr.nt];
      const uint32_t hi = (b < 7 || (size_t)8 * (tr.tile_base + tr.nt) < (size_t)8 * n_tiles)
                              ? entries_scanned[ebase + (size_t)(b + 1) * tr.nt]
                              : small[SM_ETOTAL];
      leaf_child[b] = (K_leaf < 0 || (int64_t)(hi - lo) <= K_leaf) ? 1u : 0u;
    }
    // destination of the first item with digit b of this tile, relative to the node start
    uint32_t run = entries_scanned[ebase + (size_t)b * tr.nt + tr.tl] - entries_scanned[ebase];
#pragma unroll
    for (int w = 0; w < LV_WAVES; ++w) {
      const uint32_t t = cnt[w][b];
      cnt[w][b] = run;
      run += t;
    }
  }
  __syncthreads();
  const int32_t first_child = child_base + 8 * tr.s;
#pragma unroll
  for (int r = 0; r < LV_IPT; ++r) {
    const uint32_t i = wbase + r * 64 + lane;
    if (i < nend) {
      const uint32_t dst = nstart + cnt[wave][d[r]] + rank[r];
      idx_out[dst] = v[r];
      path_out[dst] = pw[r];
      if (leaf_child[d[r]]) pos_node[dst] = first_child + (int32_t)d[r];
    }
  }
}

// one thread per (split node, child): create the 8 children (octree.py:177-191)
__global__ __launch_bounds__(256) void k_make_children(
    const int32_t* __restrict__ split_nodes, const uint32_t* __restrict__ tile_base, int ns,
    uint32_t n_tiles, NodePtrs nd, const uint32_t* __restrict__ entries_scanned,
    const uint32_t* __restrict__ small, const uint32_t* __restrict__ child_sc, int all_scheme,
    int32_t child_base, int keep_mode, int cur_epoch, const int32_t* __restrict__ old_fc,
    const int32_t* __restrict__ old_epoch) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (int64_t)8 * ns) return;
  const int s = (int)(g >> 3), j = (int)(g & 7);
  const int32_t node = split_nodes[s];
  const uint32_t tb = tile_base[s];
  const uint32_t nt = ((s + 1 < ns) ? tile_base[s + 1] : n_tiles) - tb;
  uint32_t off = 0, cnt = 0;
  if (nt > 0) {
    const size_t ebase = (size_t)8 * tb;
    const uint32_t a = entries_scanned[ebase + (size_t)j * nt];
    const uint32_t b = (j < 7 || (size_t)8 * (tb + nt) < (size_t)8 * n_tiles)
                           ? entries_scanned[ebase + (size_t)(j + 1) * nt]
                           : small[SM_ETOTAL];
    off = a - entries_scanned[ebase];
    cnt = b - a;
  }
  const int64_t c = (int64_t)child_base + 8 * (int64_t)s + j;
  nd.start[c] = nd.start[node] + off;
  nd.count[c] = cnt;
  nd.scount[c] = all_scheme ? cnt : child_sc[(size_t)8 * s + j];
  nd.depth[c] = nd.depth[node] + 1;
  nd.voxel[c] = nd.voxel[node];
  nd.parent[c] = node;
  nd.first_child[c] = -1;
  nd.epoch[c] = 0;
  const int32_t po = nd.old_id[node];
  const bool parent_was_internal = (po >= 0) && (old_fc[po] >= 0);
  nd.old_id[c] = parent_was_internal ? old_fc[po] + j : -1;
  // child_edge_length = edge / np.float_(2); corner = corner_min + offset, offsets from
  // itertools.product([0, child_edge], repeat=3): x is the slowest axis
  const double h = nd.edge[node] / 2.0;
  nd.edge[c] = h;
  nd.corner[3 * c + 0] = nd.corner[3 * (int64_t)node + 0] + ((j & 4) ? h : 0.0);
  nd.corner[3 * c + 1] = nd.corner[3 * (int64_t)node + 1] + ((j & 2) ? h : 0.0);
  nd.corner[3 * c + 2] = nd.corner[3 * (int64_t)node + 2] + ((j & 1) ? h : 0.0);
  if (j == 0) {
    nd.first_child[node] = child_base + 8 * s;
    nd.epoch[node] = parent_was_internal ? old_epoch[po] : (keep_mode ? 0 : cur_epoch);
  }
}

// ---------------------------------------------------------------------------------------------
// finalize
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_finalize(const int32_t* __restrict__ pos_node,
                                                  const int32_t* __restrict__ depth,
                                                  const uint32_t* __restrict__ idx_a,
                                                  const uint32_t* __restrict__ idx_b,
                                                  const double* __restrict__ xyz, int64_t n_alive,
                                                  uint32_t* __restrict__ ord_idx,
                                                  double* __restrict__ xyz_ord) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_alive) return;
  const int d = depth[pos_node[i]];
  const uint32_t v = ((d & 1) ? idx_b[i] : idx_a[i]) & IDX_MASK;
  ord_idx[i] = v;
  const double x = xyz[3 * (int64_t)v], y = xyz[3 * (int64_t)v + 1], z = xyz[3 * (int64_t)v + 2];
  xyz_ord[3 * i] = x;
  xyz_ord[3 * i + 1] = y;
  xyz_ord[3 * i + 2] = z;
}

// the same for the points of the roots a resumed build has subdivided (root_flag), everything else
// was written by the bucket build
__global__ __launch_bounds__(256) void k_finalize_marked(const int32_t* __restrict__ pos_node,
                                                         const int32_t* __restrict__ depth,
                                                         const int32_t* __restrict__ voxel,
                                                         const uint8_t* __restrict__ root_flag,
                                                         const uint32_t* __restrict__ idx_a,
                                                         const uint32_t* __restrict__ idx_b,
                                                         const double* __restrict__ xyz, int64_t n_alive,
                                                         uint32_t* __restrict__ ord_idx,
                                                         double* __restrict__ xyz_ord) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_alive) return;
  const int32_t node = pos_node[i];
  if (!root_flag[voxel[node]]) return;
  const int d = depth[node];
  const uint32_t v = ((d & 1) ? idx_b[i] : idx_a[i]) & IDX_MASK;
  ord_idx[i] = v;
  xyz_ord[3 * i] = xyz[3 * (int64_t)v];
  xyz_ord[3 * i + 1] = xyz[3 * (int64_t)v + 1];
  xyz_ord[3 * i + 2] = xyz[3 * (int64_t)v + 2];
}

// ---- (leaf, pose) block table of the leaf-ordered arrays ------------------------------------------------------
// A block starts where the leaf or the pose of a position differs from the position in front of it.  One
// workgroup per tile of BLK_TILE consecutive positions, eight per thread; the poses' first indices sit in LDS
// (pose of a stored index = binary search there).  k_block_count<false> leaves the heads per tile,
// k_block_count<true> - after a scan over the TILES, not over the positions - writes the blocks.  (Round 2
// flagged every position, scanned n flags and searched the poses four times per position in global memory:
// 1.67 ms of BASELINE config 4's 8.9 ms, 64 M positions.)
constexpr int BLK_IPT = 8;
constexpr int BLK_TILE = 256 * BLK_IPT;
constexpr int BLK_LDS_POSES = 2048;

template <bool FILL>
__global__ __launch_bounds__(256) void k_block_tiles(const int32_t* __restrict__ pos_node,
                                                     const uint32_t* __restrict__ ord_idx,
                                                     const int64_t* __restrict__ pose_off, int n_poses,
                                                     int64_t n_alive, uint32_t* __restrict__ tile_cnt,
                                                     int32_t* __restrict__ blk_node, int32_t* __restrict__ blk_slot,
                                                     uint32_t* __restrict__ blk_start) {
  __shared__ int64_t s_off[BLK_LDS_POSES + 1];
  __shared__ uint32_t s_w[4];
  const bool in_lds = n_poses <= BLK_LDS_POSES;
  if (in_lds)
    for (int p = threadIdx.x; p <= n_poses; p += 256) s_off[p] = pose_off[p];
  __syncthreads();
  auto slot_of = [&](uint32_t idx) {
    if (n_poses <= 1) return 0;
    int lo = 0, hi = n_poses;  // off[lo] <= idx < off[hi]
    if (in_lds) {
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_off[mid] <= (int64_t)idx) lo = mid; else hi = mid;
      }
    } else {
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (pose_off[mid] <= (int64_t)idx) lo = mid; else hi = mid;
      }
    }
    return lo;
  };
  const int64_t first = (int64_t)blockIdx.x * BLK_TILE + (int64_t)threadIdx.x * BLK_IPT;
  int32_t node[BLK_IPT];
  uint32_t idx[BLK_IPT];
  if (first + BLK_IPT <= n_alive) {  // (the arrays are 16-byte aligned, first is a multiple of 8)
    const int4* pn = reinterpret_cast<const int4*>(pos_node + first);
    const uint4* oi = reinterpret_cast<const uint4*>(ord_idx + first);
    const int4 a = pn[0], b = pn[1];
    const uint4 c = oi[0], d = oi[1];
    node[0] = a.x; node[1] = a.y; node[2] = a.z; node[3] = a.w;
    node[4] = b.x; node[5] = b.y; node[6] = b.z; node[7] = b.w;
    idx[0] = c.x; idx[1] = c.y; idx[2] = c.z; idx[3] = c.w;
    idx[4] = d.x; idx[5] = d.y; idx[6] = d.z; idx[7] = d.w;
  } else {
#pragma unroll
    for (int q = 0; q < BLK_IPT; ++q) {
      const int64_t i = first + q;
      node[q] = i < n_alive ? pos_node[i] : -1;
      idx[q] = i < n_alive ? ord_idx[i] : 0u;
    }
  }
  // the position in front of this thread's run
  int32_t pnode = -1;
  int pslot = -1;
  if (first > 0 && first < n_alive) {
    pnode = pos_node[first - 1];
    pslot = slot_of(ord_idx[first - 1]);
  }
  uint32_t heads = 0;  // bit q: position first + q starts a block
  int slot[BLK_IPT];
#pragma unroll
  for (int q = 0; q < BLK_IPT; ++q) {
    const bool live = first + q < n_alive;
    // (inside a leaf the positions are in index order: the pose changes only where the index passes the end
    //  of the previous position's pose, so most positions inherit the slot without a search)
    int sl = pslot;
    if (live && (pslot < 0 || (int64_t)idx[q] >= (in_lds ? s_off[pslot + 1] : pose_off[pslot + 1]) ||
                 (int64_t)idx[q] < (in_lds ? s_off[pslot] : pose_off[pslot])))
      sl = slot_of(idx[q]);
    slot[q] = sl;
    if (live && (first + q == 0 || node[q] != pnode || sl != pslot)) heads |= 1u << q;
    pnode = node[q];
    pslot = sl;
  }
  const uint32_t mine = (uint32_t)__popc(heads);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t inc = wave_inclusive_add(mine);
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  if (!FILL) {
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    return;
  }
  uint32_t b = tile_cnt[blockIdx.x] + inc - mine;  // (tile_cnt: scanned, exclusive)
  for (int w = 0; w < wave; ++w) b += s_w[w];
#pragma unroll
  for (int q = 0; q < BLK_IPT; ++q) {
    if ((heads >> q) & 1u) {
      blk_node[b] = node[q];
      blk_slot[b] = slot[q];
      blk_start[b] = (uint32_t)(first + q);
      ++b;
    }
  }
}

__global__ __launch_bounds__(256) void k_block_sizes(const uint32_t* __restrict__ blk_start,
                                                     const uint32_t* __restrict__ nb_dev,
                                                     int64_t n_alive,
                                                     int32_t* __restrict__ blk_size) {
  const int64_t nb = (int64_t)*nb_dev;
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  const uint32_t e = (b + 1 < nb) ? blk_start[b + 1] : (uint32_t)n_alive;
  blk_size[b] = (int32_t)(e - blk_start[b]);
}

inline unsigned grid_for(int64_t n, int threads = 256) { return (unsigned)ceil_div(n, threads); }

int bits_for(uint64_t max_value) {
  int b = 0;
  while (b < 64 && (max_value >> b) != 0) ++b;
  return b;
}

}  // namespace

int nodes_reserve(octl_ctx* ctx, NodeTable& t, int64_t cap) {
  if (cap <= t.cap) return OCTL_OK;
  const int64_t want = std::max<int64_t>(cap + cap / 2, 1024);
  for (DevBuf* b : {&t.start, &t.count, &t.scount, &t.depth, &t.voxel, &t.parent, &t.first_child,
                    &t.old_id, &t.epoch})
    OCTL_TRY(devbuf_reserve(ctx, *b, (size_t)want * 4, 1));
  OCTL_TRY(devbuf_reserve(ctx, t.corner, (size_t)want * 24, 1));
  OCTL_TRY(devbuf_reserve(ctx, t.edge, (size_t)want * 8, 1));
  t.cap = want;
  return OCTL_OK;
}

void nodes_free(octl_ctx* ctx, NodeTable& t) {
  for (DevBuf* b : {&t.start, &t.count, &t.scount, &t.depth, &t.voxel, &t.parent, &t.first_child,
                    &t.old_id, &t.epoch, &t.corner, &t.edge})
    devbuf_release(ctx, *b);
  t.cap = t.n = 0;
}

// read `count` uint32 scalars of the small block (synchronises the stream)
static int read_small(octl_ctx* ctx, int first, int count, uint32_t* out) {
  HIP_TRY(ctx, hipMemcpyAsync(ctx->small_host, ctx->small.as<uint32_t>() + first,
                              (size_t)count * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  std::memcpy(out, ctx->small_host, (size_t)count * 4);
  return OCTL_OK;
}

// (Re)build the (leaf, pose) block table from pos_node / ord_idx.  Asynchronous: the block count
// is left in small[SM_NBLOCKS]; forest_finish_blocks reads it (one synchronisation).
int forest_make_blocks(octl_forest* f) {
  octl_ctx* ctx = f->ctx;
  hipStream_t st = ctx->stream;
  const int64_t n = f->n_ord;
  const int n_poses = (int)f->pose_off.size() - 1;
  f->n_blocks = 0;
  uint32_t* small = ctx->small.as<uint32_t>();
  if (n <= 0) {
    HIP_TRY(ctx, hipMemsetAsync(small + SM_NBLOCKS, 0, 4, st));
    return OCTL_OK;
  }
  KTimer t(ctx, "blocks");
  const int64_t n_tiles = ceil_div(n, (int64_t)BLK_TILE);
  OCTL_TRY(devbuf_reserve(ctx, f->flags, (size_t)(n_tiles + 8) * 4));
  uint32_t* tile_cnt = f->flags.as<uint32_t>();
  const int32_t* pos_node = f->pos_node.as<int32_t>();
  // capacity = one block per point (grow-only buffers): the count is not known on the host yet
  OCTL_TRY(devbuf_reserve(ctx, f->blk_node, (size_t)n * 4));
  OCTL_TRY(devbuf_reserve(ctx, f->blk_slot, (size_t)n * 4));
  OCTL_TRY(devbuf_reserve(ctx, f->blk_start, (size_t)n * 4));
  OCTL_TRY(devbuf_reserve(ctx, f->blk_size, (size_t)n * 4));
  OCTL_LAUNCH(k_block_tiles<false>, dim3((unsigned)n_tiles), dim3(256), 0, st, pos_node,
                     (const uint32_t*)f->ord_idx.as<uint32_t>(), (const int64_t*)f->pose_off_dev.as<int64_t>(),
                     n_poses, n, tile_cnt, (int32_t*)nullptr, (int32_t*)nullptr, (uint32_t*)nullptr);
  HIP_TRY(ctx, hipGetLastError());
  OCTL_TRY(octl_exclusive_scan_u32(ctx, tile_cnt, tile_cnt, n_tiles, small + SM_NBLOCKS));
  OCTL_LAUNCH(k_block_tiles<true>, dim3((unsigned)n_tiles), dim3(256), 0, st, pos_node,
                     (const uint32_t*)f->ord_idx.as<uint32_t>(), (const int64_t*)f->pose_off_dev.as<int64_t>(),
                     n_poses, n, tile_cnt, f->blk_node.as<int32_t>(), f->blk_slot.as<int32_t>(),
                     f->blk_start.as<uint32_t>());
  HIP_TRY(ctx, hipGetLastError());
  OCTL_LAUNCH(k_block_sizes, dim3(grid_for(n)), dim3(256), 0, st,
                     (const uint32_t*)f->blk_start.as<uint32_t>(),
                     (const uint32_t*)(small + SM_NBLOCKS), n, f->blk_size.as<int32_t>());
  HIP_TRY(ctx, hipGetLastError());
  return OCTL_OK;
}

int forest_finish_blocks(octl_forest* f, uint32_t* err_out) {
  uint32_t sm[32];
  OCTL_TRY(read_small(f->ctx, 0, 32, sm));
  f->n_blocks = sm[SM_NBLOCKS];
  if (err_out) *err_out = sm[SM_ERR];
  return OCTL_OK;
}

int forest_sync_vkeys(octl_forest* f) {
  if (!f->vkeys_stale) return OCTL_OK;
  octl_ctx* ctx = f->ctx;
  if (f->vcode_valid) {  // the packed keys are on the device already (incremental.hip)
    f->vkeys.resize((size_t)f->n_voxels);
    if (f->n_voxels > 0) {
      HIP_TRY(ctx, hipMemcpyAsync(f->vkeys.data(), f->vcode_dev[0].p, (size_t)f->n_voxels * 8,
                                  hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    f->vkeys_stale = false;
    return OCTL_OK;
  }
  std::vector<uint64_t> lin((size_t)f->n_voxels);
  if (f->n_voxels > 0) {
    HIP_TRY(ctx, hipMemcpyAsync(lin.data(), f->vlin_dev.p, (size_t)f->n_voxels * 8,
                                hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  f->vkeys.resize((size_t)f->n_voxels);
  for (int64_t v = 0; v < f->n_voxels; ++v) {
    const uint64_t l = lin[v];
    const uint64_t qz = l % f->vl_nz, qy = (l / f->vl_nz) % f->vl_ny, qx = l / (f->vl_nz * f->vl_ny);
    f->vkeys[v] = vkey_pack((int64_t)qx + f->vl_min[0], (int64_t)qy + f->vl_min[1], (int64_t)qz + f->vl_min[2],
                            f->vorg);
  }
  f->vkeys_stale = false;
  return OCTL_OK;
}

// The level loop: level-synchronous recursive subdivide of the nodes [first_new, first_new + n_new) and
// of everything they spawn.  resume = true: the bucket build has already produced the node table and
// the leaf-ordered arrays; only roots that are still leaves with more than K scheme points split (the
// voxels it left behind), and the level buffers are set up for them here.
struct LevelLoop {
  octl_forest* f;
  NodeTable* nt;
  int64_t K;
  int keep_scheme;
  bool all_scheme;
  const uint8_t* scheme_dev;
  const int32_t *old_fc, *old_epoch;
  int cur_epoch, max_depth;
  int64_t n_alive;
  bool resume;
  int64_t first_new, n_new;
  int64_t n_internal;
  int level;
  std::vector<octl_forest::LevelSeg>* segs;
  // coordinates behind the loop's point indices: the store (3 doubles per point, nullptr = f->xyz) or the 32-byte
  // records of a prefix partition (4 doubles per point; the indices are record positions then)
  const double* gx = nullptr;
  int xs = 3;
};

// The first launch of a build (round 5: was the scalar block's copy, k_geom_set and - from octl_forest_clear -
// k_bbox_reset, three launches): the 32 scalars out of the pinned block, the geometry hint of the context's previous
// build behind them when the build may run under it (bucket_build.hip decides; staging it costs nothing), and the
// reset of the forest's voxel box when the build is going to fill it (a cloud taken in place).
__global__ void k_build_begin(const uint32_t* __restrict__ src, uint32_t* __restrict__ small, int hint_words,
                              int32_t* __restrict__ bbox) {
  const int t = threadIdx.x;
  if (t < 32) small[t] = src[t];
  if (t >= 32 && t - 32 < hint_words) small[SM_GEOM + t - 32] = src[SM_GEOM + t - 32];
  if (bbox && t >= 120 && t < 128) {
    const int a = t - 120;
    bbox[a] = a < 3 ? (1 << 30) : (a < 6 ? -(1 << 30) : 0);
  }
}

static int run_level_loop(LevelLoop& L) {
  octl_forest* f = L.f;
  octl_ctx* ctx = f->ctx;
  hipStream_t st = ctx->stream;
  NodeTable& nt = *L.nt;
  NodePtrs nd = node_ptrs(nt);
  uint32_t* small = ctx->small.as<uint32_t>();
  uint32_t* flags = nullptr;
  int32_t* pos_node = f->pos_node.as<int32_t>();
  const double* gx = L.gx ? L.gx : (const double*)f->xyz.as<double>();
  while (L.n_new > 0) {
    // split list of the freshly created nodes
    OCTL_TRY(devbuf_reserve(ctx, f->flags, (size_t)(std::max<int64_t>(L.n_new, L.n_alive) + 8) * 4));
    flags = f->flags.as<uint32_t>();
    for (int b = 0; b < 2; ++b) {
      OCTL_TRY(devbuf_reserve(ctx, f->split[b], (size_t)(L.n_new + 8) * 4));
      OCTL_TRY(devbuf_reserve(ctx, f->split_tiles[b], (size_t)(L.n_new + 8) * 4));
    }
    int32_t* split_nodes = f->split[0].as<int32_t>();
    uint32_t* tile_base = f->split_tiles[0].as<uint32_t>();
    {
      KTimer t(ctx, "level_prepare");
      const int leaves_only = (L.resume && L.level == 0) ? 1 : 0;
      OCTL_LAUNCH(k_split_flags, dim3(grid_for(L.n_new)), dim3(256), 0, st, nd, L.first_new,
                         L.n_new, L.keep_scheme, L.K, L.old_fc, leaves_only, flags);
      HIP_TRY(ctx, hipGetLastError());
      OCTL_TRY(octl_exclusive_scan_u32(ctx, flags, flags, L.n_new, small + SM_NSPLIT));
      // tiles of the nodes that split: scanned over L.n_new entries (zeros beyond the ns that are
      // filled) so that ns and the tile count come back in ONE readback
      HIP_TRY(ctx, hipMemsetAsync(tile_base, 0, (size_t)(L.n_new + 8) * 4, st));
      OCTL_LAUNCH(k_compact_split, dim3(grid_for(L.n_new)), dim3(256), 0, st, nd, L.first_new,
                         L.n_new, (const uint32_t*)flags, L.keep_scheme, L.K, L.old_fc, leaves_only,
                         split_nodes, tile_base);
      HIP_TRY(ctx, hipGetLastError());
      OCTL_TRY(octl_exclusive_scan_u32(ctx, tile_base, tile_base, L.n_new, small + SM_NTILES));
    }
    uint32_t lv[2];
    OCTL_TRY(read_small(ctx, SM_NSPLIT, 2, lv));
    const int ns = (int)lv[0];
    const uint32_t n_tiles = lv[1];
    if (ns == 0) break;
    if (L.level >= L.max_depth)
      return octl_set_error(ctx, OCTL_E_DEPTH,
                            "maximum depth %d exceeded (duplicate points with a count criterion "
                            "never stop subdividing)", L.max_depth);

    if (L.resume && L.level == 0) {
      // the roots that are subdivided here: their points are (re)written by k_finalize_marked
      OCTL_TRY(devbuf_reserve(ctx, f->root_up, (size_t)L.n_new));
      HIP_TRY(ctx, hipMemsetAsync(f->root_up.p, 0, (size_t)L.n_new, st));
      OCTL_LAUNCH(k_mark_nodes, dim3(grid_for(ns)), dim3(256), 0, st, (const int32_t*)split_nodes, ns,
                         f->root_up.as<uint8_t>());
      HIP_TRY(ctx, hipGetLastError());
    }
    const int64_t child_base = nt.n;
    if (child_base + 8 * (int64_t)ns >= ((int64_t)1 << 31))
      return octl_set_error(ctx, OCTL_E_NOMEM, "more than 2^31 scheme nodes");
    OCTL_TRY(nodes_reserve(ctx, nt, child_base + 8 * (int64_t)ns));
    nd = node_ptrs(nt);
    if (!L.all_scheme && !L.keep_scheme) {
      OCTL_TRY(devbuf_reserve(ctx, f->child_sc, (size_t)8 * ns * 4));
      HIP_TRY(ctx, hipMemsetAsync(f->child_sc.p, 0, (size_t)8 * ns * 4, st));
    }
    const int src = L.level & 1;
    const int shift = 29 - 3 * (L.level % PATH_LEVELS);
    uint32_t* entries = nullptr;
    if (n_tiles > 0) {
      OCTL_TRY(devbuf_reserve(ctx, f->entries, ((size_t)8 * n_tiles + 8) * 4));
      entries = f->entries.as<uint32_t>();
      if (L.resume && L.level == 0) {
        // resuming after the bucket build: the level buffers only exist for the voxels it left behind -
        // store index | scheme bit from the leaf-ordered permutation (insertion order inside a root)
        OCTL_LAUNCH(k_lv_init_idx, dim3(n_tiles), dim3(LV_THREADS), 0, st, (const int32_t*)split_nodes,
                           (const uint32_t*)tile_base, ns, n_tiles, nd,
                           (const uint32_t*)f->ord_idx.as<uint32_t>(),
                           (const int64_t*)f->pose_off_dev.as<int64_t>(), (int)f->pose_off.size() - 1,
                           L.scheme_dev, f->idxbuf[src].as<uint32_t>());
        HIP_TRY(ctx, hipGetLastError());
      }
      if ((L.level > 0 || L.resume) && L.level % PATH_LEVELS == 0) {
        KTimer t(ctx, "level_rekey");
        OCTL_LAUNCH(k_lv_rekey, dim3(n_tiles), dim3(LV_THREADS), 0, st,
                           (const int32_t*)split_nodes, (const uint32_t*)tile_base, ns, n_tiles, nd,
                           (const uint32_t*)f->idxbuf[src].as<uint32_t>(),
                           f->pathbuf[src].as<uint32_t>(), gx, L.xs);
        HIP_TRY(ctx, hipGetLastError());
      }
      {
        KTimer t(ctx, "level_hist");
        if (!L.all_scheme && !L.keep_scheme)
          OCTL_LAUNCH(k_lv_hist<true>, dim3(n_tiles), dim3(LV_THREADS), 0, st,
                             (const int32_t*)split_nodes, (const uint32_t*)tile_base, ns, n_tiles,
                             nd, (const uint32_t*)f->idxbuf[src].as<uint32_t>(),
                             (const uint32_t*)f->pathbuf[src].as<uint32_t>(),
                             gx, L.xs, shift, entries,
                             f->child_sc.as<uint32_t>(), small);
        else
          OCTL_LAUNCH(k_lv_hist<false>, dim3(n_tiles), dim3(LV_THREADS), 0, st,
                             (const int32_t*)split_nodes, (const uint32_t*)tile_base, ns, n_tiles,
                             nd, (const uint32_t*)f->idxbuf[src].as<uint32_t>(),
                             (const uint32_t*)f->pathbuf[src].as<uint32_t>(),
                             gx, L.xs, shift, entries,
                             (uint32_t*)nullptr, small);
        HIP_TRY(ctx, hipGetLastError());
      }
      {
        KTimer t(ctx, "level_scan");
        OCTL_TRY(octl_exclusive_scan_u32(ctx, entries, entries, (int64_t)8 * n_tiles,
                                         small + SM_ETOTAL));
      }
      {
        KTimer t(ctx, "level_scatter");
        OCTL_LAUNCH(k_lv_scatter, dim3(n_tiles), dim3(LV_THREADS), 0, st,
                           (const int32_t*)split_nodes, (const uint32_t*)tile_base, ns, n_tiles, nd,
                           (const uint32_t*)f->idxbuf[src].as<uint32_t>(),
                           (const uint32_t*)f->pathbuf[src].as<uint32_t>(),
                           gx, L.xs, shift, (const uint32_t*)entries,
                           (int32_t)child_base, f->idxbuf[src ^ 1].as<uint32_t>(),
                           f->pathbuf[src ^ 1].as<uint32_t>(), pos_node, small,
                           (L.all_scheme && !L.keep_scheme && !L.resume && L.K >= 0 && L.level + 1 < L.max_depth)
                               ? L.K : (int64_t)-1);
        HIP_TRY(ctx, hipGetLastError());
      }
    }
    {
      KTimer t(ctx, "level_children");
      OCTL_LAUNCH(k_make_children, dim3(grid_for((int64_t)8 * ns)), dim3(256), 0, st,
                         (const int32_t*)split_nodes, (const uint32_t*)tile_base, ns, n_tiles, nd,
                         (const uint32_t*)entries, (const uint32_t*)small,
                         (const uint32_t*)f->child_sc.as<uint32_t>(),
                         (int)(L.all_scheme || L.keep_scheme), (int32_t)child_base, L.keep_scheme,
                         L.cur_epoch, L.old_fc, L.old_epoch);
      HIP_TRY(ctx, hipGetLastError());
    }
    nt.n = child_base + 8 * (int64_t)ns;
    L.segs->push_back({child_base, nt.n, L.level + 1});
    L.first_new = child_base;
    L.n_new = 8 * (int64_t)ns;
    L.n_internal += ns;
    ++L.level;
  }
  return OCTL_OK;
}

int forest_fix_origin(octl_forest* f, const int bb[6]) {
  if (f->mode != 0 || bb[0] > bb[3]) return OCTL_OK;  // a single cube has voxel 0 only; an empty box fixes nothing
  if (!f->vorg_set) {
    f->vorg.x = bb[0];
    f->vorg.y = bb[1];
    f->vorg.z = bb[2];
    f->vorg_set = true;
  }
  if (!vkey_in_window(bb[0], bb[1], bb[2], f->vorg) || !vkey_in_window(bb[3], bb[4], bb[5], f->vorg))
    return octl_set_error(f->ctx, OCTL_E_DOMAIN,
                          "the scene has moved more than %d voxels away from where it started (voxel box "
                          "[%d..%d] x [%d..%d] x [%d..%d], origin %d %d %d): the packed voxel keys cannot hold it",
                          OCTL_VOX_BIAS, bb[0], bb[3], bb[1], bb[4], bb[2], bb[5], f->vorg.x, f->vorg.y, f->vorg.z);
  return OCTL_OK;
}

// the general path packs voxel keys before it knows the voxel box (k_keygen finds it on the way): a forest
// that has no origin yet gets it from the box the ingest kernel keeps (one small readback, first build only)
static int forest_ensure_origin(octl_forest* f) {
  if (f->mode != 0 || f->vorg_set || f->n_store == 0) return OCTL_OK;
  octl_ctx* ctx = f->ctx;
  if (f->bbox_pending) OCTL_TRY(store_compute_bbox(f));
  if (!f->bbox_dev.p) return OCTL_OK;
  int32_t* host = reinterpret_cast<int32_t*>(static_cast<char*>(ctx->small_host) + MIRROR_BBOX_WORD * 4);
  HIP_TRY(ctx, hipMemcpyAsync(host, f->bbox_dev.p, 32, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  int bb[6];
  std::memcpy(bb, host, sizeof(bb));
  if (host[6]) return OCTL_OK;  // (a non-finite coordinate: k_keygen reports it)
  return forest_fix_origin(f, bb);
}

int forest_build(octl_forest* f, int64_t K, const uint8_t* scheme_mask, int32_t n_mask,
                 int32_t keep_scheme, int32_t max_depth, octl_build_info* info) {
  octl_ctx* ctx = f->ctx;
  hipStream_t st = ctx->stream;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int64_t N = f->n_store;
  const int n_poses = (int)f->pose_off.size() - 1;
  if (N >= ((int64_t)1 << 31))
    return octl_set_error(ctx, OCTL_E_INVALID, "more than 2^31-1 points in one forest");
  if (keep_scheme && !f->built)
    return octl_set_error(ctx, OCTL_E_STATE, "keep_scheme build without a previous scheme");
  if (scheme_mask && n_mask != n_poses)
    return octl_set_error(ctx, OCTL_E_INVALID, "scheme mask has %d entries for %d poses", n_mask,
                          n_poses);
  if (f->displaced_rows && !keep_scheme)
    return octl_set_error(ctx, OCTL_E_DOMAIN,
                          "map_leaf_points left points outside the cube of their leaf: the reference keeps them "
                          "there and raises IndexError when such a leaf is subdivided (octree.py:94-98)");
  if (max_depth <= 0) max_depth = 63;
  f->fast_order_valid = false;  // (the block table is about to change)
  f->max_block_hint = INT64_MAX;
  BuildTrace trace;
  trace.on = ctx->opt.trace_build != 0;
  uint32_t* small = ctx->small.as<uint32_t>();

  bool all_scheme = true;
  if (scheme_mask && !keep_scheme)
    for (int p = 0; p < n_poses; ++p) all_scheme = all_scheme && scheme_mask[p];

  // ---- reset the scalar block -------------------------------------------------------------
  {
    uint32_t init[32];
    std::memset(init, 0, sizeof(init));
    int* bb = reinterpret_cast<int*>(init + SM_BBOX);
    bb[0] = bb[1] = bb[2] = 1 << 30;
    bb[3] = bb[4] = bb[5] = -(1 << 30);
    std::memcpy(ctx->small_host, init, sizeof(init));
    // the hint of the previous build rides along when this build may run under it (a cloud taken in place whose
    // box is still to be found: bucket_build.hip decides and, if it does, finds the record in place)
    int hint_words = 0;
    ctx->geom_hint_staged = false;
    if (f->bbox_pending && !keep_scheme && ctx->geom_hint_valid && !ctx->geom_hint_two_pass && !ctx->opt.no_geom_hint) {
      static_assert(sizeof(ctx->geom_hint) == 192, "hint words");
      std::memcpy(static_cast<char*>(ctx->small_host) + SM_GEOM * 4, ctx->geom_hint, sizeof(ctx->geom_hint));
      hint_words = (int)(sizeof(ctx->geom_hint) / 4);
      ctx->geom_hint_staged = true;
    }
    int32_t* box = nullptr;
    if (f->bbox_pending && f->bbox_stale && f->bbox_dev.p) {
      box = f->bbox_dev.as<int32_t>();
      f->bbox_stale = false;
    }
    // (a kernel copy out of page-locked memory, not a DMA: see octl_copy_from_pinned)
    OCTL_LAUNCH(k_build_begin, dim3(1), dim3(128), 0, st, static_cast<const uint32_t*>(ctx->small_host), small,
                       hint_words, box);
    HIP_TRY(ctx, hipGetLastError());
  }

  // ---- pose offsets / scheme mask on the device ----------------------------------------------
  {
    const void* before = f->pose_off_dev.p;
    OCTL_TRY(devbuf_reserve(ctx, f->pose_off_dev, (size_t)(n_poses + 1) * 8));
    if (f->pose_off_dev.p != before) f->pose_off_uploaded.clear();  // a new buffer holds nothing yet
  }
  const uint8_t* scheme_dev = nullptr;
  {
    // small uploads go through pinned staging so that no synchronisation is needed; the region
    // [0, 128 KiB) of ctx->pinned belongs to the build, [128 KiB, 256 KiB) to the RANSAC table
    const size_t off_bytes = (size_t)(n_poses + 1) * 8;
    const bool fits = off_bytes + (size_t)n_poses <= OCTL_PINNED_BYTES / 2;
    char* pin = static_cast<char*>(ctx->pinned);
    const void* src_off = f->pose_off.data();
    if (fits) {
      std::memcpy(pin, f->pose_off.data(), off_bytes);
      src_off = pin;
    }
    if (f->pose_off_uploaded != f->pose_off) {  // (the same offsets step after step: nothing to upload)
      if (fits)
        OCTL_TRY(octl_copy_from_pinned(ctx, f->pose_off_dev.p, src_off, off_bytes));
      else
        HIP_TRY(ctx, hipMemcpyAsync(f->pose_off_dev.p, src_off, off_bytes, hipMemcpyHostToDevice, st));
      f->pose_off_uploaded = f->pose_off;
    }
    if (!all_scheme) {
      OCTL_TRY(devbuf_reserve(ctx, f->scheme_dev, (size_t)n_poses));
      const void* src_m = scheme_mask;
      if (fits) {
        std::memcpy(pin + off_bytes, scheme_mask, (size_t)n_poses);
        src_m = pin + off_bytes;
      }
      HIP_TRY(ctx, hipMemcpyAsync(f->scheme_dev.p, src_m, (size_t)n_poses, hipMemcpyHostToDevice, st));
      scheme_dev = f->scheme_dev.as<uint8_t>();
    }
    if (!fits) HIP_TRY(ctx, hipStreamSynchronize(st));  // pageable sources must not change in flight
  }

  // ---- poses appended to a built forest inherit its scheme: only the new points are placed -----------
  // (K < 0, "never split", over a scheme that has no internal nodes - the state between insert_points calls
  //  before the first subdivide - is the same thing: new points into the existing roots, new voxels as
  //  new roots; only the build counter advances)
  const bool unsplit_again = !keep_scheme && K < 0 && f->built && f->n_internal == 0;
  if (keep_scheme || unsplit_again) {
    int done = 0;
    OCTL_TRY(forest_insert_incremental(f, &done, info));
    trace.mark("incremental insertion");
    if (done) {
      if (unsplit_again) f->epoch += 1;
      return OCTL_OK;
    }
    // the re-placement below keys every stored point by its coordinates again: a row that map_leaf_points moved
    // out of its leaf's cube would silently change leaf (or voxel).  The reference keeps such a row where it is
    // until that leaf is subdivided (octree.py:94-98,114-123); only the incremental path above does the same.
    if (f->displaced_rows)
      return octl_set_error(ctx, OCTL_E_DOMAIN,
                            "map_leaf_points left points outside the cube of their leaf and the stored poses have "
                            "to be placed again (a pose was extended or the scheme replaced since): the points "
                            "cannot be kept in their leaves, as the reference does (octree.py:114-123)");
  }
  const int64_t n_alive = f->n_alive;
  // ---- 0. the bucket build does insert + subdivide in one go (bucket_build.hip): a fresh forest, or a
  //         count-driven subdivide over a previous scheme (its internal nodes keep their epochs, its voxels
  //         must all be there again) ----------------------------------------------------------------------------
  const bool fresh0 = !f->built && f->vkeys.empty() && !f->vkeys_stale;
  const bool over_old = f->built && !ctx->opt.no_bucket_history;
  if (!keep_scheme && (fresh0 || over_old) && n_alive > 0) {
    NodeTable& bt = f->nodes[f->cur ^ 1];
    const int cur_epoch = f->epoch + 1;
    BucketBuildArgs ba{K, scheme_dev, cur_epoch, max_depth};
    const int64_t old_internal0 = f->built ? f->n_internal : 0;
    if (over_old) {
      OCTL_TRY(forest_sync_vcodes(f));
      NodeTable& old0 = f->nodes[f->cur];
      ba.old_fc = old0.first_child.as<int32_t>();
      ba.old_epoch = old0.epoch.as<int32_t>();
      ba.old_vcode = f->vcode_dev[0].as<uint64_t>();
      ba.old_voxels = f->n_voxels;
    }
    BucketBuildGeom geom;
    int done = 0, lv = 0;
    int64_t ni = 0, nv = 0, nblk = 0, pending = 0;
    std::vector<octl_forest::LevelSeg> segs;
    OCTL_TRY(forest_bucket_build(f, ba, bt, &done, &segs, &ni, &lv, &nv, &nblk, &pending, &geom));
    trace.mark("bucket build");
    if (done && pending > 0) {
      // some voxels were left as single leaves (more than 4096 points, deeper than 6 levels, a point
      // outside its cube): the level loop subdivides exactly those roots, the rest of the build stands
      for (int b = 0; b < 2; ++b) {
        OCTL_TRY(devbuf_reserve(ctx, f->idxbuf[b], (size_t)n_alive * 4));
        OCTL_TRY(devbuf_reserve(ctx, f->pathbuf[b], (size_t)n_alive * 4));
      }
      LevelLoop L{f, &bt, K, 0, all_scheme, scheme_dev, ba.old_fc, ba.old_epoch, cur_epoch, max_depth, n_alive,
                  true, 0, nv, 0, 0, &segs};
      OCTL_TRY(run_level_loop(L));
      trace.mark("level loop (pending voxels)");
      if (L.n_internal > 0) {
        NodePtrs nd = node_ptrs(bt);
        KTimer t(ctx, "finalize");
        OCTL_LAUNCH(k_finalize_marked, dim3(grid_for(n_alive)), dim3(256), 0, st,
                           (const int32_t*)f->pos_node.as<int32_t>(), (const int32_t*)nd.depth,
                           (const int32_t*)nd.voxel, (const uint8_t*)f->root_up.as<uint8_t>(),
                           (const uint32_t*)f->idxbuf[0].as<uint32_t>(),
                           (const uint32_t*)f->idxbuf[1].as<uint32_t>(), (const double*)f->xyz.as<double>(),
                           n_alive, f->ord_idx.as<uint32_t>(), f->xyz_ord.as<double>());
        HIP_TRY(ctx, hipGetLastError());
      }
      ni += L.n_internal;
      lv = std::max(lv, L.level);
      const int64_t n_ord_before = f->n_ord;
      f->n_ord = n_alive;
      OCTL_TRY(forest_make_blocks(f));
      uint32_t e = 0;
      OCTL_TRY(forest_finish_blocks(f, &e));
      nblk = f->n_blocks;
      if (e) {
        f->n_ord = n_ord_before;
        f->n_blocks = 0;
        f->built = false;
        return octl_set_error(ctx, OCTL_E_DOMAIN,
                              "a point lies outside the cube of a node that is being subdivided "
                              "(the reference raises IndexError or picks a wrong child here)");
      }
    }
    if (done) {
      f->cur ^= 1;
      f->n_voxels = nv;
      f->vkeys.clear();
      f->vkeys_stale = true;
      f->vl_min[0] = geom.min[0]; f->vl_min[1] = geom.min[1]; f->vl_min[2] = geom.min[2];
      f->vl_ny = geom.ny;
      f->vl_nz = geom.nz;
      f->level_segs.swap(segs);
      f->built = true;
      f->epoch = cur_epoch;
      f->n_ord = n_alive;
      f->n_blocks = nblk;
      f->n_internal = ni;
      // nodes inherit an older epoch only from a previous scheme that had internal nodes
      f->uniform_epoch = !(over_old && old_internal0 > 0);
      f->max_depth_reached = lv;
      f->mask_valid = false;
      f->store_dirty = false;
      f->vcode_valid = false;
      f->fast_order_valid = geom.order_done && pending == 0;
      // (count-driven from ALL poses: a leaf holds at most K points; what the level loop finished for the voxels left
      //  behind obeys the same rule)
      if (all_scheme && K >= 0) f->max_block_hint = K;
      f->built_store = f->n_store;
      f->built_poses = n_poses;
      f->append_only = true;
      if (info) {
        info->n_points = n_alive;
        info->n_voxels = nv;
        info->n_nodes = bt.n;
        info->n_internal = ni;
        info->n_blocks = nblk;
        info->max_depth = lv;
        info->n_levels = lv;
      }
      return OCTL_OK;
    }
  }
  OCTL_TRY(forest_sync_vkeys(f));  // the previous scheme's voxels persist (no-op when fresh)
  trace.mark("sync_vkeys");
  // ---- 1. keys -----------------------------------------------------------------------------------
  OCTL_TRY(forest_ensure_origin(f));
  // a fresh single cube with every point alive: one root, the store order is the level-0 order (k_cube_level0)
  const bool cube_fast = f->mode == 1 && N > 0 && n_alive == N && !f->built && f->vkeys.empty() &&
                         !ctx->opt.no_cube_fast;
  // ... and when it is BIG the store is first partitioned once by the digits of its first pm levels
  // (bucket_build.hip: forest_prefix_partition): the top pm levels of the tree follow from the partition's
  // histogram, the level loop starts at level pm and every later gather of coordinates stays inside the ~15 000
  // records of one depth-pm node.  BASELINE config 4 (64 M points, K = 4096, 5 levels): 4 of the 5 level passes
  // and the store-wide random gather of k_finalize (169 bytes fetched per 24-byte point) go.
  int pm = 0;
  const void* pre_recs = nullptr;
  const uint32_t* pre_bstart = nullptr;
  const uint32_t* pre_bad = nullptr;
  uint32_t pre_stride = 0;
  // (OCTL_CUBE_PREFIX_MIN: tests run this path on small clouds; OCTL_NO_CUBE_PREFIX: never)
  const int64_t prefix_min = ctx->opt.cube_prefix_min > 0 ? ctx->opt.cube_prefix_min : ((int64_t)4 << 20);
  if (cube_fast && all_scheme && !keep_scheme && K >= 0 && n_alive >= prefix_min && !ctx->opt.no_cube_prefix) {
    for (int c = 4; c >= 2 && !pm; --c)   // every node above depth pm has to split: expect >= 2 K points per depth-pm node
      if (n_alive >= 2 * std::max<int64_t>(K, 1) * ((int64_t)1 << (3 * c)) && c <= max_depth) pm = c;
    if (pm) OCTL_TRY(forest_prefix_partition(f, pm, &pre_recs, &pre_bstart, &pre_stride, &pre_bad));
  }
  if (N > 0 && !cube_fast) {
    OCTL_TRY(alive_ensure(f));
    OCTL_TRY(devbuf_reserve(ctx, f->vkey, (size_t)N * 8));
    OCTL_TRY(devbuf_reserve(ctx, f->path, (size_t)N * 4));
    KTimer t(ctx, "keygen");
    OCTL_LAUNCH(k_keygen, dim3(grid_for(N)), dim3(256), 0, st, f->xyz.as<double>(),
                       f->alive.as<uint8_t>(), N, f->mode, f->edge, f->corner[0], f->corner[1],
                       f->corner[2], f->vorg, ctx->opt.no_exact_digits ? 0 : 1, f->vkey.as<uint64_t>(),
                       f->path.as<uint32_t>(), small);
    HIP_TRY(ctx, hipGetLastError());
  }
  uint32_t sm[32];
  int bb[6] = {0, 0, 0, 0, 0, 0};  // (a single cube is voxel 0)
  if (!cube_fast) {
    OCTL_TRY(read_small(ctx, 0, 32, sm));
    if (sm[SM_ERR])
      return octl_set_error(ctx, OCTL_E_DOMAIN,
                            "a point has a non-finite coordinate, a top-level voxel index outside +-%d, or "
                            "lies more than %d voxels from where the scene started", OCTL_VOX_ABS_LIMIT,
                            OCTL_VOX_BIAS);
    std::memcpy(bb, sm + SM_BBOX, sizeof(bb));
  }
  // the voxels of the previous scheme persist even when they have lost all their points
  for (uint64_t k : f->vkeys) {
    int64_t q[3];
    vkey_decode(k, f->vorg, q);
    for (int a = 0; a < 3; ++a) {
      bb[a] = std::min<int>(bb[a], (int)q[a]);
      bb[3 + a] = std::max<int>(bb[3 + a], (int)q[a]);
    }
  }
  const bool any_voxel = bb[0] <= bb[3];
  const uint64_t nx = any_voxel ? (uint64_t)(bb[3] - bb[0] + 1) : 1;
  const uint64_t ny = any_voxel ? (uint64_t)(bb[4] - bb[1] + 1) : 1;
  const uint64_t nz = any_voxel ? (uint64_t)(bb[5] - bb[2] + 1) : 1;
  if (!any_voxel) bb[0] = bb[1] = bb[2] = 0;
  const uint64_t dead_lin = nx * ny * nz;  // < 2^63
  const int key_bits = bits_for(dead_lin);
  auto lin_of = [&](uint64_t k) {
    int64_t q[3];
    vkey_decode(k, f->vorg, q);
    return ((uint64_t)(q[0] - bb[0]) * ny + (uint64_t)(q[1] - bb[1])) * nz + (uint64_t)(q[2] - bb[2]);
  };
  auto vkey_of_lin = [&](uint64_t l) {
    const uint64_t qz = l % nz, qy = (l / nz) % ny, qx = l / (nz * ny);
    return vkey_pack((int64_t)qx + bb[0], (int64_t)qy + bb[1], (int64_t)qz + bb[2], f->vorg);
  };

  trace.mark("keys + bbox readback");
  // ---- 2. sort by top-level voxel ---------------------------------------------------------------
  int sorted = 0;
  if (cube_fast) {  // (the voxel table of one entry is staged where the sort's second buffers would be)
    OCTL_TRY(devbuf_reserve(ctx, f->lin[1], 64));
    OCTL_TRY(devbuf_reserve(ctx, f->val[1], 64));
  } else if (N > 0) {
    for (int b = 0; b < 2; ++b) {
      OCTL_TRY(devbuf_reserve(ctx, f->lin[b], (size_t)N * 8));
      OCTL_TRY(devbuf_reserve(ctx, f->val[b], (size_t)N * 4));
    }
    {
      KTimer t(ctx, "linkey");
      OCTL_LAUNCH(k_linkey, dim3(grid_for(N)), dim3(256), 0, st, f->vkey.as<uint64_t>(), N,
                         bb[0], bb[1], bb[2], ny, nz, dead_lin, f->vorg, f->pose_off_dev.as<int64_t>(),
                         n_poses, scheme_dev, f->lin[0].as<uint64_t>(), f->val[0].as<uint32_t>());
      HIP_TRY(ctx, hipGetLastError());
    }
    uint64_t* keys[2] = {f->lin[0].as<uint64_t>(), f->lin[1].as<uint64_t>()};
    uint32_t* vals[2] = {f->val[0].as<uint32_t>(), f->val[1].as<uint32_t>()};
    // a single voxel and no dead points: already "sorted"
    if (!(dead_lin == 1 && n_alive == N))
      OCTL_TRY(octl_radix_sort_u64_u32(ctx, keys, vals, N, key_bits, f->hist, &sorted));
  }
  const uint64_t* lin_sorted = f->lin[sorted].as<uint64_t>();
  const uint32_t* val_sorted = f->val[sorted].as<uint32_t>();

  trace.mark("sort (enqueue)");
  // ---- 3. roots ------------------------------------------------------------------------------------
  // (first voxel ordinal of every tile of RT_TILE positions: k_root_tiles, k_init_level0)
  const int64_t n_rtiles = ceil_div(std::max<int64_t>(n_alive, 1), (int64_t)RT_TILE);
  OCTL_TRY(devbuf_reserve(ctx, f->flags, (size_t)(n_rtiles + 8) * 4));
  uint32_t* flags = f->flags.as<uint32_t>();
  const bool have_old = f->built;
  const bool fresh = !have_old && f->vkeys.empty() && n_alive > 0;  // roots made on the device
  int64_t v_pts = 0;
  std::vector<uint64_t> vlin_h;
  std::vector<uint32_t> vstart_h;
  uint64_t* vlin_d = f->lin[sorted ^ 1].as<uint64_t>();   // free now: staging for the voxel table
  uint32_t* vstart_d = f->val[sorted ^ 1].as<uint32_t>();
  if (cube_fast) {
    HIP_TRY(ctx, hipMemsetAsync(vlin_d, 0, 8, st));    // voxel 0 ...
    HIP_TRY(ctx, hipMemsetAsync(vstart_d, 0, 4, st));  // ... starts at position 0
    v_pts = 1;
  } else if (n_alive > 0) {
    KTimer t(ctx, "roots");
    OCTL_LAUNCH(k_root_tiles<false>, dim3((unsigned)n_rtiles), dim3(256), 0, st, lin_sorted, n_alive, flags,
                       (uint64_t*)nullptr, (uint32_t*)nullptr);
    HIP_TRY(ctx, hipGetLastError());
    OCTL_TRY(octl_exclusive_scan_u32(ctx, flags, flags, n_rtiles, small + SM_NVOX));
    OCTL_LAUNCH(k_root_tiles<true>, dim3((unsigned)n_rtiles), dim3(256), 0, st, lin_sorted, n_alive, flags,
                       vlin_d, vstart_d);
    HIP_TRY(ctx, hipGetLastError());
    uint32_t nv;
    OCTL_TRY(read_small(ctx, SM_NVOX, 1, &nv));
    v_pts = nv;
    if (!fresh) {
      vlin_h.resize(v_pts);
      vstart_h.resize(v_pts);
      HIP_TRY(ctx, hipMemcpyAsync(vlin_h.data(), vlin_d, (size_t)v_pts * 8, hipMemcpyDeviceToHost, st));
      HIP_TRY(ctx, hipMemcpyAsync(vstart_h.data(), vstart_d, (size_t)v_pts * 4, hipMemcpyDeviceToHost, st));
      HIP_TRY(ctx, hipStreamSynchronize(st));
    }
  }
  NodeTable& nt = f->nodes[f->cur ^ 1];
  NodeTable& old = f->nodes[f->cur];
  const int32_t* old_fc = have_old ? old.first_child.as<int32_t>() : nullptr;
  const int32_t* old_epoch = have_old ? old.epoch.as<int32_t>() : nullptr;
  std::vector<uint64_t> new_vkeys;
  std::vector<int32_t> local2root;
  int64_t V = 0;
  NodePtrs nd;
  const int cur_epoch_new = f->epoch + (keep_scheme ? 0 : 1);
  if (pm) {
    // the complete top of the tree from the partition's bucket starts; a point outside the cube or a node above
    // depth pm that does not split (few points, very uneven cloud) sends the build down the plain path
    OCTL_TRY(nodes_reserve(ctx, nt, top_base(pm + 1)));
    nd = node_ptrs(nt);
    OCTL_LAUNCH(k_top_tree, dim3(grid_for(top_base(pm + 1))), dim3(256), 0, st, pre_bstart, pre_stride, pm,
                       n_alive, K, f->edge, f->corner[0], f->corner[1], f->corner[2], cur_epoch_new, nd,
                       small + SM_BK_MISSING);
    HIP_TRY(ctx, hipGetLastError());
    uint32_t* fl = static_cast<uint32_t*>(ctx->small_host);
    HIP_TRY(ctx, hipMemcpyAsync(fl, small + SM_BK_MISSING, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(fl + 1, pre_bad, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (fl[0] || fl[1]) {
      pm = 0;
      HIP_TRY(ctx, hipMemsetAsync(small + SM_BK_MISSING, 0, 4, st));
    }
  }
  if (pm) {
    V = 1;
    nt.n = top_base(pm + 1);
  } else if (fresh) {
    V = v_pts;
    OCTL_TRY(nodes_reserve(ctx, nt, std::max<int64_t>(V, 1)));
    nt.n = V;
    nd = node_ptrs(nt);
    OCTL_LAUNCH(k_make_roots, dim3(grid_for(V)), dim3(256), 0, st, (const uint64_t*)vlin_d,
                       (const uint32_t*)vstart_d, V, n_alive, f->mode, f->edge, f->corner[0],
                       f->corner[1], f->corner[2], bb[0], bb[1], bb[2], ny, nz,
                       (int)(all_scheme || keep_scheme), nd);
    HIP_TRY(ctx, hipGetLastError());
  } else {
    // union with the voxels of the previous scheme (both lists are sorted by lin: the packed key
    // order and the lin order are both the lexicographic (x,y,z) order)
    std::vector<uint32_t> r_start, r_count;
    std::vector<int32_t> r_old;
    local2root.resize(std::max<int64_t>(v_pts, 1));
    size_t a = 0, b = 0;
    const size_t na = (size_t)v_pts, nb_old = f->vkeys.size();
    while (a < na || b < nb_old) {
      const uint64_t la = (a < na) ? vlin_h[a] : ~0ull;
      const uint64_t lb = (b < nb_old) ? lin_of(f->vkeys[b]) : ~0ull;
      const uint64_t l = std::min(la, lb);
      const int32_t root = (int32_t)new_vkeys.size();
      new_vkeys.push_back(vkey_of_lin(l));
      if (la == l) {
        const uint32_t s0 = vstart_h[a];
        const uint32_t e0 = (a + 1 < na) ? vstart_h[a + 1] : (uint32_t)n_alive;
        r_start.push_back(s0);
        r_count.push_back(e0 - s0);
        local2root[a] = root;
        ++a;
      } else {
        r_start.push_back(0);
        r_count.push_back(0);
      }
      if (lb == l) {
        r_old.push_back((int32_t)b);  // old roots are nodes [0, V_old) in voxel order
        ++b;
      } else {
        r_old.push_back(-1);
      }
    }
    if (f->mode == 1 && new_vkeys.empty()) {  // a cube without points still has its root
      new_vkeys.push_back(vkey_pack(0, 0, 0, f->vorg));
      r_start.push_back(0);
      r_count.push_back(0);
      r_old.push_back(f->built ? 0 : -1);
    }
    V = (int64_t)new_vkeys.size();
    OCTL_TRY(nodes_reserve(ctx, nt, std::max<int64_t>(V, 1)));
    nt.n = V;
    nd = node_ptrs(nt);
    if (V > 0) {
      std::vector<int32_t> i32((size_t)V);
      std::vector<double> cor((size_t)V * 3), edg((size_t)V, f->edge);
      auto up = [&](void* dst, const void* src, size_t bytes) {
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
      };
      for (int64_t v = 0; v < V; ++v) {
        i32[v] = (int32_t)v;
        if (f->mode == 0) {
          int64_t q[3];
          vkey_decode(new_vkeys[v], f->vorg, q);
          // np.array(voxel_coordinates): int64(q * L), L integer valued (grid.py:72-76,104)
          for (int ax = 0; ax < 3; ++ax) cor[3 * v + ax] = (double)(int64_t)((double)q[ax] * f->edge);
        } else {
          for (int ax = 0; ax < 3; ++ax) cor[3 * v + ax] = f->corner[ax];
        }
      }
      HIP_TRY(ctx, up(nd.start, r_start.data(), (size_t)V * 4));
      HIP_TRY(ctx, up(nd.count, r_count.data(), (size_t)V * 4));
      HIP_TRY(ctx, up(nd.old_id, r_old.data(), (size_t)V * 4));
      HIP_TRY(ctx, up(nd.voxel, i32.data(), (size_t)V * 4));
      HIP_TRY(ctx, hipMemsetAsync(nd.depth, 0, (size_t)V * 4, st));
      HIP_TRY(ctx, hipMemsetAsync(nd.epoch, 0, (size_t)V * 4, st));
      HIP_TRY(ctx, hipMemsetAsync(nd.parent, 0xFF, (size_t)V * 4, st));
      HIP_TRY(ctx, hipMemsetAsync(nd.first_child, 0xFF, (size_t)V * 4, st));
      HIP_TRY(ctx, up(nd.corner, cor.data(), (size_t)V * 24));
      HIP_TRY(ctx, up(nd.edge, edg.data(), (size_t)V * 8));
      if (all_scheme || keep_scheme) {
        HIP_TRY(ctx, up(nd.scount, r_count.data(), (size_t)V * 4));
      } else {
        HIP_TRY(ctx, hipMemsetAsync(nd.scount, 0, (size_t)V * 4, st));
      }
      // the sources above are pageable host vectors that die with this scope
      HIP_TRY(ctx, hipStreamSynchronize(st));
    }
  }

  trace.mark("roots (+ union with old voxels)");
  // voxel keys of this build: kept on the device, decoded on the host only when someone asks
  OCTL_TRY(devbuf_reserve(ctx, f->vlin_dev, (size_t)std::max<int64_t>(V, 1) * 8));
  if (fresh && V > 0)
    HIP_TRY(ctx, hipMemcpyAsync(f->vlin_dev.p, vlin_d, (size_t)V * 8, hipMemcpyDeviceToDevice, st));

  const int cur_epoch = f->epoch + (keep_scheme ? 0 : 1);
  const int64_t old_internal = have_old ? f->n_internal : 0;
  int64_t first_new = 0, n_new = V, n_internal = 0;
  int level = 0;

  int32_t* pos_node = nullptr;
  // ---- 4. level-0 buffers ---------------------------------------------------------------------------
  for (int b = 0; b < 2; ++b) {
    OCTL_TRY(devbuf_reserve(ctx, f->idxbuf[b], (size_t)std::max<int64_t>(n_alive, 1) * 4));
    OCTL_TRY(devbuf_reserve(ctx, f->pathbuf[b], (size_t)std::max<int64_t>(n_alive, 1) * 4));
  }
  OCTL_TRY(devbuf_reserve(ctx, f->pos_node, (size_t)std::max<int64_t>(n_alive, 1) * 4));
  pos_node = f->pos_node.as<int32_t>();
  if (n_alive > 0) {
    const int32_t* l2r = nullptr;
    if (!fresh) {
      OCTL_TRY(devbuf_reserve(ctx, f->root_up, (size_t)v_pts * 4));
      HIP_TRY(ctx, hipMemcpyAsync(f->root_up.p, local2root.data(), (size_t)v_pts * 4,
                                  hipMemcpyHostToDevice, st));
      HIP_TRY(ctx, hipStreamSynchronize(st));
      l2r = f->root_up.as<int32_t>();
    }
    KTimer t(ctx, "init_level0");
    if (pm)
      OCTL_LAUNCH(k_pre_level0, dim3(grid_for(N)), dim3(256), 0, st, static_cast<const uint4*>(pre_recs), N, pm,
                         pos_node, f->idxbuf[pm & 1].as<uint32_t>(), f->pathbuf[pm & 1].as<uint32_t>());
    else if (cube_fast)
      OCTL_LAUNCH(k_cube_level0, dim3(grid_for(N)), dim3(256), 0, st, (const double*)f->xyz.as<double>(), N,
                         f->edge, f->corner[0], f->corner[1], f->corner[2],
                         (const int64_t*)f->pose_off_dev.as<int64_t>(), n_poses, scheme_dev,
                         ctx->opt.no_exact_digits ? 0 : 1, pos_node,
                         f->idxbuf[0].as<uint32_t>(), f->pathbuf[0].as<uint32_t>());
    else
      OCTL_LAUNCH(k_init_level0, dim3((unsigned)n_rtiles), dim3(256), 0, st, lin_sorted,
                         (const uint32_t*)flags, val_sorted, (const uint32_t*)f->path.as<uint32_t>(),
                         n_alive, l2r, pos_node,
                         f->idxbuf[0].as<uint32_t>(), f->pathbuf[0].as<uint32_t>());
    HIP_TRY(ctx, hipGetLastError());
    if (!all_scheme && !keep_scheme) {
      OCTL_LAUNCH(k_count_scheme, dim3((unsigned)ceil_div(n_alive, 2048)), dim3(256), 0, st,
                         (const int32_t*)pos_node, (const uint32_t*)f->idxbuf[0].as<uint32_t>(),
                         n_alive, nd.scount);
      HIP_TRY(ctx, hipGetLastError());
    }
  }

  trace.mark("level-0 buffers");
  // ---- 5. level loop ----------------------------------------------------------------------------------
  std::vector<octl_forest::LevelSeg> segs{{0, V, 0}};
  LevelLoop L{f, &nt, K, (int)keep_scheme, all_scheme, scheme_dev, old_fc, old_epoch, cur_epoch, max_depth,
              n_alive, false, first_new, n_new, 0, 0, &segs};
  if (pm) {  // levels 0 .. pm-1 are done: all of their nodes are internal
    for (int d = 1; d <= pm; ++d) segs.push_back({top_base(d), top_base(d + 1), d});
    L.first_new = top_base(pm);
    L.n_new = (int64_t)1 << (3 * pm);
    L.n_internal = top_base(pm);
    L.level = pm;
    L.gx = static_cast<const double*>(pre_recs);
    L.xs = 4;
  }
  OCTL_TRY(run_level_loop(L));
  nd = node_ptrs(nt);
  n_internal = L.n_internal;
  level = L.level;
  trace.mark("level loop");
  // ---- 6. leaf-ordered arrays and the block table ------------------------------------------------------
  if (n_alive > 0) {
    OCTL_TRY(devbuf_reserve(ctx, f->ord_idx, (size_t)n_alive * 4));
    OCTL_TRY(devbuf_reserve(ctx, f->xyz_ord, (size_t)n_alive * 24));
    {
      // (a fused gather + block-head count, eight positions per thread, was measured SLOWER on BASELINE config 4:
      //  2.39 + 0.24 ms against 1.95 + 0.39 ms - eight dependent random gathers per thread hide less latency
      //  than one per thread at full occupancy)
      KTimer t(ctx, "finalize");
      if (pm)
        OCTL_LAUNCH(k_finalize_rec, dim3(grid_for(n_alive)), dim3(256), 0, st,
                           (const int32_t*)pos_node, (const int32_t*)nd.depth,
                           (const uint32_t*)f->idxbuf[0].as<uint32_t>(),
                           (const uint32_t*)f->idxbuf[1].as<uint32_t>(), static_cast<const uint4*>(pre_recs),
                           n_alive, f->ord_idx.as<uint32_t>(), f->xyz_ord.as<double>());
      else
        OCTL_LAUNCH(k_finalize, dim3(grid_for(n_alive)), dim3(256), 0, st,
                           (const int32_t*)pos_node, (const int32_t*)nd.depth,
                           (const uint32_t*)f->idxbuf[0].as<uint32_t>(),
                           (const uint32_t*)f->idxbuf[1].as<uint32_t>(),
                           (const double*)f->xyz.as<double>(), n_alive, f->ord_idx.as<uint32_t>(),
                           f->xyz_ord.as<double>());
      HIP_TRY(ctx, hipGetLastError());
    }
  }
  trace.mark("finalize (enqueue)");
  int64_t n_blocks = 0;
  const int64_t n_ord_before = f->n_ord;
  f->n_ord = n_alive;
  OCTL_TRY(forest_make_blocks(f));
  {
    uint32_t e = 0;
    OCTL_TRY(forest_finish_blocks(f, &e));  // the build's final synchronisation
    n_blocks = f->n_blocks;
    if (e) {
      f->n_ord = n_ord_before;
      f->n_blocks = 0;
      f->built = false;  // the leaf-ordered arrays were overwritten
      return octl_set_error(ctx, OCTL_E_DOMAIN,
                            "a point lies outside the cube of a node that is being subdivided "
                            "(the reference raises IndexError or picks a wrong child here)");
    }
  }

  trace.mark("blocks + final sync");
  // ---- commit --------------------------------------------------------------------------------------------
  f->cur ^= 1;
  f->n_voxels = V;
  if (fresh) {
    f->vkeys.clear();
    f->vkeys_stale = true;
    f->vl_min[0] = bb[0]; f->vl_min[1] = bb[1]; f->vl_min[2] = bb[2];
    f->vl_ny = ny;
    f->vl_nz = nz;
  } else {
    f->vkeys.swap(new_vkeys);
    f->vkeys_stale = false;
  }
  f->level_segs.swap(segs);
  f->built = true;
  f->epoch = cur_epoch;
  f->n_ord = n_alive;
  f->n_blocks = n_blocks;
  f->n_internal = n_internal;
  // nodes inherit an older epoch only from a previous scheme that had internal nodes
  f->uniform_epoch = keep_scheme ? f->uniform_epoch : !(have_old && old_internal > 0);
  f->max_depth_reached = level;
  f->mask_valid = false;
  f->store_dirty = false;
  f->vcode_valid = false;
  f->built_store = f->n_store;
  f->built_poses = n_poses;
  f->append_only = true;
  if (info) {
    info->n_points = n_alive;
    info->n_voxels = V;
    info->n_nodes = nt.n;
    info->n_internal = n_internal;
    info->n_blocks = n_blocks;
    info->max_depth = level;
    info->n_levels = level;
  }
  return OCTL_OK;
}
