// Incremental insertion: poses appended to a BUILT forest inherit its scheme
// (OctreeManager.insert_points -> Octree.insert_points -> OctreeNode.insert_points,
// /root/reference octree_manager.py:161-171, octree/octree.py:67-100; a voxel seen for the first time
// gets a fresh single-leaf octree, grid/grid.py:96-109).
//
// Only the NEW points are touched: each one finds its top-level voxel by binary search over the
// sorted voxel codes, walks the scheme's node table down to its leaf with the reference's child-index
// arithmetic, the new points are sorted by leaf (stable: insertion order inside a leaf, pose-major
// because the store is) and APPENDED to the leaf-ordered arrays with their (leaf, pose) blocks behind
// the existing block table.  Nothing in the library relies on the blocks being sorted by leaf: the
// listing order of the reference is computed per block (order.hip), blocks only have to be contiguous
// runs in storage order.
//
// Voxels the scheme has not seen get new roots.  Roots stay the nodes [0, V) in voxel order (the
// invariant the rest of the library is written against), so this case renumbers the node table and
// shifts the leaf ids in the block table - O(nodes + blocks), nothing per stored point.
//
// Anything unusual (a point outside the voxel domain or outside a cube that is split, a store that
// was changed other than by appending poses) makes forest_insert_incremental return *done = 0 with
// the forest untouched: the caller then re-places everything through the general path of build.hip,
// which also raises the reference's errors.
#include <algorithm>
#include <cstdlib>

#include "build_common.h"
#include "forest.h"
#include "ref_arith.h"

namespace {

enum {  // words of the context's small scalar block (reset by forest_build)
  SM_INC_MISS = 28,   // new points whose voxel the scheme does not know
  SM_INC_BAD = 29,    // new points outside a cube that is split
  SM_INC_DEAD = 30,   // new points that are not alive
  SM_INC_NEWVOX = 31  // distinct new voxels
};

constexpr uint64_t KEY_MISS = 1ull << 63;
constexpr uint64_t KEY_DEAD = ~0ull;

inline unsigned grid_for(int64_t n) { return (unsigned)ceil_div(std::max<int64_t>(n, 1), 256); }

__device__ __forceinline__ int64_t lower_bound_u64(const uint64_t* __restrict__ a, int64_t n, uint64_t x) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (a[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ int slot_of(const int64_t* __restrict__ pose_off, int n_poses, int64_t i) {
  int lo = 0, hi = n_poses;  // largest p with pose_off[p] <= i
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (pose_off[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

// fresh builds keep their voxels as linear keys relative to the build's bounding box: -> packed codes
__global__ __launch_bounds__(256) void k_lin_to_code(const uint64_t* __restrict__ lin, int64_t V, int m0,
                                                     int m1, int m2, uint64_t ny, uint64_t nz, VoxOrg org,
                                                     uint64_t* __restrict__ code) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const uint64_t l = lin[v];
  const uint64_t qz = l % nz, qy = (l / nz) % ny, qx = l / (nz * ny);
  code[v] = vkey_pack((int64_t)qx + m0, (int64_t)qy + m1, (int64_t)qz + m2, org);
}

// One new point: voxel (grid/grid.py:72-76), root by binary search, walk to the leaf
// (octree/octree.py:67-100: idx = floor((p - corner) / (edge / 2)) per axis, child 4 ix + 2 iy + iz,
// restated as exact comparisons on the same rounded differences, see compute_path in build.hip).
__global__ __launch_bounds__(256) void k_inc_place(const double* __restrict__ xyz,
                                                   const uint8_t* __restrict__ alive, int64_t first,
                                                   int64_t n_new, int mode, double L, VoxOrg org,
                                                   const uint64_t* __restrict__ vcode, int64_t V,
                                                   const int32_t* __restrict__ first_child,
                                                   const double* __restrict__ corner,
                                                   const double* __restrict__ edge,
                                                   uint64_t* __restrict__ key, uint32_t* __restrict__ val,
                                                   uint32_t* __restrict__ small) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_new) return;
  const int64_t i = first + j;
  val[j] = (uint32_t)j;
  if (!alive[i]) {
    key[j] = KEY_DEAD;
    atomicAdd(&small[SM_INC_DEAD], 1u);
    return;
  }
  const double px = xyz[3 * i + 0], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
  int qx = 0, qy = 0, qz = 0;
  if (mode == 0) {
    const double fx = floor_div_exact(px, L), fy = floor_div_exact(py, L), fz = floor_div_exact(pz, L);
    const double lim = (double)OCTL_VOX_ABS_LIMIT;
    if (!((fabs(fx) < lim) && (fabs(fy) < lim) && (fabs(fz) < lim) &&
          vkey_in_window((int64_t)fx, (int64_t)fy, (int64_t)fz, org))) {  // also NaN
      atomicExch(&small[SM_ERR], (uint32_t)(-OCTL_E_DOMAIN));
      key[j] = KEY_DEAD;
      return;
    }
    qx = (int)fx;
    qy = (int)fy;
    qz = (int)fz;
  }
  const uint64_t code = vkey_pack(qx, qy, qz, org);
  const int64_t r = lower_bound_u64(vcode, V, code);
  if (r >= V || vcode[r] != code) {
    key[j] = KEY_MISS | code;
    atomicAdd(&small[SM_INC_MISS], 1u);
    return;
  }
  int32_t node = (int32_t)r;
  int32_t fc = first_child[node];
  // (children are numbered behind their parents, so the walk ends; the bound only guards against a
  //  damaged table - the general path then reports it)
  for (int depth = 0; fc >= 0; ++depth) {
    if (depth >= 64) {
      atomicAdd(&small[SM_INC_BAD], 1u);
      break;
    }
    const double cx = corner[3 * (int64_t)node + 0], cy = corner[3 * (int64_t)node + 1],
                 cz = corner[3 * (int64_t)node + 2], e = edge[node];
    const double h = e / 2.0;
    const double ax = px - cx, ay = py - cy, az = pz - cz;
    const bool ok = (ax >= 0.0) && (ax < e) && (ay >= 0.0) && (ay < e) && (az >= 0.0) && (az < e);
    if (!ok) {  // the reference raises IndexError or picks a wrong child: left to the general path
      atomicAdd(&small[SM_INC_BAD], 1u);
      break;
    }
    node = fc + ((ax >= h ? 4 : 0) | (ay >= h ? 2 : 0) | (az >= h ? 1 : 0));
    fc = first_child[node];
  }
  key[j] = (uint64_t)(uint32_t)node;
}

// sorted keys: [hits by leaf][misses by voxel code][dead].  Heads of the distinct new voxels.
__global__ __launch_bounds__(256) void k_inc_miss_heads(const uint64_t* __restrict__ key, int64_t first_miss,
                                                        int64_t n_miss, uint32_t* __restrict__ flags) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_miss) return;
  const int64_t s = first_miss + j;
  flags[j] = (j == 0 || key[s] != key[s - 1]) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_inc_new_codes(const uint64_t* __restrict__ key, int64_t first_miss,
                                                       int64_t n_miss, const uint32_t* __restrict__ scanned,
                                                       uint64_t* __restrict__ ucode) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_miss) return;
  const int64_t s = first_miss + j;
  if (j == 0 || key[s] != key[s - 1]) ucode[scanned[j]] = key[s] & ~KEY_MISS;
}

// merged voxel list: an old root r moves up by the number of new voxels in front of it
__global__ __launch_bounds__(256) void k_inc_merge_roots(const uint64_t* __restrict__ vcode, int64_t V,
                                                         const uint64_t* __restrict__ ucode, int64_t U,
                                                         uint64_t* __restrict__ merged,
                                                         int32_t* __restrict__ shift,
                                                         int32_t* __restrict__ new_root) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < V) {
    const int64_t s = lower_bound_u64(ucode, U, vcode[t]);
    shift[t] = (int32_t)s;
    merged[t + s] = vcode[t];
  } else if (t < V + U) {
    const int64_t j = t - V;
    const int64_t id = j + lower_bound_u64(vcode, V, ucode[j]);
    new_root[j] = (int32_t)id;
    merged[id] = ucode[j];
  }
}

__device__ __forceinline__ int32_t remap_node(int32_t x, const int32_t* __restrict__ shift, int64_t V, int32_t U) {
  return x < 0 ? x : (x < V ? x + shift[x] : x + U);
}

__global__ __launch_bounds__(256) void k_inc_copy_nodes(NodePtrs src, NodePtrs dst, int64_t n, int64_t V,
                                                        int32_t U, const int32_t* __restrict__ shift) {
  const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= n) return;
  const int64_t y = remap_node((int32_t)x, shift, V, U);
  dst.start[y] = src.start[x];
  dst.count[y] = src.count[x];
  dst.scount[y] = src.scount[x];
  dst.depth[y] = src.depth[x];
  dst.old_id[y] = src.old_id[x];
  dst.epoch[y] = src.epoch[x];
  const int32_t v = src.voxel[x];
  dst.voxel[y] = v + shift[v];
  dst.parent[y] = remap_node(src.parent[x], shift, V, U);
  dst.first_child[y] = remap_node(src.first_child[x], shift, V, U);
  dst.corner[3 * y + 0] = src.corner[3 * x + 0];
  dst.corner[3 * y + 1] = src.corner[3 * x + 1];
  dst.corner[3 * y + 2] = src.corner[3 * x + 2];
  dst.edge[y] = src.edge[x];
}

// a voxel seen for the first time: one leaf root (the manager's corner is np.array(voxel_coords):
// int64(q * L), grid/grid.py:96-105)
__global__ __launch_bounds__(256) void k_inc_new_roots(const uint64_t* __restrict__ ucode,
                                                       const int32_t* __restrict__ new_root, int64_t U,
                                                       int mode, double L, double c0x, double c0y, double c0z,
                                                       VoxOrg org, NodePtrs dst) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= U) return;
  const int64_t y = new_root[j];
  const uint64_t k = ucode[j];
  int64_t qd[3];
  vkey_decode(k, org, qd);
  const long long qx = qd[0], qy = qd[1], qz = qd[2];
  dst.start[y] = 0;
  dst.count[y] = 0;
  dst.scount[y] = 0;
  dst.depth[y] = 0;
  dst.old_id[y] = -1;
  dst.epoch[y] = 0;
  dst.voxel[y] = (int32_t)y;
  dst.parent[y] = -1;
  dst.first_child[y] = -1;
  if (mode == 0) {
    dst.corner[3 * y + 0] = (double)(long long)((double)qx * L);
    dst.corner[3 * y + 1] = (double)(long long)((double)qy * L);
    dst.corner[3 * y + 2] = (double)(long long)((double)qz * L);
  } else {
    dst.corner[3 * y + 0] = c0x;
    dst.corner[3 * y + 1] = c0y;
    dst.corner[3 * y + 2] = c0z;
  }
  dst.edge[y] = L;
}

__global__ __launch_bounds__(256) void k_inc_remap(int32_t* __restrict__ a, int64_t n, int64_t V, int32_t U,
                                                   const int32_t* __restrict__ shift) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = remap_node(a[i], shift, V, U);
}

// Sorted new points -> the tail of the leaf-ordered arrays (+ block heads).  U == 0: leaf ids stand.
__global__ __launch_bounds__(256) void k_inc_gather(
    const uint64_t* __restrict__ key, const uint32_t* __restrict__ val, int64_t n_live, int64_t first,
    int64_t n_ord, const double* __restrict__ xyz, const int64_t* __restrict__ pose_off, int n_poses,
    int64_t V, int32_t U, const int32_t* __restrict__ shift, int64_t first_miss,
    const uint32_t* __restrict__ miss_rank, const int32_t* __restrict__ new_root,
    uint32_t* __restrict__ ord_idx, double* __restrict__ xyz_ord, int32_t* __restrict__ new_node,
    uint32_t* __restrict__ heads) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_live) return;
  const uint64_t k = key[j];
  int32_t node;
  if (k & KEY_MISS) {
    // miss_rank: exclusive scan of the distinct-voxel heads; a head holds its own rank, a follower the
    // rank of the NEXT voxel
    const int64_t m = j - first_miss;
    const bool head = (m == 0) || key[j - 1] != k;
    node = new_root[head ? miss_rank[m] : miss_rank[m] - 1u];
  } else {
    node = (int32_t)(uint32_t)k;
    if (U > 0) node = remap_node(node, shift, V, U);
  }
  const int64_t i = first + (int64_t)val[j];
  const int64_t o = n_ord + j;
  ord_idx[o] = (uint32_t)i;
  xyz_ord[3 * o + 0] = xyz[3 * i + 0];
  xyz_ord[3 * o + 1] = xyz[3 * i + 1];
  xyz_ord[3 * o + 2] = xyz[3 * i + 2];
  new_node[j] = node;
  bool h = (j == 0) || key[j - 1] != k;
  if (!h) {
    const int64_t ip = first + (int64_t)val[j - 1];
    h = slot_of(pose_off, n_poses, ip) != slot_of(pose_off, n_poses, i);
  }
  heads[j] = h ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_inc_blocks(const uint32_t* __restrict__ heads_scanned,
                                                    const uint64_t* __restrict__ key, const uint32_t* __restrict__ val,
                                                    int64_t n_live, int64_t first, int64_t n_ord,
                                                    int64_t n_blocks, const int64_t* __restrict__ pose_off,
                                                    int n_poses, const int32_t* __restrict__ new_node,
                                                    int32_t* __restrict__ blk_node, int32_t* __restrict__ blk_slot,
                                                    uint32_t* __restrict__ blk_start) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_live) return;
  const int64_t i = first + (int64_t)val[j];
  bool h = (j == 0) || key[j - 1] != key[j];
  const int slot = slot_of(pose_off, n_poses, i);
  if (!h) h = slot_of(pose_off, n_poses, first + (int64_t)val[j - 1]) != slot;
  if (h) {
    const int64_t b = n_blocks + heads_scanned[j];
    blk_node[b] = new_node[j];
    blk_slot[b] = slot;
    blk_start[b] = (uint32_t)(n_ord + j);
  }
}

__global__ __launch_bounds__(256) void k_inc_block_sizes(const uint32_t* __restrict__ blk_start, int64_t n_blocks,
                                                         const uint32_t* __restrict__ n_new_blocks,
                                                         int64_t n_ord_total, int32_t* __restrict__ blk_size) {
  const int64_t nb = n_blocks + (int64_t)*n_new_blocks;
  const int64_t b = n_blocks + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  const uint32_t e = (b + 1 < nb) ? blk_start[b + 1] : (uint32_t)n_ord_total;
  blk_size[b] = (int32_t)(e - blk_start[b]);
}

int bits_for(uint64_t max_value) {
  int b = 0;
  while (b < 64 && (max_value >> b) != 0) ++b;
  return b;
}

int read_words(octl_ctx* ctx, int first, int count, uint32_t* out) {
  HIP_TRY(ctx, hipMemcpyAsync(ctx->small_host, ctx->small.as<uint32_t>() + first, (size_t)count * 4,
                              hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  std::memcpy(out, ctx->small_host, (size_t)count * 4);
  return OCTL_OK;
}

}  // namespace

// The sorted packed voxel codes of the current scheme on the device (roots = nodes [0, V) in this order).
int forest_sync_vcodes(octl_forest* f) {
  if (f->vcode_valid) return OCTL_OK;
  octl_ctx* ctx = f->ctx;
  const int64_t V = f->n_voxels;
  OCTL_TRY(devbuf_reserve(ctx, f->vcode_dev[0], (size_t)std::max<int64_t>(V, 1) * 8));
  if (V > 0) {
    if (f->vkeys_stale) {  // the last build left linear keys on the device
      OCTL_LAUNCH(k_lin_to_code, dim3(grid_for(V)), dim3(256), 0, ctx->stream,
                         (const uint64_t*)f->vlin_dev.as<uint64_t>(), V, f->vl_min[0], f->vl_min[1], f->vl_min[2],
                         f->vl_ny, f->vl_nz, f->vorg, f->vcode_dev[0].as<uint64_t>());
      HIP_TRY(ctx, hipGetLastError());
    } else {
      HIP_TRY(ctx, hipMemcpyAsync(f->vcode_dev[0].p, f->vkeys.data(), (size_t)V * 8, hipMemcpyHostToDevice,
                                  ctx->stream));
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // pageable source
    }
  }
  f->vcode_valid = true;
  return OCTL_OK;
}

int forest_insert_incremental(octl_forest* f, int* done, octl_build_info* info) {
  *done = 0;
  octl_ctx* ctx = f->ctx;
  hipStream_t st = ctx->stream;
  const bool disabled = ctx->opt.no_incremental != 0;  // (tests compare the two paths)
  const int n_poses = (int)f->pose_off.size() - 1;
  const int64_t first = f->built_store, n_new = f->n_store - f->built_store;
  if (disabled || !f->built || !f->append_only || n_new <= 0 || n_poses <= f->built_poses) return OCTL_OK;
  NodeTable& cur = f->nodes[f->cur];
  const int64_t V = f->n_voxels, n_nodes = cur.n, n_ord = f->n_ord, n_blocks = f->n_blocks;
  if (n_nodes + n_new >= ((int64_t)1 << 31)) return OCTL_OK;
  uint32_t* small = ctx->small.as<uint32_t>();
  OCTL_TRY(forest_sync_vcodes(f));

  // ---- 1. place ----------------------------------------------------------------------------------------------
  for (int b = 0; b < 2; ++b) {
    OCTL_TRY(devbuf_reserve(ctx, f->lin[b], (size_t)n_new * 8));
    OCTL_TRY(devbuf_reserve(ctx, f->val[b], (size_t)n_new * 4));
  }
  uint64_t* keys[2] = {f->lin[0].as<uint64_t>(), f->lin[1].as<uint64_t>()};
  uint32_t* vals[2] = {f->val[0].as<uint32_t>(), f->val[1].as<uint32_t>()};
  NodePtrs nd = node_ptrs(cur);
  {
    KTimer t(ctx, "inc_place");
    OCTL_LAUNCH(k_inc_place, dim3(grid_for(n_new)), dim3(256), 0, st, (const double*)f->xyz.as<double>(),
                       (const uint8_t*)f->alive.as<uint8_t>(), first, n_new, f->mode, f->edge, f->vorg,
                       (const uint64_t*)f->vcode_dev[0].as<uint64_t>(), V, (const int32_t*)nd.first_child,
                       (const double*)nd.corner, (const double*)nd.edge, keys[0], vals[0], small);
    HIP_TRY(ctx, hipGetLastError());
  }
  uint32_t sm[32];
  OCTL_TRY(read_words(ctx, 0, 32, sm));
  if (sm[SM_ERR] || sm[SM_INC_BAD]) {
    // forest_build's general path raises the error; its scalar block has to look untouched
    HIP_TRY(ctx, hipMemsetAsync(small + SM_ERR, 0, 4, st));
    return OCTL_OK;
  }
  const int64_t n_miss = sm[SM_INC_MISS], n_dead = sm[SM_INC_DEAD];
  const int64_t n_live = n_new - n_dead, n_hit = n_live - n_miss;
  if (n_ord + n_live != f->n_alive) return OCTL_OK;  // the tables were not those of the store: re-place

  // ---- 2. sort the new points by leaf (stable) ------------------------------------------------------------------
  int res = 0;
  {
    KTimer t(ctx, "inc_sort");
    const int bits = (n_miss == 0 && n_dead == 0) ? std::max(1, bits_for((uint64_t)n_nodes)) : 64;
    OCTL_TRY(octl_radix_sort_u64_u32(ctx, keys, vals, n_new, bits, f->hist, &res));
  }
  const uint64_t* skey = keys[res];
  const uint32_t* sval = vals[res];

  // ---- 3. new voxels: renumber the node table ------------------------------------------------------------------
  int64_t U = 0;
  const int32_t* shift = nullptr;
  const int32_t* new_root = nullptr;
  const uint32_t* miss_rank = nullptr;
  if (n_miss > 0) {
    KTimer t(ctx, "inc_new_voxels");
    OCTL_TRY(devbuf_reserve(ctx, f->flags, (size_t)(std::max(n_miss, n_live) + 8) * 4));
    // scratch: [miss_rank u32 n_miss | ucode u64 n_miss | shift i32 V | new_root i32 n_miss]
    const size_t o_ucode = (((size_t)n_miss + 8) * 4 + 15) & ~(size_t)15;
    const size_t o_shift = o_ucode + (size_t)n_miss * 8;
    const size_t o_root = o_shift + (((size_t)V + 8) * 4 + 15) / 16 * 16;
    OCTL_TRY(devbuf_reserve(ctx, f->entries, o_root + ((size_t)n_miss + 8) * 4));
    char* base = static_cast<char*>(f->entries.p);
    uint32_t* rank = reinterpret_cast<uint32_t*>(base);
    uint64_t* ucode = reinterpret_cast<uint64_t*>(base + o_ucode);
    int32_t* shift_w = reinterpret_cast<int32_t*>(base + o_shift);
    int32_t* root_w = reinterpret_cast<int32_t*>(base + o_root);
    OCTL_LAUNCH(k_inc_miss_heads, dim3(grid_for(n_miss)), dim3(256), 0, st, skey, n_hit, n_miss, rank);
    HIP_TRY(ctx, hipGetLastError());
    OCTL_TRY(octl_exclusive_scan_u32(ctx, rank, rank, n_miss, small + SM_INC_NEWVOX));
    OCTL_LAUNCH(k_inc_new_codes, dim3(grid_for(n_miss)), dim3(256), 0, st, skey, n_hit, n_miss,
                       (const uint32_t*)rank, ucode);
    HIP_TRY(ctx, hipGetLastError());
    uint32_t u32 = 0;
    OCTL_TRY(read_words(ctx, SM_INC_NEWVOX, 1, &u32));
    U = u32;
    if (n_nodes + U >= ((int64_t)1 << 31)) return OCTL_OK;
    OCTL_TRY(devbuf_reserve(ctx, f->vcode_dev[1], (size_t)(V + U) * 8));
    OCTL_LAUNCH(k_inc_merge_roots, dim3(grid_for(V + U)), dim3(256), 0, st,
                       (const uint64_t*)f->vcode_dev[0].as<uint64_t>(), V, (const uint64_t*)ucode, U,
                       f->vcode_dev[1].as<uint64_t>(), shift_w, root_w);
    HIP_TRY(ctx, hipGetLastError());
    NodeTable& nxt = f->nodes[f->cur ^ 1];
    OCTL_TRY(nodes_reserve(ctx, nxt, n_nodes + U));
    nxt.n = n_nodes + U;
    NodePtrs dst = node_ptrs(nxt);
    OCTL_LAUNCH(k_inc_copy_nodes, dim3(grid_for(n_nodes)), dim3(256), 0, st, nd, dst, n_nodes, V,
                       (int32_t)U, (const int32_t*)shift_w);
    HIP_TRY(ctx, hipGetLastError());
    OCTL_LAUNCH(k_inc_new_roots, dim3(grid_for(U)), dim3(256), 0, st, (const uint64_t*)ucode,
                       (const int32_t*)root_w, U, f->mode, f->edge, f->corner[0], f->corner[1], f->corner[2], f->vorg,
                       dst);
    HIP_TRY(ctx, hipGetLastError());
    if (n_blocks > 0) {
      OCTL_LAUNCH(k_inc_remap, dim3(grid_for(n_blocks)), dim3(256), 0, st, f->blk_node.as<int32_t>(),
                         n_blocks, V, (int32_t)U, (const int32_t*)shift_w);
      HIP_TRY(ctx, hipGetLastError());
    }
    shift = shift_w;
    new_root = root_w;
    miss_rank = rank;
  }

  // ---- 4. append to the leaf-ordered arrays, blocks behind the block table -----------------------------------------
  const int64_t n_total = n_ord + n_live;
  if (n_live > 0) {
    KTimer t(ctx, "inc_append");
    OCTL_TRY(devbuf_reserve(ctx, f->ord_idx, (size_t)n_total * 4, 1));
    OCTL_TRY(devbuf_reserve(ctx, f->xyz_ord, (size_t)n_total * 24, 1));
    OCTL_TRY(devbuf_reserve(ctx, f->pos_node, (size_t)n_live * 4));  // scratch: leaf of every sorted new point
    // (capacity as in forest_make_blocks: one block per point)
    OCTL_TRY(devbuf_reserve(ctx, f->blk_node, (size_t)n_total * 4, 1));
    OCTL_TRY(devbuf_reserve(ctx, f->blk_slot, (size_t)n_total * 4, 1));
    OCTL_TRY(devbuf_reserve(ctx, f->blk_start, (size_t)n_total * 4, 1));
    OCTL_TRY(devbuf_reserve(ctx, f->blk_size, (size_t)n_total * 4, 1));
    uint32_t* heads = f->flags.as<uint32_t>();
    if (n_miss == 0) {
      OCTL_TRY(devbuf_reserve(ctx, f->flags, (size_t)(n_live + 8) * 4));
      heads = f->flags.as<uint32_t>();
    } else {
      // the miss ranks live in f->entries, the heads go to f->flags (reserved above)
    }
    const int64_t* pose_off = f->pose_off_dev.as<int64_t>();
    OCTL_LAUNCH(k_inc_gather, dim3(grid_for(n_live)), dim3(256), 0, st, skey, sval, n_live, first, n_ord,
                       (const double*)f->xyz.as<double>(), pose_off, n_poses, V, (int32_t)U, shift, n_hit,
                       miss_rank, new_root, f->ord_idx.as<uint32_t>(), f->xyz_ord.as<double>(),
                       f->pos_node.as<int32_t>(), heads);
    HIP_TRY(ctx, hipGetLastError());
    OCTL_TRY(octl_exclusive_scan_u32(ctx, heads, heads, n_live, small + SM_NBLOCKS));
    OCTL_LAUNCH(k_inc_blocks, dim3(grid_for(n_live)), dim3(256), 0, st, (const uint32_t*)heads, skey, sval,
                       n_live, first, n_ord, n_blocks, pose_off, n_poses, (const int32_t*)f->pos_node.as<int32_t>(),
                       f->blk_node.as<int32_t>(), f->blk_slot.as<int32_t>(), f->blk_start.as<uint32_t>());
    HIP_TRY(ctx, hipGetLastError());
    OCTL_LAUNCH(k_inc_block_sizes, dim3(grid_for(n_live)), dim3(256), 0, st,
                       (const uint32_t*)f->blk_start.as<uint32_t>(), n_blocks, (const uint32_t*)(small + SM_NBLOCKS),
                       n_total, f->blk_size.as<int32_t>());
    HIP_TRY(ctx, hipGetLastError());
  }
  uint32_t nb_new = 0;
  if (n_live > 0) OCTL_TRY(read_words(ctx, SM_NBLOCKS, 1, &nb_new));

  // ---- commit ---------------------------------------------------------------------------------------------------
  if (U > 0) {
    f->cur ^= 1;
    std::swap(f->vcode_dev[0], f->vcode_dev[1]);
    f->n_voxels = V + U;
    f->vkeys_stale = true;  // forest_sync_vkeys downloads the codes
    for (auto& sg : f->level_segs) {
      if (sg.depth == 0 && sg.a == 0) {
        sg.b += U;
      } else {
        sg.a += U;
        sg.b += U;
      }
    }
  }
  f->fast_order_valid = false;
  f->n_ord = n_total;
  f->n_blocks = n_blocks + nb_new;
  f->mask_valid = false;
  f->store_dirty = false;
  f->built_store = f->n_store;
  f->built_poses = n_poses;
  f->append_only = true;
  if (info) {
    info->n_points = f->n_ord;
    info->n_voxels = f->n_voxels;
    info->n_nodes = f->nodes[f->cur].n;
    info->n_internal = f->n_internal;
    info->n_blocks = f->n_blocks;
    info->max_depth = f->max_depth_reached;
    info->n_levels = f->max_depth_reached;
  }
  *done = 1;
  return OCTL_OK;
}
