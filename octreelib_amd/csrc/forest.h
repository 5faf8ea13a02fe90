// Device-resident state of one Grid / OctreeManager / Octree ("forest" of top-level cubes).
#pragma once
#include "common.h"

// packed voxel key: (qx-ox+B)<<42 | (qy-oy+B)<<21 | (qz-oz+B), B = 2^20; lexicographic (x,y,z) order of
// the integer voxel coordinates == numeric order of the key (np.unique(axis=0) order,
// grid/grid.py:79-81).  The key is RELATIVE to the forest's voxel origin o - the minimum voxel of the first
// build after a clear (forest_fix_origin) - so that voxel indices of any size work (the reference takes any
// int64, grid.py:72-76: UTM coordinates at 1 m voxels are 5 * 10^6) as long as the scene stays within 2^20
// voxels of where it started; absolute indices are limited to +-2^30 (they travel as int32).
#define OCTL_VOX_BIAS (1 << 20)
#define OCTL_VOX_ABS_LIMIT (1 << 30)
#define OCTL_VOX_DEAD (~0ull)
struct VoxOrg {
  int32_t x = 0, y = 0, z = 0;
};
__host__ __device__ static inline uint64_t vkey_pack(int64_t qx, int64_t qy, int64_t qz, const VoxOrg& g) {
  return ((uint64_t)(qx - g.x + OCTL_VOX_BIAS) << 42) | ((uint64_t)(qy - g.y + OCTL_VOX_BIAS) << 21) |
         (uint64_t)(qz - g.z + OCTL_VOX_BIAS);
}
__host__ __device__ static inline bool vkey_in_window(int64_t qx, int64_t qy, int64_t qz, const VoxOrg& g) {
  const int64_t ax = qx - g.x, ay = qy - g.y, az = qz - g.z;
  return ax > -OCTL_VOX_BIAS && ax < OCTL_VOX_BIAS && ay > -OCTL_VOX_BIAS && ay < OCTL_VOX_BIAS &&
         az > -OCTL_VOX_BIAS && az < OCTL_VOX_BIAS;
}
__host__ __device__ static inline void vkey_decode(uint64_t k, const VoxOrg& g, int64_t q[3]) {
  q[0] = (int64_t)((k >> 42) & 0x1FFFFF) - OCTL_VOX_BIAS + g.x;
  q[1] = (int64_t)((k >> 21) & 0x1FFFFF) - OCTL_VOX_BIAS + g.y;
  q[2] = (int64_t)(k & 0x1FFFFF) - OCTL_VOX_BIAS + g.z;
}

// Structure-of-arrays scheme node table on the device.
struct NodeTable {
  DevBuf start, count, scount;  // u32: range in the level buffers, scheme-pose point count
  DevBuf depth, voxel, parent, first_child, old_id, epoch;  // i32
  DevBuf corner, edge;          // f64 x3, f64
  int64_t cap = 0, n = 0;
};

struct octl_forest {
  octl_ctx* ctx = nullptr;
  int mode = 0;  // 0 grid, 1 single cube
  double corner[3] = {0, 0, 0};
  double edge = 1.0;

  // point store: all poses concatenated in slot order (pose-major), insertion order inside
  DevBuf xyz;     // f64 [n_store][3]
  DevBuf alive;   // u8  [n_store]
  // octl_forest_add_pose_adopt: xyz.p is the CALLER's device buffer (read in place, never written, never
  // freed here); the forest's own block waits in xyz_own and the store is copied into it (store_materialize)
  // before anything appends to or reorders the store
  bool store_borrowed = false;
  DevBuf xyz_own;
  // the store's points have not been folded into bbox_dev yet (a cloud taken in place: adopted or routed).
  // Only ever set while the store is that ONE pose; the build folds the box into its histogram pass when it
  // has a geometry hint to work with, or runs the box pass first (store_compute_bbox)
  bool bbox_pending = false;
  // octl_forest_set_contents left rows OUTSIDE the cube of the leaf that holds them (map_leaf_points may return
  // anything, octree.py:114-123).  The reference keeps such rows in that leaf and raises IndexError as soon as the
  // leaf is split; a count-driven build here would silently re-bucket them by their coordinates - so it refuses.
  bool displaced_rows = false;
  std::vector<int64_t> pose_off{0};  // [P+1] offsets into the store
  int64_t n_store = 0, n_alive = 0;
  bool store_dirty = true;  // points were added/removed since the last build
  // what the last build covered, for the incremental insertion of incremental.hip: the store prefix
  // and the poses the tables describe, and whether every change since was a pose appended behind them
  int64_t built_store = 0;
  int built_poses = 0;
  bool append_only = true;
  // voxel bounding box of every point added since the last clear, kept by the ingest kernel
  // (api.hip): int32 x 8 = {min x,y,z, max x,y,z, domain-error flag, point outside a hinted box}
  DevBuf bbox_dev;
  // octl_forest_clear does not launch anything: the box on the device is STALE until something resets it - the next
  // ingest / box pass (bbox_ensure), or the first kernel of a build that finds the box itself (k_build_begin)
  bool bbox_stale = true;
  // the alive flags of a cloud taken in place (adopted, routed) are not written until something needs them
  // (alive_ensure: a second pose, a re-placement; apply_mask fills them in its first kernel): every point is alive
  bool alive_stale = false;

  // scheme of the last build
  NodeTable nodes[2];
  int cur = 0;               // nodes[cur] is the current table
  int epoch = 0;             // number of K-driven builds so far
  bool built = false;
  VoxOrg vorg;                  // voxel origin of the packed keys (fixed by the first build after a clear)
  bool vorg_set = false;
  std::vector<uint64_t> vkeys;  // packed integer coordinates of the top-level voxels, sorted
                                // (filled lazily from vlin_dev: forest_sync_vkeys)
  bool vkeys_stale = false;
  int64_t n_voxels = 0;
  DevBuf vcode_dev[2];          // u64 [n_voxels] sorted packed voxel keys on the device (forest_sync_vcodes); [1] = scratch
  bool vcode_valid = false;
  DevBuf vlin_dev;              // u64 [n_voxels] compact linear voxel keys of the last build
  int vl_min[3] = {0, 0, 0};    // decoding of vlin_dev: lin = ((qx-min0)*ny + (qy-min1))*nz + (qz-min2)
  uint64_t vl_ny = 1, vl_nz = 1;
  int64_t n_internal = 0;
  bool uniform_epoch = true;  // every internal node of the current scheme has the same epoch
  int32_t max_depth_reached = 0;

  // leaf-ordered arrays of the last build
  DevBuf ord_idx;    // u32 [n_ord] store index of the point at storage position i
  DevBuf xyz_ord;    // f64 [n_ord][3]
  DevBuf pos_node;   // i32 build scratch: scheme leaf of position i while a build runs (afterwards the block
                     // table alone says which leaf a position belongs to)
  int64_t n_ord = 0;
  // non-empty (leaf, pose) blocks in storage order
  DevBuf blk_node, blk_slot, blk_start, blk_size;  // i32, i32, u32, i32
  int64_t n_blocks = 0;
  // RANSAC state
  DevBuf mask;       // u8 [n_ord]
  DevBuf blk_eval;   // u8 [n_blocks] block was evaluated since the last apply_mask
  bool mask_valid = false;
  // octl_forest_apply_mask_async: the kernels are enqueued and the arrays swapped, the surviving point / block counts
  // are still on their way to the pinned mirror - forest_settle (every entry point's first step) waits for them
  bool totals_pending = false;
  uint32_t totals_seq = 0;      // wait sequence number the kernels publish under
  int64_t totals_n_before = 0;  // ordered points before the compaction

  // an upper bound of the points a block can hold, when the host knows one (a count-driven build from all poses
  // leaves at most K points per leaf, and masks / filters only remove points): the RANSAC launch skips the instances
  // for block sizes that cannot occur.  INT64_MAX: unknown (late poses, installed schemes, replaced contents ...)
  int64_t max_block_hint = INT64_MAX;
  DevBuf rs_scratch, rs_order, rs_hyp, rs_plane, rs_count, rs_index;  // ransac staging
  // the blocks in the reference's listing order as the bucket build left them (one pose, one epoch):
  // forest_reference_order takes this instead of computing it; any change of the blocks invalidates it
  DevBuf fast_order;
  bool fast_order_valid = false;
  // what pose_off_dev holds (offsets that have not changed are not uploaded again)
  std::vector<int64_t> pose_off_uploaded;
  DevBuf ord_idx2, xyz_ord2;  // compaction targets (swapped with the live arrays)
  DevBuf blk_node2, blk_slot2, blk_start2, blk_size2;

  // bucket build (bucket_build.hip): the cloud partitioned into buckets of consecutive voxels
  DevBuf part_xyz[2];  // 32-byte records {x, y, z, voxel | child digits, store index | scheme bit}, two passes
  DevBuf bk_table;     // u32 [digit][supertile] partition histogram (scanned in place)
  DevBuf bk_tot;       // u32 [BK_ROWS][n_buckets] per-bucket totals (scanned in place)
  DevBuf bk_chunks;    // chunk plan of the buckets with more than 4096 points: descriptors, per-chunk totals, range per bucket
  DevBuf bk_vox;       // u32 x3 staging {linear key, points | todo, scheme points} of the j-th voxel of bucket b
                       // at [bucket start + j]
  DevBuf bk_node;      // u32 x3 staging {voxel << 18 | path, level << 28 | ordinal, parent ordinal} of the j-th internal
                       // node of bucket b at [bucket start + j]
  DevBuf leafinfo;     // u32 [n_alive] per leaf-ordered point: leaf in bucket terms (parent ordinal, child digit, depth) | head flags
  // levels of the current node table as (first, end, depth) ranges of node ids: one per level when a
  // single build path numbered the nodes, two where the level loop has subdivided voxels the bucket
  // build left behind (order.hip sweeps them by depth)
  struct LevelSeg { int64_t a, b; int depth; };
  std::vector<LevelSeg> level_segs;
  // build scratch (kept between builds to avoid re-allocation)
  DevBuf vkey, path, lin[2], val[2], hist, idxbuf[2], pathbuf[2], flags, entries, split[2],
      split_tiles[2], child_sc, pose_off_dev, scheme_dev, root_up;
};

int forest_build(octl_forest* f, int64_t K, const uint8_t* scheme_mask, int32_t n_mask,
                 int32_t keep_scheme, int32_t max_depth, octl_build_info* info);
int nodes_reserve(octl_ctx* ctx, NodeTable& t, int64_t cap);
void nodes_free(octl_ctx* ctx, NodeTable& t);

// ransac.hip
int ransac_launch(octl_ctx* ctx, const double* xyz_dev, int64_t n_points,
                  const uint32_t* blk_start, const int32_t* blk_size, const int32_t* order_dev,
                  int64_t nb, const double* hyp_dev, int32_t H, int32_t k, double thr,
                  uint8_t* mask_dev, float* plane_dev, int32_t* count_dev, int32_t* index_dev,
                  uint8_t* evaluated_dev, DevBuf& scratch, int64_t max_block = INT64_MAX);
// the hypothesis table must hold draws from [0, 1) (np.random.random, cuda_ransac.py:39-41): anything
// else would index outside the block in the sampling arithmetic (cuda_ransac.py:103-107)
// api.hip: an empty store takes over a library-owned device buffer (swap) instead of copying it
int store_adopt(octl_forest* f, DevBuf& src, int64_t n, bool* adopted);
// api.hip: fold the whole store into bbox_dev (clears bbox_pending); copy a borrowed store into the forest's own block
int store_compute_bbox(octl_forest* f);
// api.hip: the box on the device is valid (reset if octl_forest_clear left it stale) / the alive flags are written
int bbox_ensure(octl_forest* f);
int alive_ensure(octl_forest* f);
int store_materialize(octl_forest* f);
int ransac_check_table(octl_ctx* ctx, const double* hyp, int32_t H, int32_t k);
// build.hip: (re)build the (leaf, pose) block table from pos_node / ord_idx
int forest_make_blocks(octl_forest* f);
// reads the block count (and the domain-error flag) left on the device: one synchronisation
int forest_finish_blocks(octl_forest* f, uint32_t* err_out);
// host copy of the voxel keys (synchronises when stale)
int forest_sync_vkeys(octl_forest* f);
// build.hip: fix the voxel origin of the packed keys from a voxel box (first build after a clear) and check
// that the box lies inside the window of the keys; OCTL_E_DOMAIN otherwise
int forest_fix_origin(octl_forest* f, const int bb[6]);
// incremental.hip: device copy of the packed voxel keys; poses appended to a built forest placed into
// its scheme in O(new points).  *done = 0: not applicable, nothing was changed
int forest_sync_vcodes(octl_forest* f);
int forest_insert_incremental(octl_forest* f, int* done, octl_build_info* info);

// api.hip: wait for the counts an asynchronous apply_mask left in flight (octl_forest_apply_mask_async) and book
// them; a no-op otherwise.  Every entry point that looks at the forest calls it first.
int forest_settle(octl_forest* f);
// ... and forget them (clear / destroy: nothing will look at the counts)
void forest_forget_pending(octl_forest* f);
