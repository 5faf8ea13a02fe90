// Declarations shared by the two build paths (build.hip: level-synchronous general path,
// bucket_build.hip: one MSD partition + one workgroup per bucket of voxels).
#pragma once
#include "forest.h"

constexpr uint32_t IDX_MASK = 0x7FFFFFFFu;

// slots of the context's small device scalar block (uint32 units)
enum {
  SM_ERR = 0,       // domain error flag
  SM_BBOX = 4,      // 6 x int32: min xyz, max xyz
  SM_NVOX = 12,     // voxels with points
  SM_NSPLIT = 13,   // nodes to split at the next level
  SM_NTILES = 14,   // tiles of the next level
  SM_ETOTAL = 15,   // total of the scanned tile histogram
  SM_NBLOCKS = 16,
  // 20, 21: kept points / blocks of apply_mask (21 also: slot-voxel count), 24: debug scan total
  SM_BK_NOORDER = 23,   // bucket build: some bucket has too many nodes / blocks for k_bucket_finish's own block order
  SM_BK_FLAGS = 25,     // bucket build: some bucket / voxel does not fit (BF_* bits)
  SM_BK_TOTAL = 26,     // bucket build: grand total of the scanned bucket table
  SM_BK_TODO = 27,      // bucket build: voxels left as one leaf for the level loop of build.hip
  SM_BK_MISSING = 28,
  SM_BK_OVERFULL = 18,  // bucket build: buckets with more than 4096 points (they are built in chunks of whole voxels)
  SM_BK_TICKET = 17,    // bucket build: workgroups of k_bucket_scan_totals that have finished (the last one forms the totals)
  SM_CK_COUNT = 29,     // bucket build: chunks of the buckets with more than 4096 points (k_bucket_plan)   // bucket build over a previous scheme: voxels of that scheme without points now
  SM_BK_LEVEL = 40,     // bucket build: internal nodes per level (7 words)
  // 64..: slot histogram (RANSAC batches), 512..: allreduce
  SM_GEOM = 64,         // bucket build: key geometry formed on the device (GeomDev, <= 192 bytes; the region is
                        // the slot histogram's during RANSAC - never at the same time)
};

struct NodePtrs {
  uint32_t *start, *count, *scount;
  int32_t *depth, *voxel, *parent, *first_child, *old_id, *epoch;
  double *corner, *edge;
};

static inline NodePtrs node_ptrs(NodeTable& t) {
  NodePtrs p;
  p.start = t.start.as<uint32_t>();
  p.count = t.count.as<uint32_t>();
  p.scount = t.scount.as<uint32_t>();
  p.depth = t.depth.as<int32_t>();
  p.voxel = t.voxel.as<int32_t>();
  p.parent = t.parent.as<int32_t>();
  p.first_child = t.first_child.as<int32_t>();
  p.old_id = t.old_id.as<int32_t>();
  p.epoch = t.epoch.as<int32_t>();
  p.corner = t.corner.as<double>();
  p.edge = t.edge.as<double>();
  return p;
}

// bucket_build.hip: complete build of a fresh forest (K-driven scheme or K < 0, no previous scheme) by
// one MSD partition into buckets of consecutive voxels + one workgroup per bucket.  *done = 0 when
// the path does not apply and the caller must run the general path.
struct BucketBuildArgs {
  int64_t K;
  const uint8_t* scheme_dev;  // per pose slot: 1 = the pose drives the scheme; nullptr = all poses
  int cur_epoch;
  int max_depth;
  // the previous scheme (subdivide on an already subdivided forest): nodes that were internal before keep
  // their epoch, and every voxel of the previous scheme has to be there again.  nullptr / 0 when fresh.
  const int32_t* old_fc = nullptr;
  const int32_t* old_epoch = nullptr;
  const uint64_t* old_vcode = nullptr;  // sorted packed voxel keys; old root r = voxel r
  int64_t old_voxels = 0;
};
struct BucketBuildGeom {  // decoding of the linear voxel keys: lin = ((qx-min0)*ny + (qy-min1))*nz + (qz-min2)
  int min[3];
  uint64_t ny, nz;
  bool order_done = false;  // f->fast_order holds the blocks in the reference's listing order
};
// *pending = number of voxels left as single leaves for the level loop of build.hip (flagged roots).
int forest_bucket_build(octl_forest* f, const BucketBuildArgs& a, NodeTable& nt, int* done,
                        std::vector<octl_forest::LevelSeg>* segs, int64_t* n_internal, int* levels,
                        int64_t* n_voxels, int64_t* n_blocks, int64_t* pending, BucketBuildGeom* geom);

// bucket_build.hip: one stable partition of a single cube's store by the child digits of its first pm levels
// (records of 32 bytes: x, y, z f64 | six digits << 1 | bad | store index + scheme bit)
int forest_prefix_partition(octl_forest* f, int pm, const void** recs_out, const uint32_t** bstart, uint32_t* bstride,
                            const uint32_t** bad_flag);
