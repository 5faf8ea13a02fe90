/*
 * octreelib_hip.h — C ABI of liboctree_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for octreelib's point-cloud -> octree-grid build-and-query
 * path.  The reference (prime-slam/octreelib, /root/reference) has no FFI of its own: its
 * boundary is the Python class surface plus ONE device operator.  Every entry point below
 * names the reference interface it replaces (paths relative to /root/reference):
 *
 *   octl_forest_*            Grid / OctreeManager / Octree state
 *                            grid/grid.py:49-56, octree_manager/octree_manager.py:21-34,
 *                            octree/octree_base.py:143-158
 *   octl_forest_add_pose     Grid.insert_points                grid/grid.py:58-109
 *                            OctreeManager.insert_points       octree_manager.py:161-171
 *                            Octree.insert_points              octree/octree.py:235-239
 *   octl_forest_build        Grid.subdivide                    grid/grid.py:244-258
 *                            OctreeManager.subdivide           octree_manager.py:36-66
 *                            OctreeNode.subdivide/_as          octree/octree.py:20-53
 *                            OctreeNode.insert_points          octree/octree.py:67-100
 *                            OctreeNode._generate_children     octree/octree.py:177-191
 *   octl_forest_ransac       Grid.map_leaf_points_cuda_ransac  grid/grid.py:124-215
 *   octl_forest_apply_mask   OctreeManager/Octree.apply_mask   octree_manager.py:173-180,
 *                                                              octree/octree.py:265-274
 *   octl_ransac_evaluate     CudaRansac.evaluate               ransac/cuda_ransac.py:43-81
 *                            kernel                            ransac/cuda_ransac.py:85-155
 *                            get_plane_from_points             ransac/util.py:27-84
 *                            measure_distance                  ransac/util.py:12-24
 *
 * Conventions
 *   - every function returns 0 on success, a negative OCTL_E_* code on failure; the text of
 *     the last failure on a context is octl_last_error(ctx).  No exception crosses the ABI.
 *   - host buffers belong to the caller and are only touched during the call; all device
 *     memory belongs to the context / forest.  Calls are blocking with respect to host
 *     buffers.  A context is bound to one device and one HIP stream and is not thread safe
 *     (one context per GPU; multi-GPU = one process per rank).
 *   - "pose slot" = index of a pose in insertion order (0..P-1); the Python layer maps the
 *     user's pose numbers to slots.
 *   - coordinates are float64 row-major (n,3), exactly as the reference stores them
 *     (internal/voxel.py:81-83).
 */
#ifndef OCTREELIB_HIP_H
#define OCTREELIB_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OCTL_ABI_VERSION 1

enum {
  OCTL_OK = 0,
  OCTL_E_INVALID = -1,    /* bad argument                                              */
  OCTL_E_HIP = -2,        /* HIP runtime error (message has the hipError string)        */
  OCTL_E_NOMEM = -3,      /* device or host allocation failed                           */
  OCTL_E_DOMAIN = -4,     /* input outside the parity domain (point outside its cube,
                             non-finite coordinate, voxel index out of range)            */
  OCTL_E_DEPTH = -5,      /* max_depth exceeded (duplicate points with a count criterion
                             recurse forever in the reference)                           */
  OCTL_E_STATE = -6,      /* call out of order (e.g. ransac before build)               */
  OCTL_E_COMM = -7        /* RCCL error                                                  */
};

typedef struct octl_ctx octl_ctx;
typedef struct octl_forest octl_forest;

/* ---- context ------------------------------------------------------------------------ */
int octl_abi_version(void);
int octl_device_count(int* count);
int octl_ctx_create(int device_id, octl_ctx** out);
void octl_ctx_destroy(octl_ctx* ctx);
const char* octl_last_error(const octl_ctx* ctx);
/* synchronise the context's stream */
int octl_ctx_sync(octl_ctx* ctx);
/* timing of the kernels launched by the last build / ransac call, by name; fills up to cap
 * entries, returns the number available through *n.  Milliseconds from hipEvents on the
 * context's stream; only recorded when profiling was enabled with octl_ctx_set_profiling:
 * 1 = every timed region, 2 = the RANSAC scoring kernel only (every hipEvent pair drains the
 * pipeline for ~10 us: a dozen of them are 2 % of a 10 M-point step), 0 = off.               */
int octl_ctx_set_profiling(octl_ctx* ctx, int enabled);
int octl_ctx_get_timings(octl_ctx* ctx, char* names, int name_stride, float* ms,
                         int64_t* launches, int cap, int* n);

/* ---- forest: a Grid (many top-level voxels) or one cube (Octree / OctreeManager) ------ */

/* mode 0: grid of top-level voxels of edge L anchored at grid_corner (GridConfig.
 *         voxel_edge_length / corner, grid/grid_base.py:70-71).  L must be integer valued
 *         (the reference truncates voxel coordinates with astype(int), grid.py:72-76).
 * mode 1: a single cube [corner, corner+edge)^3 (Octree(config, corner_min, edge_length) /
 *         OctreeManager(..., corner_min, edge_length)).                                    */
int octl_forest_create(octl_ctx* ctx, int mode, const double corner[3], double edge,
                       octl_forest** out);
void octl_forest_destroy(octl_forest* f);

/* Forget all poses and the scheme but keep the device allocations (a fresh Grid without the
 * hipMalloc cost).                                                                          */
int octl_forest_clear(octl_forest* f);

/* Append the cloud of a new pose slot (host pointer, copied to the device).  Returns the
 * slot through *slot.  Replaces Grid.insert_points' storage step.                        */
int octl_forest_add_pose(octl_forest* f, const double* xyz, int64_t n, int32_t* slot);
/* Same, from a device pointer (device-to-device copy fused with the bounding-box pass; no PCIe).  The
 * source is consumed in stream order: it must stay unchanged until a call that leaves the context's stream IDLE:
 * octl_ctx_sync and every call that downloads to host memory (octl_forest_get_mask, _get_blocks, _get_nodes,
 * _gather_blocks, ...).  NOT among them since round 5: octl_forest_build and octl_forest_apply_mask - they
 * return when the scalars they hand back have arrived in the pinned mirror, while their last kernels may still
 * be running (stream order is kept: anything enqueued on the context afterwards runs behind them).           */
int octl_forest_add_pose_device(octl_forest* f, const double* xyz_dev, int64_t n,
                                int32_t* slot);
/* Same, WITHOUT the copy: an empty forest reads the caller's device buffer in place as its first pose
 * (Grid.insert_points' storage step, grid/grid.py:58-109, for a cloud that is in HBM already).  The library
 * never writes to the buffer and never frees it; the caller keeps it alive and unchanged until the forest is
 * cleared or destroyed, or until more points are added to it (octl_forest_add_pose*, octl_forest_extend_pose
 * take a private copy first).  A forest that already holds points, or a pointer that is not 16-byte aligned,
 * takes the copying path of octl_forest_add_pose_device.  The voxel bounding box of such a cloud is found by the
 * build's own histogram pass when the context has built a cloud of the same scene before, by a box pass
 * (24 B/point read) otherwise.                                                                    */
int octl_forest_add_pose_adopt(octl_forest* f, const double* xyz_dev, int64_t n, int32_t* slot);
/* Replace the CONTENTS of the leaves and keep the scheme: OctreeNode.map_leaf_points (octree/octree.py:114-123)
 * stores whatever the caller's function returned for a leaf's cloud - fewer rows, more rows, rows outside the
 * leaf's cube.  The new non-empty (leaf, pose) blocks are given in storage order: leaf node, pose slot, size, and
 * the rows of all blocks one after the other.  The point store is rebuilt (pose-major, a pose's points in the
 * order of its blocks); a later subdivide re-places the points by their coordinates (a row outside its cube then
 * fails where the reference raises IndexError).                                                          */
int octl_forest_set_contents(octl_forest* f, int64_t n_blocks, const int32_t* blk_node, const int32_t* blk_slot,
                             const int32_t* blk_size, const double* xyz);
/* Append more points to an EXISTING pose slot, any slot (OctreeManager.insert_points on a pose that
 * already has an octree, octree_manager.py:161-171; Octree.insert_points, octree.py:235-239).  The
 * points of later poses move up in the pose-major store.                                       */
int octl_forest_extend_pose(octl_forest* f, int32_t slot, const double* xyz, int64_t n);
/* Same, from a device pointer (device-to-device copy; ordered behind an octl_dev_upload_async into it).  */
int octl_forest_extend_pose_device(octl_forest* f, int32_t slot, const double* xyz_dev, int64_t n);

typedef struct octl_build_info {
  int64_t n_points;     /* alive points that were placed                                   */
  int64_t n_voxels;     /* top-level voxels (roots)                                        */
  int64_t n_nodes;      /* all scheme nodes: roots + 8 per internal node                   */
  int64_t n_internal;   /* internal scheme nodes                                           */
  int64_t n_blocks;     /* non-empty (leaf, pose) blocks                                   */
  int32_t max_depth;    /* deepest leaf                                                    */
  int32_t n_levels;     /* subdivision levels executed                                     */
} octl_build_info;

/* Build the scheme for the count criterion len(points) > K over the union of the poses
 * whose scheme_mask[slot] != 0 (NULL = all poses), then place every alive point of every
 * pose in its scheme leaf.  keep_scheme != 0 places the points in the EXISTING scheme
 * (a pose inserted after a subdivide inherits it, octree_manager.py:161-171) and ignores
 * K / scheme_mask: when whole poses were appended since the last build only their points are
 * placed and appended behind the leaf-ordered arrays (cost independent of what is stored);
 * after octl_forest_extend_pose / octl_forest_set_scheme every point is placed again.  K < 0 means "never split" (the state right after insert_points).
 * max_depth guards the recursion the reference does not bound (<= 0: default 63).
 * On OCTL_E_DOMAIN / OCTL_E_DEPTH / OCTL_E_NOMEM / OCTL_E_HIP the forest keeps its points and
 * voxels but is left without a scheme (the next build starts from the top-level voxels).     */
int octl_forest_build(octl_forest* f, int64_t K, const uint8_t* scheme_mask, int32_t n_mask,
                      int32_t keep_scheme, int32_t max_depth, octl_build_info* info);

/* Install a host-defined subdivision scheme (for criteria that are arbitrary host callables,
 * octree.py:26: the host decides which nodes split, the device does the placement): nodes
 * [0,V) are the roots in voxel order, the 8 children of node i are first_child[i]..+7 (-1 =
 * leaf), epoch[i] = build at which node i became internal.  Follow with
 * octl_forest_build(keep_scheme = 1) to place every point in this scheme.                     */
int octl_forest_set_scheme(octl_forest* f, const int32_t* first_child, const int32_t* epoch,
                           int64_t n_nodes, int32_t new_epoch);

/* ---- results of the last build (host copies; size-query = pass NULL outputs) ---------- */

/* Scheme nodes.  For node i: voxel (root) index, depth, parent (-1 for roots), first child
 * (-1 for leaves; children are first_child..first_child+7 in child_id order, child_id =
 * 4*ix+2*iy+iz as octree.py:94-97), corner_min[3] and edge_length computed with the
 * reference's arithmetic (octree.py:181-191), epoch = build call at which the node became
 * internal (0 for leaves).  Any output pointer may be NULL.                                */
int octl_forest_get_nodes(octl_forest* f, int64_t cap, int32_t* voxel, int32_t* depth,
                          int32_t* parent, int32_t* first_child, double* corner,
                          double* edge, int32_t* epoch, int64_t* n_nodes);
/* Integer coordinates of the top-level voxels, (V,3) int64, lexicographically sorted — the
 * order of np.unique(axis=0) at grid.py:79-81.                                            */
int octl_forest_get_voxels(octl_forest* f, int64_t cap, int64_t* coords, int64_t* n_voxels);
/* Non-empty (leaf, pose) blocks in storage order (voxel, leaf path, pose): scheme node id of
 * the leaf, pose slot, start and size in the leaf-ordered arrays.                          */
int octl_forest_get_blocks(octl_forest* f, int64_t cap, int32_t* node, int32_t* slot,
                           int64_t* start, int32_t* size, int64_t* n_blocks);
/* Ranks (indices into the voxel table, ascending) of the top-level voxels in which a pose slot
 * currently has points - Grid.__pose_voxel_coordinates (grid.py:53,108) without fetching the
 * node and block tables.                                                                     */
int octl_forest_get_slot_voxels(octl_forest* f, int32_t slot, int64_t cap, int32_t* voxel_ranks,
                                int64_t* n);
/* Counters of one pose without fetching the tables (octree.py:144-175, octree_manager.py:132-159,
 * grid.py:343-362): points and non-empty leaves of the slot, reduced on the device ...            */
int octl_forest_slot_counts(octl_forest* f, int32_t slot, int64_t* n_points, int64_t* n_leaves);
/* ... and the internal scheme nodes of every top-level voxel (V int32): n_nodes of a pose is the sum of
 * 1 + 8 * internal over the voxels the pose was inserted into.                                     */
int octl_forest_internal_per_voxel(octl_forest* f, int64_t cap, int32_t* counts, int64_t* n_voxels);
/* Leaf-ordered point permutation: perm[i] = index of the point at storage position i in
 * the concatenation of all pose clouds in slot order (pose-local index = perm - offset of
 * its slot).  Within a block the order is ascending (stable).                              */
int octl_forest_get_perm(octl_forest* f, int64_t cap, int64_t* perm, int64_t* n);
/* Leaf-ordered coordinates (n,3) f64 for storage positions [start, start+count).           */
int octl_forest_get_points(octl_forest* f, int64_t start, int64_t count, double* xyz);
/* The rows of the blocks block_ids[0..m) (host array), concatenated in THAT order: one device gather and one
 * download instead of a host loop over the blocks.  *n_points = rows of the selection; rows are written only when
 * xyz != NULL and cap >= *n_points (call with xyz = NULL for the size).  Replaces the concatenation loops of
 * Grid.get_points / OctreeManager.get_points / Octree.get_points (grid.py:234-242, octree_manager.py:121-130,
 * octree.py:55-65,252-254), whose order - managers in creation order, leaves in cached order - the caller encodes
 * in block_ids.                                                                                                */
int octl_forest_gather_blocks(octl_forest* f, const int32_t* block_ids, int64_t m, int64_t cap, double* xyz,
                              int64_t* n_points);

/* ---- RANSAC on the forest (device resident) ------------------------------------------- */

/* Per-leaf RANSAC plane fit over the blocks listed in block_order (indices into the block
 * table, in the order the reference would concatenate them: grid.py:173-191) — the list is
 * one "batch"; virtual start indices are the running sum of the block sizes in that order
 * (cuda_ransac.py:64-66) and enter the sampling arithmetic exactly as in the reference
 * (cuda_ransac.py:103-107).  hypotheses: (H,k) f64 table (cuda_ransac.py:39-41), H <= 1024.
 * Outputs (any may be NULL), all indexed like block_order: plane (nb,4) f32, best_count
 * (nb) i32, best_index (nb) i32 (lowest hypothesis index attaining the maximum; the
 * reference lets any tied hypothesis win, cuda_ransac.py:140-145).  The inlier mask stays
 * on the device (octl_forest_get_mask / octl_forest_apply_mask).                           */
int octl_forest_ransac(octl_forest* f, const int32_t* block_order, int64_t nb,
                       const double* hypotheses, int32_t H, int32_t k, double threshold,
                       float* plane, int32_t* best_count, int32_t* best_index);
/* All non-empty (leaf, pose) blocks in the order in which the reference lists them: pose
 * slot major, then top-level voxel (lexicographic), then the octree's cached-leaf list
 * (octree_base.py:152-158, octree.py:183-191,256-263), computed on the device.  e0[slot]
 * (nullable = 0) is the build epoch at which the pose's octrees were created: the cached
 * list is history dependent (a pose inserted after a subdivide inherits the whole scheme at
 * once, octree_manager.py:161-171).  order (n_blocks) i32 = indices into the block table.   */
int octl_forest_reference_order(octl_forest* f, const int32_t* e0, int32_t n_e0, int64_t cap,
                                int32_t* order, int64_t* n_blocks);
/* RANSAC over ALL blocks in that order, one reference "batch" per poses_per_batch consecutive
 * slots (grid.py:126,149-157,194); order, virtual starts and the kernel all stay on the device,
 * nothing but the (H,k) table crosses PCIe.  The async launches are NOT synchronised: follow
 * with octl_ctx_sync / get_mask / apply_mask.                                               */
int octl_forest_ransac_all(octl_forest* f, int32_t poses_per_batch, const int32_t* e0,
                           int32_t n_e0, const double* hypotheses, int32_t H, int32_t k,
                           double threshold);
/* Inlier mask of the last ransac call(s), uint8 per storage position.                      */
int octl_forest_get_mask(octl_forest* f, int64_t cap, uint8_t* mask, int64_t* n);
/* Drop the points whose mask byte is 0 from the blocks that were evaluated (apply_mask,
 * octree.py:137-142): compacts the leaf-ordered arrays, updates the block table and marks
 * the points dead for later builds.  Returns the surviving point count - as soon as the count has reached the
 * host: the compaction kernel may still be running (it reads the forest's own arrays only, never a cloud that
 * was taken in place; work enqueued on the context later runs behind it).  octl_ctx_sync waits for it.        */
int octl_forest_apply_mask(octl_forest* f, int64_t* n_alive);
/* The same without waiting for the count (round 6): the kernels are enqueued and the call returns; the surviving
 * point and block counts are booked by the next call that looks at the forest (every octl_forest_* entry point does
 * so first; octl_forest_clear / _destroy drop them unread).  Grid.map_leaf_points_cuda_ransac returns nothing
 * (grid/grid.py:124-215), so a scan loop - insert, subdivide, RANSAC, apply_mask, clear - never waits for this count
 * at all and the next scan's first kernels queue up behind this one's last.  One compaction per context can be in
 * flight: a second one (any forest of the context) books the first before it starts.  A RANSAC launch that met a
 * block larger than it was promised is reported by the next wait of the context, whichever call that is.        */
int octl_forest_apply_mask_async(octl_forest* f);
/* Book the counts of an octl_forest_apply_mask_async now (a no-op otherwise); *n_alive (nullable): surviving points. */
int octl_forest_settle(octl_forest* f, int64_t* n_alive);
/* OctreeNode.filter (octree/octree.py:102-112) for point-count predicates, on the device: every leaf of
 * the poses with slot_sel[slot] != 0 whose point count is outside [lo, hi] is emptied (its points
 * leave the tree), followed by the same compaction as apply_mask.  A criterion `len(points) >= c` is
 * [c, INT64_MAX], `len(points) < c` is [0, c-1], several criteria intersect.                    */
int octl_forest_filter_count(octl_forest* f, const uint8_t* slot_sel, int32_t n_sel, int64_t lo,
                             int64_t hi, int64_t* n_alive);
/* Drop points by an explicit host mask over storage positions (filter / map_leaf_points
 * paths of the Python layer for arbitrary callables).                                       */
int octl_forest_apply_host_mask(octl_forest* f, const uint8_t* mask, int64_t n,
                                int64_t* n_alive);

/* ---- the reference's operator, stand-alone ------------------------------------------- */

/* CudaRansac(threshold, H, k).evaluate(point_cloud, block_sizes) -> mask
 * (ransac/cuda_ransac.py:43-81).  point_cloud (M,3) f64 leaf-major, block_sizes (B) i32,
 * hypotheses (H,k) f64.  mask_out (M) uint8.  Extensions (nullable): planes_out (B,4) f32,
 * best_count_out (B) i32, best_index_out (B) i32.                                          */
int octl_ransac_evaluate(octl_ctx* ctx, const double* point_cloud, int64_t M,
                         const int32_t* block_sizes, int64_t B, const double* hypotheses,
                         int32_t H, int32_t k, double threshold, uint8_t* mask_out,
                         float* planes_out, int32_t* best_count_out, int32_t* best_index_out);

/* ---- multi-GPU: shard the grid by top-level voxel, route points to their owner --------- */

/* Owner rank of a top-level voxel (pure function, also used by the CPU tests).             */
int32_t octl_voxel_owner(int64_t qx, int64_t qy, int64_t qz, int32_t n_ranks);
#define OCTL_UNIQUE_ID_BYTES 128
/* rank 0 creates the RCCL unique id; the caller distributes the bytes to all ranks.        */
int octl_comm_unique_id(uint8_t id[OCTL_UNIQUE_ID_BYTES]);
int octl_comm_init(octl_ctx* ctx, int32_t n_ranks, int32_t rank,
                   const uint8_t id[OCTL_UNIQUE_ID_BYTES]);
int octl_comm_destroy(octl_ctx* ctx);
/* Route a device-resident cloud (n,3) f64 with global indices gidx (n) i64 (may be NULL:
 * then index_base + i) to the ranks that own the points' top-level voxels (edge L, grid
 * corner c): key + count per destination, 8x8 count exchange, one grouped ncclSend/ncclRecv
 * all-to-all over xGMI.  The received cloud stays on the device inside ctx; its size is
 * returned through *n_recv and it is handed to a forest with octl_forest_add_pose_routed:
 * an empty forest takes the receive buffer over (its own store buffer goes to the router in
 * exchange, no copy), a forest that holds poses already copies the cloud behind them.  Either
 * way the routed cloud is consumed: route again before the next add_pose_routed.             */
int octl_route_points(octl_ctx* ctx, const double* xyz_dev, const int64_t* gidx_dev,
                      int64_t n, int64_t index_base, const double corner[3], double L,
                      int64_t* n_recv, int64_t* send_counts /* [n_ranks], nullable */);
int octl_forest_add_pose_routed(octl_forest* f, int32_t* slot);
/* The same with the routed cloud of ANOTHER context on the same device: routing (its kernels, the
 * RCCL communicator and its stream) can then run in a second context - from a second host thread -
 * while this forest's context builds and fits the previous cloud.  route_ctx may route the next
 * cloud as soon as this call has returned (a buffer swap in the common case; when the cloud has to
 * be copied the call waits for the copy).                                                        */
int octl_forest_add_pose_routed_from(octl_forest* f, octl_ctx* route_ctx, int32_t* slot);
/* global indices of the routed cloud of the last octl_route_points call (n_recv) i64        */
int octl_route_get_gidx(octl_ctx* ctx, int64_t cap, int64_t* gidx, int64_t* n);
/* Number of times the library has made the host wait for the device (stream / event synchronisations)
 * since it was loaded: bench.py reports the round trips per step from it.                        */
int octl_debug_host_syncs(uint64_t* count);
/* Kernel launches + asynchronous fills enqueued by the library so far (process-wide), the companion of
 * octl_debug_host_syncs: bench.py reports launches per step.                                          */
int octl_debug_launches(uint64_t* count);
/* Speculative launches of the bucket build's last kernel (k_bucket_finish enqueued before the host has seen the
 * build's totals, its tables sized from the context's previous build) that did the work / that the host had to
 * repeat the ordinary way (process-wide).  No reference counterpart: the reference has no device queue.        */
int octl_debug_spec_finish(uint64_t* held, uint64_t* missed);

/* Test hook: the communicator-independent half of octl_route_points for ANY number of ranks -
 * destination of every point (host cloud in), per-destination counts [n_ranks], and the packed
 * send buffers (points and global indices stably partitioned by destination).                */
int octl_debug_route_partition(octl_ctx* ctx, const double* xyz, int64_t n, int64_t index_base,
                               double L, int32_t n_ranks, int64_t* counts, double* xyz_out,
                               int64_t* gidx_out);
/* sum-all-reduce of small int64 vectors (counters) over the communicator                   */
int octl_comm_allreduce_i64(octl_ctx* ctx, int64_t* inout_host, int32_t n);

/* ---- small device utilities used by bench.py (inputs resident in HBM) ------------------ */
int octl_dev_alloc(octl_ctx* ctx, int64_t bytes, void** dptr);
int octl_dev_free(octl_ctx* ctx, void* dptr);
int octl_dev_upload(octl_ctx* ctx, void* dptr, const void* src, int64_t bytes);
int octl_dev_download(octl_ctx* ctx, void* dst, const void* dptr, int64_t bytes);
/* ---- asynchronous host feed ---------------------------------------------------------------------
 * Replaces the synchronous host-to-device copies of CudaRansac.evaluate (ransac/cuda_ransac.py:57-67) and of
 * Grid.insert_points' storage step for a loop over scans: the upload of scan i+1 runs on a copy stream of
 * its own while the context's compute stream builds and fits scan i.
 *   octl_host_alloc / octl_host_free   page-locked host memory (a copy out of it is a DMA the host does not
 *                                      wait for; out of pageable memory the call blocks while HIP stages it)
 *   octl_dev_upload_async              enqueue the copy host -> dptr behind the compute work enqueued so far;
 *                                      returns at once.  `src` must stay unchanged until octl_ctx_sync_uploads
 *                                      (or octl_dev_free / octl_host_free of either end) has returned.
 *   octl_ctx_sync_uploads              host waits for every upload enqueued so far
 * octl_forest_add_pose_device / octl_forest_add_pose_adopt order the compute stream behind the uploads enqueued
 * before them (on the device; the host does not wait).                                              */
int octl_host_alloc(octl_ctx* ctx, int64_t bytes, void** p);
int octl_host_free(octl_ctx* ctx, void* p);
int octl_dev_upload_async(octl_ctx* ctx, void* dptr, const void* src, int64_t bytes);
int octl_ctx_sync_uploads(octl_ctx* ctx);
/* measured device copy bandwidth (bytes/s) over `bytes`, for the roofline report           */
int octl_dev_copy_bandwidth(octl_ctx* ctx, int64_t bytes, int iters, double* bytes_per_s);

/* Test hook for the allocation-failure paths: the nth growth of a device buffer from now on (every device
 * allocation of the library is one) fails with OCTL_E_NOMEM exactly as a failed hipMalloc does; nth <= 0 disarms.
 * *seen (nullable) receives the number of growths since the hook was last armed, so a test can sweep nth over
 * everything an operation allocates.  Process-wide.                                                     */
int octl_debug_fail_alloc(int64_t nth, int64_t* seen);

/* What the communicator itself reports (ncclCommCount / ncclCommUserRank / ncclGetVersion; -1 where the collective
 * library does not export the query), and the device behind a context (PCI bus id as "dddd:bb:dd.f", the 16-byte
 * UUID of its properties, its compute units): a multi-GPU run prints these so that its line proves what ran where.
 * The reference has no counterpart (no distributed code).                                            */
int octl_comm_info(octl_ctx* ctx, int32_t* n_ranks, int32_t* user_rank, int32_t* version);
int octl_device_identity(octl_ctx* ctx, char pci_bus_id[32], uint8_t uuid[16], int32_t* cus);

/* Diagnostic switches of a context (tests and A/B runs compare code paths: "NO_BUCKET_BUILD", "BUCKET_POINTS",
 * "SYNC_GEOM", "NO_GEOM_HINT", "GEOM_MARGIN", "NO_EXACT_DIGITS", "NO_FAST_ORDER", "NO_BUCKET_HISTORY", "NO_CUBE_FAST",
 * "NO_CUBE_PREFIX", "CUBE_PREFIX_MIN", "NO_INCREMENTAL", "ROUTE_SELF_SENDRECV", "TRACE_BUILD", "SCAN",
 * "NO_FUSED_TABLES", "NO_SPIN_WAIT", "NO_SPEC_FINISH", "RANSAC_WAVES"; an "OCTL_" prefix is accepted; the table in
 * INTEGRATION.md says what each one does).  octl_ctx_create reads OCTL_<NAME> from the environment ONCE, as a
 * number: OCTL_X=0 is OFF (rounds 1-4 tested only for the variable's presence), a value that is not a number is 1,
 * and setting the environment after the context exists has no effect - this call changes a switch of a live
 * context.  0 = the shipped behaviour.  The reference has no counterpart.                                     */
int octl_debug_set_option(octl_ctx* ctx, const char* name, int64_t value);

/* The key geometry of a single-pass bucket build as a pure HOST function (no context, no GPU: the same code the
 * device evaluates, csrc/bucket_build.hip: geom_from_box): for a cloud whose true voxel box is tb = {min x, y, z,
 * max x, y, z}, `want` buckets wanted (a power of two), n_alive points, `target` points per bucket and `margin`
 * voxels of slack, it returns the padded box of the linear keys, the bucket width in keys, the bucket count - and
 * *mismatches = the number of keys of that box for which the kernels' division-free bucket index
 * (trunc(fma(key, 1/width, 0.5/width))) differs from key / width (must be 0).  *valid = 0 with the library's reason
 * code when the box is not a single-pass case.  Replaces nothing in the reference, which rebuckets every call from
 * scratch (grid/grid.py:72-90); tests/test_cpu_abi.py drives it.                                              */
int octl_debug_key_geometry(const int32_t tb[6], uint64_t want, int64_t n_alive, uint32_t target, int32_t margin,
                            int32_t bb_out[6], uint32_t* width, uint32_t* n_buckets, int32_t* valid,
                            int64_t* mismatches);

/* ---- test hooks for the device-wide primitives (host in / host out) ----------------------- */
int octl_debug_exclusive_scan(octl_ctx* ctx, const uint32_t* in, int64_t n, uint32_t* out,
                              uint32_t* total);
int octl_debug_radix_sort(octl_ctx* ctx, uint64_t* keys, uint32_t* vals, int64_t n,
                          int key_bits);
/* The plane fit's arithmetic shortcuts (csrc/ransac.hip: div3_by_norm, div_by_small_int,
 * sqrt_rn_guarded) on caller data: q3[i,:] = num3[i,:] / den[i] (den > 0, |num| <= 2^60 den),
 * ck[i] = c[i] / kdiv (1 <= kdiv <= 16), sq[i] = sqrt(c[i]).  All must equal the IEEE f64 results
 * the reference computes (util.py:42-44,76,80-82).                                           */
int octl_debug_plane_arith(octl_ctx* ctx, const double* num3, const double* den, const double* c,
                           int32_t kdiv, int64_t n, double* q3, double* ck, double* sq);
/* The same three operations without their per-lane range guards, as the plane fit runs them on a block whose
 * coordinates are all +0.0 or in [2^-30, 2^31) (the range certificate of csrc/ransac.hip: coord_in_fast_range).  The
 * caller keeps the operands in the certified ranges: den in [2^-200, 2^138], num3 zero or in [2^-552, den],
 * c = +0.0 or |c| in [2^-82, 2^35) for c / kdiv, c in [2^-400, 2^276] for sqrt(c).                  */
int octl_debug_plane_arith_certified(octl_ctx* ctx, const double* num3, const double* den, const double* c,
                                     int32_t kdiv, int64_t n, double* q3, double* ck, double* sq);

#ifdef __cplusplus
}
#endif
#endif /* OCTREELIB_HIP_H */
