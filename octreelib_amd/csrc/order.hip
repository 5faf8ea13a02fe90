// Device-side computation of the order in which the reference concatenates the leaves of a
// batch of poses before the RANSAC kernel (Grid.map_leaf_points_cuda_ransac, grid/grid.py:
// 173-191): for pose in batch; for top-level voxel in lexicographic order (grid.py:79-81,108);
// for leaf in the octree's cached-leaf list, non-empty only (octree.py:256-263).
//
// The cached-leaf list (octree_base.py:152-158) is history dependent: a node that is split is
// removed and its 8 children are appended (octree.py:183-191), and splits happen in DFS
// preorder within one subdivide / subdivide_as call (octree.py:20-53).  Hence a leaf sorts by
//     ( effective epoch of its parent, DFS-preorder rank of its parent, child index )
// where the effective epoch of a parent for a pose tree created at epoch e0 is
// max(epoch(parent), e0).  Everything is computed from the scheme node table:
//   1. bottom-up: internal nodes per subtree; 2. top-down: global preorder rank of every
//   internal node (voxel-major); 3. one key per (leaf, pose) block; 4. two stable radix sorts.
#include <algorithm>

#include "forest.h"

namespace {

__global__ __launch_bounds__(256) void k_sub_up(const int32_t* __restrict__ first_child,
                                                int64_t a, int64_t b,
                                                uint32_t* __restrict__ nint) {
  const int64_t x = a + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= b) return;
  const int32_t fc = first_child[x];
  uint32_t s = 0;
  if (fc >= 0) {
    s = 1;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += nint[fc + j];
  }
  nint[x] = s;
}

__global__ __launch_bounds__(256) void k_rank_down(const int32_t* __restrict__ first_child,
                                                   int64_t a, int64_t b,
                                                   const uint32_t* __restrict__ nint,
                                                   uint32_t* __restrict__ rank) {
  const int64_t x = a + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= b) return;
  const int32_t fc = first_child[x];
  if (fc < 0) return;
  uint32_t r = rank[x] + 1;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    rank[fc + j] = r;  // meaningful only for internal children
    r += nint[fc + j];
  }
}

// key1 = rank(parent) * 8 + child index (0 for a root leaf); key2 = (slot * V + voxel) << ebits
// | effective epoch.  val = block id.
__global__ __launch_bounds__(256) void k_block_keys(
    const int32_t* __restrict__ blk_node, const int32_t* __restrict__ blk_slot, int64_t nb,
    const int32_t* __restrict__ parent, const int32_t* __restrict__ first_child,
    const int32_t* __restrict__ voxel, const int32_t* __restrict__ epoch,
    const uint32_t* __restrict__ rank, const int32_t* __restrict__ e0, uint64_t V, int ebits,
    uint64_t* __restrict__ key1, uint64_t* __restrict__ key2, uint32_t* __restrict__ val) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  const int32_t l = blk_node[b], p = parent[l], s = blk_slot[b];
  uint64_t k1 = 0, ee = 0;
  if (p >= 0) {
    k1 = (uint64_t)rank[p] * 8u + (uint64_t)(l - first_child[p]);
    const int32_t ep = epoch[p], es = e0 ? e0[s] : 0;
    ee = (uint64_t)(ep > es ? ep : es);
  }
  key1[b] = k1;
  key2[b] = (((uint64_t)s * V + (uint64_t)voxel[l]) << ebits) | ee;
  val[b] = (uint32_t)b;
}

// ---- direct path (all internal nodes share one epoch: no history effects) ------------------------
// slot of an internal node x in the "parents" sequence: per voxel one slot for "the root is a
// leaf" followed by the voxel's internal nodes in preorder: idx(x) = rank[x] + voxel[x] + 1
__global__ __launch_bounds__(256) void k_leaf_children(const int32_t* __restrict__ first_child,
                                                       const int32_t* __restrict__ parent,
                                                       const int32_t* __restrict__ voxel,
                                                       const uint32_t* __restrict__ rank, int64_t n,
                                                       uint32_t* __restrict__ seq) {
  const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= n) return;
  const int32_t fc = first_child[x];
  if (fc >= 0) {
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) c += first_child[fc + j] < 0 ? 1u : 0u;
    seq[(size_t)rank[x] + (uint32_t)voxel[x] + 1u] = c;
  } else if (parent[x] < 0) {
    // a root that is a leaf: rank[x] holds the voxel's rank base (exclusive scan over the roots)
    seq[(size_t)rank[x] + (uint32_t)voxel[x]] = 1u;
  }
}

// position of every block in table[slot][cached leaf position]
__global__ __launch_bounds__(256) void k_block_positions(
    const int32_t* __restrict__ blk_node, const int32_t* __restrict__ blk_slot, int64_t nb,
    const int32_t* __restrict__ parent, const int32_t* __restrict__ first_child,
    const int32_t* __restrict__ voxel, const uint32_t* __restrict__ rank,
    const uint32_t* __restrict__ seq_scanned, uint64_t n_leaves_total,
    uint32_t* __restrict__ table) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  const int32_t l = blk_node[b], p = parent[l];
  uint32_t pos;
  if (p < 0) {
    pos = seq_scanned[(size_t)rank[l] + (uint32_t)voxel[l]];
  } else {
    const int32_t fc = first_child[p];
    uint32_t before = 0;
    for (int32_t c = fc; c < l; ++c) before += first_child[c] < 0 ? 1u : 0u;
    pos = seq_scanned[(size_t)rank[p] + (uint32_t)voxel[p] + 1u] + before;
  }
  table[(uint64_t)blk_slot[b] * n_leaves_total + pos] = (uint32_t)b + 1u;
}

__global__ __launch_bounds__(256) void k_table_flags(const uint32_t* __restrict__ table, int64_t n,
                                                     uint32_t* __restrict__ flags) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flags[i] = table[i] ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_table_compact(const uint32_t* __restrict__ table,
                                                       const uint32_t* __restrict__ scanned,
                                                       int64_t n, int32_t* __restrict__ order) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && table[i]) order[scanned[i]] = (int32_t)(table[i] - 1u);
}

__global__ __launch_bounds__(256) void k_gather_key2(const uint64_t* __restrict__ key2,
                                                     const uint32_t* __restrict__ val, int64_t nb,
                                                     uint64_t* __restrict__ out) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nb) out[b] = key2[val[b]];
}

// blocks per pose slot (<= 256 slots): LDS histogram per workgroup, then one global atomic per
// non-empty bin per workgroup (same-address global atomics serialise)
__global__ __launch_bounds__(256) void k_slot_hist(const int32_t* __restrict__ blk_slot, int64_t nb,
                                                   uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * 2048;
  for (int r = 0; r < 8; ++r) {
    const int64_t b = base + r * 256 + threadIdx.x;
    if (b < nb) atomicAdd(&h[blk_slot[b]], 1u);
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

inline unsigned grid_for(int64_t n) { return (unsigned)ceil_div(n, 256); }
int bits_for(uint64_t max_value) {
  int b = 0;
  while (b < 64 && (max_value >> b) != 0) ++b;
  return b;
}

}  // namespace

// Fills f->rs_order (device, int32 per block) with all blocks in the reference's order
// (slot-major).  e0_host: creation epoch of every pose tree (nullable = 0).  slot_counts_host
// receives the number of blocks of every slot.
int forest_reference_order(octl_forest* f, const int32_t* e0_host, std::vector<uint32_t>& slot_counts,
                           bool need_slot_counts) {
  octl_ctx* ctx = f->ctx;
  hipStream_t st = ctx->stream;
  const int64_t nb = f->n_blocks;
  const int n_poses = (int)f->pose_off.size() - 1;
  slot_counts.assign((size_t)std::max(n_poses, 1), 0);
  if (nb <= 0) return OCTL_OK;
  if (f->fast_order_valid && n_poses == 1) {
    // the bucket build has left the order behind (k_bucket_finish): one pose, one epoch - e0 cannot matter
    std::swap(f->rs_order, f->fast_order);
    f->fast_order_valid = false;
    slot_counts[0] = (uint32_t)nb;
    return OCTL_OK;
  }
  NodeTable& t = f->nodes[f->cur];
  const int64_t V = f->n_voxels;
  KTimer timer(ctx, "ransac_order");
  // scratch layout inside f->entries: [nint u32 n | rank u32 n]
  const size_t nn = (size_t)t.n;
  const size_t off_rank = ((nn + 8) * 4 + 15) & ~(size_t)15;
  OCTL_TRY(devbuf_reserve(ctx, f->entries, off_rank + (nn + 8) * 4));
  uint32_t* nint = f->entries.as<uint32_t>();
  uint32_t* rank = reinterpret_cast<uint32_t*>(static_cast<char*>(f->entries.p) + off_rank);
  const int32_t* fc = t.first_child.as<int32_t>();
  // bottom-up over the level ranges, deepest first; then top-down (ranges of equal depth are independent)
  std::vector<octl_forest::LevelSeg> segs = f->level_segs;
  std::stable_sort(segs.begin(), segs.end(),
                   [](const octl_forest::LevelSeg& x, const octl_forest::LevelSeg& y) { return x.depth > y.depth; });
  for (const auto& sg : segs) {
    if (sg.b > sg.a) {
      OCTL_LAUNCH(k_sub_up, dim3(grid_for(sg.b - sg.a)), dim3(256), 0, st, fc, sg.a, sg.b, nint);
      HIP_TRY(ctx, hipGetLastError());
    }
  }
  // roots: global rank base = exclusive scan of the per-voxel internal-node counts
  OCTL_TRY(octl_exclusive_scan_u32(ctx, nint, rank, V, nullptr));
  for (auto it = segs.rbegin(); it != segs.rend(); ++it) {
    if (it->b > it->a) {
      OCTL_LAUNCH(k_rank_down, dim3(grid_for(it->b - it->a)), dim3(256), 0, st, fc, it->a, it->b,
                         (const uint32_t*)nint, rank);
      HIP_TRY(ctx, hipGetLastError());
    }
  }
  // ---- direct path: no history effects and a table of P x #leaves that is small enough ----------
  const int64_t n_leaves_total = (int64_t)t.n - f->n_internal;
  const int64_t table_n = (int64_t)n_poses * n_leaves_total;
  bool e0_ok = true;  // a pose created before the scheme's (single) epoch sees the same order
  if (f->uniform_epoch && table_n > 0 && table_n <= ((int64_t)1 << 26)) {
    const size_t seq_n = (size_t)f->n_internal + (size_t)V + 8;
    const size_t off_tab = ((seq_n * 4) + 15) & ~(size_t)15;
    const size_t off_flg = off_tab + (((size_t)table_n + 8) * 4 + 15) / 16 * 16;
    OCTL_TRY(devbuf_reserve(ctx, f->hist, off_flg + ((size_t)table_n + 8) * 4));
    char* base = static_cast<char*>(f->hist.p);
    uint32_t* seq = reinterpret_cast<uint32_t*>(base);
    uint32_t* table = reinterpret_cast<uint32_t*>(base + off_tab);
    uint32_t* tflags = reinterpret_cast<uint32_t*>(base + off_flg);
    HIP_TRY(ctx, hipMemsetAsync(seq, 0, off_flg, st));  // seq and table
    OCTL_LAUNCH(k_leaf_children, dim3(grid_for(t.n)), dim3(256), 0, st, fc,
                       (const int32_t*)t.parent.as<int32_t>(), (const int32_t*)t.voxel.as<int32_t>(),
                       (const uint32_t*)rank, t.n, seq);
    HIP_TRY(ctx, hipGetLastError());
    OCTL_TRY(octl_exclusive_scan_u32(ctx, seq, seq, (int64_t)f->n_internal + V, nullptr));
    OCTL_LAUNCH(k_block_positions, dim3(grid_for(nb)), dim3(256), 0, st,
                       (const int32_t*)f->blk_node.as<int32_t>(),
                       (const int32_t*)f->blk_slot.as<int32_t>(), nb,
                       (const int32_t*)t.parent.as<int32_t>(), fc,
                       (const int32_t*)t.voxel.as<int32_t>(), (const uint32_t*)rank,
                       (const uint32_t*)seq, (uint64_t)n_leaves_total, table);
    HIP_TRY(ctx, hipGetLastError());
    OCTL_LAUNCH(k_table_flags, dim3(grid_for(table_n)), dim3(256), 0, st,
                       (const uint32_t*)table, table_n, tflags);
    HIP_TRY(ctx, hipGetLastError());
    OCTL_TRY(octl_exclusive_scan_u32(ctx, tflags, tflags, table_n, nullptr));
    OCTL_TRY(devbuf_reserve(ctx, f->rs_order, (size_t)nb * 4));
    OCTL_LAUNCH(k_table_compact, dim3(grid_for(table_n)), dim3(256), 0, st,
                       (const uint32_t*)table, (const uint32_t*)tflags, table_n,
                       f->rs_order.as<int32_t>());
    HIP_TRY(ctx, hipGetLastError());
  } else {
    e0_ok = false;
  }
  if (e0_ok) {
    // blocks per slot (only needed to cut the order into batches)
    if (need_slot_counts) {
      std::vector<int32_t> slots((size_t)nb);
      HIP_TRY(ctx, hipMemcpyAsync(slots.data(), f->blk_slot.p, (size_t)nb * 4, hipMemcpyDeviceToHost, st));
      HIP_TRY(ctx, hipStreamSynchronize(st));
      for (int32_t sl : slots) slot_counts[(size_t)sl] += 1;
    }
    return OCTL_OK;
  }
  // ---- general path: two stable radix sorts of per-block keys ------------------------------------------
  // keys
  int max_epoch = f->epoch;
  const int32_t* e0_dev = nullptr;
  if (e0_host) {
    OCTL_TRY(devbuf_reserve(ctx, f->scheme_dev, (size_t)n_poses * 4));
    if ((size_t)n_poses * 4 <= 64 * 1024) {  // pinned staging: [192 KiB, 256 KiB) of ctx->pinned
      char* pin = static_cast<char*>(ctx->pinned) + 192 * 1024;
      OCTL_TRY(pin_region_wait(ctx, 2));
      std::memcpy(pin, e0_host, (size_t)n_poses * 4);
      HIP_TRY(ctx, hipMemcpyAsync(f->scheme_dev.p, pin, (size_t)n_poses * 4, hipMemcpyHostToDevice, st));
      OCTL_TRY(pin_region_mark(ctx, 2));
    } else {
      HIP_TRY(ctx, hipMemcpyAsync(f->scheme_dev.p, e0_host, (size_t)n_poses * 4,
                                  hipMemcpyHostToDevice, st));
      HIP_TRY(ctx, hipStreamSynchronize(st));
    }
    e0_dev = f->scheme_dev.as<int32_t>();
    for (int p = 0; p < n_poses; ++p) max_epoch = std::max(max_epoch, (int)e0_host[p]);
  }
  const int ebits = std::max(1, bits_for((uint64_t)max_epoch));
  const int bits2 = bits_for((uint64_t)n_poses * (uint64_t)V) + ebits;
  if (bits2 > 63) return octl_set_error(ctx, OCTL_E_INVALID, "order key too wide");
  const int bits1 = bits_for((uint64_t)f->n_internal * 8u + 7u);
  for (int b = 0; b < 2; ++b) {
    OCTL_TRY(devbuf_reserve(ctx, f->lin[b], (size_t)nb * 8));
    OCTL_TRY(devbuf_reserve(ctx, f->val[b], (size_t)nb * 4));
  }
  OCTL_TRY(devbuf_reserve(ctx, f->vkey, (size_t)nb * 8));  // key2 in block order
  uint64_t* keys[2] = {f->lin[0].as<uint64_t>(), f->lin[1].as<uint64_t>()};
  uint32_t* vals[2] = {f->val[0].as<uint32_t>(), f->val[1].as<uint32_t>()};
  uint64_t* key2 = f->vkey.as<uint64_t>();
  OCTL_LAUNCH(k_block_keys, dim3(grid_for(nb)), dim3(256), 0, st,
                     (const int32_t*)f->blk_node.as<int32_t>(),
                     (const int32_t*)f->blk_slot.as<int32_t>(), nb,
                     (const int32_t*)t.parent.as<int32_t>(), fc,
                     (const int32_t*)t.voxel.as<int32_t>(), (const int32_t*)t.epoch.as<int32_t>(),
                     (const uint32_t*)rank, e0_dev, (uint64_t)V, ebits, keys[0], key2, vals[0]);
  HIP_TRY(ctx, hipGetLastError());
  int res = 0;
  OCTL_TRY(octl_radix_sort_u64_u32(ctx, keys, vals, nb, bits1, f->hist, &res));
  // second (more significant) key, gathered in the order of the first sort
  uint64_t* keys_b[2] = {keys[res ^ 1], keys[res]};
  uint32_t* vals_b[2] = {vals[res], vals[res ^ 1]};
  OCTL_LAUNCH(k_gather_key2, dim3(grid_for(nb)), dim3(256), 0, st, (const uint64_t*)key2,
                     (const uint32_t*)vals[res], nb, keys_b[0]);
  HIP_TRY(ctx, hipGetLastError());
  int res2 = 0;
  OCTL_TRY(octl_radix_sort_u64_u32(ctx, keys_b, vals_b, nb, bits2, f->hist, &res2));
  OCTL_TRY(devbuf_reserve(ctx, f->rs_order, (size_t)nb * 4));
  HIP_TRY(ctx, hipMemcpyAsync(f->rs_order.p, vals_b[res2], (size_t)nb * 4,
                              hipMemcpyDeviceToDevice, st));
  // blocks per slot (only needed to cut the order into batches)
  uint32_t* hist = ctx->small.as<uint32_t>() + 64;
  if (!need_slot_counts) {
    // nothing to do
  } else if (n_poses <= 256) {
    static_assert(256 <= MIRROR_COPY_WORDS, "the slot counts land in the copy area of the mirror, below its flag words");
    HIP_TRY(ctx, hipMemsetAsync(hist, 0, (size_t)n_poses * 4, st));
    OCTL_LAUNCH(k_slot_hist, dim3((unsigned)ceil_div(nb, 2048)), dim3(256), 0, st,
                       (const int32_t*)f->blk_slot.as<int32_t>(), nb, hist);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(ctx->small_host, hist, (size_t)n_poses * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    std::memcpy(slot_counts.data(), ctx->small_host, (size_t)n_poses * 4);
  } else {
    std::vector<int32_t> slots((size_t)nb);
    HIP_TRY(ctx, hipMemcpyAsync(slots.data(), f->blk_slot.p, (size_t)nb * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    for (int32_t s : slots) slot_counts[(size_t)s] += 1;
  }
  return OCTL_OK;
}

extern "C" int octl_forest_reference_order(octl_forest* f, const int32_t* e0, int32_t n_e0,
                                           int64_t cap, int32_t* order, int64_t* n_blocks) {
  if (!f || !n_blocks) return OCTL_E_INVALID;
  OCTL_TRY(forest_settle(f));
  octl_ctx* ctx = f->ctx;
  if (!f->built) return octl_set_error(ctx, OCTL_E_STATE, "no scheme has been built");
  const int n_poses = (int)f->pose_off.size() - 1;
  if (e0 && n_e0 != n_poses) return octl_set_error(ctx, OCTL_E_INVALID, "e0 size mismatch");
  std::vector<uint32_t> slot_counts;
  OCTL_TRY(forest_reference_order(f, e0, slot_counts, false));
  *n_blocks = f->n_blocks;
  const int64_t n = std::min<int64_t>(cap, f->n_blocks);
  if (n > 0 && order) {
    HIP_TRY(ctx, hipMemcpyAsync(order, f->rs_order.p, (size_t)n * 4, hipMemcpyDeviceToHost,
                                ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  return OCTL_OK;
}

static int ransac_all_impl(octl_forest* f, int32_t poses_per_batch, const int32_t* e0, int32_t n_e0,
                           const double* hypotheses, int32_t H, int32_t k, double threshold,
                           bool* mask_fresh);

extern "C" int octl_forest_ransac_all(octl_forest* f, int32_t poses_per_batch, const int32_t* e0,
                                      int32_t n_e0, const double* hypotheses, int32_t H, int32_t k,
                                      double threshold) {
  if (!f || !hypotheses) return OCTL_E_INVALID;
  OCTL_TRY(forest_settle(f));
  // A mask buffer that was NOT valid before this call only becomes valid when every batch has been
  // enqueued: a failure half way (reference order, a scratch reservation, a later batch) must not leave
  // uninitialised or partly written bytes marked valid for the next apply_mask / filter.
  bool mask_fresh = false;
  const int rc = ransac_all_impl(f, poses_per_batch, e0, n_e0, hypotheses, H, k, threshold, &mask_fresh);
  if (mask_fresh) f->mask_valid = (rc == OCTL_OK);
  return rc;
}

static int ransac_all_impl(octl_forest* f, int32_t poses_per_batch, const int32_t* e0, int32_t n_e0,
                           const double* hypotheses, int32_t H, int32_t k, double threshold,
                           bool* mask_fresh) {
  octl_ctx* ctx = f->ctx;
  if (!f->built) return octl_set_error(ctx, OCTL_E_STATE, "ransac before build");
  if (poses_per_batch < 1) return octl_set_error(ctx, OCTL_E_INVALID, "poses_per_batch < 1");
  if (H < 1 || H > 1024 || k < 1) return octl_set_error(ctx, OCTL_E_INVALID, "bad H or k");
  OCTL_TRY(ransac_check_table(ctx, hypotheses, H, k));
  const int n_poses = (int)f->pose_off.size() - 1;
  if (e0 && n_e0 != n_poses) return octl_set_error(ctx, OCTL_E_INVALID, "e0 size mismatch");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  // (no "keep" fill here: the batches below cover every block of every pose, the blocks tile the
  //  leaf-ordered arrays, and every block's mask bytes are written - by the scoring kernels, or as zeros for
  //  the blocks with fewer than k points)
  if (!f->mask_valid) {
    OCTL_TRY(devbuf_reserve(ctx, f->mask, (size_t)std::max<int64_t>(f->n_ord, 1)));
    *mask_fresh = true;  // (the caller marks it valid once everything below has been enqueued)
  }
  if (f->n_blocks == 0) return OCTL_OK;
  const size_t hyp_cap_before = ctx->hyp_dev.cap;
  OCTL_TRY(devbuf_reserve(ctx, ctx->hyp_dev, (size_t)H * k * 8));
  // the table of the previous call on this context (CudaRansac draws it once per object, cuda_ransac.py:39-41; a
  // loop over scans hands the same one over for every scan): already on the device
  const bool same_table = hyp_cap_before == ctx->hyp_dev.cap && ctx->hyp_host.size() == (size_t)H * k &&
                          std::memcmp(ctx->hyp_host.data(), hypotheses, (size_t)H * k * 8) == 0;
  if (same_table) {
    // nothing to upload
  } else if ((size_t)H * k * 8 <= 64 * 1024) {  // pinned staging: [128 KiB, 192 KiB) of ctx->pinned
    // the launches below are not synchronised: a later call must not overwrite the staging area
    // while this copy is still pending
    char* pin = static_cast<char*>(ctx->pinned) + 128 * 1024;
    OCTL_TRY(pin_region_wait(ctx, 1));
    std::memcpy(pin, hypotheses, (size_t)H * k * 8);
    OCTL_TRY(octl_copy_from_pinned(ctx, ctx->hyp_dev.p, pin, (size_t)H * k * 8));
    OCTL_TRY(pin_region_mark(ctx, 1));
  } else {
    HIP_TRY(ctx, hipMemcpyAsync(ctx->hyp_dev.p, hypotheses, (size_t)H * k * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
  }
  if (!same_table) ctx->hyp_host.assign(hypotheses, hypotheses + (size_t)H * k);
  const bool one_batch = n_poses <= poses_per_batch;
  std::vector<uint32_t> slot_counts;
  OCTL_TRY(forest_reference_order(f, e0, slot_counts, !one_batch));
  if (one_batch) {
    OCTL_TRY(ransac_launch(ctx, f->xyz_ord.as<double>(), f->n_ord, f->blk_start.as<uint32_t>(),
                           f->blk_size.as<int32_t>(), f->rs_order.as<int32_t>(), f->n_blocks,
                           ctx->hyp_dev.as<double>(), H, k, threshold, f->mask.as<uint8_t>(), nullptr,
                           nullptr, nullptr, nullptr, f->rs_scratch, f->max_block_hint));
    return OCTL_OK;
  }
  // one evaluate() per batch of poses_per_batch consecutive poses (grid.py:149-157,194)
  int64_t off = 0;
  for (int p0 = 0; p0 < n_poses; p0 += poses_per_batch) {
    int64_t nbatch = 0;
    for (int p = p0; p < std::min(n_poses, p0 + poses_per_batch); ++p) nbatch += slot_counts[p];
    if (nbatch > 0)
      OCTL_TRY(ransac_launch(ctx, f->xyz_ord.as<double>(), f->n_ord, f->blk_start.as<uint32_t>(),
                             f->blk_size.as<int32_t>(), f->rs_order.as<int32_t>() + off, nbatch,
                             ctx->hyp_dev.as<double>(), H, k, threshold, f->mask.as<uint8_t>(),
                             nullptr, nullptr, nullptr, nullptr, f->rs_scratch, f->max_block_hint));
    off += nbatch;
  }
  return OCTL_OK;
}
