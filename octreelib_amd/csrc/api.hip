// extern "C" surface of the forest (Grid / OctreeManager / Octree state) and of the stand-alone
// RANSAC operator.  See include/octreelib_hip.h for the reference interfaces each entry point
// replaces.
#include <unordered_map>

#include "forest.h"
#include "lookback.h"
#include "ref_arith.h"

namespace {

// ---- ingest: Grid.insert_points' storage step -------------------------------------------------------
// One pass over the new points of a pose: copies them into the forest's store (device sources),
// marks them alive and folds their top-level voxel indices - floor((p - corner) / L), grid.py:72-76 -
// into the forest's voxel bounding box, so that the build can form compact linear voxel keys without
// a pass of its own.
// The cloud is read as a FLAT array of doubles, 16 bytes per lane and instruction (a lane reading its
// own 48-byte pair of points touches every cache line three times): the bounding box needs the minimum
// and maximum per AXIS, and the axis of flat element i is i mod 3, whichever point it belongs to.
constexpr int ING_UNITS = 12;  // 16-byte units per thread
template <bool COPY>
__global__ __launch_bounds__(256) void k_ingest(const double* __restrict__ src, double* __restrict__ dst,
                                                uint8_t* __restrict__ alive, int64_t n, int mode,
                                                double L, int32_t* __restrict__ bbox) {
  const int big = 1 << 30;
  int mn[3] = {big, big, big}, mx[3] = {-big, -big, -big};
  bool bad = false;
  const int64_t n_flat = 3 * n, n_units = n_flat / 2;
  const int64_t u0 = (int64_t)blockIdx.x * (256 * ING_UNITS) + threadIdx.x;
  const double lim = (double)OCTL_VOX_ABS_LIMIT;
  auto fold = [&](double v, int axis) {
    const double f = L == 1.0 ? floor(v) : floor_div_exact(v, L);  // (floor_div_exact(v, 1) == floor(v))
    if (fabs(f) < lim) {  // false for NaN / inf
      const int q = (int)f;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        if (a == axis) {
          mn[a] = min(mn[a], q);
          mx[a] = max(mx[a], q);
        }
      }
    } else {
      bad = true;
    }
  };
  // (a pose behind an odd number of stored points starts 8 bytes off: scalar accesses then)
  const bool al16 = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
  double2 v[ING_UNITS];
#pragma unroll
  for (int k = 0; k < ING_UNITS; ++k) {
    const int64_t u = u0 + k * 256;
    if (u < n_units) {
      if (al16) {
        v[k] = reinterpret_cast<const double2*>(src)[u];
      } else {
        v[k].x = src[2 * u];
        v[k].y = src[2 * u + 1];
      }
    }
  }
#pragma unroll
  for (int k = 0; k < ING_UNITS; ++k) {
    const int64_t u = u0 + k * 256;
    if (u < n_units) {
      if (COPY) {
        if (al16) {
          reinterpret_cast<double2*>(dst)[u] = v[k];
        } else {
          dst[2 * u] = v[k].x;
          dst[2 * u + 1] = v[k].y;
        }
      }
      if (mode == 0) {
        const int ax = (int)((2 * u) % 3);
        fold(v[k].x, ax);
        fold(v[k].y, ax == 2 ? 0 : ax + 1);
      }
    }
  }
  if ((n_flat & 1) && u0 == 0) {  // the last double of an odd number of points
    const double t = src[n_flat - 1];
    if (COPY) dst[n_flat - 1] = t;
    if (mode == 0) fold(t, 2);
  }
  if (mode != 0 && u0 < n_units) mn[0] = mn[1] = mn[2] = mx[0] = mx[1] = mx[2] = 0;
  // alive flags: 16 per thread (nullptr: the caller sets them otherwise)
  if (alive) {
    const int64_t a0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (a0 + 16 <= n && (reinterpret_cast<uintptr_t>(alive) & 15) == 0) {
      *reinterpret_cast<uint4*>(alive + a0) = make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u);
    } else {
      for (int64_t i = a0; i < n && i < a0 + 16; ++i) alive[i] = 1;
    }
  }
  // wave + block reduction, then at most six atomics per block and only when the block widens the
  // box (same-address atomics serialise)
  __shared__ int s_bb[4][6];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      mn[a] = min(mn[a], __shfl_xor(mn[a], off));
      mx[a] = max(mx[a], __shfl_xor(mx[a], off));
    }
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicExch(reinterpret_cast<uint32_t*>(bbox + 6), 1u);
  if ((threadIdx.x & 63) == 0) {
    int* w = s_bb[threadIdx.x >> 6];
    w[0] = mn[0]; w[1] = mn[1]; w[2] = mn[2]; w[3] = mx[0]; w[4] = mx[1]; w[5] = mx[2];
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int a = threadIdx.x;
    int v = s_bb[0][a];
    for (int w = 1; w < 4; ++w) v = (a < 3) ? min(v, s_bb[w][a]) : max(v, s_bb[w][a]);
    if (a < 3) {
      if (v != big && v < __hip_atomic_load(&bbox[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMin(&bbox[a], v);
    } else {
      if (v != -big && v > __hip_atomic_load(&bbox[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(&bbox[a], v);
    }
  }
}

__global__ void k_bbox_reset(int32_t* __restrict__ bbox) {
  const int a = threadIdx.x;
  if (a < 8) bbox[a] = a < 3 ? (1 << 30) : (a < 6 ? -(1 << 30) : 0);
}

__global__ __launch_bounds__(256) void k_fill_u8(uint8_t* p, int64_t n, uint8_t v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ---- apply_mask: stream compaction of the leaf-ordered arrays and of the block table ----------------------
// kept points per 2048-point tile (the compaction's tile offsets) ...
__device__ __forceinline__ void mask_tile_count(const uint8_t* __restrict__ mask, int64_t n,
                                                uint32_t* __restrict__ tilecnt, uint32_t tile,
                                                uint8_t* __restrict__ alive_fill) {
  __shared__ uint32_t s_w[4];
  const int64_t i0 = (int64_t)tile * 2048 + (int64_t)threadIdx.x * 8;
  uint32_t c = 0;
  if (i0 + 8 <= n) {
    const uint64_t w = *reinterpret_cast<const uint64_t*>(mask + i0);  // (the mask buffer is 16-byte aligned)
    // bytes that are not zero
    const uint64_t nz = ((w & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | w;
    c = (uint32_t)__popcll(nz & 0x8080808080808080ull);
    // (alive flags that were never written - a cloud taken in place: position range = store range)
    if (alive_fill) *reinterpret_cast<uint64_t*>(alive_fill + i0) = 0x0101010101010101ull;
  } else {
    for (int64_t i = i0; i < n; ++i) {
      c += mask[i] ? 1u : 0u;
      if (alive_fill) alive_fill[i] = 1;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) tilecnt[tile] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// ... and per (leaf, pose) block: the block table is compacted block-wise, not re-derived from the points.
// ONE launch for both counts: workgroups [0, nt) take a tile of 2048 points each, the rest 256 blocks each.
__global__ __launch_bounds__(256) void k_blk_kept(const uint8_t* __restrict__ mask, int64_t n, uint32_t nt,
                                                  uint32_t* __restrict__ tilecnt,
                                                  const uint32_t* __restrict__ blk_start,
                                                  const int32_t* __restrict__ blk_size, int64_t nb,
                                                  uint32_t* __restrict__ kept, uint32_t* __restrict__ nonempty,
                                                  uint8_t* __restrict__ alive_fill) {
  if (blockIdx.x < nt) {
    mask_tile_count(mask, n, tilecnt, blockIdx.x, alive_fill);
    return;
  }
  const int64_t b = (int64_t)(blockIdx.x - nt) * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const uint32_t st = b < nb ? blk_start[b] : 0u;
  const int sz = b < nb ? blk_size[b] : 0;
  uint32_t c = 0;
  if (sz <= 256)
    for (int i = 0; i < sz; ++i) c += mask[(size_t)st + i] ? 1u : 0u;
  // large blocks (unsplit voxels, big leaves of a bare octree): the whole wave, one block at a time
  unsigned long long big = __ballot(sz > 256);
  while (big) {
    const int src = __ffsll((long long)big) - 1;
    big &= big - 1;
    const uint32_t s0 = (uint32_t)__shfl((int)st, src);
    const int z = __shfl(sz, src);
    uint32_t cc = 0;
    for (int i = lane; i < z; i += 64) cc += mask[(size_t)s0 + i] ? 1u : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cc += __shfl_xor(cc, off);
    if (lane == src) c = cc;
  }
  if (b < nb) {
    kept[b] = c;
    nonempty[b] = c ? 1u : 0u;
  }
}


// tile-wise stable compaction (8 rows of 256 points per workgroup, ballot ranks); dropped points die in the store.
// offset_of(kept points of the tile) -> kept points in front of the tile: read from the scanned table
// (k_compact_tiles) or found by look-back (the tile workgroups of k_mask_scan, small clouds: no launch of its own).
template <typename OffsetOf>
__device__ __forceinline__ void compact_tile(
    uint32_t tile, const uint8_t* __restrict__ mask, int64_t n,
    const uint32_t* __restrict__ ord_idx, const double* __restrict__ xyz_ord,
    uint32_t* __restrict__ ord_idx2, double* __restrict__ xyz_ord2, uint8_t* __restrict__ alive,
    uint8_t* __restrict__ alive_fill, uint32_t* s_cnt /* [33] */, OffsetOf offset_of) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t base = (int64_t)tile * 2048;
  const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
  uint32_t rk[8];
  uint32_t keepbits = 0;
  // (every load of the tile is issued before the first dependent instruction: 8 rounds x (index + 3 coordinates)
  //  in flight per lane; loading them behind `if (kept)` round by round left the kernel at 4.9 TB/s)
  uint32_t iv[8];
  double px[8], py[8], pz[8];
  uint8_t mk[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    mk[r] = 0;
    iv[r] = 0;
    px[r] = py[r] = pz[r] = 0.0;
    if (i < n) {
      mk[r] = mask[i];
      iv[r] = ord_idx[i];
      px[r] = xyz_ord[3 * i];
      py[r] = xyz_ord[3 * i + 1];
      pz[r] = xyz_ord[3 * i + 2];
    }
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    const bool k = i < n && mk[r] != 0;
    const unsigned long long bal = __ballot(k);
    rk[r] = (uint32_t)__popcll(bal & lt);
    keepbits |= (k ? 1u : 0u) << r;
    if (lane == 0) s_cnt[r * 4 + wave] = (uint32_t)__popcll(bal);
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    uint32_t v = lane < 32 ? s_cnt[lane] : 0u, inc = v;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
      const uint32_t t = __shfl_up(inc, off);
      if (lane >= off) inc += t;
    }
    if (lane < 32) s_cnt[lane] = inc - v;
    if (lane == 31) s_cnt[32] = inc;  // kept points of the tile
  }
  __syncthreads();
  const uint32_t toff = offset_of(s_cnt[32]);
  // (alive flags that were never written: position range = store range, every flag of the tile is written here)
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    if (i >= n) continue;
    if ((keepbits >> r) & 1u) {
      const int64_t d = (int64_t)toff + s_cnt[r * 4 + wave] + rk[r];
      ord_idx2[d] = iv[r];
      xyz_ord2[3 * d] = px[r];
      xyz_ord2[3 * d + 1] = py[r];
      xyz_ord2[3 * d + 2] = pz[r];
      if (alive_fill) alive_fill[iv[r]] = 1;
    } else {
      alive[iv[r]] = 0;
    }
  }
}

__global__ __launch_bounds__(256) void k_compact_tiles(
    const uint8_t* __restrict__ mask, const uint32_t* __restrict__ tile_off, int64_t n,
    const uint32_t* __restrict__ ord_idx, const double* __restrict__ xyz_ord,
    uint32_t* __restrict__ ord_idx2, double* __restrict__ xyz_ord2, uint8_t* __restrict__ alive) {
  __shared__ uint32_t s_cnt[33];  // [row][wave] -> exclusive offsets | total
  compact_tile(blockIdx.x, mask, n, ord_idx, xyz_ord, ord_idx2, xyz_ord2, alive, nullptr, s_cnt,
               [&](uint32_t) { return tile_off[blockIdx.x]; });
}


// apply_mask's counts, prefix sums and block-table compaction in ONE launch (round 5; before: k_blk_kept, a scan over
// [tile counts | kept per block | block non-empty], k_blk_compact).  Workgroups [0, nt) take a tile of 2048 positions:
// kept points of the tile, chained by decoupled look-back into the tile's offset, and the tile's compaction.
// Workgroups [nt, nt + nbw) take 256 blocks each: kept points and "non-empty" per block, two look-back chains over
// the block workgroups (kept points in front = the block's new start, non-empty blocks in front = its new id), and
// the surviving blocks are written straight into the compacted table.  Chains never cross: each has its own status
// words, and tiles are taken in blockIdx order inside every chain.  totals[0] / totals[1] (pinned host memory):
// kept points, surviving blocks.  fill_alive: the store's alive flags have never been written (a cloud taken in
// place): the tile workgroups write 1s over their range - position range = store range while every point is alive.
__global__ __launch_bounds__(256) void k_mask_scan(
    const uint8_t* __restrict__ mask, int64_t n, uint32_t nt,
    const uint32_t* __restrict__ ord_idx, const double* __restrict__ xyz_ord,
    uint32_t* __restrict__ ord_idx2, double* __restrict__ xyz_ord2, uint8_t* __restrict__ alive,
    const uint32_t* __restrict__ blk_start, const int32_t* __restrict__ blk_size, int64_t nb,
    const int32_t* __restrict__ blk_node, const int32_t* __restrict__ blk_slot, int32_t* __restrict__ blk_node2,
    int32_t* __restrict__ blk_slot2, uint32_t* __restrict__ blk_start2, int32_t* __restrict__ blk_size2,
    uint64_t* __restrict__ st_tiles, uint64_t* __restrict__ st_kept, uint64_t* __restrict__ st_ids, uint32_t epoch,
    uint32_t* __restrict__ mirror, uint32_t seq, uint8_t* __restrict__ alive_fill) {
  __shared__ uint32_t s_w[2][4];
  __shared__ uint32_t s_excl;
  __shared__ uint32_t s_cnt[33];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (blockIdx.x < nt) {
    // a tile of 2048 positions: its loads are in flight while the look-back finds the kept points in front of it,
    // then it compacts itself (round 6: was a count here and k_compact_tiles behind - one launch more)
    const uint32_t tile = blockIdx.x;
    compact_tile(tile, mask, n, ord_idx, xyz_ord, ord_idx2, xyz_ord2, alive, alive_fill, s_cnt,
                 [&](uint32_t total) {
                   const uint32_t excl = lookback_exclusive(st_tiles, epoch, tile, total, &s_excl);
                   if (threadIdx.x == 0 && tile == nt - 1) {
                     mirror[MIRROR_MASK_TOTALS] = excl + total;
                     mirror_publish(mirror, MIRROR_FLAG_MASK0, seq);
                   }
                   return excl;
                 });
    return;
  }
  const uint32_t bw = blockIdx.x - nt;
  const int64_t b = (int64_t)bw * 256 + threadIdx.x;
  const uint32_t st = b < nb ? blk_start[b] : 0u;
  const int sz = b < nb ? blk_size[b] : 0;
  uint32_t c = 0;
  if (sz <= 256)
    for (int i = 0; i < sz; ++i) c += mask[(size_t)st + i] ? 1u : 0u;
  // large blocks (unsplit voxels, big leaves of a bare octree): the whole wave, one block at a time
  unsigned long long big = __ballot(sz > 256);
  while (big) {
    const int src = __ffsll((long long)big) - 1;
    big &= big - 1;
    const uint32_t s0 = (uint32_t)__shfl((int)st, src);
    const int z = __shfl(sz, src);
    uint32_t cc = 0;
    for (int i = lane; i < z; i += 64) cc += mask[(size_t)s0 + i] ? 1u : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cc += __shfl_xor(cc, off);
    if (lane == src) c = cc;
  }
  // exclusive prefixes inside the workgroup: kept points, non-empty blocks
  const uint32_t ne = c ? 1u : 0u;
  uint32_t ic = c, ie = ne;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t tc = __shfl_up(ic, off), te = __shfl_up(ie, off);
    if (lane >= off) {
      ic += tc;
      ie += te;
    }
  }
  if (lane == 63) {
    s_w[0][wave] = ic;
    s_w[1][wave] = ie;
  }
  __syncthreads();
  uint32_t pc = ic - c, pe = ie - ne, tot_c = 0, tot_e = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (w < wave) {
      pc += s_w[0][w];
      pe += s_w[1][w];
    }
    tot_c += s_w[0][w];
    tot_e += s_w[1][w];
  }
  __syncthreads();   // (s_excl is used by both chains)
  const uint32_t ex_c = lookback_exclusive(st_kept, epoch, bw, tot_c, &s_excl);
  __syncthreads();
  const uint32_t ex_e = lookback_exclusive(st_ids, epoch, bw, tot_e, &s_excl);
  if (threadIdx.x == 0 && bw == gridDim.x - nt - 1) {
    mirror[MIRROR_MASK_TOTALS + 1] = ex_e + tot_e;
    mirror_publish(mirror, MIRROR_FLAG_MASK1, seq);
  }
  if (b < nb && c) {
    const uint32_t id = ex_e + pe;
    blk_node2[id] = blk_node[b];
    blk_slot2[id] = blk_slot[b];
    blk_start2[id] = ex_c + pc;
    blk_size2[id] = (int32_t)c;
  }
}

// The three prefix sums of apply_mask come out of ONE scan over [tile counts | kept per block | block
// non-empty]: the second and third segment carry the totals of the segments in front of them, which are
// read from their first entries.
__global__ __launch_bounds__(256) void k_blk_compact(
    const uint32_t* __restrict__ raw, const uint32_t* __restrict__ scanned, const uint32_t* __restrict__ grand_total,
    int64_t nt, int64_t nb, const int32_t* __restrict__ blk_node,
    const int32_t* __restrict__ blk_slot, int32_t* __restrict__ blk_node2, int32_t* __restrict__ blk_slot2,
    uint32_t* __restrict__ blk_start2, int32_t* __restrict__ blk_size2, uint32_t* __restrict__ mirror,
    uint32_t seq) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t base_kept = scanned[nt], base_id = scanned[nt + nb];
  if (b == 0) {
    mirror[MIRROR_MASK_TOTALS] = base_id - base_kept;          // kept points
    mirror[MIRROR_MASK_TOTALS + 1] = *grand_total - base_id;   // non-empty blocks
    mirror_publish(mirror, MIRROR_FLAG_MASK0, seq);
    mirror_publish(mirror, MIRROR_FLAG_MASK1, seq);
  }
  if (b >= nb) return;
  const uint32_t c = raw[nt + b];
  if (!c) return;
  const uint32_t id = scanned[nt + nb + b] - base_id;
  blk_node2[id] = blk_node[b];
  blk_slot2[id] = blk_slot[b];
  blk_start2[id] = scanned[nt + b] - base_kept;
  blk_size2[id] = (int32_t)c;
}

// OctreeNode.filter for count predicates (octree.py:102-112): a leaf of a selected pose whose point
// count lies outside [lo, hi] is emptied - one wavefront per block clears its mask bytes
__global__ __launch_bounds__(256) void k_filter_blocks(const uint32_t* __restrict__ blk_start,
                                                       const int32_t* __restrict__ blk_size,
                                                       const int32_t* __restrict__ blk_slot, int64_t nb,
                                                       const uint8_t* __restrict__ slot_sel, int64_t lo,
                                                       int64_t hi, uint8_t* __restrict__ mask) {
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= nb) return;
  const int64_t n = blk_size[b];
  if (!slot_sel[blk_slot[b]] || (n >= lo && n <= hi)) return;
  const int64_t s0 = blk_start[b];
  for (int64_t i = threadIdx.x & 63; i < n; i += 64) mask[s0 + i] = 0;
}

// counters of one pose (octree.py:144-175, grid.py:343-362) without fetching the block table:
// out[0] = points, out[1] = non-empty leaves of the slot
__global__ __launch_bounds__(256) void k_slot_counts(const int32_t* __restrict__ blk_slot,
                                                     const int32_t* __restrict__ blk_size, int64_t nb,
                                                     int32_t slot, unsigned long long* __restrict__ out) {
  __shared__ unsigned long long part[2][4];
  unsigned long long pts = 0, lv = 0;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += (int64_t)gridDim.x * blockDim.x) {
    if (blk_slot[b] == slot) {
      pts += (unsigned long long)blk_size[b];
      lv += 1;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    pts += __shfl_xor(pts, off);
    lv += __shfl_xor(lv, off);
  }
  if ((threadIdx.x & 63) == 0) {
    part[0][threadIdx.x >> 6] = pts;
    part[1][threadIdx.x >> 6] = lv;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    const unsigned long long t = part[threadIdx.x][0] + part[threadIdx.x][1] + part[threadIdx.x][2] + part[threadIdx.x][3];
    if (t) atomicAdd(&out[threadIdx.x], t);
  }
}

// internal scheme nodes per top-level voxel (n_nodes of a pose = sum over its voxels of 1 + 8 * internal)
__global__ __launch_bounds__(256) void k_internal_per_voxel(const int32_t* __restrict__ first_child,
                                                            const int32_t* __restrict__ voxel, int64_t n,
                                                            int32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && first_child[i] >= 0) atomicAdd(&out[voxel[i]], 1);
}

// voxels in which a pose slot has at least one block
__global__ __launch_bounds__(256) void k_slot_voxel_flags(const int32_t* __restrict__ blk_node,
                                                          const int32_t* __restrict__ blk_slot,
                                                          int64_t nb, int32_t slot,
                                                          const int32_t* __restrict__ node_voxel,
                                                          uint32_t* __restrict__ flags) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nb && blk_slot[b] == slot) flags[node_voxel[blk_node[b]]] = 1u;
}

__global__ __launch_bounds__(256) void k_flag_indices(const uint32_t* __restrict__ flags,
                                                      const uint32_t* __restrict__ scanned, int64_t n,
                                                      int32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flags[i]) out[scanned[i]] = (int32_t)i;
}

__global__ __launch_bounds__(256) void k_widen_u32_i64(const uint32_t* __restrict__ in, int64_t n,
                                                       int64_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (int64_t)in[i];
}

inline unsigned grid_for(int64_t n) { return (unsigned)ceil_div(n, 256); }

// octl_forest_gather_blocks: sizes of the selected blocks, then one wavefront per block copies its rows to where the
// prefix sum of the sizes puts them
__global__ __launch_bounds__(256) void k_gather_sizes(const int32_t* __restrict__ ids, int64_t m,
                                                      const int32_t* __restrict__ blk_size, uint32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) out[i] = (uint32_t)blk_size[ids[i]];
}
__global__ __launch_bounds__(256) void k_gather_rows(const int32_t* __restrict__ ids, int64_t m,
                                                     const uint32_t* __restrict__ blk_start,
                                                     const int32_t* __restrict__ blk_size,
                                                     const uint32_t* __restrict__ offs, const double* __restrict__ xyz,
                                                     double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= m) return;
  const int32_t b = ids[i];
  const double* src = xyz + 3 * (size_t)blk_start[b];
  double* dst = out + 3 * (size_t)offs[i];
  const int n3 = 3 * blk_size[b];
  for (int j = threadIdx.x & 63; j < n3; j += 64) dst[j] = src[j];
}

}  // namespace

int bbox_ensure(octl_forest* f) {
  octl_ctx* ctx = f->ctx;
  // (no page-locked mirror per forest: hipHostMalloc / hipHostFree synchronise the whole device - also the copy
  //  stream's upload of the next scan; the one readback of the box goes through the context's scalar mirror)
  if (!f->bbox_dev.p) {
    OCTL_TRY(devbuf_reserve(ctx, f->bbox_dev, 32));
    f->bbox_stale = true;
  }
  if (!f->bbox_stale) return OCTL_OK;
  OCTL_LAUNCH(k_bbox_reset, dim3(1), dim3(64), 0, ctx->stream, f->bbox_dev.as<int32_t>());
  HIP_TRY(ctx, hipGetLastError());
  f->bbox_stale = false;
  return OCTL_OK;
}

int alive_ensure(octl_forest* f) {
  if (!f->alive_stale) return OCTL_OK;
  if (f->n_store > 0) HIP_TRY(f->ctx, hipMemsetAsync(f->alive.p, 1, (size_t)f->n_store, f->ctx->stream));
  f->alive_stale = false;
  return OCTL_OK;
}

int forest_settle(octl_forest* f) {
  if (!f->totals_pending) return OCTL_OK;
  octl_ctx* ctx = f->ctx;
  f->totals_pending = false;
  if (ctx->pending_mask_forest == f) ctx->pending_mask_forest = nullptr;
  const int flags[2] = {MIRROR_FLAG_MASK0, MIRROR_FLAG_MASK1};
  const int64_t n = f->totals_n_before;
  OCTL_TRY(octl_wait_mirror_flags(ctx, flags, 2, f->totals_seq, 500 + n / 2000));
  uint32_t res[2];
  std::memcpy(res, static_cast<uint32_t*>(ctx->small_host) + MIRROR_MASK_TOTALS, 8);
  f->n_alive -= (n - (int64_t)res[0]);
  f->n_ord = res[0];
  f->n_blocks = res[1];
  return OCTL_OK;
}

void forest_forget_pending(octl_forest* f) {
  if (!f->totals_pending) return;
  f->totals_pending = false;
  if (f->ctx->pending_mask_forest == f) f->ctx->pending_mask_forest = nullptr;
}


namespace {

// Device sources are consumed in stream order (no synchronisation: the caller keeps the buffer
// unchanged until the next synchronising call on the context); host sources are copied before the
// call returns.
int store_append(octl_forest* f, const double* xyz, int64_t n, bool from_device) {
  octl_ctx* ctx = f->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (n < 0 || (n > 0 && !xyz)) return octl_set_error(ctx, OCTL_E_INVALID, "bad point buffer");
  const int64_t total = f->n_store + n;
  if (total >= ((int64_t)1 << 31))
    return octl_set_error(ctx, OCTL_E_INVALID, "more than 2^31-1 points in one forest");
  // a store that is read in place from the caller's buffer becomes the forest's own before it grows; a
  // cloud whose box has not been taken yet is folded in now (the kernel below only adds the new points)
  if (f->store_borrowed) OCTL_TRY(store_materialize(f));
  if (f->bbox_pending) OCTL_TRY(store_compute_bbox(f));
  OCTL_TRY(alive_ensure(f));   // (the flags of the points in front of the new ones: the buffer may move)
  OCTL_TRY(devbuf_reserve(ctx, f->xyz, (size_t)std::max<int64_t>(total, 1) * 24 + 16, 1));
  OCTL_TRY(devbuf_reserve(ctx, f->alive, (size_t)std::max<int64_t>(total, 1) + 2, 1));
  OCTL_TRY(bbox_ensure(f));
  if (n > 0) {
    hipStream_t st = ctx->stream;
    double* dst = f->xyz.as<double>() + 3 * f->n_store;
    uint8_t* alive = f->alive.as<uint8_t>() + f->n_store;
    const unsigned grid = (unsigned)std::max<int64_t>(1, ceil_div(3 * n / 2, 256 * ING_UNITS));
    KTimer t(ctx, "ingest");
    // (an odd store offset would misalign the 16-byte accesses of the pair-wise kernel: such a pose
    //  goes through the plain copy + the in-place form on its own, 8-byte aligned, pointer)
    const bool aligned = (f->n_store % 2) == 0 && (reinterpret_cast<uintptr_t>(xyz) % 16) == 0;
    if (from_device && xyz == dst) {
      // an adopted buffer (store_adopt): the points are in place already
      OCTL_LAUNCH(k_ingest<false>, dim3(grid), dim3(256), 0, st, (const double*)dst, dst, alive, n,
                         f->mode, f->edge, f->bbox_dev.as<int32_t>());
    } else if (from_device && aligned) {
      OCTL_LAUNCH(k_ingest<true>, dim3(grid), dim3(256), 0, st, xyz, dst, alive, n, f->mode,
                         f->edge, f->bbox_dev.as<int32_t>());
    } else {
      HIP_TRY(ctx, hipMemcpyAsync(dst, xyz, (size_t)n * 24,
                                  from_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
      if ((f->n_store % 2) == 0) {
        OCTL_LAUNCH(k_ingest<false>, dim3(grid), dim3(256), 0, st, (const double*)dst, dst, alive,
                           n, f->mode, f->edge, f->bbox_dev.as<int32_t>());
      } else {
        // first point alone, then the aligned rest
        OCTL_LAUNCH(k_ingest<false>, dim3(1), dim3(256), 0, st, (const double*)dst, dst, alive,
                           (int64_t)1, f->mode, f->edge, f->bbox_dev.as<int32_t>());
        if (n > 1)
          OCTL_LAUNCH(k_ingest<false>, dim3((unsigned)std::max<int64_t>(1, ceil_div(3 * (n - 1) / 2, 256 * ING_UNITS))), dim3(256),
                             0, st, (const double*)(dst + 3), dst + 3, alive + 1, n - 1, f->mode, f->edge,
                             f->bbox_dev.as<int32_t>());
      }
    }
    HIP_TRY(ctx, hipGetLastError());
    // (the box stays on the device: the build forms the key geometry there, or fetches it when it has to)
    if (!from_device) HIP_TRY(ctx, hipStreamSynchronize(st));  // the host buffer is the caller's again
  }
  return OCTL_OK;
}

// An EMPTY forest takes the n points that f->xyz already holds (a swapped-in routed buffer, a borrowed
// caller buffer) as its first pose without touching them: alive flags by memset, the voxel box left to the
// build (bbox_pending).
int store_take_in_place(octl_forest* f, int64_t n) {
  octl_ctx* ctx = f->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (n >= ((int64_t)1 << 31))
    return octl_set_error(ctx, OCTL_E_INVALID, "more than 2^31-1 points in one forest");
  OCTL_TRY(devbuf_reserve(ctx, f->alive, (size_t)std::max<int64_t>(n, 1) + 2, 0));
  if (!f->bbox_dev.p) {
    OCTL_TRY(devbuf_reserve(ctx, f->bbox_dev, 32));
    f->bbox_stale = true;
  }
  // (no launch here: the flags are written when something needs them, the box is reset by whoever fills it)
  f->alive_stale = true;
  f->bbox_pending = true;
  return OCTL_OK;
}

// The block table describes the leaf-ordered arrays exactly (every producer leaves it that way), so both
// are compacted together: points tile-wise, blocks block-wise.  One synchronisation (kept points, blocks).
// (async: return behind the last launch - forest_settle books the counts when somebody looks at the forest again)
int apply_device_mask(octl_forest* f, int64_t* n_alive_out, bool async = false) {
  octl_ctx* ctx = f->ctx;
  hipStream_t st = ctx->stream;
  // (the counts travel through two words of the context's mirror: one compaction in flight per context)
  if (ctx->pending_mask_forest && ctx->pending_mask_forest != f) OCTL_TRY(forest_settle(ctx->pending_mask_forest));
  OCTL_TRY(forest_settle(f));
  const int64_t n = f->n_ord, nb = f->n_blocks;
  f->mask_valid = false;
  f->fast_order_valid = false;  // block ids change
  if (n > 0 && nb > 0) {
    KTimer t(ctx, "apply_mask");
    uint32_t* small = ctx->small.as<uint32_t>();
    const int64_t nt = ceil_div(n, 2048);
    // scratch: in [tile counts nt | kept nb | non-empty nb], out (scanned) the same layout
    const int64_t n_all = nt + 2 * nb;
    const size_t o_out = (((size_t)n_all + 8) * 4 + 15) & ~(size_t)15;
    OCTL_TRY(devbuf_reserve(ctx, f->flags, 2 * o_out));
    uint32_t* raw = f->flags.as<uint32_t>();
    uint32_t* scanned = reinterpret_cast<uint32_t*>(static_cast<char*>(f->flags.p) + o_out);
    const uint8_t* mask = f->mask.as<uint8_t>();
    // the fused form chains its workgroups by look-back: beyond ~1000 of them the chain costs more than the
    // separate scan (10 M points: 65 us against 35)
    const bool fused = !ctx->opt.no_fused_tables && nt + 2 * ceil_div(nb, 256) <= 1024;
    const uint32_t wait_seq = octl_wait_next_seq(ctx);
    // (alive flags that were never written - a cloud taken in place - are filled by the tile workgroups of the
    //  first kernel: by position in k_blk_kept - k_compact_tiles, a later launch, then clears the dropped points' -
    //  and by store index, each flag once, where k_mask_scan compacts in the same launch)
    uint8_t* fill = nullptr;
    if (f->alive_stale && f->n_ord == f->n_store) {
      fill = f->alive.as<uint8_t>();
      f->alive_stale = false;
    } else {
      OCTL_TRY(alive_ensure(f));
    }
    if (!fused) {
      OCTL_LAUNCH(k_blk_kept, dim3((unsigned)nt + grid_for(nb)), dim3(256), 0, st, mask, n, (uint32_t)nt, raw,
                         (const uint32_t*)f->blk_start.as<uint32_t>(), (const int32_t*)f->blk_size.as<int32_t>(), nb,
                         raw + nt, raw + nt + nb, fill);
      HIP_TRY(ctx, hipGetLastError());
      OCTL_TRY(octl_exclusive_scan_u32(ctx, raw, scanned, n_all, small + 22));
    }
    OCTL_TRY(devbuf_reserve(ctx, f->ord_idx2, (size_t)n * 4));
    OCTL_TRY(devbuf_reserve(ctx, f->xyz_ord2, (size_t)n * 24));
    // (block buffers keep the capacity convention of forest_make_blocks: one block per point)
    OCTL_TRY(devbuf_reserve(ctx, f->blk_node2, (size_t)n * 4));
    OCTL_TRY(devbuf_reserve(ctx, f->blk_slot2, (size_t)n * 4));
    OCTL_TRY(devbuf_reserve(ctx, f->blk_start2, (size_t)n * 4));
    OCTL_TRY(devbuf_reserve(ctx, f->blk_size2, (size_t)n * 4));
    if (fused) {
      // counts + the three prefix sums + the block table's compaction: one launch (k_mask_scan)
      const int64_t nbw = ceil_div(nb, 256);
      uint64_t* status = nullptr;
      uint32_t epoch = 0;
      OCTL_TRY(octl_scan_status_acquire(ctx, nt + 2 * nbw, &status, &epoch));
      OCTL_LAUNCH(k_mask_scan, dim3((unsigned)(nt + nbw)), dim3(256), 0, st, mask, n, (uint32_t)nt,
                         (const uint32_t*)f->ord_idx.as<uint32_t>(), (const double*)f->xyz_ord.as<double>(),
                         f->ord_idx2.as<uint32_t>(), f->xyz_ord2.as<double>(), f->alive.as<uint8_t>(),
                         (const uint32_t*)f->blk_start.as<uint32_t>(), (const int32_t*)f->blk_size.as<int32_t>(), nb,
                         (const int32_t*)f->blk_node.as<int32_t>(), (const int32_t*)f->blk_slot.as<int32_t>(),
                         f->blk_node2.as<int32_t>(), f->blk_slot2.as<int32_t>(), f->blk_start2.as<uint32_t>(),
                         f->blk_size2.as<int32_t>(), status, status + nt, status + nt + nbw, epoch,
                         static_cast<uint32_t*>(ctx->small_host), wait_seq, fill);
      HIP_TRY(ctx, hipGetLastError());
    }
    if (!fused) {
      OCTL_LAUNCH(k_compact_tiles, dim3((unsigned)nt), dim3(256), 0, st, mask, (const uint32_t*)scanned, n,
                         (const uint32_t*)f->ord_idx.as<uint32_t>(), (const double*)f->xyz_ord.as<double>(),
                         f->ord_idx2.as<uint32_t>(), f->xyz_ord2.as<double>(), f->alive.as<uint8_t>());
      HIP_TRY(ctx, hipGetLastError());
      OCTL_LAUNCH(k_blk_compact, dim3(grid_for(nb)), dim3(256), 0, st, (const uint32_t*)raw,
                         (const uint32_t*)scanned, (const uint32_t*)(small + 22), nt, nb,
                         (const int32_t*)f->blk_node.as<int32_t>(), (const int32_t*)f->blk_slot.as<int32_t>(),
                         f->blk_node2.as<int32_t>(), f->blk_slot2.as<int32_t>(), f->blk_start2.as<uint32_t>(),
                         f->blk_size2.as<int32_t>(), static_cast<uint32_t*>(ctx->small_host), wait_seq);
      HIP_TRY(ctx, hipGetLastError());
    }
    // (the two totals and their flags are written into the pinned mirror by the kernels themselves: the host polls
    //  for them - the compaction of the points may still be running when this returns, everything behind it is
    //  ordered by the stream)
    std::swap(f->ord_idx, f->ord_idx2);
    std::swap(f->xyz_ord, f->xyz_ord2);
    std::swap(f->blk_node, f->blk_node2);
    std::swap(f->blk_slot, f->blk_slot2);
    std::swap(f->blk_start, f->blk_start2);
    std::swap(f->blk_size, f->blk_size2);
    f->totals_pending = true;
    f->totals_seq = wait_seq;
    f->totals_n_before = n;
    ctx->pending_mask_forest = f;
    if (!async) OCTL_TRY(forest_settle(f));
  }
  if (n_alive_out) *n_alive_out = f->n_ord;
  return OCTL_OK;
}

int ensure_mask(octl_forest* f) {
  octl_ctx* ctx = f->ctx;
  if (f->mask_valid) return OCTL_OK;
  OCTL_TRY(devbuf_reserve(ctx, f->mask, (size_t)std::max<int64_t>(f->n_ord, 1)));
  if (f->n_ord > 0)
    HIP_TRY(ctx, hipMemsetAsync(f->mask.p, 1, (size_t)f->n_ord, ctx->stream));
  f->mask_valid = true;
  return OCTL_OK;
}

}  // namespace

// An empty store takes over a library-owned device buffer that holds the cloud (and hands its own
// buffer back in exchange) instead of copying it: the routed cloud of the multi-GPU path.
int store_adopt(octl_forest* f, DevBuf& src, int64_t n, bool* adopted) {
  *adopted = !(f->n_store != 0 || n <= 0 || src.cap < (size_t)n * 24 + 16 || f->store_borrowed);
  if (!*adopted) return store_append(f, src.as<double>(), n, true);
  std::swap(f->xyz, src);
  return store_take_in_place(f, n);
}

int store_compute_bbox(octl_forest* f) {
  octl_ctx* ctx = f->ctx;
  f->bbox_pending = false;
  const int64_t n = f->n_store;
  if (n <= 0) return OCTL_OK;
  OCTL_TRY(bbox_ensure(f));
  KTimer t(ctx, "ingest");
  const unsigned grid = (unsigned)std::max<int64_t>(1, ceil_div(3 * n / 2, 256 * ING_UNITS));
  double* p = f->xyz.as<double>();
  OCTL_LAUNCH(k_ingest<false>, dim3(grid), dim3(256), 0, ctx->stream, (const double*)p, p,
                     (uint8_t*)nullptr, n, f->mode, f->edge, f->bbox_dev.as<int32_t>());
  HIP_TRY(ctx, hipGetLastError());
  return OCTL_OK;
}

int store_materialize(octl_forest* f) {
  octl_ctx* ctx = f->ctx;
  if (!f->store_borrowed) return OCTL_OK;
  DevBuf own = f->xyz_own;
  f->xyz_own = DevBuf{};
  const int rc = devbuf_reserve(ctx, own, (size_t)std::max<int64_t>(f->n_store, 1) * 24 + 16, 0);
  if (rc != OCTL_OK) {
    f->xyz_own = own;
    return rc;
  }
  if (f->n_store > 0)
    HIP_TRY(ctx, hipMemcpyAsync(own.p, f->xyz.p, (size_t)f->n_store * 24, hipMemcpyDeviceToDevice, ctx->stream));
  f->xyz = own;
  f->store_borrowed = false;
  return OCTL_OK;
}


extern "C" {

int octl_forest_create(octl_ctx* ctx, int mode, const double corner[3], double edge,
                       octl_forest** out) {
  if (!ctx || !out || !corner) return OCTL_E_INVALID;
  *out = nullptr;
  if (mode != 0 && mode != 1) return octl_set_error(ctx, OCTL_E_INVALID, "mode must be 0 or 1");
  if (!(edge > 0.0)) return octl_set_error(ctx, OCTL_E_INVALID, "edge length must be positive");
  if (mode == 0) {
    if (corner[0] != 0.0 || corner[1] != 0.0 || corner[2] != 0.0)
      return octl_set_error(ctx, OCTL_E_INVALID,
                            "a non-zero grid corner is outside the parity domain: the reference "
                            "places its octrees relative to the corner but subtracts absolute "
                            "points (grid.py:96-105 vs octree.py:74)");
    if (edge != (double)(int64_t)edge)
      return octl_set_error(ctx, OCTL_E_INVALID,
                            "voxel_edge_length must be integer valued: the reference truncates "
                            "voxel coordinates with astype(int) (grid.py:72-76)");
  }
  octl_forest* f = new octl_forest();
  f->ctx = ctx;
  f->mode = mode;
  for (int a = 0; a < 3; ++a) f->corner[a] = corner[a];
  f->edge = edge;
  *out = f;
  return OCTL_OK;
}

void octl_forest_destroy(octl_forest* f) {
  if (!f) return;
  forest_forget_pending(f);
  (void)hipSetDevice(f->ctx->device);
  (void)hipStreamSynchronize(f->ctx->stream);
  nodes_free(f->ctx, f->nodes[0]);
  nodes_free(f->ctx, f->nodes[1]);
  if (f->store_borrowed) {  // (the caller's buffer is not ours to release)
    f->xyz = f->xyz_own;
    f->xyz_own = DevBuf{};
    f->store_borrowed = false;
  }
  for (DevBuf* b :
       {&f->bbox_dev, &f->part_xyz[0], &f->part_xyz[1], &f->bk_table, &f->bk_tot, &f->bk_chunks, &f->bk_vox, &f->bk_node, &f->leafinfo,
        &f->xyz, &f->alive, &f->ord_idx, &f->xyz_ord, &f->pos_node, &f->blk_node, &f->blk_slot,
        &f->blk_start, &f->blk_size, &f->blk_node2, &f->blk_slot2, &f->blk_start2, &f->blk_size2, &f->mask, &f->blk_eval, &f->rs_scratch, &f->rs_order, &f->fast_order,
        &f->rs_hyp, &f->rs_plane, &f->rs_count, &f->rs_index, &f->ord_idx2, &f->xyz_ord2,
        &f->vkey, &f->path, &f->lin[0], &f->lin[1], &f->val[0], &f->val[1],
        &f->hist, &f->idxbuf[0], &f->idxbuf[1], &f->pathbuf[0], &f->pathbuf[1], &f->flags,
        &f->entries, &f->split[0], &f->split[1], &f->split_tiles[0], &f->split_tiles[1],
        &f->child_sc, &f->pose_off_dev, &f->scheme_dev, &f->root_up, &f->vlin_dev, &f->vcode_dev[0],
        &f->vcode_dev[1]})
    devbuf_release(f->ctx, *b);
  delete f;
}

int octl_forest_clear(octl_forest* f) {
  if (!f) return OCTL_E_INVALID;
  forest_forget_pending(f);   // (counts of a compaction still in flight: nobody will ask for them)
  f->max_block_hint = INT64_MAX;
  f->bbox_stale = true;   // (nothing is launched: whoever fills the box next resets it first)
  f->alive_stale = false;
  if (f->store_borrowed) {  // back to the forest's own block; the caller's buffer is the caller's again
    f->xyz = f->xyz_own;
    f->xyz_own = DevBuf{};
    f->store_borrowed = false;
  }
  f->bbox_pending = false;
  f->displaced_rows = false;
  f->pose_off.assign(1, 0);
  f->n_store = f->n_alive = 0;
  f->store_dirty = true;
  f->nodes[0].n = f->nodes[1].n = 0;
  f->cur = 0;
  f->epoch = 0;
  f->built = false;
  f->vkeys.clear();
  f->vkeys_stale = false;
  f->vorg_set = false;
  f->vcode_valid = false;
  f->fast_order_valid = false;
  f->built_store = 0;
  f->built_poses = 0;
  f->append_only = true;
  f->n_voxels = 0;
  f->level_segs.clear();
  f->n_internal = 0;
  f->max_depth_reached = 0;
  f->n_ord = 0;
  f->n_blocks = 0;
  f->mask_valid = false;
  return OCTL_OK;
}

int octl_forest_add_pose(octl_forest* f, const double* xyz, int64_t n, int32_t* slot) {
  if (!f) return OCTL_E_INVALID;
  OCTL_TRY(forest_settle(f));
  OCTL_TRY(store_append(f, xyz, n, false));
  if (slot) *slot = (int32_t)f->pose_off.size() - 1;
  f->n_store += n;
  f->n_alive += n;
  f->pose_off.push_back(f->n_store);
  f->store_dirty = true;
  return OCTL_OK;
}

int octl_forest_add_pose_device(octl_forest* f, const double* xyz_dev, int64_t n, int32_t* slot) {
  if (!f) return OCTL_E_INVALID;
  OCTL_TRY(forest_settle(f));
  // (the cloud may be the target of an octl_dev_upload_async that is still in flight)
  if (n > 0) OCTL_TRY(ctx_wait_uploads(f->ctx, xyz_dev, (size_t)n * 24));
  OCTL_TRY(store_append(f, xyz_dev, n, true));
  if (slot) *slot = (int32_t)f->pose_off.size() - 1;
  f->n_store += n;
  f->n_alive += n;
  f->pose_off.push_back(f->n_store);
  f->store_dirty = true;
  return OCTL_OK;
}

int octl_forest_add_pose_adopt(octl_forest* f, const double* xyz_dev, int64_t n, int32_t* slot) {
  if (!f) return OCTL_E_INVALID;
  OCTL_TRY(forest_settle(f));
  // only the first pose of an empty forest can be read in place (the store is one contiguous array); the
  // kernels read 16 bytes at a time from the start of the cloud
  if (f->n_store != 0 || n <= 0 || !xyz_dev || (reinterpret_cast<uintptr_t>(xyz_dev) & 15) != 0)
    return octl_forest_add_pose_device(f, xyz_dev, n, slot);
  // (the cloud may be the target of an octl_dev_upload_async that is still in flight)
  OCTL_TRY(ctx_wait_uploads(f->ctx, xyz_dev, (size_t)n * 24));
  if (!f->store_borrowed) f->xyz_own = f->xyz;
  f->xyz = DevBuf{const_cast<double*>(xyz_dev), 0};
  f->store_borrowed = true;
  const int rc = store_take_in_place(f, n);
  if (rc != OCTL_OK) {
    f->xyz = f->xyz_own;
    f->xyz_own = DevBuf{};
    f->store_borrowed = false;
    return rc;
  }
  if (slot) *slot = (int32_t)f->pose_off.size() - 1;
  f->n_store += n;
  f->n_alive += n;
  f->pose_off.push_back(f->n_store);
  f->store_dirty = true;
  return OCTL_OK;
}

static int extend_pose_impl(octl_forest* f, int32_t slot, const double* xyz, int64_t n, bool from_device) {
  if (!f) return OCTL_E_INVALID;
  octl_ctx* ctx = f->ctx;
  const int n_poses = (int)f->pose_off.size() - 1;
  if (slot < 0 || slot >= n_poses) return octl_set_error(ctx, OCTL_E_INVALID, "bad pose slot");
  const int64_t tail = f->n_store - f->pose_off[slot + 1];  // points of the later poses
  // (a device cloud may be the target of an octl_dev_upload_async that is still in flight)
  if (from_device && n > 0) OCTL_TRY(ctx_wait_uploads(ctx, xyz, (size_t)n * 24));
  OCTL_TRY(store_append(f, xyz, n, from_device));  // lands behind everything (bounding box, alive flags)
  if (tail > 0 && n > 0) {
    // The store is pose-major: rotate the new points in front of the later poses' points (they were
    // appended at the end).  The whole range [tail | new] goes through the partition scratch and comes
    // back as [new | tail]: with more new points than later points the two pieces overlap in the
    // store, so nothing is copied store-to-store.  The next build re-derives every table from the store.
    hipStream_t st = ctx->stream;
    const int64_t at = f->pose_off[slot + 1];
    const int64_t span = tail + n;
    OCTL_TRY(devbuf_reserve(ctx, f->part_xyz[0], (size_t)span * 25));
    char* tmp = static_cast<char*>(f->part_xyz[0].p);
    char* tmp_al = tmp + (size_t)span * 24;
    double* xs = f->xyz.as<double>();
    uint8_t* al = f->alive.as<uint8_t>();
    HIP_TRY(ctx, hipMemcpyAsync(tmp, xs + 3 * at, (size_t)span * 24, hipMemcpyDeviceToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(tmp_al, al + at, (size_t)span, hipMemcpyDeviceToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(xs + 3 * at, tmp + (size_t)tail * 24, (size_t)n * 24, hipMemcpyDeviceToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(al + at, tmp_al + tail, (size_t)n, hipMemcpyDeviceToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(xs + 3 * (at + n), tmp, (size_t)tail * 24, hipMemcpyDeviceToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(al + at + n, tmp_al, (size_t)tail, hipMemcpyDeviceToDevice, st));
  }
  f->n_store += n;
  f->n_alive += n;
  for (int p = slot + 1; p <= n_poses; ++p) f->pose_off[p] += n;
  f->store_dirty = true;
  f->append_only = false;  // the store was rotated: the next build re-places everything
  return OCTL_OK;
}

int octl_forest_extend_pose(octl_forest* f, int32_t slot, const double* xyz, int64_t n) {
  if (f) OCTL_TRY(forest_settle(f));
  return extend_pose_impl(f, slot, xyz, n, false);
}

int octl_forest_extend_pose_device(octl_forest* f, int32_t slot, const double* xyz_dev, int64_t n) {
  if (f) OCTL_TRY(forest_settle(f));
  return extend_pose_impl(f, slot, xyz_dev, n, true);
}

int octl_forest_build(octl_forest* f, int64_t K, const uint8_t* scheme_mask, int32_t n_mask,
                      int32_t keep_scheme, int32_t max_depth, octl_build_info* info) {
  if (!f) return OCTL_E_INVALID;
  OCTL_TRY(forest_settle(f));
  const int rc = forest_build(f, K, scheme_mask, n_mask, keep_scheme, max_depth, info);
  if (rc != OCTL_OK && rc != OCTL_E_INVALID && rc != OCTL_E_STATE) {
    // a build that failed half way has overwritten scratch the previous tables referred to:
    // the forest is left WITHOUT a scheme (its points and voxels are kept); the next build starts
    // from the top-level voxels again
    f->built = false;
    f->n_ord = 0;
    f->n_blocks = 0;
    f->n_internal = 0;
    f->mask_valid = false;
    f->store_dirty = true;
  }
  return rc;
}

int octl_forest_set_scheme(octl_forest* f, const int32_t* first_child, const int32_t* epoch,
                           int64_t n_nodes, int32_t new_epoch) {
  if (!f || !first_child || !epoch) return OCTL_E_INVALID;
  OCTL_TRY(forest_settle(f));
  octl_ctx* ctx = f->ctx;
  if (!f->built) return octl_set_error(ctx, OCTL_E_STATE, "set_scheme needs a built forest");
  const int64_t V = f->n_voxels;
  if (n_nodes < V || (n_nodes - V) % 8 != 0 || n_nodes >= ((int64_t)1 << 31))
    return octl_set_error(ctx, OCTL_E_INVALID, "scheme must have V roots + 8 nodes per internal node");
  int64_t n_internal = 0;
  for (int64_t i = 0; i < n_nodes; ++i) {
    const int32_t c = first_child[i];
    if (c >= 0) {
      if (c < V || (int64_t)c + 8 > n_nodes || (c - V) % 8 != 0)
        return octl_set_error(ctx, OCTL_E_INVALID, "first_child[%lld] = %d is not a valid child group",
                              (long long)i, c);
      ++n_internal;
    }
  }
  if (V + 8 * n_internal != n_nodes)
    return octl_set_error(ctx, OCTL_E_INVALID, "node count does not match the number of internal nodes");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  NodeTable& t = f->nodes[f->cur];
  OCTL_TRY(nodes_reserve(ctx, t, n_nodes));
  HIP_TRY(ctx, hipMemcpyAsync(t.first_child.p, first_child, (size_t)n_nodes * 4, hipMemcpyHostToDevice,
                              ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(t.epoch.p, epoch, (size_t)n_nodes * 4, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  t.n = n_nodes;
  f->n_internal = n_internal;
  f->uniform_epoch = false;
  if (new_epoch > f->epoch) f->epoch = new_epoch;
  // only first_child / epoch of this table are meaningful until the next (keep_scheme) build,
  // which has to place every point again
  f->append_only = false;
  f->fast_order_valid = false;
  f->n_ord = 0;
  f->n_blocks = 0;
  f->mask_valid = false;
  return OCTL_OK;
}

int octl_forest_set_contents(octl_forest* f, int64_t n_blocks, const int32_t* blk_node, const int32_t* blk_slot,
                             const int32_t* blk_size, const double* xyz) {
  if (!f) return OCTL_E_INVALID;
  OCTL_TRY(forest_settle(f));
  octl_ctx* ctx = f->ctx;
  if (!f->built || f->store_dirty)
    return octl_set_error(ctx, OCTL_E_STATE, "set_contents needs a forest whose tables are up to date (build first)");
  if (n_blocks < 0 || (n_blocks > 0 && (!blk_node || !blk_slot || !blk_size)))
    return octl_set_error(ctx, OCTL_E_INVALID, "bad block arrays");
  const int n_poses = (int)f->pose_off.size() - 1;
  const int64_t n_nodes = f->nodes[f->cur].n;
  // new pose offsets: the store is pose-major, a pose's points in the storage order of its blocks
  std::vector<int64_t> per_slot((size_t)std::max(n_poses, 1), 0);
  int64_t total = 0;
  for (int64_t b = 0; b < n_blocks; ++b) {
    if (blk_size[b] <= 0) return octl_set_error(ctx, OCTL_E_INVALID, "block %lld is empty", (long long)b);
    if (blk_slot[b] < 0 || blk_slot[b] >= n_poses || blk_node[b] < 0 || blk_node[b] >= n_nodes)
      return octl_set_error(ctx, OCTL_E_INVALID, "block %lld: bad node or pose slot", (long long)b);
    per_slot[(size_t)blk_slot[b]] += blk_size[b];
    total += blk_size[b];
  }
  if (total >= ((int64_t)1 << 31)) return octl_set_error(ctx, OCTL_E_INVALID, "more than 2^31-1 points in one forest");
  if (total > 0 && !xyz) return octl_set_error(ctx, OCTL_E_INVALID, "null point array");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  // The blocks must name LEAVES of the current scheme, every (leaf, pose) pair at most once, in the storage order
  // of the block table they replace: the (leaf, pose) pairs of the table this forest holds, some of them possibly
  // missing (emptied leaves), none added out of place.  A table that breaks this would be committed as it is and
  // corrupt every later query, so it is checked against the forest's own table first.
  if (n_blocks > 0) {
    std::vector<int32_t> fc((size_t)n_nodes);
    HIP_TRY(ctx, hipMemcpyAsync(fc.data(), f->nodes[f->cur].first_child.p, (size_t)n_nodes * 4, hipMemcpyDeviceToHost, st));
    const int64_t nb_old = f->n_blocks;
    std::vector<int32_t> on((size_t)std::max<int64_t>(nb_old, 1)), os((size_t)std::max<int64_t>(nb_old, 1));
    if (nb_old > 0) {
      HIP_TRY(ctx, hipMemcpyAsync(on.data(), f->blk_node.p, (size_t)nb_old * 4, hipMemcpyDeviceToHost, st));
      HIP_TRY(ctx, hipMemcpyAsync(os.data(), f->blk_slot.p, (size_t)nb_old * 4, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(ctx, hipStreamSynchronize(st));
    // rank of a (leaf, pose) pair in the current table's storage order; pairs the table does not hold (a leaf that
    // was empty for the pose) sort by their leaf's first occurrence - unknown leaves keep the caller's order
    // among themselves only as far as (node, slot) is strictly increasing
    std::unordered_map<uint64_t, int64_t> rank;
    rank.reserve((size_t)nb_old * 2 + 1);
    for (int64_t b = 0; b < nb_old; ++b) rank.emplace(((uint64_t)(uint32_t)on[(size_t)b] << 32) | (uint32_t)os[(size_t)b], b);
    int64_t last_rank = -1;
    uint64_t last_unknown = 0;
    bool have_unknown = false;
    std::unordered_map<uint64_t, char> seen;
    seen.reserve((size_t)n_blocks * 2 + 1);
    for (int64_t b = 0; b < n_blocks; ++b) {
      if (fc[(size_t)blk_node[b]] >= 0)
        return octl_set_error(ctx, OCTL_E_INVALID, "block %lld: node %d is not a leaf of the scheme", (long long)b,
                              blk_node[b]);
      const uint64_t key = ((uint64_t)(uint32_t)blk_node[b] << 32) | (uint32_t)blk_slot[b];
      if (!seen.emplace(key, 1).second)
        return octl_set_error(ctx, OCTL_E_INVALID, "block %lld: (leaf %d, pose slot %d) appears twice", (long long)b,
                              blk_node[b], blk_slot[b]);
      const auto it = rank.find(key);
      if (it != rank.end()) {
        if (it->second <= last_rank)
          return octl_set_error(ctx, OCTL_E_INVALID,
                                "block %lld: (leaf %d, pose slot %d) is out of the storage order of the block table",
                                (long long)b, blk_node[b], blk_slot[b]);
        last_rank = it->second;
        have_unknown = false;
      } else {
        if (have_unknown && key <= last_unknown)
          return octl_set_error(ctx, OCTL_E_INVALID,
                                "block %lld: (leaf %d, pose slot %d) is out of order", (long long)b, blk_node[b], blk_slot[b]);
        last_unknown = key;
        have_unknown = true;
      }
    }
  }
  std::vector<int64_t> off((size_t)n_poses + 1, 0);
  for (int p = 0; p < n_poses; ++p) off[(size_t)p + 1] = off[(size_t)p] + per_slot[(size_t)p];
  std::vector<int64_t> cursor(off.begin(), off.end() - 1);
  std::vector<double> store((size_t)std::max<int64_t>(total, 1) * 3);
  std::vector<uint32_t> ord((size_t)std::max<int64_t>(total, 1)), starts((size_t)std::max<int64_t>(n_blocks, 1));
  int64_t pos = 0;
  for (int64_t b = 0; b < n_blocks; ++b) {
    starts[(size_t)b] = (uint32_t)pos;
    int64_t& c = cursor[(size_t)blk_slot[b]];
    std::memcpy(store.data() + 3 * c, xyz + 3 * pos, (size_t)blk_size[b] * 24);
    for (int32_t i = 0; i < blk_size[b]; ++i) ord[(size_t)(pos + i)] = (uint32_t)(c + i);
    c += blk_size[b];
    pos += blk_size[b];
  }
  // rows outside the cube of their leaf (the reference's test: floor((p - corner) / (edge / 2)) in {0, 1} per axis,
  // octree.py:73-75,94-98 - i.e. 0 <= p - corner < edge with the rounded difference)
  bool displaced = false;
  if (total > 0) {
    NodeTable& t = f->nodes[f->cur];
    std::vector<double> corner((size_t)n_nodes * 3), edge((size_t)n_nodes);
    HIP_TRY(ctx, hipMemcpyAsync(corner.data(), t.corner.p, (size_t)n_nodes * 24, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(edge.data(), t.edge.p, (size_t)n_nodes * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    int64_t q = 0;
    for (int64_t b = 0; b < n_blocks && !displaced; ++b) {
      const double* c = corner.data() + 3 * (size_t)blk_node[b];
      const double e = edge[(size_t)blk_node[b]];
      for (int32_t i = 0; i < blk_size[b] && !displaced; ++i, ++q) {
        for (int a = 0; a < 3; ++a) {
          const double d = xyz[3 * q + a] - c[a];
          if (!(d >= 0.0 && d < e)) displaced = true;  // (also NaN)
        }
      }
      if (displaced) break;
    }
  }
  // The new store is the forest's own.  While the old one is BORROWED (the caller's buffer, never written) the
  // new points go to the forest's own block and the borrow ends only at the commit below; every allocation
  // comes before the first byte is overwritten, so that a failed allocation leaves the forest as it was.
  DevBuf& own = f->store_borrowed ? f->xyz_own : f->xyz;
  const size_t n1 = (size_t)std::max<int64_t>(total, 1), nb1 = (size_t)std::max<int64_t>(n_blocks, 1);
  OCTL_TRY(devbuf_reserve(ctx, own, n1 * 24 + 16));
  OCTL_TRY(devbuf_reserve(ctx, f->alive, n1 + 2));
  OCTL_TRY(devbuf_reserve(ctx, f->ord_idx, n1 * 4));
  OCTL_TRY(devbuf_reserve(ctx, f->xyz_ord, n1 * 24));
  // (block buffers keep the capacity convention of forest_make_blocks: one block per point)
  for (DevBuf* b : {&f->blk_node, &f->blk_slot, &f->blk_start, &f->blk_size})
    OCTL_TRY(devbuf_reserve(ctx, *b, std::max(n1, nb1) * 4));
  auto upload = [&]() -> hipError_t {
    hipError_t e = hipSuccess;
    auto up = [&](void* dst, const void* src, size_t bytes) {
      if (e == hipSuccess && bytes > 0) e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
    };
    up(own.p, store.data(), (size_t)total * 24);
    up(f->xyz_ord.p, xyz, (size_t)total * 24);
    up(f->ord_idx.p, ord.data(), (size_t)total * 4);
    if (e == hipSuccess && total > 0) e = hipMemsetAsync(f->alive.p, 1, (size_t)total, st);
    up(f->blk_node.p, blk_node, (size_t)n_blocks * 4);
    up(f->blk_slot.p, blk_slot, (size_t)n_blocks * 4);
    up(f->blk_size.p, blk_size, (size_t)n_blocks * 4);
    up(f->blk_start.p, starts.data(), (size_t)n_blocks * 4);
    if (e == hipSuccess) e = hipStreamSynchronize(st);  // (pageable sources)
    return e;
  };
  if (const hipError_t e = upload(); e != hipSuccess) {
    // the tables (and an own store) are partly overwritten: the forest is left EMPTY but valid rather than
    // with sizes that describe arrays which no longer hold them
    (void)octl_forest_clear(f);
    return octl_set_error(ctx, OCTL_E_HIP, "set_contents: upload failed (%s); the forest was cleared",
                          hipGetErrorString(e));
  }
  // ---- commit ------------------------------------------------------------------------------------------------
  if (f->store_borrowed) {
    f->xyz = f->xyz_own;
    f->xyz_own = DevBuf{};
    f->store_borrowed = false;
  }
  f->pose_off = off;
  f->n_store = f->n_alive = f->n_ord = total;
  f->n_blocks = n_blocks;
  f->built_store = total;
  f->built_poses = n_poses;
  f->append_only = true;
  f->store_dirty = false;
  f->mask_valid = false;
  f->fast_order_valid = false;
  // the voxel box of the new points is not known (rows may have left their cubes): the next build finds it
  f->max_block_hint = INT64_MAX;
  f->bbox_stale = true;
  f->alive_stale = false;   // (written above)
  f->bbox_pending = total > 0;
  f->displaced_rows = displaced;
  return OCTL_OK;
}

int octl_forest_get_nodes(octl_forest* f, int64_t cap, int32_t* voxel, int32_t* depth,
                          int32_t* parent, int32_t* first_child, double* corner, double* edge,
                          int32_t* epoch, int64_t* n_nodes) {
  if (!f || !n_nodes) return OCTL_E_INVALID;
  OCTL_TRY(forest_settle(f));
  octl_ctx* ctx = f->ctx;
  if (!f->built) return octl_set_error(ctx, OCTL_E_STATE, "no scheme has been built");
  NodeTable& t = f->nodes[f->cur];
  *n_nodes = t.n;
  const int64_t n = std::min<int64_t>(cap, t.n);
  if (n <= 0) return OCTL_OK;
  hipStream_t st = ctx->stream;
  auto dl = [&](void* dst, const DevBuf& src, size_t bytes) {
    return dst ? hipMemcpyAsync(dst, src.p, bytes, hipMemcpyDeviceToHost, st) : hipSuccess;
  };
  HIP_TRY(ctx, dl(voxel, t.voxel, (size_t)n * 4));
  HIP_TRY(ctx, dl(depth, t.depth, (size_t)n * 4));
  HIP_TRY(ctx, dl(parent, t.parent, (size_t)n * 4));
  HIP_TRY(ctx, dl(first_child, t.first_child, (size_t)n * 4));
  HIP_TRY(ctx, dl(corner, t.corner, (size_t)n * 24));
  HIP_TRY(ctx, dl(edge, t.edge, (size_t)n * 8));
  HIP_TRY(ctx, dl(epoch, t.epoch, (size_t)n * 4));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  return OCTL_OK;
}

int octl_forest_get_voxels(octl_forest* f, int64_t cap, int64_t* coords, int64_t* n_voxels) {
  if (!f || !n_voxels) return OCTL_E_INVALID;
  OCTL_TRY(forest_settle(f));
  if (!f->built) return octl_set_error(f->ctx, OCTL_E_STATE, "no scheme has been built");
  *n_voxels = f->n_voxels;
  if (!coords) return OCTL_OK;
  OCTL_TRY(forest_sync_vkeys(f));
  const int64_t n = std::min<int64_t>(cap, *n_voxels);
  for (int64_t v = 0; v < n; ++v) {
    int64_t q[3];
    vkey_decode(f->vkeys[v], f->vorg, q);
    // the reference's voxel coordinates are int(q * L): the corner, not the index
    for (int a = 0; a < 3; ++a)
      coords[3 * v + a] = f->mode == 0 ? (int64_t)((double)q[a] * f->edge) : 0;
  }
  return OCTL_OK;
}

int octl_forest_get_blocks(octl_forest* f, int64_t cap, int32_t* node, int32_t* slot,
                           int64_t* start, int32_t* size, int64_t* n_blocks) {
  if (!f || !n_blocks) return OCTL_E_INVALID;
  OCTL_TRY(forest_settle(f));
  octl_ctx* ctx = f->ctx;
  if (!f->built) return octl_set_error(ctx, OCTL_E_STATE, "no scheme has been built");
  *n_blocks = f->n_blocks;
  const int64_t n = std::min<int64_t>(cap, f->n_blocks);
  if (n <= 0) return OCTL_OK;
  hipStream_t st = ctx->stream;
  if (node) HIP_TRY(ctx, hipMemcpyAsync(node, f->blk_node.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  if (slot) HIP_TRY(ctx, hipMemcpyAsync(slot, f->blk_slot.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  if (size) HIP_TRY(ctx, hipMemcpyAsync(size, f->blk_size.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  if (start) {
    OCTL_TRY(devbuf_reserve(ctx, f->rs_scratch, (size_t)n * 8));
    OCTL_LAUNCH(k_widen_u32_i64, dim3(grid_for(n)), dim3(256), 0, st,
                       (const uint32_t*)f->blk_start.as<uint32_t>(), n,
                       f->rs_scratch.as<int64_t>());
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(start, f->rs_scratch.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
  }
  HIP_TRY(ctx, hipStreamSynchronize(st));
  return OCTL_OK;
}

int octl_forest_get_slot_voxels(octl_forest* f, int32_t slot, int64_t cap, int32_t* voxel_ranks,
                                int64_t* n_out) {
  if (!f || !n_out) return OCTL_E_INVALID;
  OCTL_TRY(forest_settle(f));
  octl_ctx* ctx = f->ctx;
  if (!f->built) return octl_set_error(ctx, OCTL_E_STATE, "no scheme has been built");
  const int n_poses = (int)f->pose_off.size() - 1;
  if (slot < 0 || slot >= n_poses) return octl_set_error(ctx, OCTL_E_INVALID, "bad pose slot");
  *n_out = 0;
  const int64_t V = f->n_voxels, nb = f->n_blocks;
  if (V <= 0 || nb <= 0) return OCTL_OK;
  hipStream_t st = ctx->stream;
  // scratch inside rs_scratch: [flags u32 V+8 | scanned u32 V+8 | out i32 V]
  const size_t seg = (((size_t)V + 8) * 4 + 15) & ~(size_t)15;
  OCTL_TRY(devbuf_reserve(ctx, f->rs_scratch, 3 * seg));
  char* base = static_cast<char*>(f->rs_scratch.p);
  uint32_t* flags = reinterpret_cast<uint32_t*>(base);
  uint32_t* scanned = reinterpret_cast<uint32_t*>(base + seg);
  int32_t* out = reinterpret_cast<int32_t*>(base + 2 * seg);
  uint32_t* total = ctx->small.as<uint32_t>() + 21;
  HIP_TRY(ctx, hipMemsetAsync(flags, 0, seg, st));
  OCTL_LAUNCH(k_slot_voxel_flags, dim3(grid_for(nb)), dim3(256), 0, st,
                     (const int32_t*)f->blk_node.as<int32_t>(), (const int32_t*)f->blk_slot.as<int32_t>(),
                     nb, slot, (const int32_t*)f->nodes[f->cur].voxel.as<int32_t>(), flags);
  HIP_TRY(ctx, hipGetLastError());
  OCTL_TRY(octl_exclusive_scan_u32(ctx, flags, scanned, V, total));
  OCTL_LAUNCH(k_flag_indices, dim3(grid_for(V)), dim3(256), 0, st, (const uint32_t*)flags,
                     (const uint32_t*)scanned, V, out);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(ctx->small_host, total, 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  uint32_t cnt;
  std::memcpy(&cnt, ctx->small_host, 4);
  *n_out = cnt;
  const int64_t m = std::min<int64_t>(cap, cnt);
  if (m > 0 && voxel_ranks) {
    HIP_TRY(ctx, hipMemcpyAsync(voxel_ranks, out, (size_t)m * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
  }
  return OCTL_OK;
}

int octl_forest_slot_counts(octl_forest* f, int32_t slot, int64_t* n_points, int64_t* n_leaves) {
  if (!f || !n_points || !n_leaves) return OCTL_E_INVALID;
  OCTL_TRY(forest_settle(f));
  octl_ctx* ctx = f->ctx;
  if (!f->built) return octl_set_error(ctx, OCTL_E_STATE, "no scheme has been built");
  const int n_poses = (int)f->pose_off.size() - 1;
  if (slot < 0 || slot >= n_poses) return octl_set_error(ctx, OCTL_E_INVALID, "bad pose slot");
  *n_points = *n_leaves = 0;
  if (f->n_blocks <= 0) return OCTL_OK;
  hipStream_t st = ctx->stream;
  unsigned long long* out = reinterpret_cast<unsigned long long*>(ctx->small.as<uint32_t>() + 28);
  HIP_TRY(ctx, hipMemsetAsync(out, 0, 16, st));
  OCTL_LAUNCH(k_slot_counts, dim3((unsigned)std::min<int64_t>(1024, ceil_div(f->n_blocks, 256))), dim3(256),
                     0, st, (const int32_t*)f->blk_slot.as<int32_t>(), (const int32_t*)f->blk_size.as<int32_t>(),
                     f->n_blocks, slot, out);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(ctx->small_host, out, 16, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  unsigned long long r[2];
  std::memcpy(r, ctx->small_host, 16);
  *n_points = (int64_t)r[0];
  *n_leaves = (int64_t)r[1];
  return OCTL_OK;
}

int octl_forest_internal_per_voxel(octl_forest* f, int64_t cap, int32_t* counts, int64_t* n_voxels) {
  if (!f || !n_voxels) return OCTL_E_INVALID;
  OCTL_TRY(forest_settle(f));
  octl_ctx* ctx = f->ctx;
  if (!f->built) return octl_set_error(ctx, OCTL_E_STATE, "no scheme has been built");
  *n_voxels = f->n_voxels;
  const int64_t V = std::min<int64_t>(cap, f->n_voxels);
  if (V <= 0 || !counts) return OCTL_OK;
  hipStream_t st = ctx->stream;
  NodeTable& t = f->nodes[f->cur];
  OCTL_TRY(devbuf_reserve(ctx, f->rs_scratch, (size_t)f->n_voxels * 4));
  HIP_TRY(ctx, hipMemsetAsync(f->rs_scratch.p, 0, (size_t)f->n_voxels * 4, st));
  OCTL_LAUNCH(k_internal_per_voxel, dim3(grid_for(t.n)), dim3(256), 0, st,
                     (const int32_t*)t.first_child.as<int32_t>(), (const int32_t*)t.voxel.as<int32_t>(), t.n,
                     f->rs_scratch.as<int32_t>());
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(counts, f->rs_scratch.p, (size_t)V * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  return OCTL_OK;
}

int octl_forest_get_perm(octl_forest* f, int64_t cap, int64_t* perm, int64_t* n_out) {
  if (!f || !n_out) return OCTL_E_INVALID;
  OCTL_TRY(forest_settle(f));
  octl_ctx* ctx = f->ctx;
  if (!f->built) return octl_set_error(ctx, OCTL_E_STATE, "no scheme has been built");
  *n_out = f->n_ord;
  const int64_t n = std::min<int64_t>(cap, f->n_ord);
  if (n <= 0 || !perm) return OCTL_OK;
  hipStream_t st = ctx->stream;
  OCTL_TRY(devbuf_reserve(ctx, f->rs_scratch, (size_t)n * 8));
  OCTL_LAUNCH(k_widen_u32_i64, dim3(grid_for(n)), dim3(256), 0, st,
                     (const uint32_t*)f->ord_idx.as<uint32_t>(), n, f->rs_scratch.as<int64_t>());
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(perm, f->rs_scratch.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  return OCTL_OK;
}

int octl_forest_get_points(octl_forest* f, int64_t start, int64_t count, double* xyz) {
  if (!f) return OCTL_E_INVALID;
  OCTL_TRY(forest_settle(f));
  octl_ctx* ctx = f->ctx;
  if (!f->built) return octl_set_error(ctx, OCTL_E_STATE, "no scheme has been built");
  if (start < 0 || count < 0 || start + count > f->n_ord)
    return octl_set_error(ctx, OCTL_E_INVALID, "point range out of bounds");
  if (count == 0) return OCTL_OK;
  if (!xyz) return OCTL_E_INVALID;
  HIP_TRY(ctx, hipMemcpyAsync(xyz, f->xyz_ord.as<double>() + 3 * start, (size_t)count * 24,
                              hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return OCTL_OK;
}

int octl_forest_gather_blocks(octl_forest* f, const int32_t* block_ids, int64_t m, int64_t cap, double* xyz,
                              int64_t* n_points) {
  if (!f || !n_points || m < 0 || (m > 0 && !block_ids)) return OCTL_E_INVALID;
  OCTL_TRY(forest_settle(f));
  octl_ctx* ctx = f->ctx;
  if (!f->built) return octl_set_error(ctx, OCTL_E_STATE, "no scheme has been built");
  *n_points = 0;
  if (m == 0) return OCTL_OK;
  for (int64_t i = 0; i < m; ++i)
    if (block_ids[i] < 0 || block_ids[i] >= f->n_blocks) return octl_set_error(ctx, OCTL_E_INVALID, "bad block id");
  hipStream_t st = ctx->stream;
  // scratch (f->hist): [ids i32 m | sizes -> offsets u32 m + 1 | total]
  const size_t o_sz = (((size_t)m * 4) + 15) & ~(size_t)15;
  const size_t o_tot = o_sz + ((((size_t)m + 1) * 4) + 15) / 16 * 16;
  OCTL_TRY(devbuf_reserve(ctx, f->hist, o_tot + 16));
  char* base = static_cast<char*>(f->hist.p);
  int32_t* ids_d = reinterpret_cast<int32_t*>(base);
  uint32_t* offs = reinterpret_cast<uint32_t*>(base + o_sz);
  uint32_t* total = reinterpret_cast<uint32_t*>(base + o_tot);
  HIP_TRY(ctx, hipMemcpyAsync(ids_d, block_ids, (size_t)m * 4, hipMemcpyHostToDevice, st));
  OCTL_LAUNCH(k_gather_sizes, dim3(grid_for(m)), dim3(256), 0, st, (const int32_t*)ids_d, m,
                     (const int32_t*)f->blk_size.as<int32_t>(), offs);
  HIP_TRY(ctx, hipGetLastError());
  OCTL_TRY(octl_exclusive_scan_u32(ctx, offs, offs, m, total));
  uint32_t tot_h = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&tot_h, total, 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));  // (also: the pageable id list is the caller's again)
  *n_points = tot_h;
  if (tot_h == 0 || !xyz || cap < (int64_t)tot_h) return OCTL_OK;  // (size query, or nothing to copy)
  // the gathered rows go through the compaction target of apply_mask (free between calls)
  OCTL_TRY(devbuf_reserve(ctx, f->xyz_ord2, (size_t)tot_h * 24));
  OCTL_LAUNCH(k_gather_rows, dim3((unsigned)ceil_div(m, 4)), dim3(256), 0, st, (const int32_t*)ids_d, m,
                     (const uint32_t*)f->blk_start.as<uint32_t>(), (const int32_t*)f->blk_size.as<int32_t>(),
                     (const uint32_t*)offs, (const double*)f->xyz_ord.as<double>(), f->xyz_ord2.as<double>());
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(xyz, f->xyz_ord2.p, (size_t)tot_h * 24, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  return OCTL_OK;
}

int octl_forest_ransac(octl_forest* f, const int32_t* block_order, int64_t nb,
                       const double* hypotheses, int32_t H, int32_t k, double threshold,
                       float* plane, int32_t* best_count, int32_t* best_index) {
  if (!f) return OCTL_E_INVALID;
  OCTL_TRY(forest_settle(f));
  octl_ctx* ctx = f->ctx;
  if (!f->built) return octl_set_error(ctx, OCTL_E_STATE, "ransac before build");
  if (nb < 0 || (nb > 0 && !block_order) || !hypotheses)
    return octl_set_error(ctx, OCTL_E_INVALID, "bad ransac arguments");
  if (H < 1 || H > 1024 || k < 1) return octl_set_error(ctx, OCTL_E_INVALID, "bad H or k");
  OCTL_TRY(ransac_check_table(ctx, hypotheses, H, k));
  for (int64_t b = 0; b < nb; ++b)
    if (block_order[b] < 0 || block_order[b] >= f->n_blocks)
      return octl_set_error(ctx, OCTL_E_INVALID, "block index out of range");
  hipStream_t st = ctx->stream;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  OCTL_TRY(ensure_mask(f));
  if (nb == 0) return OCTL_OK;
  OCTL_TRY(devbuf_reserve(ctx, f->rs_order, (size_t)nb * 4));
  OCTL_TRY(devbuf_reserve(ctx, f->rs_hyp, (size_t)H * k * 8));
  HIP_TRY(ctx, hipMemcpyAsync(f->rs_order.p, block_order, (size_t)nb * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(f->rs_hyp.p, hypotheses, (size_t)H * k * 8, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  float* plane_d = nullptr;
  int32_t *count_d = nullptr, *index_d = nullptr;
  if (plane) {
    OCTL_TRY(devbuf_reserve(ctx, f->rs_plane, (size_t)nb * 16));
    plane_d = f->rs_plane.as<float>();
  }
  if (best_count) {
    OCTL_TRY(devbuf_reserve(ctx, f->rs_count, (size_t)nb * 4));
    count_d = f->rs_count.as<int32_t>();
  }
  if (best_index) {
    OCTL_TRY(devbuf_reserve(ctx, f->rs_index, (size_t)nb * 4));
    index_d = f->rs_index.as<int32_t>();
  }
  OCTL_TRY(ransac_launch(ctx, f->xyz_ord.as<double>(), f->n_ord, f->blk_start.as<uint32_t>(),
                         f->blk_size.as<int32_t>(), f->rs_order.as<int32_t>(), nb,
                         f->rs_hyp.as<double>(), H, k, threshold, f->mask.as<uint8_t>(), plane_d,
                         count_d, index_d, nullptr, f->rs_scratch, f->max_block_hint));
  if (plane) HIP_TRY(ctx, hipMemcpyAsync(plane, plane_d, (size_t)nb * 16, hipMemcpyDeviceToHost, st));
  if (best_count) HIP_TRY(ctx, hipMemcpyAsync(best_count, count_d, (size_t)nb * 4, hipMemcpyDeviceToHost, st));
  if (best_index) HIP_TRY(ctx, hipMemcpyAsync(best_index, index_d, (size_t)nb * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  return OCTL_OK;
}

int octl_forest_get_mask(octl_forest* f, int64_t cap, uint8_t* mask, int64_t* n_out) {
  if (!f || !n_out) return OCTL_E_INVALID;
  OCTL_TRY(forest_settle(f));
  octl_ctx* ctx = f->ctx;
  if (!f->built) return octl_set_error(ctx, OCTL_E_STATE, "no scheme has been built");
  *n_out = f->n_ord;
  const int64_t n = std::min<int64_t>(cap, f->n_ord);
  if (n <= 0 || !mask) return OCTL_OK;
  OCTL_TRY(ensure_mask(f));
  HIP_TRY(ctx, hipMemcpyAsync(mask, f->mask.p, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return OCTL_OK;
}

int octl_forest_apply_mask(octl_forest* f, int64_t* n_alive) {
  if (!f) return OCTL_E_INVALID;
  OCTL_TRY(forest_settle(f));
  if (!f->built) return octl_set_error(f->ctx, OCTL_E_STATE, "apply_mask before build");
  OCTL_TRY(ensure_mask(f));
  return apply_device_mask(f, n_alive);
}

int octl_forest_apply_mask_async(octl_forest* f) {
  if (!f) return OCTL_E_INVALID;
  OCTL_TRY(forest_settle(f));
  if (!f->built) return octl_set_error(f->ctx, OCTL_E_STATE, "apply_mask before build");
  OCTL_TRY(ensure_mask(f));
  return apply_device_mask(f, nullptr, true);
}

int octl_forest_settle(octl_forest* f, int64_t* n_alive) {
  if (!f) return OCTL_E_INVALID;
  OCTL_TRY(forest_settle(f));
  if (n_alive) *n_alive = f->n_ord;
  return OCTL_OK;
}

int octl_forest_filter_count(octl_forest* f, const uint8_t* slot_sel, int32_t n_sel, int64_t lo,
                             int64_t hi, int64_t* n_alive) {
  if (!f || !slot_sel) return OCTL_E_INVALID;
  OCTL_TRY(forest_settle(f));
  octl_ctx* ctx = f->ctx;
  if (!f->built) return octl_set_error(ctx, OCTL_E_STATE, "filter before build");
  const int n_poses = (int)f->pose_off.size() - 1;
  if (n_sel != n_poses) return octl_set_error(ctx, OCTL_E_INVALID, "slot selection has %d entries for %d poses", n_sel, n_poses);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  OCTL_TRY(ensure_mask(f));
  if (f->n_blocks > 0) {
    OCTL_TRY(devbuf_reserve(ctx, f->scheme_dev, (size_t)std::max(n_poses, 1)));
    HIP_TRY(ctx, hipMemcpyAsync(f->scheme_dev.p, slot_sel, (size_t)n_poses, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));  // (a pageable source)
    KTimer t(ctx, "filter");
    OCTL_LAUNCH(k_filter_blocks, dim3((unsigned)ceil_div(f->n_blocks, 4)), dim3(256), 0, st,
                       (const uint32_t*)f->blk_start.as<uint32_t>(), (const int32_t*)f->blk_size.as<int32_t>(),
                       (const int32_t*)f->blk_slot.as<int32_t>(), f->n_blocks,
                       (const uint8_t*)f->scheme_dev.as<uint8_t>(), lo, hi, f->mask.as<uint8_t>());
    HIP_TRY(ctx, hipGetLastError());
  }
  return apply_device_mask(f, n_alive);
}

int octl_forest_apply_host_mask(octl_forest* f, const uint8_t* mask, int64_t n, int64_t* n_alive) {
  if (!f) return OCTL_E_INVALID;
  OCTL_TRY(forest_settle(f));
  octl_ctx* ctx = f->ctx;
  if (!f->built) return octl_set_error(ctx, OCTL_E_STATE, "apply_mask before build");
  if (n != f->n_ord || (n > 0 && !mask))
    return octl_set_error(ctx, OCTL_E_INVALID, "mask has %lld entries for %lld points",
                          (long long)n, (long long)f->n_ord);
  OCTL_TRY(devbuf_reserve(ctx, f->mask, (size_t)std::max<int64_t>(n, 1)));
  if (n > 0) {
    HIP_TRY(ctx, hipMemcpyAsync(f->mask.p, mask, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  f->mask_valid = true;
  return apply_device_mask(f, n_alive);
}

int octl_ransac_evaluate(octl_ctx* ctx, const double* point_cloud, int64_t M,
                         const int32_t* block_sizes, int64_t B, const double* hypotheses,
                         int32_t H, int32_t k, double threshold, uint8_t* mask_out,
                         float* planes_out, int32_t* best_count_out, int32_t* best_index_out) {
  if (!ctx) return OCTL_E_INVALID;
  if (M < 0 || B < 0 || (M > 0 && !point_cloud) || (B > 0 && !block_sizes) || !hypotheses ||
      (M > 0 && !mask_out))
    return octl_set_error(ctx, OCTL_E_INVALID, "bad ransac_evaluate arguments");
  if (H < 1 || H > 1024 || k < 1) return octl_set_error(ctx, OCTL_E_INVALID, "bad H or k");
  OCTL_TRY(ransac_check_table(ctx, hypotheses, H, k));
  if (M >= ((int64_t)1 << 31)) return octl_set_error(ctx, OCTL_E_INVALID, "cloud too large");
  int64_t total = 0;
  std::vector<uint32_t> starts((size_t)std::max<int64_t>(B, 1));
  for (int64_t b = 0; b < B; ++b) {
    if (block_sizes[b] < 0) return octl_set_error(ctx, OCTL_E_INVALID, "negative block size");
    starts[b] = (uint32_t)total;  // np.cumsum([0] + sizes[:-1]) (cuda_ransac.py:64-66)
    total += block_sizes[b];
  }
  if (total > M)
    return octl_set_error(ctx, OCTL_E_INVALID, "block sizes add up to %lld > %lld points",
                          (long long)total, (long long)M);
  if (M == 0) return OCTL_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  DevBuf xyz, sizes, starts_d, hyp, mask, plane, count, index, scratch;
  int rc = OCTL_OK;
  auto cleanup = [&]() {
    // (back to the context's pool, not to the allocator: hipFree synchronises the device, and the next
    //  evaluate() of a loop over batches wants the same blocks again)
    (void)hipStreamSynchronize(st);
    for (DevBuf* b : {&xyz, &sizes, &starts_d, &hyp, &mask, &plane, &count, &index, &scratch})
      devbuf_release(ctx, *b);
  };
#define EV_TRY(expr)            \
  do {                          \
    rc = (expr);                \
    if (rc != OCTL_OK) {        \
      cleanup();                \
      return rc;                \
    }                           \
  } while (0)
#define EV_HIP(expr)                                                                        \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      cleanup();                                                                            \
      return octl_set_error(ctx, OCTL_E_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    }                                                                                       \
  } while (0)
  EV_TRY(devbuf_reserve(ctx, xyz, (size_t)M * 24));
  EV_TRY(devbuf_reserve(ctx, mask, (size_t)M));
  EV_TRY(devbuf_reserve(ctx, hyp, (size_t)H * k * 8));
  EV_HIP(hipMemcpyAsync(xyz.p, point_cloud, (size_t)M * 24, hipMemcpyHostToDevice, st));
  EV_HIP(hipMemcpyAsync(hyp.p, hypotheses, (size_t)H * k * 8, hipMemcpyHostToDevice, st));
  EV_HIP(hipMemsetAsync(mask.p, 0, (size_t)M, st));  // np.zeros (cuda_ransac.py:57)
  if (B > 0) {
    EV_TRY(devbuf_reserve(ctx, sizes, (size_t)B * 4));
    EV_TRY(devbuf_reserve(ctx, starts_d, (size_t)B * 4));
    EV_HIP(hipMemcpyAsync(sizes.p, block_sizes, (size_t)B * 4, hipMemcpyHostToDevice, st));
    EV_HIP(hipMemcpyAsync(starts_d.p, starts.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
    if (planes_out) EV_TRY(devbuf_reserve(ctx, plane, (size_t)B * 16));
    if (best_count_out) EV_TRY(devbuf_reserve(ctx, count, (size_t)B * 4));
    if (best_index_out) EV_TRY(devbuf_reserve(ctx, index, (size_t)B * 4));
    EV_HIP(hipStreamSynchronize(st));
    int64_t max_size = 0;   // (the caller's block sizes are host data: the largest one is known)
    for (int64_t b = 0; b < B; ++b) max_size = std::max<int64_t>(max_size, block_sizes[b]);
    EV_TRY(ransac_launch(ctx, xyz.as<double>(), M, starts_d.as<uint32_t>(), sizes.as<int32_t>(),
                         nullptr, B, hyp.as<double>(), H, k, threshold, mask.as<uint8_t>(),
                         planes_out ? plane.as<float>() : nullptr,
                         best_count_out ? count.as<int32_t>() : nullptr,
                         best_index_out ? index.as<int32_t>() : nullptr, nullptr, scratch, max_size));
    if (planes_out) EV_HIP(hipMemcpyAsync(planes_out, plane.p, (size_t)B * 16, hipMemcpyDeviceToHost, st));
    if (best_count_out) EV_HIP(hipMemcpyAsync(best_count_out, count.p, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    if (best_index_out) EV_HIP(hipMemcpyAsync(best_index_out, index.p, (size_t)B * 4, hipMemcpyDeviceToHost, st));
  }
  EV_HIP(hipMemcpyAsync(mask_out, mask.p, (size_t)M, hipMemcpyDeviceToHost, st));
  EV_HIP(hipStreamSynchronize(st));
  cleanup();
#undef EV_TRY
#undef EV_HIP
  return OCTL_OK;
}

}  // extern "C"

// ---- test hooks for the device-wide primitives (tests/test_gpu_primitives.py) ----------------
extern "C" int octl_debug_exclusive_scan(octl_ctx* ctx, const uint32_t* in, int64_t n,
                                         uint32_t* out, uint32_t* total) {
  if (!ctx || n < 0 || (n > 0 && (!in || !out))) return OCTL_E_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  DevBuf buf;
  OCTL_TRY(devbuf_reserve(ctx, buf, (size_t)(n + 8) * 4));
  hipStream_t st = ctx->stream;
  uint32_t* tot_d = ctx->small.as<uint32_t>() + 24;
  int rc = OCTL_OK;
  if (n > 0 && hipMemcpyAsync(buf.p, in, (size_t)n * 4, hipMemcpyHostToDevice, st) != hipSuccess)
    rc = OCTL_E_HIP;
  if (rc == OCTL_OK) rc = octl_exclusive_scan_u32(ctx, buf.as<uint32_t>(), buf.as<uint32_t>(), n, tot_d);
  if (rc == OCTL_OK && n > 0 &&
      hipMemcpyAsync(out, buf.p, (size_t)n * 4, hipMemcpyDeviceToHost, st) != hipSuccess)
    rc = OCTL_E_HIP;
  if (rc == OCTL_OK && total &&
      hipMemcpyAsync(total, tot_d, 4, hipMemcpyDeviceToHost, st) != hipSuccess)
    rc = OCTL_E_HIP;
  if (hipStreamSynchronize(st) != hipSuccess) rc = OCTL_E_HIP;
  devbuf_free(buf);
  return rc;
}

extern "C" int octl_debug_radix_sort(octl_ctx* ctx, uint64_t* keys, uint32_t* vals, int64_t n,
                                     int key_bits) {
  if (!ctx || n < 0 || (n > 0 && (!keys || !vals))) return OCTL_E_INVALID;
  if (n == 0) return OCTL_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  DevBuf k[2], v[2], hist;
  int rc = OCTL_OK;
  for (int b = 0; b < 2 && rc == OCTL_OK; ++b) {
    rc = devbuf_reserve(ctx, k[b], (size_t)n * 8);
    if (rc == OCTL_OK) rc = devbuf_reserve(ctx, v[b], (size_t)n * 4);
  }
  if (rc == OCTL_OK) {
    if (hipMemcpyAsync(k[0].p, keys, (size_t)n * 8, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(v[0].p, vals, (size_t)n * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
      rc = OCTL_E_HIP;
  }
  int res = 0;
  if (rc == OCTL_OK) {
    uint64_t* kk[2] = {k[0].as<uint64_t>(), k[1].as<uint64_t>()};
    uint32_t* vv[2] = {v[0].as<uint32_t>(), v[1].as<uint32_t>()};
    rc = octl_radix_sort_u64_u32(ctx, kk, vv, n, key_bits, hist, &res);
  }
  if (rc == OCTL_OK) {
    if (hipMemcpyAsync(keys, k[res].p, (size_t)n * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(vals, v[res].p, (size_t)n * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
      rc = OCTL_E_HIP;
  }
  for (int b = 0; b < 2; ++b) {
    devbuf_free(k[b]);
    devbuf_free(v[b]);
  }
  devbuf_free(hist);
  return rc;
}
