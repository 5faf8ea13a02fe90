// Bucket build: insert + subdivide of a fresh forest with ONE move of the coordinates.
//
// Replaces, for the count criterion (paths relative to /root/reference):
//   Grid.insert_points          grid/grid.py:58-109      top-level voxel bucketing
//   OctreeManager.subdivide     octree_manager.py:36-66  scheme from the union of the scheme poses
//   OctreeNode.subdivide        octree/octree.py:20-32   recursive 8-way split while count > K
//   OctreeNode.insert_points    octree/octree.py:67-100  child index arithmetic
//   OctreeNode._generate_children octree/octree.py:177-191
//
// The reference sorts the cloud by voxel (np.unique + argsort, grid.py:79-90) and then re-sorts every
// node's points at every level.  A device radix sort by voxel followed by a per-voxel gather reads the
// 24-byte points at random: 128-byte lines for 24 useful bytes (measured 147 B per point).  Here the
// points move through HBM once:
//   k_part_hist / k_part_scatter   MSD partition of (linear voxel key, index, xyz) into <= 4096
//                                  BUCKETS of consecutive voxels (one 12-bit digit, LDS histograms,
//                                  wave64 ballot ranks: stable) - the only scatter of coordinates;
//   k_bucket_build                 one workgroup per bucket (<= 4096 points, a few voxels): the bucket
//                                  is sorted in LDS by (voxel, 21-bit child-digit path), the leaf of
//                                  every point follows from segment scans over the sorted keys (a node
//                                  splits while its scheme-pose count exceeds K), a second LDS sort
//                                  restores insertion order inside the leaves, and the leaf-ordered
//                                  permutation + coordinates are written with coalesced stores;
//   k_bucket_scan                  bucket totals -> voxel / node / block numbering bases;
//   k_bucket_nodes                 one wavefront per bucket: scheme nodes in the level-major numbering
//                                  of the level-synchronous path, position -> leaf map, block table.
// Everything is HBM-bound integer / compare work; the f64 arithmetic is the reference's own
// (exact comparisons on the rounded differences it forms).  Results are bit-identical to the
// level-synchronous path of build.hip.
#include "build_common.h"
#include "ref_arith.h"
#include "wave_utils.h"

namespace {

constexpr int PT_THREADS = 256;
constexpr int PT_IPT = 16;
constexpr int PT_TILE = PT_THREADS * PT_IPT;     // 4096 items per tile
constexpr int PT_ST_TILES = 4;                   // tiles per supertile (one workgroup, one table column)
constexpr int PT_ST = PT_TILE * PT_ST_TILES;     // 16384 items
constexpr int PT_BITS = 12;
constexpr int PT_BINS = 1 << PT_BITS;            // buckets

constexpr int BB_THREADS = 256;
constexpr int BB_IPT = 16;
constexpr int BB_CAP = BB_THREADS * BB_IPT;      // points per bucket handled in LDS
constexpr int BB_LEVELS = 7;                     // child digits per point (21 bits)
constexpr uint32_t PATH_MASK = 0x1FFFFFu;

// rows of the per-bucket totals table bk_tot[row][bucket]
enum { BK_NVOX = 0, BK_NINT = 1 /* .. 7 */, BK_NBLK = 8, BK_FLAGS = 9, BK_ROWS = 10 };
// leafinfo word of a leaf-ordered point
constexpr uint32_t LI_VHEAD = 1u << 24;   // first point of its top-level voxel
constexpr uint32_t LI_BHEAD = 1u << 25;   // first point of its (leaf, pose) block
// bucket flags
constexpr uint32_t BF_OVERFLOW = 1u;      // more than BB_CAP points
constexpr uint32_t BF_DEEP = 2u;          // a node at level 7 still exceeds K
constexpr uint32_t BF_BAD = 4u;           // a point outside its cube (exact slow path needed)

struct LinParams {
  int mode;          // 0 grid, 1 single cube
  double L;          // voxel edge
  double c0x, c0y, c0z;  // cube corner (mode 1)
  int minx, miny, minz;  // voxel bounding box
  uint32_t ny, nz;
  int shift;         // bucket = lin >> shift
};

// top-level voxel of a point: floor((p - corner) / L) with corner = 0 (grid.py:72-76), as a compact
// linear key in lexicographic (x, y, z) order - the order of np.unique(axis=0) (grid.py:79-81)
__device__ __forceinline__ uint32_t lin_of(const LinParams& lp, double x, double y, double z) {
  if (lp.mode != 0) return 0u;
  const int qx = (int)floor_div_exact(x, lp.L), qy = (int)floor_div_exact(y, lp.L),
            qz = (int)floor_div_exact(z, lp.L);
  return ((uint32_t)(qx - lp.minx) * lp.ny + (uint32_t)(qy - lp.miny)) * lp.nz + (uint32_t)(qz - lp.minz);
}

// ---------------------------------------------------------------------------------------------
// partition
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PT_THREADS) void k_part_hist(const double* __restrict__ xyz,
                                                          const uint8_t* __restrict__ alive, int64_t N,
                                                          LinParams lp, uint32_t nst, uint32_t nd,
                                                          uint32_t* __restrict__ table) {
  __shared__ uint32_t hist[PT_BINS];
  for (uint32_t d = threadIdx.x; d < nd; d += PT_THREADS) hist[d] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * PT_ST;
#pragma unroll 4
  for (int r = 0; r < PT_ST / PT_THREADS; ++r) {
    const int64_t i = base + (int64_t)r * PT_THREADS + threadIdx.x;  // coalesced; order is irrelevant here
    if (i < N && alive[i]) {
      const uint32_t lin = lin_of(lp, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
      atomicAdd(&hist[lin >> lp.shift], 1u);
    }
  }
  __syncthreads();
  for (uint32_t d = threadIdx.x; d < nd; d += PT_THREADS) table[(size_t)d * nst + blockIdx.x] = hist[d];
}

// stable rank inside one wave's stream with 16-bit counters: a wave's counters are touched by that
// wave only and its rounds are sequential, so the leader of a digit updates them with plain accesses
template <int BITS>
__device__ __forceinline__ uint32_t wave_rank_u16(uint32_t digit, bool valid, uint16_t* cnt) {
  const uint64_t peers = wave_match<BITS>(digit, valid);
  const uint32_t rank_in_round = __popcll(peers & lanemask_lt());
  const int leader = __ffsll((unsigned long long)peers) - 1;
  uint32_t old = 0;
  if (valid && (int)(threadIdx.x & 63u) == leader) {
    old = cnt[digit];
    cnt[digit] = (uint16_t)(old + (uint32_t)__popcll(peers));
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  old = __shfl(old, leader < 0 ? 0 : leader);
  return old + rank_in_round;
}

__global__ __launch_bounds__(PT_THREADS) void k_part_scatter(
    const double* __restrict__ xyz, const uint8_t* __restrict__ alive, int64_t N, LinParams lp,
    uint32_t nst, uint32_t nd, const uint32_t* __restrict__ table_scanned,
    const int64_t* __restrict__ pose_off, int n_poses, const uint8_t* __restrict__ scheme,
    uint32_t* __restrict__ out_lin, uint32_t* __restrict__ out_idx, double* __restrict__ out_xyz) {
  __shared__ uint32_t base[PT_BINS];                 // running destination of every bucket
  __shared__ uint16_t cnt[PT_THREADS / 64][PT_BINS]; // per wave, per tile
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (uint32_t d = threadIdx.x; d < nd; d += PT_THREADS)
    base[d] = table_scanned[(size_t)d * nst + blockIdx.x];
  for (int t = 0; t < PT_ST_TILES; ++t) {
    const int64_t tbase = (int64_t)blockIdx.x * PT_ST + (int64_t)t * PT_TILE;
    if (tbase >= N) break;
    for (uint32_t d = threadIdx.x; d < nd; d += PT_THREADS) {
#pragma unroll
      for (int w = 0; w < PT_THREADS / 64; ++w) cnt[w][d] = 0;
    }
    __syncthreads();
    // wave w owns items [tbase + w*1024, +1024) in 16 rounds of 64 consecutive items: stream order
    // == memory order, so the partition is stable
    const int64_t wbase = tbase + (int64_t)wave * (64 * PT_IPT);
    double x[PT_IPT], y[PT_IPT], z[PT_IPT];
    uint32_t lin[PT_IPT], rank[PT_IPT];
#pragma unroll
    for (int r = 0; r < PT_IPT; ++r) {
      const int64_t i = wbase + r * 64 + lane;
      const bool valid = i < N && alive[i];
      lin[r] = 0;
      x[r] = y[r] = z[r] = 0.0;
      if (valid) {
        x[r] = xyz[3 * i];
        y[r] = xyz[3 * i + 1];
        z[r] = xyz[3 * i + 2];
        lin[r] = lin_of(lp, x[r], y[r], z[r]);
      }
      rank[r] = wave_rank_u16<PT_BITS>(lin[r] >> lp.shift, valid, cnt[wave]) | (valid ? 0x80000000u : 0u);
    }
    __syncthreads();
    // per bucket: exclusive offsets of the waves inside this tile; the tile's total moves the running
    // base once every item is placed
    uint32_t tile_tot[PT_BINS / PT_THREADS];
#pragma unroll
    for (int q = 0; q < PT_BINS / PT_THREADS; ++q) {
      const uint32_t d = q * PT_THREADS + threadIdx.x;
      uint32_t run = 0;
      if (d < nd) {
#pragma unroll
        for (int w = 0; w < PT_THREADS / 64; ++w) {
          const uint32_t c = cnt[w][d];
          cnt[w][d] = (uint16_t)run;
          run += c;
        }
      }
      tile_tot[q] = run;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PT_IPT; ++r) {
      if (rank[r] >> 31) {
        const int64_t i = wbase + r * 64 + lane;
        const uint32_t d = lin[r] >> lp.shift;
        const uint32_t dst = base[d] + cnt[wave][d] + (rank[r] & 0x7FFFFFFFu);
        uint32_t v = (uint32_t)i;
        if (scheme) {
          int lo = 0, hi = n_poses;
          while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pose_off[mid] <= i) lo = mid; else hi = mid;
          }
          if (scheme[lo]) v |= 0x80000000u;
        } else {
          v |= 0x80000000u;
        }
        out_lin[dst] = lin[r];
        out_idx[dst] = v;
        out_xyz[3 * (size_t)dst] = x[r];
        out_xyz[3 * (size_t)dst + 1] = y[r];
        out_xyz[3 * (size_t)dst + 2] = z[r];
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PT_BINS / PT_THREADS; ++q) {
      const uint32_t d = q * PT_THREADS + threadIdx.x;
      if (d < nd) base[d] += tile_tot[q];
    }
    // (the next tile reads base only after the barrier that follows its counter reset)
  }
}

// ---------------------------------------------------------------------------------------------
// LDS radix sort pass (stable, 8-bit digit) over n <= BB_CAP keys, wave-striped ownership
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_inclusive_add(uint32_t v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(v, off);
    if (lane >= off) v += t;
  }
  return v;
}

// exclusive prefix of one value per thread over the 256-thread block; *total = block sum.
// scratch: >= 4 words of LDS; two barriers inside.
__device__ __forceinline__ uint32_t block_excl_add(uint32_t v, uint32_t* total, uint32_t* scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t inc = wave_inclusive_add(v);
  if (lane == 63) scratch[wave] = inc;
  __syncthreads();
  uint32_t basev = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < BB_THREADS / 64; ++w) {
    const uint32_t s = scratch[w];
    if (w < wave) basev += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return basev + inc - v;
}

template <typename T>
__device__ __forceinline__ void lds_sort_pass(const T* __restrict__ src, T* __restrict__ dst, int n,
                                              int shift, uint32_t (*cnt)[256], uint32_t* scratch) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int w = 0; w < BB_THREADS / 64; ++w) cnt[w][threadIdx.x] = 0;
  __syncthreads();
  T key[BB_IPT];
  uint32_t rank[BB_IPT];
#pragma unroll
  for (int r = 0; r < BB_IPT; ++r) {
    const int i = wave * (64 * BB_IPT) + r * 64 + lane;
    const bool valid = i < n;
    key[r] = valid ? src[i] : (T)0;
    const uint32_t d = (uint32_t)(key[r] >> shift) & 0xFFu;
    rank[r] = wave_stable_rank<8>(d, valid, cnt[wave]);
  }
  __syncthreads();
  {
    const int d = threadIdx.x;
    uint32_t tot = 0;
#pragma unroll
    for (int w = 0; w < BB_THREADS / 64; ++w) tot += cnt[w][d];
    uint32_t all;
    uint32_t run = block_excl_add(tot, &all, scratch);
#pragma unroll
    for (int w = 0; w < BB_THREADS / 64; ++w) {
      const uint32_t c = cnt[w][d];
      cnt[w][d] = run;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < BB_IPT; ++r) {
    const int i = wave * (64 * BB_IPT) + r * 64 + lane;
    if (i < n) {
      const uint32_t d = (uint32_t)(key[r] >> shift) & 0xFFu;
      dst[cnt[wave][d] + rank[r]] = key[r];
    }
  }
  __syncthreads();
}

// child digits of BB_LEVELS levels below the cube (c, e): the exact comparisons of compute_path in
// build.hip (octree/octree.py:73-75,94-97,181-191); *bad when the point is not inside the cube
__device__ __forceinline__ uint32_t path21_of(double px, double py, double pz, double cx, double cy,
                                              double cz, double e, bool* bad) {
  uint32_t path = 0;
  double h = e / 2.0;
#pragma unroll 1
  for (int j = 0; j < BB_LEVELS; ++j) {
    const double ax = px - cx, ay = py - cy, az = pz - cz;
    const bool ok = (ax >= 0.0) && (ax < e) && (ay >= 0.0) && (ay < e) && (az >= 0.0) && (az < e);
    if (!ok) {
      *bad = true;
      return path;
    }
    const bool bx = ax >= h, by = ay >= h, bz = az >= h;
    path |= ((bx ? 4u : 0u) | (by ? 2u : 0u) | (bz ? 1u : 0u)) << (18 - 3 * j);
    cx = cx + (bx ? h : 0.0);
    cy = cy + (by ? h : 0.0);
    cz = cz + (bz ? h : 0.0);
    e = h;
    h = e / 2.0;
  }
  return path;
}

__device__ __forceinline__ int find_slot_dev(const int64_t* __restrict__ pose_off, int n_poses,
                                             int64_t idx) {
  int lo = 0, hi = n_poses;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (pose_off[mid] <= idx) lo = mid; else hi = mid;
  }
  return lo;
}

struct BkParams {
  LinParams lp;
  int64_t K;
  uint32_t nst;       // stride of the partition table
  uint32_t nb;        // buckets
  uint32_t n_alive;
  int n_poses;
  int all_scheme;
};

// ---------------------------------------------------------------------------------------------
// one workgroup per bucket
// ---------------------------------------------------------------------------------------------
// Key of a point inside its bucket (64 bit): [33+s .. 33) voxel inside the bucket | [33 .. 12) path21
// | [12 .. 0) slot = position in the partitioned bucket (insertion order); bit 62 = pose is in the
// scheme, bit 63 = bad point.  Sorting on bits [12, 33+s) orders the points by (voxel, path); a node
// of level l is a run of keys with equal top s + 3l sort bits.
__global__ __launch_bounds__(BB_THREADS) void k_bucket_build(
    const uint32_t* __restrict__ part_lin, const uint32_t* __restrict__ part_idx,
    const double* __restrict__ part_xyz, const uint32_t* __restrict__ table, BkParams P,
    const int64_t* __restrict__ pose_off, uint32_t* __restrict__ ord_idx, double* __restrict__ xyz_ord,
    uint32_t* __restrict__ leafinfo, uint32_t* __restrict__ bk_vox, uint32_t* __restrict__ bk_tot) {
  __shared__ uint64_t s_key[2][BB_CAP];
  __shared__ uint32_t s_cnt[BB_THREADS / 64][256];
  __shared__ uint32_t s_scr[8];
  __shared__ uint32_t s_tot[BK_ROWS];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const uint32_t b = blockIdx.x;
  const uint32_t start = table[(size_t)b * P.nst];
  const uint32_t end = (b + 1 < P.nb) ? table[(size_t)(b + 1) * P.nst] : P.n_alive;
  const int n = (int)(end - start);
  if (tid < BK_ROWS) s_tot[tid] = 0;
  if (n == 0 || n > BB_CAP) {
    if (tid < BK_ROWS) bk_tot[(size_t)tid * P.nb + b] = (tid == BK_FLAGS && n > BB_CAP) ? BF_OVERFLOW : 0u;
    return;
  }
  const int s = P.lp.shift;
  const uint32_t lin0 = b << s;

  // ---- 1. keys ----------------------------------------------------------------------------------------
  bool bad_any = false;
#pragma unroll 4
  for (int r = 0; r < BB_IPT; ++r) {
    const int i = wave * (64 * BB_IPT) + r * 64 + lane;
    if (i < n) {
      const size_t g = (size_t)start + i;
      const uint32_t lin = part_lin[g];
      const double x = part_xyz[3 * g], y = part_xyz[3 * g + 1], z = part_xyz[3 * g + 2];
      double cx = P.lp.c0x, cy = P.lp.c0y, cz = P.lp.c0z;
      if (P.lp.mode == 0) {
        // the manager's corner is np.array(voxel_coords): int64(q * L) (grid.py:72-76,96-105)
        cx = (double)(long long)(floor_div_exact(x, P.lp.L) * P.lp.L);
        cy = (double)(long long)(floor_div_exact(y, P.lp.L) * P.lp.L);
        cz = (double)(long long)(floor_div_exact(z, P.lp.L) * P.lp.L);
      }
      bool bad = false;
      const uint32_t path = path21_of(x, y, z, cx, cy, cz, P.lp.L, &bad);
      uint64_t k = ((uint64_t)(lin - lin0) << 33) | ((uint64_t)path << 12) | (uint64_t)i;
      if (part_idx[g] >> 31) k |= 1ull << 62;
      if (bad) k |= 1ull << 63;
      bad_any = bad_any || bad;
      s_key[0][i] = k;
    }
  }
  __syncthreads();

  // ---- 2. sort by (voxel, path) -------------------------------------------------------------------------
  // K < 0 never splits: the path does not matter, only the voxel
  const int lo_bit = P.K < 0 ? 33 : 12;
  const int hi_bit = 33 + s;
  int cur = 0;
  for (int sh = lo_bit; sh < hi_bit; sh += 8) {
    lds_sort_pass<uint64_t>(s_key[cur], s_key[cur ^ 1], n, sh, s_cnt, s_scr);
    cur ^= 1;
  }
  const uint64_t* __restrict__ KS = s_key[cur];
  char* KT = reinterpret_cast<char*>(s_key[cur ^ 1]);
  // scratch carved out of the other key buffer
  int8_t* c8 = reinterpret_cast<int8_t*>(KT);                       // [0, n]: common levels with j-1
  uint16_t* pfx = reinterpret_cast<uint16_t*>(KT + 4608);           // [0, n]: scheme points before j
  uint16_t* lf16 = reinterpret_cast<uint16_t*>(KT + 4608 + 8704);   // leaf ordinal of sorted position j

  // ---- 3. structure -----------------------------------------------------------------------------------------
  // c[j] = number of leading levels (0..7) that sorted element j shares with j-1, -1 across voxels;
  // c[0] = c[n] = -1.  Blocked ownership from here on: thread t owns positions [16t, 16t+16).
  for (int j = tid; j <= n; j += BB_THREADS) {
    int c = -1;
    if (j > 0 && j < n) {
      const uint64_t x = ((KS[j] ^ KS[j - 1]) >> 12) & ((1ull << (21 + 12)) - 1ull);
      if ((x >> 21) == 0) {
        const uint32_t xp = (uint32_t)x & PATH_MASK;
        c = xp == 0 ? 7 : (20 - (31 - __clz((int)xp))) / 3;
      }
    }
    c8[j] = (int8_t)c;
  }
  const int j0 = tid * BB_IPT;
  uint64_t key[BB_IPT];
#pragma unroll
  for (int e = 0; e < BB_IPT; ++e) key[e] = (j0 + e < n) ? KS[j0 + e] : 0ull;
  if (!P.all_scheme) {
    uint32_t mine = 0;
#pragma unroll
    for (int e = 0; e < BB_IPT; ++e) mine += (j0 + e < n) ? (uint32_t)((key[e] >> 62) & 1ull) : 0u;
    uint32_t all;
    uint32_t run = block_excl_add(mine, &all, s_scr);
#pragma unroll
    for (int e = 0; e < BB_IPT; ++e) {
      if (j0 + e <= n) pfx[j0 + e] = (uint16_t)run;
      run += (j0 + e < n) ? (uint32_t)((key[e] >> 62) & 1ull) : 0u;
    }
    if (tid == BB_THREADS - 1 && n == BB_CAP) pfx[n] = (uint16_t)run;
  }
  __syncthreads();

  uint32_t act = 0;  // bit e: element e has not reached its leaf yet
#pragma unroll
  for (int e = 0; e < BB_IPT; ++e) act |= (j0 + e < n) ? (1u << e) : 0u;
  uint32_t dep_pk[2] = {0, 0};       // 4 bits per element: leaf depth
  uint32_t first_leaf = 0;           // bit e: the element's leaf starts its voxel
  uint16_t vlo[BB_IPT], vcnt[BB_IPT];
  uint32_t nint_loc[BB_LEVELS];
#pragma unroll
  for (int l = 0; l < BB_LEVELS; ++l) nint_loc[l] = 0;
  bool deep = false;
#pragma unroll 1
  for (int l = 0; l <= BB_LEVELS; ++l) {
    // heads of level l: positions i with c[i] < l.  lo(j) = last head <= j, hi(j) = first head > j.
    int8_t cc[BB_IPT + 1];
#pragma unroll
    for (int e = 0; e <= BB_IPT; ++e) cc[e] = (j0 + e <= n) ? c8[j0 + e] : (int8_t)-1;
    int last = -1, first = 0x7FFF;
#pragma unroll
    for (int e = 0; e < BB_IPT; ++e) {
      if (cc[e] < l) {
        last = j0 + e;
        if (first == 0x7FFF) first = j0 + e;
      }
    }
    // exclusive max-scan of `last` over lower threads, exclusive min-scan of `first` over higher ones
    int pl = last, nf = first;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int a = __shfl_up(pl, off), bq = __shfl_down(nf, off);
      if (lane >= off) pl = max(pl, a);
      if (lane + off < 64) nf = min(nf, bq);
    }
    int* sw = reinterpret_cast<int*>(s_scr);
    if (lane == 63) sw[wave] = pl;
    if (lane == 0) sw[4 + wave] = nf;
    __syncthreads();
    int carry_lo = -1, carry_hi = 0x7FFF;
#pragma unroll
    for (int w = 0; w < BB_THREADS / 64; ++w) {
      if (w < wave) carry_lo = max(carry_lo, sw[w]);
      if (w > wave) carry_hi = min(carry_hi, sw[4 + w]);
    }
    {
      const int a = __shfl_up(pl, 1), bq = __shfl_down(nf, 1);
      carry_lo = max(carry_lo, lane > 0 ? a : -1);
      carry_hi = min(carry_hi, lane < 63 ? bq : 0x7FFF);
    }
    carry_hi = min(carry_hi, n);
    int lo_e[BB_IPT], hi_e[BB_IPT];
    int run = carry_lo;
#pragma unroll
    for (int e = 0; e < BB_IPT; ++e) {
      if (cc[e] < l) run = j0 + e;
      lo_e[e] = run;
    }
    run = carry_hi;
#pragma unroll
    for (int e = BB_IPT - 1; e >= 0; --e) {
      hi_e[e] = (cc[e + 1] < l) ? j0 + e + 1 : run;
      run = hi_e[e];
    }
    bool any_split = false;
#pragma unroll
    for (int e = 0; e < BB_IPT; ++e) {
      if (l == 0) {
        vlo[e] = (uint16_t)lo_e[e];
        vcnt[e] = (uint16_t)(min(hi_e[e], n) - lo_e[e]);
      }
      if (act & (1u << e)) {
        const int lo = lo_e[e], hi = min(hi_e[e], n);
        const int64_t cntv = P.all_scheme ? (int64_t)(hi - lo) : (int64_t)(pfx[hi] - pfx[lo]);
        const bool split = P.K >= 0 && cntv > P.K;
        if (split && l < BB_LEVELS) {
          if (j0 + e == lo) nint_loc[l < BB_LEVELS ? l : 0] += 1;
          any_split = true;
        } else {
          if (split) deep = true;  // level 7 and still too many points: not representable here
          act &= ~(1u << e);
          dep_pk[e >> 3] |= (uint32_t)l << (4 * (e & 7));
          if (lo == (int)vlo[e]) first_leaf |= 1u << e;
        }
      }
    }
    if (!__syncthreads_or(any_split ? 1 : 0)) break;
  }

  // ---- 4. leaf / voxel ordinals, totals -------------------------------------------------------------------
  uint32_t heads = 0;  // low half: leaf heads, high half: voxel heads among my elements
  uint32_t lh_bits = 0, vh_bits = 0;
#pragma unroll
  for (int e = 0; e < BB_IPT; ++e) {
    if (j0 + e < n) {
      const int d = (int)((dep_pk[e >> 3] >> (4 * (e & 7))) & 15u);
      const int c = (int)c8[j0 + e];
      if (c < d) lh_bits |= 1u << e;
      if (c < 0) vh_bits |= 1u << e;
    }
  }
  heads = (uint32_t)__popc(lh_bits) | ((uint32_t)__popc(vh_bits) << 16);
  uint32_t tot_heads;
  uint32_t hrun = block_excl_add(heads, &tot_heads, s_scr);
  const int n_leaves = (int)(tot_heads & 0xFFFFu);
  uint32_t lrun = hrun & 0xFFFFu, vrun = hrun >> 16;
#pragma unroll
  for (int e = 0; e < BB_IPT; ++e) {
    if (j0 + e < n) {
      if (lh_bits & (1u << e)) ++lrun;
      if (vh_bits & (1u << e)) {
        // staging of the j-th voxel of this bucket: linear key and point count
        const uint32_t vl = (uint32_t)((key[e] >> 33) & 0xFFFu);
        bk_vox[2 * ((size_t)start + vrun)] = lin0 + vl;
        bk_vox[2 * ((size_t)start + vrun) + 1] = (uint32_t)vcnt[e];
        ++vrun;
      }
      lf16[j0 + e] = (uint16_t)(lrun - 1u);
    }
  }
  {
    // per-level internal node counts, bad / deep flags
#pragma unroll
    for (int l = 0; l < BB_LEVELS; ++l) {
      uint32_t v = nint_loc[l];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
      if (lane == 0 && v) atomicAdd(&s_tot[BK_NINT + l], v);
    }
    const uint32_t fl = (__any(deep) ? BF_DEEP : 0u) | (__any(bad_any) ? BF_BAD : 0u);
    if (lane == 0 && fl) atomicOr(&s_tot[BK_FLAGS], fl);
    if (tid == 0) s_tot[BK_NVOX] = tot_heads >> 16;
  }
  __syncthreads();  // lf16 complete; every read of KS / c8 is done

  // ---- 5. back to insertion order: INFO[slot], (leaf ordinal, slot) words ------------------------------------
  uint32_t* INFO = reinterpret_cast<uint32_t*>(s_key[cur]);             // [BB_CAP]
  uint32_t* SB0 = reinterpret_cast<uint32_t*>(s_key[cur]) + BB_CAP;     // [BB_CAP]
  uint32_t* SB1 = reinterpret_cast<uint32_t*>(s_key[cur ^ 1]);          // [BB_CAP] (over c8 / pfx: dead)
  uint16_t lf_loc[BB_IPT];
#pragma unroll
  for (int e = 0; e < BB_IPT; ++e) lf_loc[e] = (j0 + e < n) ? lf16[j0 + e] : (uint16_t)0;
  __syncthreads();
#pragma unroll
  for (int e = 0; e < BB_IPT; ++e) {
    if (j0 + e < n) {
      const uint32_t slot = (uint32_t)(key[e] & 0xFFFull);
      const uint32_t d = (dep_pk[e >> 3] >> (4 * (e & 7))) & 15u;
      const uint32_t path = (uint32_t)(key[e] >> 12) & PATH_MASK;
      INFO[slot] = path | (d << 21) | ((first_leaf >> e) & 1u ? (1u << 30) : 0u);
      SB0[slot] = ((uint32_t)lf_loc[e] << 12) | slot;
    }
  }
  __syncthreads();
  // ---- 6. stable sort by leaf ordinal from insertion order ---------------------------------------------------
  uint32_t* sb[2] = {SB0, SB1};
  int sc = 0;
  for (int sh = 12; sh < 12 + 13 && (sh == 12 || (n_leaves - 1) >> (sh - 12)); sh += 8) {
    lds_sort_pass<uint32_t>(sb[sc], sb[sc ^ 1], n, sh, s_cnt, s_scr);
    sc ^= 1;
  }
  const uint32_t* __restrict__ RS = sb[sc];

  // ---- 7. outputs (coalesced) ------------------------------------------------------------------------------------
  uint32_t nblk = 0;
#pragma unroll 4
  for (int r = 0; r < BB_IPT; ++r) {
    const int f = r * BB_THREADS + tid;
    if (f < n) {
      const uint32_t w = RS[f];
      const uint32_t slot = w & 0xFFFu;
      const uint32_t info = INFO[slot];
      const bool leaf_head = f == 0 || (RS[f - 1] >> 12) != (w >> 12);
      const size_t g = (size_t)start + slot;
      const uint32_t idx = part_idx[g] & IDX_MASK;
      bool blk_head = leaf_head;
      if (!leaf_head && P.n_poses > 1) {
        const uint32_t pidx = part_idx[(size_t)start + (RS[f - 1] & 0xFFFu)] & IDX_MASK;
        blk_head = find_slot_dev(pose_off, P.n_poses, idx) != find_slot_dev(pose_off, P.n_poses, pidx);
      }
      nblk += blk_head ? 1u : 0u;
      const size_t o = (size_t)start + f;
      leafinfo[o] = (info & 0xFFFFFFu) | ((leaf_head && (info >> 30 & 1u)) ? LI_VHEAD : 0u) |
                    (blk_head ? LI_BHEAD : 0u);
      ord_idx[o] = idx;
      xyz_ord[3 * o] = part_xyz[3 * g];
      xyz_ord[3 * o + 1] = part_xyz[3 * g + 1];
      xyz_ord[3 * o + 2] = part_xyz[3 * g + 2];
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) nblk += __shfl_xor(nblk, off);
  if (lane == 0 && nblk) atomicAdd(&s_tot[BK_NBLK], nblk);
  __syncthreads();
  if (tid < BK_ROWS) bk_tot[(size_t)tid * P.nb + b] = s_tot[tid];
}

// ---------------------------------------------------------------------------------------------
// bucket totals -> numbering bases (one workgroup; nb <= 4096)
// ---------------------------------------------------------------------------------------------
// In place: row BK_NVOX -> first voxel (root) of every bucket; rows BK_NINT+l -> first internal node of
// level l of every bucket in the LEVEL-MAJOR order of the level-synchronous path (all internal nodes of
// level 0 in voxel order, then level 1, ...); row BK_NBLK -> first block.  Totals -> small[].
__global__ __launch_bounds__(1024) void k_bucket_scan(uint32_t* __restrict__ bk_tot, uint32_t nb,
                                                      uint32_t* __restrict__ small) {
  __shared__ uint32_t s_w[16];
  __shared__ uint32_t s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t level_base = 0, flags = 0;
  for (int row = 0; row < BK_ROWS; ++row) {
    if (tid == 0) s_carry = (row >= BK_NINT && row < BK_NINT + BB_LEVELS) ? level_base : 0u;
    __syncthreads();
    uint32_t row_total = 0;
    for (uint32_t b0 = 0; b0 < nb; b0 += 1024) {
      const uint32_t bb = b0 + tid;
      const uint32_t v = bb < nb ? bk_tot[(size_t)row * nb + bb] : 0u;
      if (row == BK_FLAGS) {
        flags |= v;
        continue;
      }
      uint32_t inc = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
      }
      if (lane == 63) s_w[wave] = inc;
      __syncthreads();
      uint32_t basev = s_carry, tot = 0;
      for (int w = 0; w < 16; ++w) {
        if (w < wave) basev += s_w[w];
        tot += s_w[w];
      }
      if (bb < nb) bk_tot[(size_t)row * nb + bb] = basev + inc - v;
      row_total += tot;
      __syncthreads();
      if (tid == 0) s_carry += tot;
      __syncthreads();
    }
    if (row == BK_FLAGS) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) flags |= __shfl_xor(flags, off);
      if (lane == 0 && flags) atomicOr(&small[SM_BK_FLAGS], flags);
    } else if (tid == 0) {
      if (row == BK_NVOX) small[SM_NVOX] = row_total;
      if (row == BK_NBLK) small[SM_NBLOCKS] = row_total;
      if (row >= BK_NINT && row < BK_NINT + BB_LEVELS) small[SM_BK_LEVEL + (row - BK_NINT)] = row_total;
    }
    if (row >= BK_NINT && row < BK_NINT + BB_LEVELS) level_base += row_total;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// scheme nodes, position -> leaf, block table: one wavefront per bucket
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t lanemask_le64() {
  const unsigned lane = threadIdx.x & 63u;
  return (lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull);
}
__device__ __forceinline__ uint32_t digit_at(uint32_t path21, int level) {
  return (path21 >> (18 - 3 * level)) & 7u;
}

struct NodeParams {
  LinParams lp;
  uint32_t nst, nb, n_alive;
  int n_poses, all_scheme, cur_epoch;
  int64_t node_cap;   // capacity of the node table (nodes); a larger table is needed -> overflow flag
};

__global__ __launch_bounds__(256) void k_bucket_nodes(
    NodePtrs nd, NodeParams P, const uint32_t* __restrict__ table, const uint32_t* __restrict__ bk_base,
    const uint32_t* __restrict__ leafinfo,
    const uint32_t* __restrict__ ord_idx, const uint32_t* __restrict__ bk_vox,
    const int64_t* __restrict__ pose_off, int32_t* __restrict__ pos_node, uint64_t* __restrict__ vlin,
    int32_t* __restrict__ blk_node, int32_t* __restrict__ blk_slot, uint32_t* __restrict__ blk_start,
    uint32_t* small) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t b = blockIdx.x * 4 + wave;
  if (b >= P.nb) return;  // no workgroup barrier below: waves are independent
  const uint32_t start = table[(size_t)b * P.nst];
  const uint32_t end = (b + 1 < P.nb) ? table[(size_t)(b + 1) * P.nst] : P.n_alive;
  const int n = (int)(end - start);
  if (n == 0 || n > BB_CAP) return;
  const int64_t V = (int64_t)small[SM_NVOX];
  int64_t n_int = 0;
#pragma unroll
  for (int l = 0; l < BB_LEVELS; ++l) n_int += small[SM_BK_LEVEL + l];
  if (V + 8 * n_int > P.node_cap) {
    if (lane == 0) atomicOr(&small[SM_BK_FLAGS], 0x100u);  // the host grows the table and launches again
    return;
  }
  const uint32_t vbase = bk_base[(size_t)BK_NVOX * P.nb + b];
  uint32_t lbase[BB_LEVELS];
#pragma unroll
  for (int l = 0; l < BB_LEVELS; ++l) lbase[l] = bk_base[(size_t)(BK_NINT + l) * P.nb + b];
  const uint32_t bbase = bk_base[(size_t)BK_NBLK * P.nb + b];

  uint32_t vcarry = 0, bcarry = 0;
  uint32_t carry[BB_LEVELS], tail_pref[BB_LEVELS];
  bool tail_act[BB_LEVELS];
#pragma unroll
  for (int l = 0; l < BB_LEVELS; ++l) {
    carry[l] = 0;
    tail_pref[l] = 0;
    tail_act[l] = false;
  }
  for (int f0 = 0; f0 < n; f0 += 64) {
    const int f = f0 + lane;
    const bool valid = f < n;
    const uint32_t li = valid ? leafinfo[(size_t)start + f] : 0u;
    const uint32_t pw = li & PATH_MASK;
    const uint32_t dep = (li >> 21) & 7u;
    const bool vhead = valid && (li & LI_VHEAD);
    const bool bhead = valid && (li & LI_BHEAD);
    // voxel of every lane
    const uint64_t vb = __ballot(vhead);
    const uint32_t vo = vcarry + (uint32_t)__popcll(vb & lanemask_le64()) - 1u;  // ordinal inside the bucket
    vcarry += (uint32_t)__popcll(vb);
    const int32_t v = (int32_t)(vbase + vo);
    // root geometry (only the lanes that need it read the staging record)
    double c0x = P.lp.c0x, c0y = P.lp.c0y, c0z = P.lp.c0z;
    uint32_t lin = 0;
    if (valid && P.lp.mode == 0) {
      lin = bk_vox[2 * ((size_t)start + vo)];
      const uint32_t qz = lin % P.lp.nz, qy = (lin / P.lp.nz) % P.lp.ny, qx = lin / (P.lp.nz * P.lp.ny);
      // np.array(voxel_coordinates): int64(q * L), L integer valued (grid.py:72-76,104)
      c0x = (double)(long long)((double)((int)qx + P.lp.minx) * P.lp.L);
      c0y = (double)(long long)((double)((int)qy + P.lp.miny) * P.lp.L);
      c0z = (double)(long long)((double)((int)qz + P.lp.minz) * P.lp.L);
    }
    if (vhead) {
      const uint32_t cntv = bk_vox[2 * ((size_t)start + vo) + 1];
      nd.start[v] = start + (uint32_t)f;
      nd.count[v] = cntv;
      nd.scount[v] = P.all_scheme ? cntv : 0u;
      nd.depth[v] = 0;
      nd.voxel[v] = v;
      nd.parent[v] = -1;
      nd.old_id[v] = -1;
      nd.edge[v] = P.lp.L;
      nd.corner[3 * (int64_t)v] = c0x;
      nd.corner[3 * (int64_t)v + 1] = c0y;
      nd.corner[3 * (int64_t)v + 2] = c0z;
      vlin[v] = (uint64_t)lin;
    }
    int32_t leaf = v;  // dep == 0: the root is the leaf
    int32_t cb_prev = 0;
#pragma unroll
    for (int l = 0; l < BB_LEVELS; ++l) {
      const bool actl = valid && dep > (uint32_t)l;  // inside an internal node of level l
      const uint32_t pref = l == 0 ? 0u : (pw >> (21 - 3 * l));
      uint32_t pp = __shfl_up(pref, 1);
      bool pa = __shfl_up(actl ? 1 : 0, 1) != 0;
      if (lane == 0) {
        pp = tail_pref[l];
        pa = tail_act[l];
      }
      const bool head = actl && (vhead || !pa || pp != pref);
      const uint64_t hb = __ballot(head);
      const uint32_t ord = carry[l] + (uint32_t)__popcll(hb & lanemask_le64()) - 1u;
      carry[l] += (uint32_t)__popcll(hb);
      tail_pref[l] = __shfl(pref, 63);
      tail_act[l] = __shfl(actl ? 1 : 0, 63) != 0;
      if (actl) {
        const int32_t cb = (int32_t)(V + 8 * (int64_t)(lbase[l] + ord));  // its 8 children
        const int32_t xid = l == 0 ? v : cb_prev + (int32_t)digit_at(pw, l - 1);
        if (head) {
          nd.first_child[xid] = cb;
          nd.epoch[xid] = P.cur_epoch;
          // corner / edge: descend from the root with the reference's arithmetic
          // (corner + offset, edge / 2: octree.py:181-191)
          double cx = c0x, cy = c0y, cz = c0z, e = P.lp.L;
          for (int t = 0; t < l; ++t) {
            const uint32_t d = digit_at(pw, t);
            const double h = e / 2.0;
            cx = cx + ((d & 4u) ? h : 0.0);
            cy = cy + ((d & 2u) ? h : 0.0);
            cz = cz + ((d & 1u) ? h : 0.0);
            e = h;
          }
          const double h = e / 2.0;
          for (int j = 0; j < 8; ++j) {
            const int64_t c = (int64_t)cb + j;
            nd.start[c] = 0;  // ranges are only meaningful inside the level-synchronous path
            nd.count[c] = 0;
            nd.scount[c] = 0;
            nd.depth[c] = l + 1;
            nd.voxel[c] = v;
            nd.parent[c] = xid;
            nd.old_id[c] = -1;
            nd.edge[c] = h;
            nd.corner[3 * c + 0] = cx + ((j & 4) ? h : 0.0);
            nd.corner[3 * c + 1] = cy + ((j & 2) ? h : 0.0);
            nd.corner[3 * c + 2] = cz + ((j & 1) ? h : 0.0);
          }
        }
        if (dep == (uint32_t)l + 1u) leaf = cb + (int32_t)digit_at(pw, l);
        cb_prev = cb;
      }
    }
    if (valid) pos_node[(size_t)start + f] = leaf;
    // (leaf, pose) blocks
    const uint64_t bbm = __ballot(bhead);
    if (bhead) {
      const uint32_t bo = bbase + bcarry + (uint32_t)__popcll(bbm & lanemask_le64()) - 1u;
      blk_node[bo] = leaf;
      blk_slot[bo] = P.n_poses > 1 ? find_slot_dev(pose_off, P.n_poses, ord_idx[(size_t)start + f]) : 0;
      blk_start[bo] = start + (uint32_t)f;
    }
    bcarry += (uint32_t)__popcll(bbm);
  }
}

__global__ __launch_bounds__(256) void k_block_sizes_dev(const uint32_t* __restrict__ blk_start,
                                                         const uint32_t* __restrict__ nb_dev,
                                                         uint32_t n_alive, int32_t* __restrict__ blk_size) {
  const int64_t nb = (int64_t)*nb_dev;
  for (int64_t bq = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; bq < nb;
       bq += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t e = (bq + 1 < nb) ? blk_start[bq + 1] : n_alive;
    blk_size[bq] = (int32_t)(e - blk_start[bq]);
  }
}

int ceil_log2_u64(uint64_t v) {
  int b = 0;
  while (b < 64 && ((uint64_t)1 << b) < v) ++b;
  return b;
}

}  // namespace

// Complete build of a fresh forest (no previous scheme): *done = 1 when the scheme, the leaf-ordered
// arrays, pos_node and the block table are complete; *done = 0 when this path does not apply (the
// caller then runs the general path; nothing it relies on has been modified).
int forest_bucket_build(octl_forest* f, const BucketBuildArgs& a, NodeTable& nt, int* done,
                        std::vector<int64_t>* level_first, int64_t* n_internal, int* levels,
                        int64_t* n_voxels, int64_t* n_blocks, BucketBuildGeom* geom) {
  octl_ctx* ctx = f->ctx;
  hipStream_t st = ctx->stream;
  *done = 0;
  const int64_t N = f->n_store, n_alive = f->n_alive;
  // (OCTL_NO_VOXEL_BUILD: tests compare this path with the level-synchronous one)
  if (n_alive <= 0 || !f->bbox_dev.p || getenv("OCTL_NO_BUCKET_BUILD") || getenv("OCTL_NO_VOXEL_BUILD"))
    return OCTL_OK;
  const int n_poses = (int)f->pose_off.size() - 1;
  // ---- voxel bounding box (kept by the ingest kernel; the copy was enqueued with the last ingest) ------
  HIP_TRY(ctx, hipEventSynchronize(f->bbox_event));
  int bb[6];
  std::memcpy(bb, f->bbox_host, sizeof(bb));
  if (f->bbox_host[6])
    return octl_set_error(ctx, OCTL_E_DOMAIN,
                          "a point has a non-finite coordinate or a top-level voxel index outside +-%d",
                          OCTL_VOX_BIAS);
  if (bb[0] > bb[3]) return OCTL_OK;
  const uint64_t nx = (uint64_t)(bb[3] - bb[0] + 1), ny = (uint64_t)(bb[4] - bb[1] + 1),
                 nz = (uint64_t)(bb[5] - bb[2] + 1);
  if (nx * ny > (1ull << 32) || nx * ny * nz >= (1ull << 32)) return OCTL_OK;  // keys would not fit 32 bits
  const uint64_t R = nx * ny * nz;
  // buckets: runs of 2^s consecutive voxel keys, sized for ~2500 points on average
  uint64_t want = 1;
  while (want < PT_BINS && want * 2560 < (uint64_t)n_alive) want <<= 1;
  if (want * 2560 < (uint64_t)n_alive && R > (uint64_t)PT_BINS) return OCTL_OK;  // needs two levels: not yet
  const int s = std::min(12, std::max(0, ceil_log2_u64(R) - ceil_log2_u64(want)));
  if (((R - 1) >> s) + 1 > (uint64_t)PT_BINS) return OCTL_OK;  // a sparse scene: more than 2^24 voxel keys
  const uint32_t nb = (uint32_t)(((R - 1) >> s) + 1);

  LinParams lp;
  lp.mode = f->mode;
  lp.L = f->edge;
  lp.c0x = f->corner[0];
  lp.c0y = f->corner[1];
  lp.c0z = f->corner[2];
  lp.minx = bb[0];
  lp.miny = bb[1];
  lp.minz = bb[2];
  lp.ny = (uint32_t)ny;
  lp.nz = (uint32_t)nz;
  lp.shift = s;
  uint32_t* small = ctx->small.as<uint32_t>();
  const uint32_t nst = (uint32_t)ceil_div(N, PT_ST);
  // ---- scratch ------------------------------------------------------------------------------------------------
  OCTL_TRY(devbuf_reserve(ctx, f->part_lin[0], (size_t)n_alive * 4));
  OCTL_TRY(devbuf_reserve(ctx, f->part_idx[0], (size_t)n_alive * 4));
  OCTL_TRY(devbuf_reserve(ctx, f->part_xyz[0], (size_t)n_alive * 24));
  OCTL_TRY(devbuf_reserve(ctx, f->bk_table, ((size_t)nb * nst + 8) * 4));
  OCTL_TRY(devbuf_reserve(ctx, f->bk_tot, (size_t)BK_ROWS * nb * 4));
  OCTL_TRY(devbuf_reserve(ctx, f->bk_vox, (size_t)n_alive * 8));
  OCTL_TRY(devbuf_reserve(ctx, f->leafinfo, (size_t)n_alive * 4));
  OCTL_TRY(devbuf_reserve(ctx, f->ord_idx, (size_t)n_alive * 4));
  OCTL_TRY(devbuf_reserve(ctx, f->xyz_ord, (size_t)n_alive * 24));
  OCTL_TRY(devbuf_reserve(ctx, f->pos_node, (size_t)n_alive * 4));
  OCTL_TRY(devbuf_reserve(ctx, f->blk_node, (size_t)n_alive * 4));
  OCTL_TRY(devbuf_reserve(ctx, f->blk_slot, (size_t)n_alive * 4));
  OCTL_TRY(devbuf_reserve(ctx, f->blk_start, (size_t)n_alive * 4));
  OCTL_TRY(devbuf_reserve(ctx, f->blk_size, (size_t)n_alive * 4));
  uint32_t* table = f->bk_table.as<uint32_t>();
  HIP_TRY(ctx, hipMemsetAsync(small + SM_BK_FLAGS, 0, 4, st));
  // ---- partition ----------------------------------------------------------------------------------------------
  {
    KTimer t(ctx, "part_hist");
    hipLaunchKernelGGL(k_part_hist, dim3(nst), dim3(PT_THREADS), 0, st, (const double*)f->xyz.as<double>(),
                       (const uint8_t*)f->alive.as<uint8_t>(), N, lp, nst, nb, table);
    HIP_TRY(ctx, hipGetLastError());
  }
  {
    KTimer t(ctx, "part_scan");
    OCTL_TRY(octl_exclusive_scan_u32(ctx, table, table, (int64_t)nb * nst, nullptr));
  }
  {
    KTimer t(ctx, "part_scatter");
    hipLaunchKernelGGL(k_part_scatter, dim3(nst), dim3(PT_THREADS), 0, st,
                       (const double*)f->xyz.as<double>(), (const uint8_t*)f->alive.as<uint8_t>(), N, lp,
                       nst, nb, (const uint32_t*)table, (const int64_t*)f->pose_off_dev.as<int64_t>(),
                       n_poses, a.scheme_dev, f->part_lin[0].as<uint32_t>(), f->part_idx[0].as<uint32_t>(),
                       f->part_xyz[0].as<double>());
    HIP_TRY(ctx, hipGetLastError());
  }
  // ---- buckets ------------------------------------------------------------------------------------------------
  BkParams bp;
  bp.lp = lp;
  bp.K = a.K;
  bp.nst = nst;
  bp.nb = nb;
  bp.n_alive = (uint32_t)n_alive;
  bp.n_poses = n_poses;
  bp.all_scheme = a.scheme_dev ? 0 : 1;
  {
    KTimer t(ctx, "bucket_build");
    hipLaunchKernelGGL(k_bucket_build, dim3(nb), dim3(BB_THREADS), 0, st,
                       (const uint32_t*)f->part_lin[0].as<uint32_t>(),
                       (const uint32_t*)f->part_idx[0].as<uint32_t>(),
                       (const double*)f->part_xyz[0].as<double>(), (const uint32_t*)table, bp,
                       (const int64_t*)f->pose_off_dev.as<int64_t>(), f->ord_idx.as<uint32_t>(),
                       f->xyz_ord.as<double>(), f->leafinfo.as<uint32_t>(), f->bk_vox.as<uint32_t>(),
                       f->bk_tot.as<uint32_t>());
    HIP_TRY(ctx, hipGetLastError());
  }
  {
    KTimer t(ctx, "bucket_scan");
    hipLaunchKernelGGL(k_bucket_scan, dim3(1), dim3(1024), 0, st, f->bk_tot.as<uint32_t>(), nb, small);
    HIP_TRY(ctx, hipGetLastError());
  }
  uint32_t sm[64];
  HIP_TRY(ctx, hipMemcpyAsync(ctx->small_host, small, sizeof(sm), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  std::memcpy(sm, ctx->small_host, sizeof(sm));
  if (sm[SM_BK_FLAGS]) return OCTL_OK;  // some bucket / voxel does not fit: the caller runs the general path
  const int64_t V = sm[SM_NVOX];
  int64_t n_int = 0;
  int depth = 0;
  level_first->assign({0, V});
  for (int l = 0; l < BB_LEVELS; ++l) {
    if (sm[SM_BK_LEVEL + l] == 0) break;
    n_int += sm[SM_BK_LEVEL + l];
    depth = l + 1;
    level_first->push_back(V + 8 * n_int);
  }
  if (depth > a.max_depth) return octl_set_error(ctx, OCTL_E_DEPTH, "maximum depth %d exceeded", a.max_depth);
  const int64_t total = V + 8 * n_int;
  if (total >= ((int64_t)1 << 31)) return octl_set_error(ctx, OCTL_E_NOMEM, "more than 2^31 scheme nodes");
  OCTL_TRY(nodes_reserve(ctx, nt, total));
  OCTL_TRY(devbuf_reserve(ctx, f->vlin_dev, (size_t)std::max<int64_t>(V, 1) * 8));
  NodePtrs nd = node_ptrs(nt);
  HIP_TRY(ctx, hipMemsetAsync(nd.first_child, 0xFF, (size_t)total * 4, st));
  HIP_TRY(ctx, hipMemsetAsync(nd.epoch, 0, (size_t)total * 4, st));
  NodeParams np;
  np.lp = lp;
  np.nst = nst;
  np.nb = nb;
  np.n_alive = (uint32_t)n_alive;
  np.n_poses = n_poses;
  np.all_scheme = bp.all_scheme;
  np.cur_epoch = a.cur_epoch;
  np.node_cap = nt.cap;
  {
    KTimer t(ctx, "bucket_nodes");
    hipLaunchKernelGGL(k_bucket_nodes, dim3((unsigned)ceil_div(nb, 4)), dim3(256), 0, st, nd, np,
                       (const uint32_t*)table, (const uint32_t*)f->bk_tot.as<uint32_t>(),
                       (const uint32_t*)f->leafinfo.as<uint32_t>(),
                       (const uint32_t*)f->ord_idx.as<uint32_t>(), (const uint32_t*)f->bk_vox.as<uint32_t>(),
                       (const int64_t*)f->pose_off_dev.as<int64_t>(), f->pos_node.as<int32_t>(),
                       f->vlin_dev.as<uint64_t>(), f->blk_node.as<int32_t>(), f->blk_slot.as<int32_t>(),
                       f->blk_start.as<uint32_t>(), small);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_block_sizes_dev, dim3(1024), dim3(256), 0, st,
                       (const uint32_t*)f->blk_start.as<uint32_t>(), (const uint32_t*)(small + SM_NBLOCKS),
                       (uint32_t)n_alive, f->blk_size.as<int32_t>());
    HIP_TRY(ctx, hipGetLastError());
  }
  nt.n = total;
  *n_internal = n_int;
  *levels = depth;
  *n_voxels = V;
  *n_blocks = sm[SM_NBLOCKS];
  geom->min[0] = bb[0];
  geom->min[1] = bb[1];
  geom->min[2] = bb[2];
  geom->ny = ny;
  geom->nz = nz;
  *done = 1;
  return OCTL_OK;
}
