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
//   k_part_hist / k_part_scatter   partition of (voxel, child digits, index, xyz) records into BUCKETS of
//                                  consecutive voxels: one stable pass on a 12-bit digit (LDS
//                                  histograms, wave64 ballot ranks) for up to 4096 buckets - the only
//                                  scatter of coordinates - and a second one on the next 12 bits for
//                                  clouds that need more buckets (> 10 M points);
//   k_bucket_build                 one workgroup per bucket (<= 4096 points, a few voxels; larger buckets in
//                                  chunks of whole voxels): level by level the undecided points add
//                                  themselves to an LDS histogram over (node ordinal, child digit), bins
//                                  above K become the nodes of the next level (a node splits while its
//                                  scheme-pose count exceeds K) - no point moves until every point knows
//                                  its leaf; then ONE stable LDS radix sort by (voxel, leaf path) and
//                                  coalesced stores of permutation, coordinates and a leaf word per
//                                  point; voxels, internal nodes and leaves are numbered INSIDE the
//                                  bucket (staging records per voxel and per internal node);
//   one exclusive scan             bucket totals -> voxel / node / block numbering bases (k_bucket_scan_totals; over few
//                                  buckets the speculative k_bucket_finish<true> scans the table itself);
//   k_bucket_finish                one workgroup per bucket, parallel sweeps: roots, internal nodes and
//                                  their children in the level-major numbering of the level-synchronous
//                                  path, (leaf, pose) block table, position -> leaf map when needed.
// Everything is HBM-bound integer / compare work; the f64 arithmetic is the reference's own
// (exact comparisons on the rounded differences it forms).  Results are bit-identical to the
// level-synchronous path of build.hip.
#include "build_common.h"
#include "ref_arith.h"
#include "lookback.h"
#include "wave_utils.h"

namespace {

constexpr int PT_THREADS = 256;
// a workgroup partitions one SUPERTILE (one column of the histogram table) tile by tile; the number
// of tiles per supertile is chosen at run time so that one round of workgroups covers the cloud
constexpr int PT_BITS = 12;
constexpr int PT_BINS = 1 << PT_BITS;            // buckets

// The bucket kernel reads
// every record ONCE - the coordinates stay in registers from the first pass to the output, which goes through an
// LDS window - instead of the tail first and the whole record again for the output.  It needs 8 instead of 16
// items per thread (42 VGPRs of coordinates for BB_KEEP_ROUNDS = 7 of the 8 rounds; a bucket fuller than that reads
// its last round again) and so 512 threads per bucket, two workgroups per CU instead of three.  (The second argument
// of __launch_bounds__ is WAVES PER SIMD with this compiler, not workgroups per CU: 4 = 2 workgroups of 8 waves.)
constexpr int BB_THREADS = 512;
constexpr int BB_IPT = 8;
constexpr int BB_WGS = 4;   // waves per SIMD asked of the compiler (2 workgroups of 8 waves per CU)
#define BB_LEAF_UNROLL 2
#define BB_KEEP_ROUNDS 7
constexpr int BB_KEEP = BB_KEEP_ROUNDS < BB_IPT ? BB_KEEP_ROUNDS : BB_IPT;   // rounds whose coordinates stay in registers 
constexpr int BB_CAP = BB_THREADS * BB_IPT;      // points per bucket handled in LDS
static_assert(BB_CAP == 4096, "leaf words, 16-bit item ids and the chunk plan assume 4096 points per piece");
constexpr int BB_LEVELS = 7;                     // child digits per point (21 bits)
constexpr uint32_t PATH_MASK = 0x1FFFFFu;

// rows of the per-bucket totals table bk_tot[row][bucket]
enum { BK_NVOX = 0, BK_NINT = 1 /* .. 7 */, BK_NBLK = 8, BK_ROWS = 9 };
// leafinfo word of a leaf-ordered point: bits 0..15 ordinal (inside the bucket) of the leaf's parent among
// the internal nodes of its level - or of the voxel when the root is the leaf -, child digit, leaf depth
constexpr int LC_DIGIT = 16, LC_DEPTH = 19;
constexpr uint32_t LI_VHEAD = 1u << 24;   // first point of its top-level voxel
constexpr uint32_t LI_BHEAD = 1u << 25;   // first point of its (leaf, pose) block
// bucket flags
constexpr uint32_t FO_MAX = 1024;       // internal nodes / blocks per bucket k_bucket_finish can order itself
constexpr uint32_t BF_OVERFLOW = 1u;      // a bucket beyond what the oversize launch handles: whole build -> general path

struct LinParams {
  int mode;          // 0 grid, 1 single cube, 2 single cube keyed by the child digits of its first pm levels
  int pm;            // mode 2: levels of the key (the "voxels" of the partition are the cube's depth-pm nodes)
  double L;          // voxel edge
  double c0x, c0y, c0z;  // cube corner (mode 1)
  int minx, miny, minz;  // origin of the linear keys: the voxel box the geometry was formed for, PADDED (see geom_from_box)
  uint32_t ny, nz;
  int shift;         // bits of a voxel's position inside its bucket (width <= 1 << shift); two passes: bucket = lin >> shift
  // digit of the current partition pass: (lin >> dshift) & dmask.  One pass: the bucket itself.  More
  // than 4096 buckets: two stable passes, the low 12 bits of the bucket first, then the rest.
  int dshift;
  uint32_t dmask;
  int raw_vp;        // 1: this pass is not the last one, records carry the full linear key
  int exact_digits;  // 0: child digits always level by level (OCTL_NO_EXACT_DIGITS, tests)
  // A single pass whose geometry was formed by geom_from_box: bucket = lin / width for ANY width <= 4096 (round 6:
  // the box is padded by a margin so that a scene drifting by a voxel keeps its geometry, and a padded box's key
  // count is no power of two times the bucket count).  winv = fl(1 / width) and whalf = winv / 2:
  // trunc(fma(lin, winv, whalf)) = floor((lin + 0.5) / width) = floor(lin / width) exactly for lin < 2^24 - the true
  // value is at least 0.5 / width >= 2^-13 away from an integer, the rounding errors are below 2^-27.
  // winv == 0: the shift form (host-formed geometries: two passes, single cubes, OCTL_SYNC_GEOM).
  uint32_t width;
  double winv, whalf;
};
__device__ __forceinline__ uint32_t bucket_of(const LinParams& lp, uint32_t lin) {
  return lp.winv != 0.0 ? (uint32_t)fma((double)lin, lp.winv, lp.whalf) : lin >> lp.shift;
}
__device__ __forceinline__ uint32_t bucket_lin0(const LinParams& lp, uint32_t b) {
  return lp.winv != 0.0 ? b * lp.width : b << lp.shift;
}
__device__ __forceinline__ uint32_t digit_of(const LinParams& lp, uint32_t lin) {
  return lp.winv != 0.0 ? (uint32_t)fma((double)lin, lp.winv, lp.whalf) : (lin >> lp.dshift) & lp.dmask;
}

// The geometry of the linear voxel keys, formed ON THE DEVICE from the bounding box the ingest kernel
// keeps: clouds that need one partition pass (up to 4096 buckets - ~10 M points) are partitioned and
// built without the host ever waiting for the box; it reads this record together with the bucket totals.
struct GeomDev {
  LinParams lp;
  int32_t bb[6];    // the box of the linear keys: the true box padded by the margin
  int32_t tb[6];    // the TRUE voxel box of the cloud this build partitioned (the next build's hint is formed from it)
  uint32_t valid;   // 1: the kernels run; 0: they return at once, and the host acts on `reason`
  uint32_t reason;  // 1 a point outside the voxel domain, 2 nothing alive, 3 not a single-pass case,
                    // 4 the hinted geometry did not hold (the box is known now: build again)
};
enum { GEOM_DOMAIN = 1, GEOM_EMPTY = 2, GEOM_RETRY = 3, GEOM_REHASH = 4 };
static_assert(sizeof(GeomDev) <= 192, "GeomDev lives in the scalar block");

// what a build asks of the geometry (host -> kernels that form or validate one)
struct GeomAsk {
  uint64_t want;      // buckets wanted (a power of two, n_alive / target rounded up)
  int64_t n_alive;
  uint32_t target;    // points per bucket aimed at
  int32_t margin;     // voxels of slack on every side of the true box
};

__host__ __device__ inline int geom_ceil_log2(uint64_t v) {
  int b = 0;
  while (b < 64 && ((uint64_t)1 << b) < v) ++b;
  return b;
}
// buckets a single pass has tables for: twice what the points ask for - the padded box's empty keys take buckets
// too, and a bucket must not grow past what one workgroup holds because of them - at least 64, at most 4096
__host__ __device__ inline uint64_t geom_bucket_cap(uint64_t want) {
  const uint64_t c = 2 * want;
  return c < 64 ? 64 : (c > (uint64_t)(1u << 12) ? (uint64_t)(1u << 12) : c);
}
// the true box widened by the margin on every side (indices stay far inside int32: |q| <= 2^30, margin <= 64)
__host__ __device__ inline void geom_pad_box(const int32_t* tb, int margin, int32_t* pb) {
  for (int a = 0; a < 3; ++a) {
    pb[a] = tb[a] - margin;
    pb[3 + a] = tb[3 + a] + margin;
  }
}

// Geometry of a single pass for the cloud whose TRUE voxel box is tb (valid = 1), or the reason why this is not a
// single-pass case.  Round 6: the keys are linearised over the box PADDED by ask.margin voxels on every side and a
// bucket is a run of `width` consecutive keys for any width (not a power of two), so that
//   * a geometry is a function of the true box alone - the same on the host (the next build's hint) and on the
//     device (k_bucket_geom, the validation of a hint);
//   * a scene whose box drifts by up to `margin` voxels between two scans fits the geometry of the previous scan:
//     the hinted histogram pass finds the true box on its way and no scan pays a box pass of its own
//     (grid.py:72-90 rebuckets from scratch on every call - there is no state to carry; here the state is a hint
//     that the build verifies);
//   * the width follows the points: about ask.target points per bucket of a box that is full, and never more
//     buckets than the tables hold (geom_bucket_cap).
__host__ __device__ inline void geom_from_box(const int32_t* tb, bool domain_error, const GeomAsk& ask,
                                              const LinParams& base, GeomDev& o) {
  o.lp = base;
  for (int a = 0; a < 6; ++a) o.tb[a] = o.bb[a] = tb[a];
  o.valid = 0;
  o.reason = 0;
  if (domain_error) {
    o.reason = GEOM_DOMAIN;
  } else if (tb[0] > tb[3]) {
    o.reason = GEOM_EMPTY;
  } else {
    geom_pad_box(tb, ask.margin, o.bb);
    const uint64_t nx = (uint64_t)((int64_t)o.bb[3] - o.bb[0] + 1), ny = (uint64_t)((int64_t)o.bb[4] - o.bb[1] + 1),
                   nz = (uint64_t)((int64_t)o.bb[5] - o.bb[2] + 1);
    if (nx * ny > (1ull << 32) || nx * ny * nz >= (1ull << 32)) {
      o.reason = GEOM_RETRY;  // (the host path decides: keys would not fit 32 bits)
    } else {
      const uint64_t R = nx * ny * nz;
      const uint64_t Rt = (uint64_t)((int64_t)tb[3] - tb[0] + 1) * (uint64_t)((int64_t)tb[4] - tb[1] + 1) *
                          (uint64_t)((int64_t)tb[5] - tb[2] + 1);
      // (the host sizes its tables for as many buckets, see bucket_build_impl)
      const uint64_t cap = geom_bucket_cap(ask.want);
      // keys per bucket: `target` points in a box that is full ...
      const uint64_t n = ask.n_alive > 0 ? (uint64_t)ask.n_alive : 1;
      uint64_t w = ((uint64_t)ask.target * Rt + n / 2) / n;
      if (w < 1) w = 1;
      // ... and no more buckets than there is room for
      const uint64_t wc = (R + cap - 1) / cap;
      if (wc > w) w = wc;
      if (w > (uint64_t)(1u << 12)) {
        o.reason = GEOM_RETRY;  // a sparse scene: more buckets than the single pass has room for
      } else {
        o.lp.minx = o.bb[0];
        o.lp.miny = o.bb[1];
        o.lp.minz = o.bb[2];
        o.lp.ny = (uint32_t)ny;
        o.lp.nz = (uint32_t)nz;
        o.lp.width = (uint32_t)w;
        o.lp.winv = 1.0 / (double)w;
        o.lp.whalf = 0.5 * o.lp.winv;
        o.lp.shift = geom_ceil_log2(w);
        o.lp.dshift = o.lp.shift;
        o.lp.dmask = 0xFFFFFFFFu;
        o.lp.raw_vp = 0;
        o.valid = 1;
      }
    }
  }
}

__global__ void k_bucket_geom(const int32_t* __restrict__ bbox, GeomAsk ask, LinParams base,
                              GeomDev* __restrict__ g) {
  if (threadIdx.x != 0) return;
  GeomDev o;
  geom_from_box(bbox, bbox[6] != 0, ask, base, o);
  *g = o;
}

// Hinted geometry.  A cloud that was taken in place (octl_forest_add_pose_adopt, a routed cloud) has not been
// through the box pass of the ingest kernel.  When the context has the true box of its previous single-pass
// build (a SLAM loop feeds scans of the same scene), the histogram pass runs under the geometry of THAT box and
// finds this cloud's true box on the way (k_part_hist<true>); the validation keeps the hint when every point fell
// inside its (padded) box and its buckets are within a factor of two of what this cloud's own geometry would use
// - otherwise it writes the geometry of the true box with valid = 0 / reason = GEOM_REHASH, every later kernel of
// the build returns at once, and the host runs the build again from the (now known) box.  A miss costs one
// histogram pass and one round trip; a scene that moves by up to the margin per scan never misses.
__global__ void k_geom_set(GeomDev hint, GeomDev* __restrict__ g) {
  if (threadIdx.x == 0) *g = hint;
}
// (the decision alone: does the hinted geometry of width hint_width stand for the cloud whose box the histogram
//  pass found?  o: the record to leave behind when it does not)
__device__ __forceinline__ bool geom_validate_decide(const int32_t* __restrict__ bbox, const GeomAsk& ask,
                                                     const LinParams& base, uint32_t hint_width, GeomDev& o) {
  geom_from_box(bbox, bbox[6] != 0, ask, base, o);
  const bool inside = bbox[7] == 0;  // no point outside the hint's box
  if (o.valid && inside && hint_width <= 2u * o.lp.width && o.lp.width <= 2u * hint_width) return true;
  if (o.valid) {
    o.valid = 0;
    o.reason = GEOM_REHASH;
  }
  return false;
}
__device__ __forceinline__ void geom_validate_body(const int32_t* __restrict__ bbox, const GeomAsk& ask,
                                                   const LinParams& base, GeomDev* __restrict__ g) {
  GeomDev o;
  if (geom_validate_decide(bbox, ask, base, g->lp.width, o)) {
    for (int a = 0; a < 6; ++a) g->tb[a] = bbox[a];  // the hint stands; the host forms the next one from this box
    return;
  }
  *g = o;
}
__global__ void k_geom_validate(const int32_t* __restrict__ bbox, GeomAsk ask, LinParams base,
                                GeomDev* __restrict__ g) {
  if (threadIdx.x == 0) geom_validate_body(bbox, ask, base, g);
}

// The same for a hinted TWO-pass build (more than 4096 buckets: 125 M points of one rank's shard).  There the host
// forms the geometry - it needs the bucket count for its tables - from the hint's (padded) box instead of waiting
// for the box pass; the hint stands when every point fell inside its box and the true box, padded, asks for the same
// bucket width.  Otherwise the true box is left in the record (valid = 0, GEOM_REHASH) and the host builds again.
__device__ __forceinline__ void geom_validate2_body(const int32_t* __restrict__ bbox, const GeomAsk& ask,
                                                    GeomDev* __restrict__ g) {
  GeomDev o = *g;
  const bool domain_error = bbox[6] != 0, inside = bbox[7] == 0;
  bool same = false;
  if (!domain_error && inside && bbox[0] <= bbox[3]) {
    int32_t pb[6];
    geom_pad_box(bbox, ask.margin, pb);
    const uint64_t nx = (uint64_t)((int64_t)pb[3] - pb[0] + 1), ny = (uint64_t)((int64_t)pb[4] - pb[1] + 1),
                   nz = (uint64_t)((int64_t)pb[5] - pb[2] + 1);
    if (!(nx * ny > (1ull << 32) || nx * ny * nz >= (1ull << 32))) {
      // (the bucket width follows the TRUE box: a padded 128^3 box has more than 2^21 keys, and a width taken from
      //  that would be twice what the points ask for - the host forms it the same way)
      const uint64_t Rt = (uint64_t)((int64_t)bbox[3] - bbox[0] + 1) * (uint64_t)((int64_t)bbox[4] - bbox[1] + 1) *
                          (uint64_t)((int64_t)bbox[5] - bbox[2] + 1);
      int s = geom_ceil_log2(Rt) - geom_ceil_log2(ask.want);
      s = s < 0 ? 0 : (s > 12 ? 12 : s);
      same = s == o.lp.shift;
    }
  }
  if (same) {  // the hint stands
    for (int a = 0; a < 6; ++a) g->tb[a] = bbox[a];
    return;
  }
  for (int a = 0; a < 6; ++a) o.tb[a] = o.bb[a] = bbox[a];
  o.valid = 0;
  o.reason = domain_error ? GEOM_DOMAIN : (bbox[0] > bbox[3] ? GEOM_EMPTY : GEOM_REHASH);
  *g = o;
}
__global__ void k_geom_validate2(const int32_t* __restrict__ bbox, GeomAsk ask, GeomDev* __restrict__ g) {
  if (threadIdx.x == 0) geom_validate2_body(bbox, ask, g);
}

// ---------------------------------------------------------------------------------------------
// partition
// ---------------------------------------------------------------------------------------------
// One partitioned point: 32 bytes, written and read as two 16-byte accesses.  (Five separate stores
// per point - key, index, three coordinates - cost one partial-sector write each: measured 1.08 GB
// written for 0.32 GB of records.)
struct __attribute__((aligned(32))) PartRec {
  double x, y, z;
  uint32_t vp;    // voxel inside the bucket << 19 | first 6 child digits << 1 | bad point
  uint32_t idx;   // store index | bit 31: the pose drives the scheme
};
// The child digits are computed by the partition kernel, whose waves otherwise wait for their
// scattered stores: the arithmetic is free there, in the bucket kernel it was a third of the VALU work.
constexpr int PATH_EAGER = 6;          // child digits carried by the record; deeper ones on demand

__device__ __forceinline__ double floor_div_fast(double a, double L) {
  return L == 1.0 ? floor(a) : floor_div_exact(a, L);  // (floor_div_exact(a, 1) == floor(a))
}

// child digits of levels [l0, l1) below the cube (c, e): the exact comparisons of compute_path in
// build.hip (octree/octree.py:73-75,94-97,181-191) - the reference's floor((p - corner) / (edge / 2))
// in {0, 1} restated as comparisons on the same rounded difference.  All levels from 0 are walked
// (the corner of level l follows from the digits above it); digits below l0 are not returned.
// *bad when the point is not inside the cube at some level.
__device__ __forceinline__ uint32_t path_levels(double px, double py, double pz, double cx, double cy,
                                                double cz, double e, int l1, bool* bad) {
  uint32_t path = 0;
  double h = e / 2.0;
#pragma unroll 1
  for (int j = 0; j < l1; ++j) {
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

// top-level voxel of a point: floor((p - corner) / L) with corner = 0 (grid.py:72-76), as a compact
// linear key in lexicographic (x, y, z) order - the order of np.unique(axis=0) (grid.py:79-81)
__device__ __forceinline__ uint32_t lin_of(const LinParams& lp, double x, double y, double z) {
  if (lp.mode != 0) return 0u;
  const int qx = (int)floor_div_fast(x, lp.L), qy = (int)floor_div_fast(y, lp.L),
            qz = (int)floor_div_fast(z, lp.L);
  return ((uint32_t)(qx - lp.minx) * lp.ny + (uint32_t)(qy - lp.miny)) * lp.nz + (uint32_t)(qz - lp.minz);
}

// the same, plus the corner of the voxel as the reference's OctreeManager holds it:
// np.array(voxel_coords) = int64(q * L) (grid.py:72-76,96-105)
__device__ __forceinline__ uint32_t lin_corner_of(const LinParams& lp, double x, double y, double z,
                                                  double& cx, double& cy, double& cz) {
  if (lp.mode != 0) {
    cx = lp.c0x;
    cy = lp.c0y;
    cz = lp.c0z;
    return 0u;
  }
  const double fx = floor_div_fast(x, lp.L), fy = floor_div_fast(y, lp.L), fz = floor_div_fast(z, lp.L);
  // int64(q * L) back as a double: truncation towards zero, and a zero comes back as +0.0 (the `+ 0.0`).
  // v_trunc_f64 instead of the two emulated f64 <-> i64 conversions (~14 f64-rate instructions per axis; the
  // key arithmetic is 54 % of k_part_scatter's time).  Equal for |q * L| < 2^63 (|q| <= 2^30 here).
  cx = trunc(fx * lp.L) + 0.0;
  cy = trunc(fy * lp.L) + 0.0;
  cz = trunc(fz * lp.L) + 0.0;
  return ((uint32_t)((int)fx - lp.minx) * lp.ny + (uint32_t)((int)fy - lp.miny)) * lp.nz +
         (uint32_t)((int)fz - lp.minz);
}

constexpr int PH_THREADS = 1024;
// The table is written [supertile][digit] - one coalesced row per workgroup - and transposed for the scan
// (digit-major: the prefix of (digit, supertile) is where that supertile's share of the bucket starts) and
// back for the scatter kernel: stored digit-major directly it was 4 bytes per 128-byte line, 93 MB of
// write traffic for 12 MB of counts, and as much again when k_part_scatter fetched its column.
// BBOX: the cloud has not been through the box pass; G holds a HINTED geometry.  Points outside the hint's
// box are not counted (the flag bbox[7] invalidates the pass), the true box is reduced on the way.
// (Round 5, measured: this pass takes 70 us for 10 M points and the plain one 40 - NOT because of what it does per
//  point: without the box, without the domain test, without the inside test, with the loads two pairs ahead in three
//  fixed register sets it stays at 69-73 us.  It is the step's first reader and pays for the write-back of the dirty
//  lines the previous kernels left in the memory-side cache: tools/probes/stream_probe.hip reads a cold cloud with this
//  kernel's skeleton in 46 us behind clean reads and in 106 us behind 768 MB of writes.  The plain pass runs behind the
//  box pass, which has paid that bill.)
template <bool BBOX, bool LONE>
__global__ __launch_bounds__(PH_THREADS) void k_part_hist(const double* __restrict__ xyz,
                                                          const uint8_t* __restrict__ alive, int64_t N,
                                                          LinParams lp, const GeomDev* __restrict__ G,
                                                          uint32_t nst, uint32_t nd,
                                                          int64_t st_items, uint32_t* __restrict__ table_t,
                                                          int32_t* __restrict__ bbox) {
  __shared__ uint32_t hist[PT_BINS];
  __shared__ int s_bb[PH_THREADS / 64][8];
  int hx = 0, hy = 0, hz = 0;  // extent of the hint's box in voxels (BBOX)
  if (G) {  // geometry formed on the device (k_bucket_geom / k_geom_set)
    if (!G->valid) return;
    lp = G->lp;
    if (BBOX) {
      hx = G->bb[3] - G->bb[0] + 1;
      hy = G->bb[4] - G->bb[1] + 1;
      hz = G->bb[5] - G->bb[2] + 1;
    }
  }
  for (uint32_t d = threadIdx.x; d < nd; d += PH_THREADS) hist[d] = 0;
  __syncthreads();
  const int big = 1 << 30;
  int mn[3] = {big, big, big}, mx[3] = {-big, -big, -big};
  bool bad = false, outside = false;
  // (kernel-uniform; mode 2 only: see k_part_scatter)
  const bool exact_cube = lp.exact_digits && nonneg_integer_below_2p45(lp.L) && lp.L >= 1.0 &&
                          nonneg_integer_below_2p45(lp.c0x) && nonneg_integer_below_2p45(lp.c0y) &&
                          nonneg_integer_below_2p45(lp.c0z);
  const bool edge_pow2 = (__double_as_longlong(lp.L) & 0xFFFFFFFFFFFFFll) == 0;
  const double inv64 = 64.0 / lp.L;
  // LONE: voxel edge 1 (compile time: the test inside the loop kept the loads of the next points behind it)
  auto fdiv = [&](double v) { return LONE ? floor(v) : floor_div_exact(v, lp.L); };
  auto count = [&](double x, double y, double z, bool live) {
    double fx = 0.0, fy = 0.0, fz = 0.0;
    if (lp.mode == 0) {
      fx = fdiv(x);
      fy = fdiv(y);
      fz = fdiv(z);
    }
    if (!BBOX) {
      // (the box pass has validated the domain)
      uint32_t lin = lp.mode != 0 ? 0u
                                  : ((uint32_t)((int)fx - lp.minx) * lp.ny + (uint32_t)((int)fy - lp.miny)) * lp.nz +
                                        (uint32_t)((int)fz - lp.minz);
      if (lp.mode == 2) {
        // a big single cube partitioned by its first pm levels (build.hip: cube_prefix_build): the key is the
        // digits themselves; a point outside the cube makes the caller take the plain path (flag in bbox[0])
        bool pbad = false;
        uint32_t d18;
        if (exact_cube && coord_takes_exact_digits(x) && coord_takes_exact_digits(y) && coord_takes_exact_digits(z))
          d18 = digits18_exact(x, y, z, lp.c0x, lp.c0y, lp.c0z, lp.L, edge_pow2, inv64, &pbad);
        else
          d18 = path_levels(x, y, z, lp.c0x, lp.c0y, lp.c0z, lp.L, PATH_EAGER, &pbad) >> 3;
        lin = d18 >> (18 - 3 * lp.pm);
        bad = bad || (live && pbad);
      }
      if (live) atomicAdd(&hist[digit_of(lp, lin)], 1u);
      return;
    }
    // |f| < 2^30, not NaN / inf - on the exponent fields (integer compares): exponent < 1023 + 30
    auto small_enough = [](double f) { return (((uint32_t)__double2hiint(f) >> 20) & 0x7FFu) < 1023u + 30u; };
    const bool dom = small_enough(fx) && small_enough(fy) && small_enough(fz);
    const bool use = live && dom;
    const int qx = use ? (int)fx : 0, qy = use ? (int)fy : 0, qz = use ? (int)fz : 0;
    mn[0] = min(mn[0], use ? qx : big); mx[0] = max(mx[0], use ? qx : -big);
    mn[1] = min(mn[1], use ? qy : big); mx[1] = max(mx[1], use ? qy : -big);
    mn[2] = min(mn[2], use ? qz : big); mx[2] = max(mx[2], use ? qz : -big);
    const uint32_t ax = (uint32_t)(qx - lp.minx), ay = (uint32_t)(qy - lp.miny), az = (uint32_t)(qz - lp.minz);
    const bool in = lp.mode != 0 || (ax < (uint32_t)hx && ay < (uint32_t)hy && az < (uint32_t)hz);
    bad = bad || (live && !dom);
    outside = outside || (use && !in);
    const uint32_t lin = lp.mode != 0 ? 0u : (ax * lp.ny + ay) * lp.nz + az;
    if (use && in) atomicAdd(&hist[digit_of(lp, lin)], 1u);
  };
  const int64_t base = (int64_t)blockIdx.x * st_items;
  // two points per thread and step: 48 contiguous bytes as three 16-byte loads (the store is 16-byte aligned and
  // base is even).  Software pipelined by hand: the loads of the next pair are in flight while this one is counted.
  const int64_t lim_i = min(N, base + st_items);  // (st_items is even)
  const int64_t stride = 2 * PH_THREADS;
  auto load = [&](int64_t j, double2& a, double2& b2, double2& c, uint32_t& al) {
    const double2* s2 = reinterpret_cast<const double2*>(xyz + 3 * j);
    a = s2[0];
    b2 = s2[1];
    c = s2[2];
    al = alive ? (uint32_t)*reinterpret_cast<const uint16_t*>(alive + j) : 0x0101u;
  };
  int64_t i = base + 2 * (int64_t)threadIdx.x;
  bool have = i < lim_i && i + 1 < N;
  double2 a = double2{0, 0}, b2 = a, c = a, na = a, nb2 = a, nc = a;
  uint32_t al = 0, nal = 0;
  if (have) load(i, a, b2, c, al);
  while (have) {
    const int64_t j = i + stride;
    const bool have_next = j < lim_i && j + 1 < N;
    if (have_next) load(j, na, nb2, nc, nal);
    count(a.x, a.y, b2.x, (al & 0xFF) != 0);
    count(b2.y, c.x, c.y, (al >> 8) != 0);
    a = na;
    b2 = nb2;
    c = nc;
    al = nal;
    i = j;
    have = have_next;
  }
  // the last point of an odd number of points
  if ((N & 1) && threadIdx.x == 0 && N - 1 >= base && N - 1 < lim_i)
    count(xyz[3 * (N - 1)], xyz[3 * (N - 1) + 1], xyz[3 * (N - 1) + 2], !alive || alive[N - 1]);
  __syncthreads();
  for (uint32_t d = threadIdx.x; d < nd; d += PH_THREADS) table_t[(size_t)blockIdx.x * nd + d] = hist[d];
  if (!BBOX && lp.mode == 2 && bbox && __any(bad) && (threadIdx.x & 63) == 0) atomicExch(reinterpret_cast<uint32_t*>(bbox), 1u);
  if (BBOX) {
    // wave + block reduction, then at most six atomics per block and only when the block widens the box
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        mn[ax] = min(mn[ax], __shfl_xor(mn[ax], off));
        mx[ax] = max(mx[ax], __shfl_xor(mx[ax], off));
      }
    }
    const bool wbad = __any(bad), wout = __any(outside);
    if ((threadIdx.x & 63) == 0) {
      int* w = s_bb[threadIdx.x >> 6];
      w[0] = mn[0]; w[1] = mn[1]; w[2] = mn[2]; w[3] = mx[0]; w[4] = mx[1]; w[5] = mx[2];
      // (a flag that is up already is not written again: under a hint that does not fit, a third of all waves have
      //  points outside its box, and their same-address atomics were 90 us of a 160 us pass)
      if (wbad && __hip_atomic_load(&bbox[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
        atomicExch(reinterpret_cast<uint32_t*>(bbox + 6), 1u);
      if (wout && __hip_atomic_load(&bbox[7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
        atomicExch(reinterpret_cast<uint32_t*>(bbox + 7), 1u);
    }
    __syncthreads();
    if (threadIdx.x < 6) {
      const int ax = threadIdx.x;
      int v = s_bb[0][ax];
      for (int w = 1; w < PH_THREADS / 64; ++w) v = (ax < 3) ? min(v, s_bb[w][ax]) : max(v, s_bb[w][ax]);
      if (ax < 3) {
        if (v != big && v < __hip_atomic_load(&bbox[ax], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
          atomicMin(&bbox[ax], v);
      } else {
        if (v != -big && v > __hip_atomic_load(&bbox[ax], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
          atomicMax(&bbox[ax], v);
      }
    }
  }
}

// u32 matrix transpose through LDS tiles: in [R][C] -> out [C][R]
__global__ __launch_bounds__(256) void k_transpose_u32(const uint32_t* __restrict__ in, uint32_t R, uint32_t C,
                                                       uint32_t* __restrict__ out) {
  __shared__ uint32_t tile[64][65];
  const uint32_t c0 = blockIdx.x * 64u, r0 = blockIdx.y * 64u;
  const uint32_t tx = threadIdx.x & 63u, ty = threadIdx.x >> 6;
#pragma unroll
  for (uint32_t j = 0; j < 64; j += 4) {
    const uint32_t r = r0 + ty + j, c = c0 + tx;
    if (r < R && c < C) tile[ty + j][tx] = in[(size_t)r * C + c];
  }
  __syncthreads();
#pragma unroll
  for (uint32_t j = 0; j < 64; j += 4) {
    const uint32_t c = c0 + ty + j, r = r0 + tx;
    if (r < R && c < C) out[(size_t)c * R + r] = tile[tx][ty + j];
  }
}

// The partition table in ONE launch (round 5; before: transpose, scan, transpose back - three launches around
// 25 MB of traffic for 8 MB of counts).  table[supertile][digit] holds counts; wanted is the exclusive prefix in
// (digit, supertile) order - where supertile st's share of bucket d starts - IN PLACE and in the same layout, the
// one k_part_scatter reads its row from, plus the start of every bucket (bucket_start[d], bucket_start[nd] = total).
// A workgroup owns 64 neighbouring digit columns: lane = column, wave w = the w-th quarter of the rows; a row segment
// is 256 contiguous bytes.  Pass 1 sums the columns, the workgroups' totals are chained by decoupled look-back
// (<= 64 workgroups), pass 2 walks the rows again (L2) and writes the running offsets.
// VALIDATE: the workgroup 0 also decides about the HINTED geometry the histogram ran under (what k_geom_validate /
// k_geom_validate2 did in a launch of their own): nothing in this kernel reads the verdict, the next launch does.
constexpr int TS_COLS = 64;
constexpr int TS_THREADS = 512;
constexpr int TS_WAVES = TS_THREADS / 64;
constexpr int TS_UNROLL = 8;             // rows in flight per lane
constexpr uint32_t TS_MAX_ROWS = 2048;   // beyond: the three-launch form (a workgroup would walk its rows for too long)
__global__ __launch_bounds__(TS_THREADS) void k_table_scan(uint32_t* __restrict__ table, uint32_t nst, uint32_t nd,
                                                          uint32_t* __restrict__ bucket_start,
                                                          uint64_t* __restrict__ status, uint32_t epoch, int validate,
                                                          const int32_t* __restrict__ bbox, GeomAsk ask, LinParams base,
                                                          GeomDev* __restrict__ g) {
  __shared__ uint32_t s_w[TS_WAVES][TS_COLS];   // column sums per slice of the rows
  __shared__ uint32_t s_col[TS_COLS];           // exclusive prefix over the workgroup's columns
  __shared__ uint32_t s_tot, s_excl;
  if (validate && blockIdx.x == 0 && threadIdx.x == 0) {
    if (validate == 1) geom_validate_body(bbox, ask, base, g); else geom_validate2_body(bbox, ask, g);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t d = blockIdx.x * TS_COLS + lane;
  const bool col = d < nd;
  const uint32_t per = (nst + TS_WAVES - 1) / TS_WAVES;
  const uint32_t r0 = min(nst, (uint32_t)wave * per), r1 = min(nst, r0 + per);
  uint32_t sum = 0;
  if (col) {
    uint32_t r = r0;
    for (; r + TS_UNROLL <= r1; r += TS_UNROLL) {
      const uint32_t* p0 = table + (size_t)r * nd + d;
      uint32_t v[TS_UNROLL];
#pragma unroll
      for (int u = 0; u < TS_UNROLL; ++u) v[u] = p0[(size_t)u * nd];
#pragma unroll
      for (int u = 0; u < TS_UNROLL; ++u) sum += v[u];
    }
    for (; r < r1; ++r) sum += table[(size_t)r * nd + d];
  }
  s_w[wave][lane] = sum;
  __syncthreads();
  if (wave == 0) {
    uint32_t t = 0;
#pragma unroll
    for (int w = 0; w < TS_WAVES; ++w) t += s_w[w][lane];
    const uint32_t inc = wave_inclusive_add(t);
    s_col[lane] = inc - t;
    if (lane == 63) s_tot = inc;
  }
  __syncthreads();
  const uint32_t total = s_tot;
  const uint32_t excl = lookback_exclusive(status, epoch, blockIdx.x, total, &s_excl);
  if (!col) return;
  uint32_t run = excl + s_col[lane];
  if (wave == 0) {
    bucket_start[d] = run;
    if (d == nd - 1) bucket_start[nd] = excl + total;
  }
  for (int w = 0; w < wave; ++w) run += s_w[w][lane];   // the slices in front of this one inside the column
  uint32_t r = r0;
  for (; r + TS_UNROLL <= r1; r += TS_UNROLL) {
    uint32_t* p0 = table + (size_t)r * nd + d;
    uint32_t v[TS_UNROLL];
#pragma unroll
    for (int u = 0; u < TS_UNROLL; ++u) v[u] = p0[(size_t)u * nd];
#pragma unroll
    for (int u = 0; u < TS_UNROLL; ++u) {
      p0[(size_t)u * nd] = run;
      run += v[u];
    }
  }
  for (; r < r1; ++r) {
    const uint32_t a = table[(size_t)r * nd + d];
    table[(size_t)r * nd + d] = run;
    run += a;
  }
}

// second pass: the records of the first pass (they carry the full linear key)
struct PartRec;
__global__ __launch_bounds__(PH_THREADS) void k_part_hist_rec(const uint4* __restrict__ recs, int64_t N,
                                                              LinParams lp, const GeomDev* __restrict__ G,
                                                              uint32_t nst, uint32_t nd,
                                                              int64_t st_items, uint32_t* __restrict__ table) {
  __shared__ uint32_t hist[PT_BINS];
  if (G && !G->valid) return;  // (a hinted geometry that did not hold: lp is the host's, G only gates)
  for (uint32_t d = threadIdx.x; d < nd; d += PH_THREADS) hist[d] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * st_items;
  const int64_t lim = min(N, base + st_items);
  for (int64_t i = base + threadIdx.x; i < lim; i += PH_THREADS)
    atomicAdd(&hist[digit_of(lp, recs[2 * i + 1].z)], 1u);
  __syncthreads();
  for (uint32_t d = threadIdx.x; d < nd; d += PH_THREADS) table[(size_t)blockIdx.x * nd + d] = hist[d];
}

// -DBB_STAMPS: phase clocks of bucket_chunk (thread 0 of every workgroup adds its s_memtime deltas to a device
// array; tools/bb_stamps.py reads it through octl_debug_bb_stamps).  Experiments only: not in the shipped library.
#ifdef BB_STAMPS
__device__ unsigned long long g_bb_stamps[16];
#define BB_STAMP(k)                                                              \
  do {                                                                           \
    if (threadIdx.x == 0) {                                                      \
      const unsigned long long _t = __builtin_readcyclecounter();                \
      atomicAdd(&g_bb_stamps[k], _t - _stamp_t0);                                \
      _stamp_t0 = _t;                                                            \
    }                                                                            \
  } while (0)
#define BB_STAMP_INIT unsigned long long _stamp_t0 = __builtin_readcyclecounter()
#else
#define BB_STAMP(k) do {} while (0)
#define BB_STAMP_INIT do {} while (0)
#endif
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

// MBITS: ballot rounds of the in-wave rank = bits of the pass's digit (12 for a single pass over up to 4096 buckets,
// 8 when a pass of a two-pass partition has at most 256 digits: a third of the rounds less)
// OWN_SCAN (round 6, small clouds: <= 256 digits, <= PS_OWN_MAX_ROWS supertiles): the table is the RAW one of the
// histogram pass and every workgroup forms its own row of the scanned one - column totals over all rows, their
// exclusive scan over the digits, plus the rows in front of its own - and decides about a hinted geometry itself
// (the same pure function on the same box: the same verdict everywhere); workgroup 0 also leaves behind what
// k_table_scan left: the buckets' starts and the verdict / the true box in the geometry record.  k_table_scan was
// 6.6 us of a 100 k-point scan in front of this kernel for a table of 50 rows.
struct ScatterOwnScan {
  uint32_t* bucket_start;
  int validate;            // 0 / 1 (geom_validate_decide)
  const int32_t* bbox;
  GeomAsk ask;
  LinParams base;
  GeomDev* g;
};
constexpr uint32_t PS_OWN_MAX_ROWS = 64;
template <int PT_IPT, bool FROM_REC, int MBITS, bool OWN_SCAN = false>
__global__ __launch_bounds__(PT_THREADS, MBITS <= 8 ? 3 : 2) void k_part_scatter(
    const double* __restrict__ xyz, const uint8_t* __restrict__ alive, int64_t N, LinParams lp,
    const GeomDev* __restrict__ G, uint32_t nst, uint32_t nd, int st_tiles,
    const uint32_t* __restrict__ table_scanned,
    const int64_t* __restrict__ pose_off, int n_poses, const uint8_t* __restrict__ scheme,
    PartRec* __restrict__ out, ScatterOwnScan own) {
  // (sized by the pass's digit: 48 KB for the 4096 digits of a single pass, 3 KB for the <= 256 of a two-pass
  //  partition - whose tiles then also zero and scan a sixteenth of the counters)
  constexpr int NB = 1 << MBITS;
  static_assert(NB % PT_THREADS == 0 && NB <= PT_BINS, "digits per thread in the offset scan");
  __shared__ uint32_t base[NB];                 // running destination of every bucket
  __shared__ uint16_t cnt[PT_THREADS / 64][NB]; // per wave, per tile
  if (G) {
    if (!G->valid) return;
    if (!FROM_REC) lp = G->lp;  // (the second pass of a two-pass partition runs under the host's digit parameters)
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if constexpr (OWN_SCAN) {
    static_assert(NB == PT_THREADS, "one digit column per thread");
    __shared__ uint32_t s_ws[PT_THREADS / 64];
    __shared__ int s_stands;
    // (the rows first: their loads are in flight while thread 0 decides about the hint - 64-bit divisions)
    const uint32_t d = threadIdx.x;
    uint32_t tot = 0, before = 0;
    if (d < nd) {
      // (all of a column's rows in flight at once: 4 batches of 16 loads)
      for (uint32_t r0 = 0; r0 < nst; r0 += 16) {
        uint32_t v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = r0 + u < nst ? table_scanned[(size_t)(r0 + u) * nd + d] : 0u;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          tot += v[u];
          before += r0 + u < blockIdx.x ? v[u] : 0u;
        }
      }
    }
    if (own.validate && threadIdx.x == PT_THREADS - 1) {
      GeomDev o;
      const bool stands = geom_validate_decide(own.bbox, own.ask, own.base, lp.width, o);
      s_stands = stands ? 1 : 0;
      if (blockIdx.x == 0) {
        if (stands) {
          for (int a = 0; a < 6; ++a) own.g->tb[a] = own.bbox[a];
        } else {
          *own.g = o;   // (the other workgroups reach the same verdict themselves and return)
        }
      }
    }
    const uint32_t inc = wave_inclusive_add(tot);
    if (lane == 63) s_ws[wave] = inc;
    __syncthreads();
    if (own.validate && !s_stands) return;
    uint32_t excl = inc - tot;
    for (int w = 0; w < wave; ++w) excl += s_ws[w];
    if (d < nd) {
      base[d] = excl + before;
      if (blockIdx.x == 0) {
        own.bucket_start[d] = excl;
        if (d == nd - 1) own.bucket_start[nd] = excl + tot;
      }
    }
  } else {
    for (uint32_t d = threadIdx.x; d < nd; d += PT_THREADS)
      base[d] = table_scanned[(size_t)blockIdx.x * nd + d];  // ([supertile][digit]: see k_part_hist)
  }
  constexpr int PT_TILE = PT_THREADS * PT_IPT;
  // (kernel-uniform) the cubes the digits are taken against have integer-valued, non-negative corners and edges:
  // always in a grid (corner = q L, edge = L, L integer) for a non-negative point; a single cube when its own
  // corner and edge are; OCTL_NO_EXACT_DIGITS (host: lp.exact_digits) switches the short form off for tests
  const bool exact_cube = lp.exact_digits && nonneg_integer_below_2p45(lp.L) && lp.L >= 1.0 &&
                          (lp.mode == 0 || (nonneg_integer_below_2p45(lp.c0x) && nonneg_integer_below_2p45(lp.c0y) &&
                                            nonneg_integer_below_2p45(lp.c0z)));
  const bool edge_pow2 = (__double_as_longlong(lp.L) & 0xFFFFFFFFFFFFFll) == 0;
  const double inv64 = 64.0 / lp.L;
  BB_STAMP_INIT;
  for (int t = 0; t < st_tiles; ++t) {
    const int64_t tbase = ((int64_t)blockIdx.x * st_tiles + t) * PT_TILE;
    if (tbase >= N) break;
    for (uint32_t d = threadIdx.x; d < nd; d += PT_THREADS) {
#pragma unroll
      for (int w = 0; w < PT_THREADS / 64; ++w) cnt[w][d] = 0;
    }
    __syncthreads();
    BB_STAMP(9);   // scatter: counters reset
    // wave w owns items [tbase + w * 64 * PT_IPT, + 64 * PT_IPT) in PT_IPT rounds of 64 consecutive items: stream order
    // == memory order, so the partition is stable
    const int64_t wbase = tbase + (int64_t)wave * (64 * PT_IPT);
    double x[PT_IPT], y[PT_IPT], z[PT_IPT];
    uint32_t lin[PT_IPT], rank[PT_IPT], pbits[PT_IPT], idxv[PT_IPT], dig[PT_IPT];
    uint8_t live[PT_IPT];
    // every load of the tile is issued before the first use: a load behind `if (alive[i])` waits for
    // the flag first, and 16 rounds of two dependent HBM latencies were 50 us per 4096-record tile
#pragma unroll
    for (int r = 0; r < PT_IPT; ++r) {
      const int64_t i = min(wbase + r * 64 + lane, N - 1);
      if (FROM_REC) {
        const uint4* q = reinterpret_cast<const uint4*>(xyz) + 2 * i;
        const uint4 a = q[0], b = q[1];
        x[r] = __longlong_as_double((long long)(((uint64_t)a.y << 32) | a.x));
        y[r] = __longlong_as_double((long long)(((uint64_t)a.w << 32) | a.z));
        z[r] = __longlong_as_double((long long)(((uint64_t)b.y << 32) | b.x));
        lin[r] = b.z;
        idxv[r] = b.w;
        live[r] = 1;
      } else {
        live[r] = alive ? alive[i] : (uint8_t)1;
        x[r] = xyz[3 * i];
        y[r] = xyz[3 * i + 1];
        z[r] = xyz[3 * i + 2];
        idxv[r] = 0;
      }
    }
#ifdef BB_STAMPS
    __builtin_amdgcn_s_waitcnt(0);  // (experiments: the loads' latency apart from the keys and ranks)
    BB_STAMP(15);  // scatter: loads arrived
#endif
#pragma unroll
    for (int r = 0; r < PT_IPT; ++r) {
      const int64_t i = wbase + r * 64 + lane;
      const bool valid = i < N && live[r];
      if (!FROM_REC) lin[r] = 0;
      pbits[r] = 0;
      if (valid) {
        double cx, cy, cz;
        const uint32_t l = lin_corner_of(lp, x[r], y[r], z[r], cx, cy, cz);
        if (!FROM_REC) lin[r] = l;
        if (!lp.raw_vp) {
          bool bad = false;
          uint32_t d18;
          // non-negative coordinates below an integer cube: the six rounded subtractions per axis of the
          // reference are exact and the digits are the leading bits of p - corner (ref_arith.h: digits18_exact);
          // anything else (negative coordinates, a cube at a fractional corner) walks the levels
          if (exact_cube && coord_takes_exact_digits(x[r]) && coord_takes_exact_digits(y[r]) &&
              coord_takes_exact_digits(z[r])) {
            d18 = digits18_exact(x[r], y[r], z[r], cx, cy, cz, lp.L, edge_pow2, inv64, &bad);
          } else {
            d18 = path_levels(x[r], y[r], z[r], cx, cy, cz, lp.L, PATH_EAGER, &bad) >> 3;
          }
          pbits[r] = (d18 << 1) | (bad ? 1u : 0u);
          if (!FROM_REC && lp.mode == 2) lin[r] = d18 >> (18 - 3 * lp.pm);  // (the key IS the first pm digits)
        }
      }
      dig[r] = digit_of(lp, lin[r]);
      rank[r] = wave_rank_u16<MBITS>(dig[r], valid, cnt[wave]) | (valid ? 0x80000000u : 0u);
    }
    BB_STAMP(10);  // scatter: loads, keys, ranks (wave 0)
    __syncthreads();
    BB_STAMP(11);  // scatter: wait for the other waves
    // per bucket: exclusive offsets of the waves inside this tile; the tile's total moves the running
    // base once every item is placed
    uint32_t tile_tot[NB / PT_THREADS];
#pragma unroll
    for (int q = 0; q < NB / PT_THREADS; ++q) {
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
    BB_STAMP(12);  // scatter: wave offsets
#pragma unroll
    for (int r = 0; r < PT_IPT; ++r) {
      if (rank[r] >> 31) {
        const int64_t i = wbase + r * 64 + lane;
        const uint32_t d = dig[r];
        const uint32_t dst = base[d] + cnt[wave][d] + (rank[r] & 0x7FFFFFFFu);
        uint32_t v = (uint32_t)i;
        if (FROM_REC) {
          v = idxv[r];
        } else if (scheme) {
          int lo = 0, hi = n_poses;
          while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pose_off[mid] <= i) lo = mid; else hi = mid;
          }
          if (scheme[lo]) v |= 0x80000000u;
        } else {
          v |= 0x80000000u;
        }
        uint4* o = reinterpret_cast<uint4*>(out + dst);
        const uint64_t xb = (uint64_t)__double_as_longlong(x[r]), yb = (uint64_t)__double_as_longlong(y[r]),
                       zb = (uint64_t)__double_as_longlong(z[r]);
        o[0] = uint4{(uint32_t)xb, (uint32_t)(xb >> 32), (uint32_t)yb, (uint32_t)(yb >> 32)};
        // the voxel's position inside its bucket (a single pass: d IS the bucket; records of a first pass carry lin)
        const uint32_t vl = lp.winv != 0.0 ? lin[r] - d * lp.width : lin[r] & ((1u << lp.shift) - 1u);
        o[1] = uint4{(uint32_t)zb, (uint32_t)(zb >> 32), lp.raw_vp ? lin[r] : ((vl << 19) | pbits[r]), v};
      }
    }
    BB_STAMP(13);  // scatter: stores issued
    __syncthreads();
    BB_STAMP(14);  // scatter: barrier behind the stores
#pragma unroll
    for (int q = 0; q < NB / PT_THREADS; ++q) {
      const uint32_t d = q * PT_THREADS + threadIdx.x;
      if (d < nd) base[d] += tile_tot[q];
    }
    // (the next tile reads base only after the barrier that follows its counter reset)
  }
}

// first record of every bucket after a two-pass partition (one pass: the scanned table has them): bstart[b] for b
// in [0, nb] = the first record whose bucket is >= b, by binary search over the sorted records' own coordinates.
// (Round 3 streamed all records through one thread each to find the boundaries: 4 GB of reads for 65 537 numbers,
//  0.66 ms at 125 M points; the search reads ~27 records per boundary.)
__global__ __launch_bounds__(256) void k_bucket_bounds(const uint4* __restrict__ recs, uint32_t n,
                                                       LinParams lp, const GeomDev* __restrict__ G, uint32_t nb,
                                                       uint32_t* __restrict__ bstart) {
  const uint32_t bq = blockIdx.x * 256u + threadIdx.x;
  if (bq > nb || (G && !G->valid)) return;
  auto bucket_at = [&](uint32_t j) {
    const uint4 a = recs[2 * (size_t)j], b = recs[2 * (size_t)j + 1];
    return bucket_of(lp, lin_of(lp, __longlong_as_double((long long)(((uint64_t)a.y << 32) | a.x)),
                                __longlong_as_double((long long)(((uint64_t)a.w << 32) | a.z)),
                                __longlong_as_double((long long)(((uint64_t)b.y << 32) | b.x))));
  };
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (bucket_at(mid) < bq) lo = mid + 1; else hi = mid;
  }
  bstart[bq] = lo;
}

// ---------------------------------------------------------------------------------------------
// helpers of the bucket kernel
// ---------------------------------------------------------------------------------------------
// exclusive prefix of one value per thread over the NT-thread block; *total = block sum.
// scratch: >= NT / 64 words of LDS; two barriers inside.
template <int NT = 256>
__device__ __forceinline__ uint32_t block_excl_add(uint32_t v, uint32_t* total, uint32_t* scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t inc = wave_inclusive_add(v);
  if (lane == 63) scratch[wave] = inc;
  __syncthreads();
  uint32_t basev = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) {
    const uint32_t s = scratch[w];
    if (w < wave) basev += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return basev + inc - v;
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
  uint32_t bstride;   // first record of bucket b = bstart[b * bstride]
  uint32_t nb;        // buckets
  uint32_t n_alive;
  int n_poses;
  int all_scheme;
};

__device__ __forceinline__ void load_rec(const PartRec* __restrict__ p, double& x, double& y, double& z,
                                         uint32_t& lin, uint32_t& idx) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 a = q[0], b = q[1];
  x = __longlong_as_double((long long)(((uint64_t)a.y << 32) | a.x));
  y = __longlong_as_double((long long)(((uint64_t)a.w << 32) | a.z));
  z = __longlong_as_double((long long)(((uint64_t)b.y << 32) | b.x));
  lin = b.z;
  idx = b.w;
}

// ---------------------------------------------------------------------------------------------
// one workgroup per bucket
// ---------------------------------------------------------------------------------------------
// The subdivision of every voxel of the bucket (OctreeNode.subdivide, octree/octree.py:20-32: a node
// splits while its scheme-pose count exceeds K) is found WITHOUT moving points: level by level the
// still undecided points add themselves to an LDS histogram over (node ordinal, child digit); bins
// above K become the nodes of the next level and get ordinals from a block scan.  Nothing is sorted
// until every point knows its leaf depth d; then ONE stable LDS radix sort by (voxel, first d digits)
// puts the bucket in the final order - voxel, leaf path, insertion order inside the leaf - and the
// outputs are written with coalesced stores.
constexpr int BB_BINS = 8192;          // histogram bins per level (nodes of a level x 8)
#define BB_OUT_UNROLL 4
constexpr int BB_SORT_BITS = 9;        // digit of the in-bucket radix sort
constexpr int BB_SORT_BINS = 1 << BB_SORT_BITS;
constexpr int BB_SORT_DPT = BB_SORT_BINS / BB_THREADS;  // sort digits per thread in the offset scan
static_assert(BB_SORT_DPT * BB_THREADS == BB_SORT_BINS && BB_SORT_DPT >= 1, "whole sort digits per thread");
constexpr uint32_t NOT_OVER = 0xFFFFFFFFu;

// One chunk of a bucket: n <= BB_CAP points (a run of whole voxels), records part[SRC[i]] (CHUNKED)
// or part[i]; outputs at positions out_base + [0, n), voxel staging records from vox_stage on.
// Returns 0 or the BF_* flags that send the build down the general path.
template <bool CHUNKED>
__device__ __forceinline__ uint32_t bucket_chunk(
    const PartRec* __restrict__ part, const uint16_t* __restrict__ SRC, const int n, const uint32_t out_base,
    const uint32_t vox_stage, const uint32_t lin0, const BkParams& P, const int64_t* __restrict__ pose_off,
    uint32_t* __restrict__ ord_idx, double* __restrict__ xyz_ord, uint32_t* __restrict__ leafinfo,
    uint32_t* __restrict__ bk_vox, uint32_t* __restrict__ bk_node, const uint32_t node_stage, const uint32_t node_room,
    uint32_t* s_bins, uint16_t (*s_slot)[BB_CAP], uint16_t (*s_cnt)[BB_SORT_BINS],
    uint32_t* s_scr, uint32_t* s_tot, uint32_t* s_base, uint32_t* s_todo, uint32_t* __restrict__ small,
    uint32_t* s_idx) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int s = P.lp.shift;
  BB_STAMP_INIT;
  // ordinals are counted per chunk; the bucket-wide ones add what the earlier chunks of the bucket hold
  // (s_tot is stable here: the previous chunk ended with a barrier, this one updates it behind barriers)
  if (tid < BK_ROWS) s_base[tid] = s_tot[tid];
  __syncthreads();
  const uint32_t basev = s_base[BK_NVOX];
  uint32_t rec_base = 0;
#pragma unroll
  for (int l = 0; l < BB_LEVELS; ++l) rec_base += s_base[BK_NINT + l];
  const uint32_t* basel = s_base + BK_NINT;
  // vord << 18 | path prefix of the overfull nodes of the previous / the current level, kept by the owners
  // of the histogram bins (the sort buffers are free while the levels run)
  uint32_t* s_ninfo = reinterpret_cast<uint32_t*>(&s_slot[0][0]);   // [2][1024]
  bool rec_overflow = false;
  // s_todo: one bit per voxel of the bucket - the voxel cannot be finished here (a point outside its
  // cube, a tree deeper than the digits the records carry, more nodes per level than the histogram
  // holds).  Such a voxel stays ONE leaf (its root, points in insertion order) and is flagged in its
  // staging record; the host lets the level loop of build.hip subdivide exactly those voxels.
  // (the bitmap is zero on entry: cleared by the kernel's prologue and again at the end of every chunk)
  const uint32_t snap = tid < BK_ROWS ? s_tot[tid] : 0u;  // totals before this chunk (for the second attempt)
  // wave w owns items [w * per_wave, (w+1) * per_wave) in `rounds` rounds of 64 consecutive items:
  // item order == (wave, round, lane) order, which the stable ranks below rely on
  const int rounds = (n + BB_THREADS - 1) / BB_THREADS;  // <= 16
  const int per_wave = rounds * 64;

  // ---- 1. points -> (voxel inside the bucket, child digits) ----------------------------------------------
  // pth: bits 0..20 child digits (level j at bits 20-3j..18-3j), bit 30 bad point, bit 31 scheme pose
  // stt: bit 31 undecided, bits 16..27 ordinal of the point's voxel among the voxels of the chunk;
  //      undecided: bits 0..15 ordinal of the point's node among the overfull nodes of the current level;
  //      decided: bits 28..30 leaf depth d, bits 0..15 ordinal of the leaf's PARENT among the overfull nodes
  //      of level d - 1 (d = 0: the voxel ordinal again)
  uint32_t pth[BB_IPT], vlv[BB_IPT], stt[BB_IPT];
  // the records' coordinates: read once, held to the output - for the first BB_KEEP rounds; a bucket of more
  // than BB_KEEP * BB_THREADS points (the average bucket holds 60 % of BB_CAP) reads the coordinates of the
  // rest again at the output, when the pyramid's registers are free (held through, they spill: 132 B of
  // scratch per lane were +166 MB of reads and +197 MB of writes per 10 M points)
  double cxr[BB_KEEP], cyr[BB_KEEP], czr[BB_KEEP];
  bool bad_any = false;
#pragma unroll
  for (int r = 0; r < BB_IPT; ++r) {
    pth[r] = 0;
    vlv[r] = 0;
    stt[r] = 0;
    if (r < BB_KEEP) cxr[r] = cyr[r] = czr[r] = 0.0;
    if (r < rounds) {
      const int i = wave * per_wave + r * 64 + lane;
      if (i < n) {
        // (16-byte loads: the second half of the record holds voxel, child digits and index)
        const uint4* q4 = reinterpret_cast<const uint4*>(part + (CHUNKED ? (int)SRC[i] : i));
        const uint4 w4 = q4[1];
        if (r < BB_KEEP) {
          const uint4 a4 = q4[0];
          cxr[r] = __longlong_as_double((long long)(((uint64_t)a4.y << 32) | a4.x));
          cyr[r] = __longlong_as_double((long long)(((uint64_t)a4.w << 32) | a4.z));
          czr[r] = __longlong_as_double((long long)(((uint64_t)w4.y << 32) | w4.x));
        }
        s_idx[i] = w4.w;
        const uint2 w = uint2{w4.z, w4.w};
        const bool bad = w.x & 1u;
        pth[r] = (((w.x >> 1) & 0x3FFFFu) << 3) | (w.y & 0x80000000u) | (bad ? 0x40000000u : 0u);
        vlv[r] = w.x >> 19;
        stt[r] = 0x80000000u;
        bad_any = bad_any || bad;
      }
    }
  }
  BB_STAMP(0);  // records' tails loaded
  // (after the loop: a conditional LDS access between the loads serialises them)
  if (__any(bad_any)) {
#pragma unroll
    for (int r = 0; r < BB_IPT; ++r)
      if (pth[r] & 0x40000000u) atomicOr(&s_todo[vlv[r] >> 5], 1u << (vlv[r] & 31u));
  }
  int dmax = 0;
  bool again = false;   // some voxel turned out to be unfinishable here: second attempt without it
  // (a lambda instantiated twice rather than a loop: the second attempt is rare and a loop back-edge
  //  over the per-item register arrays cost 20 % of the kernel)
  auto run_levels = [&]() {
  again = false;
  rec_overflow = false;

  // ---- 2. level 0: the voxels ------------------------------------------------------------------------------------
  // bin = voxel inside the bucket; low half: scheme-pose points, high half: all points
  const int nbins0 = 1 << s;
  for (int d = tid; d < nbins0; d += BB_THREADS) s_bins[d] = 0;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < BB_IPT; ++r)
    if (stt[r] >> 31) atomicAdd(&s_bins[vlv[r]], 0x10000u + (pth[r] >> 31));
  __syncthreads();
  uint32_t n_over;
  {
    const int per = (nbins0 + BB_THREADS - 1) / BB_THREADS;
    auto is_over = [&](int d, uint32_t c) {
      return P.K >= 0 && (int64_t)(c & 0xFFFFu) > P.K && !((s_todo[d >> 5] >> (d & 31)) & 1u);
    };
    uint32_t mine = 0;  // low half: voxels, high half: overfull voxels
#pragma unroll 1
    for (int q = 0; q < per; ++q) {
      const int d = tid * per + q;
      if (d < nbins0) {
        const uint32_t c = s_bins[d];
        mine += ((c >> 16) ? 1u : 0u) + (is_over(d, c) ? 0x10000u : 0u);
      }
    }
    uint32_t tot;
    uint32_t run = block_excl_add<BB_THREADS>(mine, &tot, s_scr);
#pragma unroll 1
    for (int q = 0; q < per; ++q) {
      const int d = tid * per + q;
      if (d < nbins0) {
        const uint32_t c = s_bins[d];
        if (c >> 16) {  // staging of the j-th voxel of this chunk: linear key, points (| todo), scheme points
          const size_t at = 3 * ((size_t)vox_stage + (run & 0xFFFFu));
          const bool todo = (s_todo[d >> 5] >> (d & 31)) & 1u;
          bk_vox[at] = lin0 + (uint32_t)d;
          bk_vox[at + 1] = (c >> 16) | (todo ? 0x80000000u : 0u);
          bk_vox[at + 2] = c & 0xFFFFu;
          run += 1u;
        }
        const bool over = is_over(d, c);
        // (run & 0xFFFF counts this voxel already when it has points)
        const uint32_t vord = (run & 0xFFFFu) - ((c >> 16) ? 1u : 0u);
        s_bins[d] = (vord << 16) | (over ? (run >> 16) : 0xFFFFu);
        if (over) {
          const uint32_t q = run >> 16;
          if (q < 1024u) s_ninfo[q] = (basev + vord) << 18;
          if (rec_base + q < node_room) {
            const size_t at = 3 * ((size_t)node_stage + rec_base + q);
            bk_node[at] = (basev + vord) << 18;
            bk_node[at + 1] = basel[0] + q;          // level 0 | ordinal inside the bucket
            bk_node[at + 2] = 0u;
          } else {
            rec_overflow = true;
          }
          run += 0x10000u;
        }
      }
    }
    n_over = tot >> 16;
    if (tid == 0) {
      s_tot[BK_NVOX] += tot & 0xFFFFu;
      s_tot[BK_NINT] += n_over;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < BB_IPT; ++r) {
    if (stt[r] >> 31) {
      const uint32_t v = s_bins[vlv[r]];
      const uint32_t vo = (v >> 16) << 16;
      stt[r] = (v & 0xFFFFu) == 0xFFFFu ? (vo | (v >> 16)) : (0x80000000u | vo | (v & 0xFFFFu));
    }
  }
  __syncthreads();

  // ---- 3. deeper levels: bins = (overfull node of the level above, child digit) -----------------------------
  dmax = 0;
  uint32_t nrec = n_over;  // node records of this chunk so far
#pragma unroll 1
  for (int l = 1; n_over > 0; ++l) {
    const int nbl = 8 * (int)n_over;
    // a node of level 6 still exceeds K (deeper than the digits the records carry), or more nodes on
    // this level than the histogram holds (tiny K, many points): the voxels of the still undecided
    // points are left to the level loop of build.hip, which has neither limit
    if (l > PATH_EAGER || nbl > BB_BINS) {
#pragma unroll
      for (int r = 0; r < BB_IPT; ++r)
        if (stt[r] >> 31) atomicOr(&s_todo[vlv[r] >> 5], 1u << (vlv[r] & 31u));
      again = true;
      break;
    }
    dmax = l;
    for (int d = tid; d < nbl; d += BB_THREADS) s_bins[d] = 0;
    __syncthreads();
    const int dsh = 18 - 3 * (l - 1);
#pragma unroll
    for (int r = 0; r < BB_IPT; ++r)
      if ((stt[r] >> 31) && (pth[r] >> 31))
        atomicAdd(&s_bins[(stt[r] & 0xFFFFu) * 8u + ((pth[r] >> dsh) & 7u)], 1u);
    __syncthreads();
    {
      const int per = (nbl + BB_THREADS - 1) / BB_THREADS;
      uint32_t mine = 0;
#pragma unroll 1
      for (int q = 0; q < per; ++q) {
        const int d = tid * per + q;
        if (d < nbl) mine += (int64_t)s_bins[d] > P.K ? 1u : 0u;
      }
      uint32_t tot;
      uint32_t run = block_excl_add<BB_THREADS>(mine, &tot, s_scr);
      const uint32_t* ninfo_up = s_ninfo + 1024 * ((l - 1) & 1);
      uint32_t* ninfo_me = s_ninfo + 1024 * (l & 1);
#pragma unroll 1
      for (int q = 0; q < per; ++q) {
        const int d = tid * per + q;
        if (d < nbl) {
          const bool over = (int64_t)s_bins[d] > P.K;
          s_bins[d] = over ? run : NOT_OVER;
          if (over) {
            // the node's voxel and path: its parent's (nodes of the level above: <= 1024) + its own digit
            const uint32_t up = ninfo_up[d >> 3];
            const uint32_t info = (up & 0xFFFC0000u) | (((up & 0x3FFFFu) << 3) | (uint32_t)(d & 7));
            if (run < 1024u) ninfo_me[run] = info;
            if (l < BB_LEVELS && rec_base + nrec + run < node_room) {
              const size_t at = 3 * ((size_t)node_stage + rec_base + nrec + run);
              bk_node[at] = info;
              bk_node[at + 1] = ((uint32_t)l << 28) | (basel[l] + run);
              bk_node[at + 2] = basel[l - 1] + (uint32_t)(d >> 3);
            } else {
              rec_overflow = true;
            }
            ++run;
          }
        }
      }
      n_over = tot;
      nrec += tot;
      if (tid == 0 && l < BB_LEVELS) s_tot[BK_NINT + l] += tot;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < BB_IPT; ++r) {
      if (stt[r] >> 31) {
        const uint32_t v = s_bins[(stt[r] & 0xFFFFu) * 8u + ((pth[r] >> dsh) & 7u)];
        stt[r] = v == NOT_OVER ? (((uint32_t)l << 28) | (stt[r] & 0x0FFFFFFFu))
                               : (0x80000000u | (stt[r] & 0x0FFF0000u) | v);
      }
    }
    __syncthreads();
  }
  if (again) __syncthreads();  // (uniform) the flagged voxels are read below
  };  // run_levels
  run_levels();
  if (again) {
    if (tid < BK_ROWS) s_tot[tid] = snap;
#pragma unroll
    for (int r = 0; r < BB_IPT; ++r) stt[r] = (r < rounds && wave * per_wave + r * 64 + lane < n) ? 0x80000000u : 0u;
    __syncthreads();
    run_levels();
    if (again) return BF_OVERFLOW;  // (cannot happen: the second attempt runs without the flagged voxels)
  }
  // more internal nodes than points in the bucket (K of one or two and trees six levels deep): the staging
  // area of the node records is sized by the points
  if (__syncthreads_or(rec_overflow ? 1 : 0)) return BF_OVERFLOW;
  // voxels left to the general path (counted once, by the owners of their bins)
  {
    uint32_t c = 0;
    for (int d = tid; d < (1 << s); d += BB_THREADS) c += (s_todo[d >> 5] >> (d & 31)) & 1u;
    // (a flagged voxel of ANOTHER chunk of this bucket cannot be set here: the bitmap is per chunk)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if (lane == 0 && c) atomicAdd(&small[SM_BK_TODO], c);
  }
  BB_STAMP(1);  // level pyramid
  const int kshift = 3 * dmax;          // key = voxel << kshift | first d digits, left aligned in dmax digits
  const int kbits = s + kshift;
  if (kbits > 32) return BF_OVERFLOW;   // (s = 12 and 7 levels: cannot happen with 6)
  // ---- 4. sort keys ------------------------------------------------------------------------------------------------
  uint32_t* KEY = s_bins;
  uint32_t* INFO = s_bins + BB_CAP;
#pragma unroll
  for (int r = 0; r < BB_IPT; ++r) {
    if (r < rounds) {
      const int i = wave * per_wave + r * 64 + lane;
      if (i < n) {
        const uint32_t d = (stt[r] >> 28) & 7u;
        const uint32_t path = pth[r] & PATH_MASK;
        const uint32_t trunc = d == 0 ? 0u : ((path >> (21 - 3 * d)) << (3 * (dmax - (int)d)));
        KEY[i] = (kshift ? (vlv[r] << kshift) : vlv[r]) | trunc;
        // the leaf in bucket terms: depth, ordinal of its parent among the internal nodes of level d - 1 of
        // the bucket (d = 0: ordinal of the voxel) and its child digit - k_bucket_finish adds the bases
        const uint32_t ob = (stt[r] & 0xFFFFu) + (d == 0 ? basev : basel[d - 1]);
        INFO[i] = ob | (d == 0 ? 0u : (((path >> (21 - 3 * d)) & 7u) << LC_DIGIT)) | (d << LC_DEPTH);
      }
    }
  }
  __syncthreads();
  int cur = 0;
  if (kbits == 0) {
    for (int i = tid; i < n; i += BB_THREADS) s_slot[0][i] = (uint16_t)i;
    __syncthreads();
  }
  // stable LSD radix sort of the item ids, BB_SORT_BITS = 9 bits per pass: the common bucket - 8 voxels, leaves two
  // levels deep: 3 + 6 key bits - is sorted in ONE pass (with 8-bit digits it took two).  Per-wave digit counters
  // are 16 bits wide (<= 4096 items) and touched by their wave only: plain accesses by the digit's leader lane.
  for (int sh = 0; sh < kbits; sh += BB_SORT_BITS) {
    const bool first = sh == 0;
    for (int d = tid; d < BB_SORT_BINS; d += BB_THREADS) {
#pragma unroll
      for (int w = 0; w < BB_THREADS / 64; ++w) s_cnt[w][d] = 0;
    }
    __syncthreads();
    uint16_t rank[BB_IPT];
#pragma unroll
    for (int r = 0; r < BB_IPT; ++r) {
      rank[r] = 0;
      if (r < rounds) {
        const int i = wave * per_wave + r * 64 + lane;
        const bool valid = i < n;
        const uint32_t it = valid ? (first ? (uint32_t)i : (uint32_t)s_slot[cur][i]) : 0u;
        const uint32_t d = valid ? (KEY[it] >> sh) & (uint32_t)(BB_SORT_BINS - 1) : 0u;
        rank[r] = (uint16_t)wave_rank_u16<BB_SORT_BITS>(d, valid, s_cnt[wave]);
      }
    }
    __syncthreads();
    {
      // exclusive offsets: digits d = DPT tid ... DPT tid + DPT - 1 (all waves of a digit one after the other)
      uint32_t c[BB_SORT_DPT][BB_THREADS / 64], tot = 0;
#pragma unroll
      for (int q = 0; q < BB_SORT_DPT; ++q) {
#pragma unroll
        for (int w = 0; w < BB_THREADS / 64; ++w) {
          c[q][w] = s_cnt[w][BB_SORT_DPT * tid + q];
          tot += c[q][w];
        }
      }
      uint32_t all;
      uint32_t run = block_excl_add<BB_THREADS>(tot, &all, s_scr);
#pragma unroll
      for (int q = 0; q < BB_SORT_DPT; ++q) {
#pragma unroll
        for (int w = 0; w < BB_THREADS / 64; ++w) {
          s_cnt[w][BB_SORT_DPT * tid + q] = (uint16_t)run;
          run += c[q][w];
        }
      }
    }
    __syncthreads();
    const int dst_buf = first ? 0 : (cur ^ 1);
#pragma unroll
    for (int r = 0; r < BB_IPT; ++r) {
      if (r < rounds) {
        const int i = wave * per_wave + r * 64 + lane;
        if (i < n) {
          // (item and digit are re-read: cheaper than 32 registers held across the barriers)
          const uint32_t it = first ? (uint32_t)i : (uint32_t)s_slot[cur][i];
          const uint32_t d = (KEY[it] >> sh) & (uint32_t)(BB_SORT_BINS - 1);
          s_slot[dst_buf][(uint32_t)s_cnt[wave][d] + rank[r]] = (uint16_t)it;
        }
      }
    }
    __syncthreads();
    cur = dst_buf;
  }
  BB_STAMP(2);  // keys + sort
  const uint16_t* __restrict__ RS = s_slot[cur];

  // ---- 5. outputs (coalesced) ------------------------------------------------------------------------------------
  uint32_t nblk = 0;
  // item -> final position, in the sort buffer that is free now
  uint16_t* INV = s_slot[cur ^ 1];
  // (these two loops touch no per-item register array: rolled or nearly so - BB_LEAF_UNROLL - they leave the
  //  registers to the coordinates that are held across them)
#pragma unroll BB_LEAF_UNROLL
  for (int r = 0; r < BB_IPT; ++r) {
    const int f = r * BB_THREADS + tid;
    if (f < n) INV[RS[f]] = (uint16_t)f;
  }
  // leaf words and the permutation in sorted order (index words from LDS: nothing is read from the records again)
#pragma unroll BB_LEAF_UNROLL
  for (int r = 0; r < BB_IPT; ++r) {
    const int f = r * BB_THREADS + tid;
    if (f < n) {
      const uint32_t it = RS[f];
      const uint32_t key = KEY[it];
      const uint32_t pit = f > 0 ? (uint32_t)RS[f - 1] : 0u;
      const uint32_t pkey = f > 0 ? KEY[pit] : ~key;
      const bool leaf_head = key != pkey;
      const bool vox_head = f == 0 || (key >> kshift) != (pkey >> kshift);
      const uint32_t idx = s_idx[it] & IDX_MASK;
      bool blk_head = leaf_head;
      if (!leaf_head && P.n_poses > 1)
        blk_head = find_slot_dev(pose_off, P.n_poses, idx) != find_slot_dev(pose_off, P.n_poses, s_idx[pit] & IDX_MASK);
      nblk += blk_head ? 1u : 0u;
      const size_t o = (size_t)out_base + f;
      leafinfo[o] = INFO[it] | (vox_head ? LI_VHEAD : 0u) | (blk_head ? LI_BHEAD : 0u);
      ord_idx[o] = idx;
    }
  }
  __syncthreads();  // KEY / INFO are dead: their 32 KB become the coordinate window; INV is complete
  {
    double* XW = reinterpret_cast<double*>(s_bins);
    constexpr int W = (BB_BINS * 4) / 24;  // positions per window round (1365)
    // (the coordinates of the rounds past BB_KEEP: a full bucket only)
    constexpr int LATE = BB_IPT > BB_KEEP ? BB_IPT - BB_KEEP : 1;
    double lxr[LATE], lyr[LATE], lzr[LATE];
    lxr[0] = lyr[0] = lzr[0] = 0.0;
#pragma unroll
    for (int r = BB_KEEP; r < BB_IPT; ++r) {
      lxr[r - BB_KEEP] = lyr[r - BB_KEEP] = lzr[r - BB_KEEP] = 0.0;
      if (r < rounds) {
        const int i = wave * per_wave + r * 64 + lane;
        if (i < n) {
          const PartRec* q = part + (CHUNKED ? (int)SRC[i] : i);
          const uint4 a4 = reinterpret_cast<const uint4*>(q)[0];
          const uint2 b2 = reinterpret_cast<const uint2*>(q)[2];
          lxr[r - BB_KEEP] = __longlong_as_double((long long)(((uint64_t)a4.y << 32) | a4.x));
          lyr[r - BB_KEEP] = __longlong_as_double((long long)(((uint64_t)a4.w << 32) | a4.z));
          lzr[r - BB_KEEP] = __longlong_as_double((long long)(((uint64_t)b2.y << 32) | b2.x));
        }
      }
    }
    for (int w0 = 0; w0 < n; w0 += W) {
      const int cnt = min(W, n - w0);
#pragma unroll
      for (int r = 0; r < BB_IPT; ++r) {
        if (r < rounds) {
          const int i = wave * per_wave + r * 64 + lane;
          if (i < n) {
            const int f = (int)INV[i] - w0;
            if ((unsigned)f < (unsigned)cnt) {
              XW[3 * f] = r < BB_KEEP ? cxr[r < BB_KEEP ? r : 0] : lxr[r < BB_KEEP ? 0 : r - BB_KEEP];
              XW[3 * f + 1] = r < BB_KEEP ? cyr[r < BB_KEEP ? r : 0] : lyr[r < BB_KEEP ? 0 : r - BB_KEEP];
              XW[3 * f + 2] = r < BB_KEEP ? czr[r < BB_KEEP ? r : 0] : lzr[r < BB_KEEP ? 0 : r - BB_KEEP];
            }
          }
        }
      }
      __syncthreads();
      double* __restrict__ dst = xyz_ord + 3 * ((size_t)out_base + w0);
      for (int e = tid; e < 3 * cnt; e += BB_THREADS) dst[e] = XW[e];
      __syncthreads();
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) nblk += __shfl_xor(nblk, off);
  if (lane == 0 && nblk) atomicAdd(&s_tot[BK_NBLK], nblk);
  if (tid < (1 << PT_BITS) / 32) s_todo[tid] = 0;
  __syncthreads();  // the LDS arrays are free for the next chunk
  BB_STAMP(3);  // outputs
  return 0u;
}

// A bucket with more than BB_CAP points (an over-dense region of a skewed scene; the hash ownership of the
// multi-GPU shards puts 0...16 owned voxels into a bucket) is cut into CHUNKS: runs of whole voxels with at most
// BB_CAP points each, planned per bucket (k_bucket_plan) and then worked off by ONE WORKGROUP PER CHUNK
// (k_bucket_chunks) - round 2 had the bucket's own workgroup work its chunks off one after the other, which left a
// dozen workgroups running for milliseconds behind an otherwise finished build (sparse_scene: 3.4 of 4.2 ms).
// Chunks number their voxels, internal nodes and blocks from zero; k_bucket_finish adds the prefix over the chunks
// in front of them inside the bucket on top of the bucket's bases from the global scan.
struct __attribute__((aligned(16))) ChunkDesc {
  uint32_t bucket;
  uint32_t v_first, v_end;   // voxels (inside the bucket) [v_first, v_end)
  uint32_t n;                // points
  uint32_t cofs;             // first output position, relative to the bucket's start
  uint32_t cvox;             // voxels with points in front of the chunk inside the bucket
  uint32_t big;              // a single voxel with more than BB_CAP points: copied through as one leaf
  uint32_t pad;
};
constexpr uint32_t CK_CAP = 1u << 16;  // chunks per build (beyond: the general path)

// normal buckets: at most BB_CAP points
__global__ __launch_bounds__(BB_THREADS, BB_WGS) void k_bucket_build(
    const PartRec* __restrict__ part, const uint32_t* __restrict__ bstart, BkParams P,
    const GeomDev* __restrict__ G,
    const int64_t* __restrict__ pose_off, uint32_t* __restrict__ ord_idx, double* __restrict__ xyz_ord,
    uint32_t* __restrict__ leafinfo, uint32_t* __restrict__ bk_vox, uint32_t* __restrict__ bk_node,
    uint32_t* __restrict__ bk_tot, uint32_t* __restrict__ small, int chunks_later) {
  if (G) {
    if (!G->valid) return;
    P.lp = G->lp;
  }
  __shared__ uint32_t s_bins[BB_BINS];            // pyramid bins; afterwards KEY[BB_CAP] | INFO[BB_CAP]
  __shared__ uint32_t s_base[BK_ROWS];            // totals of the bucket before the current chunk
  __shared__ uint16_t s_slot[2][BB_CAP];          // sort buffers: positions -> item
  __shared__ uint16_t s_cnt[BB_THREADS / 64][BB_SORT_BINS];
  __shared__ uint32_t s_scr[16];
  __shared__ uint32_t s_tot[BK_ROWS];
  __shared__ uint32_t s_todo[(1 << PT_BITS) / 32];
  __shared__ uint16_t s_src[1];
  __shared__ uint32_t s_idx[BB_CAP];              // index | scheme bit of every item (the records are read once)
  const int tid = threadIdx.x;
  const uint32_t b = blockIdx.x;
  const uint32_t start = bstart[(size_t)b * P.bstride];
  const uint32_t end = (b + 1 < P.nb) ? bstart[(size_t)(b + 1) * P.bstride] : P.n_alive;
  const int n = (int)(end - start);
  if (n > BB_CAP) {
    // k_bucket_plan / k_bucket_chunks: running beside this kernel already, or (chunks_later) launched by the host
    // when the totals tell it that there are such buckets - their rows are zero until the chunks add to them
    if (chunks_later && tid < BK_ROWS) bk_tot[(size_t)tid * P.nb + b] = 0u;
    if (tid == 0) atomicAdd(&small[SM_BK_OVERFULL], 1u);
    return;
  }
  if (tid < BK_ROWS) s_tot[tid] = 0;
  if (tid < (1 << PT_BITS) / 32) s_todo[tid] = 0;
  if (n == 0) {
    if (tid < BK_ROWS) bk_tot[(size_t)tid * P.nb + b] = 0u;
    return;
  }
  __syncthreads();
  const uint32_t lin0 = bucket_lin0(P.lp, b);
  const PartRec* __restrict__ recs = part + start;
  const uint32_t fl = bucket_chunk<false>(recs, s_src, n, start, start, lin0, P, pose_off, ord_idx, xyz_ord, leafinfo,
                                          bk_vox, bk_node, start, (uint32_t)n, s_bins, s_slot, s_cnt, s_scr, s_tot,
                                          s_base, s_todo, small, s_idx);
  if (fl) {  // the host runs the general path instead
    if (tid < BK_ROWS) bk_tot[(size_t)tid * P.nb + b] = 0u;
    if (tid == 0) atomicOr(&small[SM_BK_FLAGS], fl);
    return;
  }
  __syncthreads();
  if (tid < BK_ROWS) bk_tot[(size_t)tid * P.nb + b] = s_tot[tid];
  if (tid == 0) {
    uint32_t nrec = 0;
    for (int l = 0; l < BB_LEVELS; ++l) nrec += s_tot[BK_NINT + l];
    // (k_bucket_finish ranks the internal nodes and the blocks of a bucket among themselves in LDS)
    if (nrec > FO_MAX || s_tot[BK_NBLK] > FO_MAX) atomicOr(&small[SM_BK_NOORDER], 1u);
  }
}

// plan of the buckets with more than BB_CAP points: points per voxel, then greedy runs of voxels with at most
// BB_CAP points; the chunk descriptors go to one global list (a range per bucket, claimed with one atomic)
__global__ __launch_bounds__(BB_THREADS) void k_bucket_plan(
    const PartRec* __restrict__ part, const uint32_t* __restrict__ bstart, BkParams P, const GeomDev* __restrict__ G,
    ChunkDesc* __restrict__ ck_desc, uint2* __restrict__ ck_of_bucket, uint32_t* __restrict__ bk_tot,
    uint32_t* __restrict__ small) {
  if (G) {
    if (!G->valid) return;
    P.lp = G->lp;
  }
  __shared__ uint32_t s_bins[1 << PT_BITS];
  const int tid = threadIdx.x;
  const uint32_t b = blockIdx.x;
  const uint32_t start = bstart[(size_t)b * P.bstride];
  const uint32_t end = (b + 1 < P.nb) ? bstart[(size_t)(b + 1) * P.bstride] : P.n_alive;
  const int n = (int)(end - start);
  if (n <= BB_CAP) {
    if (tid == 0) ck_of_bucket[b] = uint2{0u, 0u};
    return;
  }
  if (tid < BK_ROWS) bk_tot[(size_t)tid * P.nb + b] = 0u;  // (the chunks add their totals)
  if (n > 65535) {  // (chunk items are 16-bit positions inside the bucket)
    if (tid == 0) {
      ck_of_bucket[b] = uint2{0u, 0u};
      atomicOr(&small[SM_BK_FLAGS], BF_OVERFLOW);
    }
    return;
  }
  const int nbins0 = 1 << P.lp.shift;
  for (int d = tid; d < nbins0; d += BB_THREADS) s_bins[d] = 0;
  __syncthreads();
  const PartRec* __restrict__ recs = part + start;
  // (four loads in flight per lane: the bucket is up to 16 times the kernel's threads)
  for (int i0 = 0; i0 < n; i0 += 4 * BB_THREADS) {
    uint32_t v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * BB_THREADS + tid;
      v[u] = i < n ? reinterpret_cast<const uint4*>(recs + i)[1].z >> 19 : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (v[u] != 0xFFFFFFFFu) atomicAdd(&s_bins[v[u]], 1u);
  }
  __syncthreads();
  // one lane walks the <= 4096 bins (the cut depends on the running size) - eight bins per LDS round trip - and
  // leaves the chunks in LDS; the workgroup claims their range in the list and copies them out
  constexpr int PLAN_MAX = 64;  // (two neighbouring chunks hold more than BB_CAP points and n <= 65535: <= 34 chunks)
  __shared__ ChunkDesc s_desc[PLAN_MAX];
  __shared__ uint32_t s_plan[2];  // chunks, base in the list (0xFFFFFFFF: given up)
  if (tid == 0) {
    uint32_t c = 0, size = 0, vox = 0, cum = 0, first = 0, cofs = 0, cvox = 0;
    auto close_chunk = [&](uint32_t v_end, uint32_t big) {
      if (c < (uint32_t)PLAN_MAX) s_desc[c] = ChunkDesc{b, first, v_end, size, cofs, cvox, big, 0u};
      ++c;
      size = 0;
      first = v_end;
      cofs = cum;
      cvox = vox;
    };
    for (int d0 = 0; d0 < nbins0; d0 += 8) {
      uint32_t cn[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) cn[u] = d0 + u < nbins0 ? s_bins[d0 + u] : 0u;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t cnt = cn[u];
        const int d = d0 + u;
        if (cnt == 0) continue;
        const bool big = cnt > (uint32_t)BB_CAP;  // a single voxel beyond the LDS capacity: one leaf, copied through
        if (size > 0 && (big || size + cnt > (uint32_t)BB_CAP)) close_chunk((uint32_t)d, 0u);
        size += cnt;
        cum += cnt;
        ++vox;
        if (big) close_chunk((uint32_t)d + 1u, (uint32_t)d + 1u);
      }
    }
    if (size > 0) close_chunk((uint32_t)nbins0, 0u);
    uint32_t base = 0xFFFFFFFFu;
    if (c <= (uint32_t)PLAN_MAX) {
      base = atomicAdd(&small[SM_CK_COUNT], c);
      if (base + c > CK_CAP) base = 0xFFFFFFFFu;
    }
    if (base == 0xFFFFFFFFu) {
      ck_of_bucket[b] = uint2{0u, 0u};
      atomicOr(&small[SM_BK_FLAGS], BF_OVERFLOW);
    } else {
      ck_of_bucket[b] = uint2{base, c};
    }
    s_plan[0] = c;
    s_plan[1] = base;
    atomicOr(&small[SM_BK_NOORDER], 1u);  // (a chunked bucket's blocks are ordered by order.hip)
  }
  __syncthreads();
  if (s_plan[1] != 0xFFFFFFFFu) {
    // (a descriptor is two 16-byte words)
    const uint4* src = reinterpret_cast<const uint4*>(s_desc);
    uint4* dst = reinterpret_cast<uint4*>(ck_desc + s_plan[1]);
    for (uint32_t e = tid; e < 2u * s_plan[0]; e += BB_THREADS) dst[e] = src[e];
  }
}

// one workgroup per chunk (the grid strides over the list: its length is on the device)
__global__ __launch_bounds__(BB_THREADS, 1) void k_bucket_chunks(
    const PartRec* __restrict__ part, const uint32_t* __restrict__ bstart, BkParams P,
    const GeomDev* __restrict__ G, const ChunkDesc* __restrict__ ck_desc, uint32_t* __restrict__ ck_tot,
    const int64_t* __restrict__ pose_off, uint32_t* __restrict__ ord_idx, double* __restrict__ xyz_ord,
    uint32_t* __restrict__ leafinfo, uint32_t* __restrict__ bk_vox, uint32_t* __restrict__ bk_node,
    uint32_t* __restrict__ bk_tot, uint32_t* __restrict__ small) {
  if (G) {
    if (!G->valid) return;
    P.lp = G->lp;
  }
  __shared__ uint32_t s_bins[BB_BINS];
  __shared__ uint32_t s_base[BK_ROWS];
  __shared__ uint16_t s_slot[2][BB_CAP];
  __shared__ uint16_t s_cnt[BB_THREADS / 64][BB_SORT_BINS];
  __shared__ uint32_t s_scr[16];
  __shared__ uint32_t s_tot[BK_ROWS];
  __shared__ uint32_t s_todo[(1 << PT_BITS) / 32];
  __shared__ uint16_t s_src[BB_CAP];  // chunk item -> record of the bucket
  __shared__ uint32_t s_idx[BB_CAP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const uint32_t total = min(small[SM_CK_COUNT], CK_CAP);
  if (small[SM_BK_FLAGS] & BF_OVERFLOW) return;  // (the plan gave up: the general path takes the build)
  for (uint32_t ci = blockIdx.x; ci < total; ci += gridDim.x) {
    const ChunkDesc cd = ck_desc[ci];
    const uint32_t b = cd.bucket;
    const uint32_t start = bstart[(size_t)b * P.bstride];
    const uint32_t end = (b + 1 < P.nb) ? bstart[(size_t)(b + 1) * P.bstride] : P.n_alive;
    const int n = (int)(end - start);
    const uint32_t lin0 = bucket_lin0(P.lp, b);
    const PartRec* __restrict__ recs = part + start;
    if (tid < BK_ROWS) s_tot[tid] = 0;
    if (tid < (1 << PT_BITS) / 32) s_todo[tid] = 0;
    if (tid < 16) s_scr[tid] = 0;
    __syncthreads();
    uint32_t fl = 0;
    if (cd.big) {
      // ---- one voxel with more than BB_CAP points: copied through in insertion order as ONE leaf (its
      //      root), flagged for the level loop of build.hip --------------------------------------------------
      const uint32_t vl = cd.big - 1u;
      const uint32_t out_base = start + cd.cofs;
      uint32_t* tmp = s_bins;  // [BB_THREADS] indices of the selected records of one pass
      uint32_t basec = 0, last_idx = 0, nblk = 0, nsch = 0;
      for (int i0 = 0; i0 < n; i0 += BB_THREADS) {
        const int i = i0 + tid;
        uint4 a = uint4{0, 0, 0, 0}, bq = uint4{0, 0, 0, 0};
        if (i < n) {
          a = reinterpret_cast<const uint4*>(recs + i)[0];
          bq = reinterpret_cast<const uint4*>(recs + i)[1];
        }
        const bool sel = i < n && (bq.z >> 19) == vl;
        const uint64_t m = __ballot(sel);
        if (lane == 0) s_scr[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t off = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < BB_THREADS / 64; ++w) {
          if (w < wave) off += s_scr[w];
          tot += s_scr[w];
        }
        off += (uint32_t)__popcll(m & lanemask_lt());
        if (sel) tmp[off] = bq.w;
        __syncthreads();
        if (sel) {
          const uint32_t idx = bq.w & IDX_MASK;
          const bool first = basec + off == 0;
          bool bhead = first;
          if (!first && P.n_poses > 1) {
            const uint32_t pidx = (off > 0 ? tmp[off - 1] : last_idx) & IDX_MASK;
            bhead = find_slot_dev(pose_off, P.n_poses, idx) != find_slot_dev(pose_off, P.n_poses, pidx);
          }
          const size_t o = (size_t)out_base + basec + off;
          leafinfo[o] = 0u | (first ? LI_VHEAD : 0u) | (bhead ? LI_BHEAD : 0u);  // depth 0, voxel ordinal 0 of the chunk
          ord_idx[o] = idx;
          reinterpret_cast<uint2*>(xyz_ord + 3 * o)[0] = uint2{a.x, a.y};
          reinterpret_cast<uint2*>(xyz_ord + 3 * o)[1] = uint2{a.z, a.w};
          reinterpret_cast<uint2*>(xyz_ord + 3 * o)[2] = uint2{bq.x, bq.y};
          nblk += bhead ? 1u : 0u;
          nsch += bq.w >> 31;
        }
        if (tot > 0) last_idx = tmp[tot - 1];
        basec += tot;
        __syncthreads();
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        nblk += __shfl_xor(nblk, off);
        nsch += __shfl_xor(nsch, off);
      }
      if (lane == 0) {
        atomicAdd(&s_tot[BK_NBLK], nblk);
        atomicAdd(&s_scr[12], nsch);  // (slots 0..7 are the waves' counts)
      }
      __syncthreads();
      if (tid == 0) {
        const size_t at = 3 * ((size_t)start + cd.cvox);
        bk_vox[at] = lin0 + vl;
        bk_vox[at + 1] = basec | 0x80000000u;
        bk_vox[at + 2] = s_scr[12];
        s_tot[BK_NVOX] += 1;
        atomicAdd(&small[SM_BK_TODO], 1u);
      }
      __syncthreads();
    } else {
      // stable compaction of the chunk's records: SRC[k] = k-th record of the bucket whose voxel is in the chunk.
      // Every wave takes one contiguous stretch of the bucket and goes over it twice - count, then place behind
      // the waves in front of it - with ONE barrier in between (a barrier pair per BB_THREADS records, as before
      // round 4, made the workgroup's ~65 steps over a 33 k-point bucket a chain of barrier latencies).
      {
        constexpr int W = BB_THREADS / 64, U = 8;
        const int seg = (((n + W - 1) / W) + 63) & ~63;
        const int lo = wave * seg, hi = min(n, lo + seg);
        auto selected = [&](const int i) {
          uint32_t vl = 0xFFFFFFFFu;
          if (i < hi) vl = reinterpret_cast<const uint4*>(recs + i)[1].z >> 19;
          return i < hi && vl >= cd.v_first && vl < cd.v_end;
        };
        uint32_t mine = 0;
        for (int i0 = lo; i0 < hi; i0 += 64 * U) {
          bool sel[U];
#pragma unroll
          for (int u = 0; u < U; ++u) sel[u] = selected(i0 + u * 64 + lane);
#pragma unroll
          for (int u = 0; u < U; ++u) mine += (uint32_t)__popcll(__ballot(sel[u]));
        }
        if (lane == 0) s_scr[wave] = mine;
        __syncthreads();
        uint32_t off = 0;
#pragma unroll
        for (int w = 0; w < W; ++w)
          if (w < wave) off += s_scr[w];
        for (int i0 = lo; i0 < hi; i0 += 64 * U) {
          bool sel[U];
#pragma unroll
          for (int u = 0; u < U; ++u) sel[u] = selected(i0 + u * 64 + lane);
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const uint64_t m = __ballot(sel[u]);
            if (sel[u]) s_src[off + (uint32_t)__popcll(m & lanemask_lt())] = (uint16_t)(i0 + u * 64 + lane);
            off += (uint32_t)__popcll(m);
          }
        }
        __syncthreads();
      }
      fl = bucket_chunk<true>(recs, s_src, (int)cd.n, start + cd.cofs, start + cd.cvox, lin0, P, pose_off, ord_idx,
                              xyz_ord, leafinfo, bk_vox, bk_node, start + cd.cofs, cd.n, s_bins, s_slot, s_cnt, s_scr,
                              s_tot, s_base, s_todo, small, s_idx);
    }
    if (fl) {
      if (tid == 0) atomicOr(&small[SM_BK_FLAGS], fl);
    } else {
      __syncthreads();
      if (tid < BK_ROWS) {
        ck_tot[(size_t)ci * BK_ROWS + tid] = s_tot[tid];
        if (s_tot[tid]) atomicAdd(&bk_tot[(size_t)tid * P.nb + b], s_tot[tid]);
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// bucket totals -> numbering bases
// ---------------------------------------------------------------------------------------------
// ONE exclusive scan over the row-major table bk_tot[row][bucket] numbers everything: inside row
// BK_NVOX the prefix is the first voxel (root) of a bucket; rows BK_NINT .. BK_NINT+6 follow each other,
// so the prefix minus the value at the head of row BK_NINT is the first internal node of level l of a
// bucket in the LEVEL-MAJOR order of the level-synchronous path (all internal nodes of level 0 in voxel
// order, then level 1, ...); row BK_NBLK minus its head is the first block.  The row heads are the totals.
// (mirror: the context's pinned host block - the scalars the host waits for are stored there by this kernel, `words`
//  32-bit words of `small` from its start, so that no copy stands between the kernel and the host's wait)
__global__ void k_bucket_totals(const uint32_t* __restrict__ scanned, uint32_t nb,
                                const uint32_t* __restrict__ grand_total, uint32_t* __restrict__ small,
                                uint32_t* __restrict__ mirror, int words, uint32_t seq) {
  const int t = threadIdx.x;
  if (t == 0) small[SM_NVOX] = scanned[(size_t)BK_NINT * nb];
  if (t < BB_LEVELS) {
    const uint32_t a = scanned[(size_t)(BK_NINT + t) * nb], bq = scanned[(size_t)(BK_NINT + t + 1) * nb];
    small[SM_BK_LEVEL + t] = bq - a;
  }
  if (t == 8) small[SM_NBLOCKS] = *grand_total - scanned[(size_t)BK_NBLK * nb];
  __threadfence();
  __syncthreads();
  for (int w = t; w < words; w += (int)blockDim.x) mirror[w] = small[w];
  __threadfence_system();
  __syncthreads();
  if (t == 0) mirror_publish(mirror, MIRROR_FLAG_BUILD, seq);
}

// The scan over the per-bucket totals and k_bucket_totals in ONE launch (round 5): a single-pass look-back scan
// (2048 entries per workgroup) whose LAST workgroup to finish - a ticket in the scalar block, zeroed with it at the
// start of the build - forms the totals and copies the scalar block to the pinned mirror.  The table is 36 K entries
// for 4096 buckets: the device-scope fence in front of the ticket costs nothing here (it did on the 8 MB partition
// table).
constexpr int BT_THREADS = 256, BT_IPT = 8, BT_TILE = BT_THREADS * BT_IPT;
__global__ __launch_bounds__(BT_THREADS) void k_bucket_scan_totals(const uint32_t* __restrict__ raw,
                                                                  uint32_t* __restrict__ tot, uint32_t n, uint32_t nb,
                                                                  uint64_t* __restrict__ status, uint32_t epoch,
                                                                  uint32_t* __restrict__ small,
                                                                  uint32_t* __restrict__ mirror, int words, uint32_t seq) {
  __shared__ uint32_t s_wave[BT_THREADS / 64];
  __shared__ uint32_t s_excl, s_last;
  const uint32_t tile = blockIdx.x;
  const uint32_t i0 = tile * BT_TILE + threadIdx.x * BT_IPT;
  uint32_t x[BT_IPT];
#pragma unroll
  for (int j = 0; j < BT_IPT; ++j) x[j] = (i0 + j < n) ? raw[i0 + j] : 0u;
  uint32_t sum = 0;
#pragma unroll
  for (int j = 0; j < BT_IPT; ++j) sum += x[j];
  const uint32_t inc = wave_inclusive_add(sum);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  uint32_t pre = inc - sum, total = 0;
#pragma unroll
  for (int w = 0; w < BT_THREADS / 64; ++w) {
    if (w < wave) pre += s_wave[w];
    total += s_wave[w];
  }
  const uint32_t excl = lookback_exclusive(status, epoch, tile, total, &s_excl);
  pre += excl;
#pragma unroll
  for (int j = 0; j < BT_IPT; ++j) {
    if (i0 + j < n) tot[i0 + j] = pre;
    pre += x[j];
  }
  if (tile == gridDim.x - 1 && threadIdx.x == 0) small[SM_BK_TOTAL] = excl + total;
  // ---- the last workgroup to get here forms the totals ----------------------------------------------------------
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) s_last = (atomicAdd(&small[SM_BK_TICKET], 1u) == gridDim.x - 1) ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  // (the ticket is ready for the next launch: a build whose hinted geometry is rejected runs this kernel twice
  //  between two resets of the scalar block)
  if (threadIdx.x == 0) small[SM_BK_TICKET] = 0u;
  __threadfence();
  auto rd = [&](size_t i) { return __hip_atomic_load(&tot[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  const uint32_t grand = __hip_atomic_load(&small[SM_BK_TOTAL], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int t = threadIdx.x;
  if (t == 0) small[SM_NVOX] = rd((size_t)BK_NINT * nb);
  if (t < BB_LEVELS) small[SM_BK_LEVEL + t] = rd((size_t)(BK_NINT + t + 1) * nb) - rd((size_t)(BK_NINT + t) * nb);
  if (t == 8) small[SM_NBLOCKS] = grand - rd((size_t)BK_NBLK * nb);
  __threadfence();
  __syncthreads();
  for (int w = t; w < words; w += BT_THREADS)
    mirror[w] = __hip_atomic_load(&small[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __threadfence_system();
  __syncthreads();
  if (t == 0) mirror_publish(mirror, MIRROR_FLAG_BUILD, seq);
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
  uint32_t bstride, nb, n_alive;
  int n_poses, all_scheme, cur_epoch;
  int64_t node_cap;   // capacity of the node table (nodes); a larger table is needed -> overflow flag
  int write_pos;      // position -> leaf is only read by the level loop that finishes the voxels left behind
  int32_t* order_out;  // nullable: the blocks in the reference's listing order (single pose, one epoch)
  // previous scheme (nullptr: none): epochs of the nodes that were internal before, roots' old ids
  const int32_t* old_fc;
  const int32_t* old_epoch;
  const uint64_t* old_vcode;
  int64_t old_voxels;
  VoxOrg org;  // voxel origin of the packed keys (old_vcode)
  // Speculative launch (round 5): the kernel is enqueued BEFORE the host has seen the build's totals, so that the
  // host's wait for them runs beside it instead of in front of it.  What the host would have decided from the totals
  // is decided here from the same device words, identically for every workgroup: the geometry record is valid, nothing
  // was left to the level loop / the general path / order.hip, no bucket needs the chunk kernels that have not run,
  // and the tables the host sized from the context's previous build are large enough.  If not, every workgroup
  // returns at once and the host - which checks the same words in the mirror - launches again the ordinary way.
  const GeomDev* spec_geom;  // nullptr: an ordinary launch (lp above is final); else lp is read from this record
  int spec_chunks;           // the chunk kernels ran beside the bucket kernel
  int64_t vox_cap, blk_cap;  // capacity of vlin / order_out (entries)
};

// One workgroup per bucket, three independent sweeps (nothing is a serial chain any more: the bucket
// kernel has already numbered voxels, internal nodes and leaves INSIDE the bucket, this kernel adds the
// bases that the scan over all buckets produced):
//   roots    one thread per voxel (staging record: linear key, points, scheme points)
//   nodes    eight lanes per internal node (record: voxel, path, level, own / parent ordinal): the node's
//            first_child / epoch and its eight children (octree.py:177-191)
//   blocks   the leafinfo words in storage order: (leaf, pose) block table, and position -> leaf when
//            the level loop of build.hip is going to resume
// packed key (forest.h) of the voxel with linear key lin; root of that voxel in the previous scheme or -1
__device__ __forceinline__ int32_t old_root_of(const NodeParams& P, uint32_t lin) {
  int64_t qx = 0, qy = 0, qz = 0;
  if (P.lp.mode == 0) {
    qz = (int64_t)(lin % P.lp.nz) + P.lp.minz;
    qy = (int64_t)((lin / P.lp.nz) % P.lp.ny) + P.lp.miny;
    qx = (int64_t)(lin / (P.lp.nz * P.lp.ny)) + P.lp.minx;
  }
  const uint64_t code = vkey_pack(qx, qy, qz, P.org);
  int64_t lo = 0, hi = P.old_voxels;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (P.old_vcode[mid] < code) lo = mid + 1; else hi = mid;
  }
  return (lo < P.old_voxels && P.old_vcode[lo] == code) ? (int32_t)lo : -1;
}

// every voxel of the previous scheme must be a voxel again (they persist even without points, which this
// path cannot express): counts the ones that are not
__global__ __launch_bounds__(256) void k_old_voxels_missing(const uint64_t* __restrict__ old_vcode, int64_t old_V,
                                                            const uint64_t* __restrict__ new_vlin, int64_t new_V,
                                                            LinParams lp, int nx, VoxOrg org,
                                                            uint32_t* __restrict__ missing) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= old_V) return;
  int64_t qd[3];
  vkey_decode(old_vcode[r], org, qd);
  const int64_t qx = qd[0], qy = qd[1], qz = qd[2];
  bool found = false;
  if (lp.mode != 0) {
    found = new_V > 0;
  } else {
    const int64_t ax = qx - lp.minx, ay = qy - lp.miny, az = qz - lp.minz;
    if (ax >= 0 && ax < nx && ay >= 0 && ay < (int64_t)lp.ny && az >= 0 && az < (int64_t)lp.nz) {
      const uint64_t lin = ((uint64_t)ax * lp.ny + (uint64_t)ay) * lp.nz + (uint64_t)az;
      int64_t lo = 0, hi = new_V;
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (new_vlin[mid] < lin) lo = mid + 1; else hi = mid;
      }
      found = lo < new_V && new_vlin[lo] == lin;
    }
  }
  if (!found) atomicAdd(missing, 1u);
}

constexpr int BF_SCAN_IPT = 18, BF_SCAN_MAX = 256 * BF_SCAN_IPT;  // words of totals a finish launch scans itself
// k_bucket_finish<true>'s own scan of the totals (see there; that instance runs over few buckets and is compiled for
// 128 registers - the ordinary one keeps its 64 and its 16 bytes of scratch)
__device__ __forceinline__ void finish_own_scan(const uint32_t* __restrict__ bk_raw, uint32_t n_tot, uint32_t nb,
                                             uint32_t* s_tab, uint32_t* s_scr, bool publish,
                                             uint32_t* __restrict__ bk_scan_out, uint32_t* small,
                                             uint32_t* __restrict__ mirror, int mirror_words, uint32_t seq) {
  const int tid = threadIdx.x;
  const uint32_t i0 = (uint32_t)tid * BF_SCAN_IPT;
  uint32_t x[BF_SCAN_IPT], sum = 0;
#pragma unroll
  for (int j = 0; j < BF_SCAN_IPT; ++j) {
    x[j] = i0 + j < n_tot ? bk_raw[i0 + j] : 0u;
    sum += x[j];
  }
  const uint32_t inc = wave_inclusive_add(sum);
  if ((tid & 63) == 63) s_scr[tid >> 6] = inc;
  __syncthreads();
  uint32_t pre = inc - sum;
  for (int w = 0; w < (tid >> 6); ++w) pre += s_scr[w];
#pragma unroll
  for (int j = 0; j < BF_SCAN_IPT; ++j) {
    if (i0 + j < n_tot) s_tab[i0 + j] = pre;
    pre += x[j];
  }
  if (tid == 255) s_tab[n_tot] = pre;
  __syncthreads();
  if (!publish) return;
  const uint32_t grand = s_tab[n_tot];
  for (uint32_t i = tid; i < n_tot; i += 256) bk_scan_out[i] = s_tab[i];
  if (tid == 0) {
    small[SM_BK_TOTAL] = grand;
    small[SM_NVOX] = s_tab[(size_t)BK_NINT * nb];
  }
  if (tid < BB_LEVELS)
    small[SM_BK_LEVEL + tid] = s_tab[(size_t)(BK_NINT + tid + 1) * nb] - s_tab[(size_t)(BK_NINT + tid) * nb];
  if (tid == 8) small[SM_NBLOCKS] = grand - s_tab[(size_t)BK_NBLK * nb];
  __threadfence();
  __syncthreads();
  for (int w = tid; w < mirror_words; w += 256)
    mirror[w] = __hip_atomic_load(&small[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __threadfence_system();
  __syncthreads();
  if (tid == 0) mirror_publish(mirror, MIRROR_FLAG_BUILD, seq);
}

#define BF_WAVES 8   // waves per SIMD asked of the compiler for k_bucket_finish: 64 VGPRs, 16 B of scratch per lane (A/B on one box: 4 waves at 128 VGPRs 0.109 ms, 5: 0.121, 6: 0.104, 8: 0.092 - the kernel is latency bound)
template <bool OWN_SCAN>
__global__ __launch_bounds__(256, OWN_SCAN ? 4 : BF_WAVES) void k_bucket_finish(
    NodePtrs nd, NodeParams P, const uint32_t* __restrict__ bstart, const uint32_t* __restrict__ bk_base,
    const uint32_t* __restrict__ grand_total, const uint32_t* __restrict__ leafinfo,
    const uint32_t* __restrict__ ord_idx, const uint32_t* __restrict__ bk_vox,
    const uint32_t* __restrict__ bk_node, const int64_t* __restrict__ pose_off,
    const ChunkDesc* __restrict__ ck_desc, const uint32_t* __restrict__ ck_tot, const uint2* __restrict__ ck_of_bucket,
    int32_t* __restrict__ pos_node, uint64_t* __restrict__ vlin,
    int32_t* __restrict__ blk_node, int32_t* __restrict__ blk_slot, uint32_t* __restrict__ blk_start,
    int32_t* __restrict__ blk_size, uint32_t* small,
    const uint32_t* __restrict__ bk_raw, uint32_t* __restrict__ bk_scan_out, uint32_t* __restrict__ mirror,
    int mirror_words, uint32_t seq) {
  // (dynamic: only a launch that scans the totals itself - bk_raw != nullptr - asks for it)
  extern __shared__ uint32_t s_tab[];                 // the scanned table of totals | its grand total
  __shared__ uint32_t s_scr[8];
  __shared__ unsigned long long s_bal[BB_CAP / 64];   // block-head ballots of the piece, position order
  __shared__ uint32_t s_hp[256];                      // head positions of one round (pieces beyond BB_CAP)
  // the reference's listing order of the bucket's blocks (P.order_out): preorder keys of its internal nodes,
  // (level, ordinal) -> voxel << 16 | preorder rank, block keys
  __shared__ unsigned long long s_rk[FO_MAX];
  __shared__ uint32_t s_map[FO_MAX];
  __shared__ uint32_t s_bkey[FO_MAX];
  __shared__ uint32_t lvl_off[BB_LEVELS];  // first slot of every level in s_map
  __shared__ uint32_t s_lvlf[BB_LEVELS];   // first internal node of every level of the current piece (forest-wide)
  __shared__ uint32_t s_hc[65];            // block heads per (round, wave) of the piece, then their exclusive prefix | total
  const int tid = threadIdx.x;
  const uint32_t b = blockIdx.x;
  const bool spec = P.spec_geom != nullptr;
  // A launch over few buckets scans the table of totals ITSELF (round 6: k_bucket_scan_totals was 9-12 us in front of
  // a 100 k-point scan's finish for a table of 2 304 words; up to BF_SCAN_MAX words = 512 buckets): every workgroup with points scans the raw table into
  // LDS - BF_SCAN_IPT words per thread - and workgroup 0 also writes what the scan kernel wrote: the scanned table
  // (the ordinary launch that follows a refused speculative one reads it), the build's scalars and the pinned mirror
  // the host is waiting for.
  constexpr bool own_scan = OWN_SCAN;   // (an instance of its own: the ordinary one keeps its 64 registers to itself)
  const uint32_t n_tot = (uint32_t)BK_ROWS * P.nb;
  const uint32_t bucket_start = bstart[(size_t)b * P.bstride];
  const uint32_t bucket_end = (b + 1 < P.nb) ? bstart[(size_t)(b + 1) * P.bstride] : P.n_alive;
  if constexpr (own_scan) {
    if (b != 0 && bucket_end == bucket_start) return;
    finish_own_scan(bk_raw, n_tot, P.nb, s_tab, s_scr, b == 0, bk_scan_out, small, mirror, mirror_words, seq);
  }
  // entries of the scanned table: row r, bucket b; the entry behind the last one of the table is the total
  auto at = [&](int r, uint32_t q) {
    const size_t i = (size_t)r * P.nb + q;
    if constexpr (own_scan) return s_tab[i];
    else return i < (size_t)n_tot ? bk_base[i] : *grand_total;
  };
  if (spec) {
    // (kernel-uniform; the words are final: k_bucket_scan_totals is in front of this launch, or the scan above)
    if (!P.spec_geom->valid) return;
    if (small[SM_BK_FLAGS] | small[SM_BK_TODO] | small[SM_BK_NOORDER]) return;
    if (!P.spec_chunks && small[SM_BK_OVERFULL]) return;
    const int64_t nvox_all = own_scan ? (int64_t)at(BK_NINT, 0) : (int64_t)small[SM_NVOX];
    const int64_t nblk_all = own_scan ? (int64_t)(at(BK_ROWS, 0) - at(BK_NBLK, 0)) : (int64_t)small[SM_NBLOCKS];
    if (nvox_all > P.vox_cap || nblk_all > P.blk_cap) return;
    P.lp = P.spec_geom->lp;
  }
  if (bucket_end == bucket_start) return;
  const uint32_t head_int = at(BK_NINT, 0), head_blk = at(BK_NBLK, 0);
  const int64_t V = (int64_t)head_int;                     // total of row BK_NVOX
  const int64_t n_int = (int64_t)(head_blk - head_int);    // total of the level rows
  if (V + 8 * n_int > P.node_cap) {
    if (tid == 0 && !spec) atomicOr(&small[SM_BK_FLAGS], 0x100u);  // the host grows the table and launches again
    return;
  }
  // A bucket is ONE piece, or - when it held more than BB_CAP points - the chunks k_bucket_chunks built one by
  // one: each piece numbered its voxels / internal nodes / blocks from zero, the bases below add the bucket's
  // share of the global scan and the pieces in front of it inside the bucket.
  const uint2 cko = ck_of_bucket ? ck_of_bucket[b] : uint2{0u, 0u};
  const uint32_t n_pieces = cko.y ? cko.y : 1u;
  uint32_t pre_vox = 0, pre_blk = 0, pre_lvl[BB_LEVELS];
#pragma unroll
  for (int l = 0; l < BB_LEVELS; ++l) pre_lvl[l] = 0;
  BB_STAMP_INIT;
  for (uint32_t piece = 0; piece < n_pieces; ++piece) {
  uint32_t start = bucket_start, vox_stage = bucket_start, node_stage = bucket_start;
  int n = (int)(bucket_end - bucket_start);
  uint32_t nvox, lvl_cnt[BB_LEVELS], nblk_piece = 0;
  if (cko.y) {
    const ChunkDesc cd = ck_desc[cko.x + piece];
    start = bucket_start + cd.cofs;
    vox_stage = bucket_start + cd.cvox;
    node_stage = start;
    n = (int)cd.n;
    const uint32_t* t = ck_tot + (size_t)(cko.x + piece) * BK_ROWS;
    nvox = t[BK_NVOX];
#pragma unroll
    for (int l = 0; l < BB_LEVELS; ++l) lvl_cnt[l] = t[BK_NINT + l];
    nblk_piece = t[BK_NBLK];
  } else {
    nvox = at(BK_NVOX, b + 1) - at(BK_NVOX, b);
#pragma unroll
    for (int l = 0; l < BB_LEVELS; ++l) lvl_cnt[l] = at(BK_NINT + l, b + 1) - at(BK_NINT + l, b);
  }
  const uint32_t vbase = at(BK_NVOX, b) + pre_vox;
  uint32_t nrec = 0;
#pragma unroll
  for (int l = 0; l < BB_LEVELS; ++l) nrec += lvl_cnt[l];
  const uint32_t bbase = at(BK_NBLK, b) - head_blk + pre_blk;
  // first internal node of level l of this piece, level-major over the whole forest (LDS: indexed at run time)
#pragma unroll
  for (int l = 0; l < BB_LEVELS; ++l)
    if (tid == l) s_lvlf[l] = at(BK_NINT + l, b) - head_int + pre_lvl[l];
  __syncthreads();
  auto lvl_first = [&](int l) { return s_lvlf[l]; };

  // ---- roots --------------------------------------------------------------------------------------------------
  // (first position of a voxel = piece start + points of the voxels in front of it)
  uint32_t run = 0;
  for (uint32_t j0 = 0; j0 < nvox; j0 += 256) {
    const uint32_t j = j0 + tid;
    uint32_t lin = 0, cntw = 0, sc = 0;
    if (j < nvox) {
      lin = bk_vox[3 * ((size_t)vox_stage + j)];
      cntw = bk_vox[3 * ((size_t)vox_stage + j) + 1];
      sc = bk_vox[3 * ((size_t)vox_stage + j) + 2];
    }
    const uint32_t cntv = cntw & 0x7FFFFFFFu;  // (bit 31: the voxel is left to the level loop of build.hip)
    uint32_t tot;
    const uint32_t pre = block_excl_add(cntv, &tot, s_scr);
    if (j < nvox) {
      const int32_t v = (int32_t)(vbase + j);
      double c0x = P.lp.c0x, c0y = P.lp.c0y, c0z = P.lp.c0z;
      if (P.lp.mode == 0) {
        const uint32_t qz = lin % P.lp.nz, qy = (lin / P.lp.nz) % P.lp.ny, qx = lin / (P.lp.nz * P.lp.ny);
        // np.array(voxel_coordinates): int64(q * L), L integer valued (grid.py:72-76,104)
        c0x = (double)(long long)((double)((int)qx + P.lp.minx) * P.lp.L);
        c0y = (double)(long long)((double)((int)qy + P.lp.miny) * P.lp.L);
        c0z = (double)(long long)((double)((int)qz + P.lp.minz) * P.lp.L);
      }
      nd.start[v] = start + run + pre;
      nd.count[v] = cntv;
      nd.scount[v] = sc;
      nd.depth[v] = 0;
      nd.voxel[v] = v;
      nd.parent[v] = -1;
      nd.first_child[v] = -1;  // (a leaf until the sweep over the internal nodes says otherwise - behind a barrier)
      nd.epoch[v] = 0;
      nd.old_id[v] = P.old_fc ? old_root_of(P, lin) : -1;
      nd.edge[v] = P.lp.L;
      nd.corner[3 * (int64_t)v] = c0x;
      nd.corner[3 * (int64_t)v + 1] = c0y;
      nd.corner[3 * (int64_t)v + 2] = c0z;
      vlin[v] = (uint64_t)lin;
    }
    run += tot;
  }

  BB_STAMP(4);  // finish: bases + roots
  // ---- internal nodes and their children -----------------------------------------------------------------
  for (uint32_t j = (uint32_t)tid >> 3; j < nrec; j += 32) {
    const int c = tid & 7;
    const size_t r = 3 * ((size_t)node_stage + j);
    const uint32_t info = bk_node[r], w1 = bk_node[r + 1], up = bk_node[r + 2];
    const int l = (int)(w1 >> 28);
    const uint32_t own = w1 & 0xFFFFu, vord = info >> 18, prefix = info & 0x3FFFFu;  // l digits
    const int32_t v = (int32_t)(vbase + vord);
    const int32_t cb = (int32_t)(V + 8 * (int64_t)(lvl_first(l) + own));
    const int32_t xid =
        l == 0 ? v : (int32_t)(V + 8 * (int64_t)(lvl_first(l - 1) + up)) + (int32_t)(prefix & 7u);
    double cx = P.lp.c0x, cy = P.lp.c0y, cz = P.lp.c0z, e = P.lp.L;
    if (P.lp.mode == 0) {
      const uint32_t lin = bk_vox[3 * ((size_t)vox_stage + vord)];
      const uint32_t qz = lin % P.lp.nz, qy = (lin / P.lp.nz) % P.lp.ny, qx = lin / (P.lp.nz * P.lp.ny);
      cx = (double)(long long)((double)((int)qx + P.lp.minx) * P.lp.L);
      cy = (double)(long long)((double)((int)qy + P.lp.miny) * P.lp.L);
      cz = (double)(long long)((double)((int)qz + P.lp.minz) * P.lp.L);
    }
    // corner / edge: descend from the root with the reference's arithmetic
    // (corner + offset, edge / 2: octree.py:181-191)
    for (int t = 0; t < l; ++t) {
      const uint32_t d = (prefix >> (3 * (l - 1 - t))) & 7u;
      const double h = e / 2.0;
      cx = cx + ((d & 4u) ? h : 0.0);
      cy = cy + ((d & 2u) ? h : 0.0);
      cz = cz + ((d & 1u) ? h : 0.0);
      e = h;
    }
    const double h = e / 2.0;
    const int64_t ch = (int64_t)cb + c;
    nd.start[ch] = 0;  // ranges are only meaningful inside the level-synchronous path
    nd.count[ch] = 0;
    nd.scount[ch] = 0;
    nd.depth[ch] = l + 1;
    nd.voxel[ch] = v;
    nd.parent[ch] = xid;
    nd.first_child[ch] = -1;  // (k_node_init's job until round 5: every node is written by exactly one owner here)
    nd.epoch[ch] = 0;
    nd.old_id[ch] = -1;
    nd.edge[ch] = h;
    nd.corner[3 * ch + 0] = cx + ((c & 4) ? h : 0.0);
    nd.corner[3 * ch + 1] = cy + ((c & 2) ? h : 0.0);
    nd.corner[3 * ch + 2] = cz + ((c & 1) ? h : 0.0);
  }
  // the internal nodes themselves: first_child / epoch, one thread per record, behind the defaults written above
  // (a node's defaults come from its parent's lanes or the roots' sweep - other threads of this workgroup)
  __syncthreads();
  for (uint32_t j = (uint32_t)tid; j < nrec; j += 256) {
    const size_t r = 3 * ((size_t)node_stage + j);
    const uint32_t info = bk_node[r], w1 = bk_node[r + 1], up = bk_node[r + 2];
    const int l = (int)(w1 >> 28);
    const uint32_t own = w1 & 0xFFFFu, vord = info >> 18, prefix = info & 0x3FFFFu;
    const int32_t v = (int32_t)(vbase + vord);
    const int32_t cb = (int32_t)(V + 8 * (int64_t)(lvl_first(l) + own));
    const int32_t xid =
        l == 0 ? v : (int32_t)(V + 8 * (int64_t)(lvl_first(l - 1) + up)) + (int32_t)(prefix & 7u);
    nd.first_child[xid] = cb;
    int32_t ep = P.cur_epoch;
    if (P.old_fc) {
      // the same node (voxel, path) of the previous scheme: internal there -> it keeps its epoch
      // (k_make_children of build.hip: epoch = old_epoch[old id] when the old node had children)
      int32_t o = old_root_of(P, bk_vox[3 * ((size_t)vox_stage + vord)]);
      for (int t = 0; t < l && o >= 0; ++t) {
        const int32_t ofc = P.old_fc[o];
        o = ofc >= 0 ? ofc + (int32_t)((prefix >> (3 * (l - 1 - t))) & 7u) : -1;
      }
      if (o >= 0 && P.old_fc[o] >= 0) ep = P.old_epoch[o];
    }
    nd.epoch[xid] = ep;
  }

  // ---- preorder ranks of the internal nodes (for the block order) -------------------------------------------
  // Octree.get_leaf_points lists the cached leaves: a leaf sorts by (preorder rank of its parent among the
  // internal nodes of its voxel, child index) - octree_base.py:152-158, octree.py:183-191 (see order.hip) -
  // voxels in lexicographic order.  All blocks of a voxel are in this bucket, and with ONE pose and ONE epoch
  // the order of the whole forest is the buckets' orders one after the other.  (Never for a chunked bucket:
  // the host does not ask for the order then, SM_BK_NOORDER.)
  BB_STAMP(5);  // finish: internal nodes + children
  const bool want_order = P.order_out != nullptr && cko.y == 0;
  if (want_order) {
    if (tid == 0) {
      uint32_t o = 0;
      for (int l = 0; l < BB_LEVELS; ++l) {
        lvl_off[l] = o;
        o += lvl_cnt[l];
      }
    }
    for (uint32_t j = tid; j < nrec; j += 256) {
      const size_t r = 3 * ((size_t)node_stage + j);
      const uint32_t info = bk_node[r], w1 = bk_node[r + 1];
      const int l = (int)(w1 >> 28);
      const uint32_t prefix = info & 0x3FFFFu;
      // voxel, then the path as (digit + 1) nibbles, left aligned: an ancestor sorts in front of its descendants
      unsigned long long key = (unsigned long long)(info >> 18) << 28;
      for (int t = 0; t < l; ++t) key |= (unsigned long long)(((prefix >> (3 * (l - 1 - t))) & 7u) + 1u) << (4 * (6 - t));
      s_rk[j] = key;
    }
    __syncthreads();
    for (uint32_t j = tid; j < nrec; j += 256) {
      const unsigned long long key = s_rk[j];
      uint32_t rk = 0;
      for (uint32_t i = 0; i < nrec; ++i) rk += s_rk[i] < key ? 1u : 0u;
      const size_t r = 3 * ((size_t)node_stage + j);
      const uint32_t w1 = bk_node[r + 1];
      s_map[lvl_off[w1 >> 28] + (w1 & 0xFFFFu)] = ((bk_node[r] >> 18) << 16) | rk;
    }
    __syncthreads();
  }

  BB_STAMP(6);  // finish: preorder ranks
  // ---- (leaf, pose) blocks, position -> leaf ------------------------------------------------------------------
  uint32_t brun = 0;
  // (size: points of the block that starts at f = distance to the next head, or to the end of the piece; ~0u: the
  //  caller fills it in later - pieces beyond BB_CAP)
  auto emit = [&](int f, uint32_t li, bool bhead, uint32_t blk_index, uint32_t size) {
    const uint32_t dep = (li >> LC_DEPTH) & 7u, ob = li & 0xFFFFu;
    const int32_t leaf =
        dep == 0 ? (int32_t)(vbase + ob)
                 : (int32_t)(V + 8 * (int64_t)(lvl_first((int)dep - 1) + ob)) + (int32_t)((li >> LC_DIGIT) & 7u);
    if (P.write_pos) pos_node[(size_t)start + f] = leaf;
    if (bhead) {
      const uint32_t bo = bbase + blk_index;
      if (want_order) {
        // key: voxel, preorder rank of the parent, child digit (a root that is a leaf: the voxel alone)
        uint32_t key;
        if (dep == 0) {
          key = ob << 13;
        } else {
          const uint32_t m = s_map[lvl_off[dep - 1] + ob];
          key = ((m >> 16) << 13) | ((m & 0xFFFFu) << 3) | ((li >> LC_DIGIT) & 7u);
        }
        s_bkey[blk_index] = key;
      }
      blk_node[bo] = leaf;
      blk_slot[bo] = P.n_poses > 1 ? find_slot_dev(pose_off, P.n_poses, ord_idx[(size_t)start + f]) : 0;
      blk_start[bo] = start + (uint32_t)f;
      if (size != ~0u) blk_size[bo] = (int32_t)size;
    }
  };
  if (n <= BB_CAP) {
    // The usual piece (<= 4096 positions): the leaf words of all <= 16 rounds are loaded at once, block heads are
    // ranked by ballots, and ONE scan over the (round, wave) counts replaces a block scan (two barriers) per round.
    constexpr int FR = BB_CAP / 256;
    const int wave = tid >> 6, lane = tid & 63;
    const uint64_t lt = lanemask_lt();
    // (the words are held only as far as the ballots: the emitting pass below finds its heads in the ballots and
    //  loads a head's word again - an L2 hit for one lane in ~17 - instead of keeping 16 words and 16 ranks per lane
    //  through it: 128 -> fewer VGPRs, more buckets in flight per CU; the kernel is latency bound)
    {
      uint32_t li[FR];
#pragma unroll
      for (int r = 0; r < FR; ++r) {
        const int f = r * 256 + tid;
        li[r] = f < n ? leafinfo[(size_t)start + f] : 0u;
      }
#pragma unroll
      for (int r = 0; r < FR; ++r) {
        const uint64_t bal = __ballot((li[r] & LI_BHEAD) != 0u);   // (positions behind n hold 0)
        if (lane == 0) {
          s_hc[r * 4 + wave] = (uint32_t)__popcll(bal);
          s_bal[r * 4 + wave] = bal;
        }
      }
    }
    __syncthreads();
    if (tid < 64) {   // exclusive prefix over the FR x 4 counts (position order: round, wave)
      const uint32_t v = s_hc[tid];
      uint32_t inc = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off);
        if (tid >= off) inc += t;
      }
      s_hc[tid] = inc - v;
      if (tid == 63) s_hc[64] = inc;
    }
    __syncthreads();
#pragma unroll 2
    for (int r = 0; r < FR; ++r) {
      const int f = r * 256 + tid;
      if (r * 256 >= n) break;
      const unsigned long long bal = s_bal[r * 4 + wave];
      const bool bhead = ((bal >> lane) & 1ull) != 0ull;
      uint32_t size = 0;
      if (bhead) {
        // the next head: later in this wave's ballot, or the first one of a following ballot; none = end of the piece
        int q = r * 4 + wave;
        unsigned long long rest = bal & ~lt & ~(1ull << lane);
        while (rest == 0ull && ++q < FR * 4) rest = s_bal[q];
        const int next = rest ? q * 64 + (__ffsll((long long)rest) - 1) : n;
        size = (uint32_t)(next - f);
      }
      if (f < n && (bhead || P.write_pos))
        emit(f, leafinfo[(size_t)start + f], bhead, s_hc[r * 4 + wave] + (uint32_t)__popcll(bal & lt), size);
    }
    brun = s_hc[64];
  } else {
    // (a piece beyond BB_CAP: an unsplit voxel, round by round; a block's size is known when the next head shows up)
    uint32_t open_blk = ~0u, open_pos = 0;   // the last head so far: its block still waits for its size (uniform)
    for (int f0 = 0; f0 < n; f0 += 256) {
      const int f = f0 + tid;
      const uint32_t li = f < n ? leafinfo[(size_t)start + f] : 0u;
      const bool bhead = f < n && (li & LI_BHEAD);
      uint32_t tot;
      const uint32_t pre = block_excl_add(bhead ? 1u : 0u, &tot, s_scr);
      if (bhead) s_hp[pre] = (uint32_t)f;
      if (f < n && (bhead || P.write_pos)) emit(f, li, bhead, brun + pre, ~0u);
      __syncthreads();
      if (tot > 0) {
        if (bhead && pre + 1 < tot) blk_size[bbase + brun + pre] = (int32_t)(s_hp[pre + 1] - (uint32_t)f);
        if (tid == 0 && open_blk != ~0u) blk_size[bbase + open_blk] = (int32_t)(s_hp[0] - open_pos);
        open_blk = brun + tot - 1;
        open_pos = s_hp[tot - 1];
      }
      __syncthreads();
      brun += tot;
    }
    if (tid == 0 && open_blk != ~0u) blk_size[bbase + open_blk] = (int32_t)((uint32_t)n - open_pos);
  }
  BB_STAMP(7);  // finish: blocks
  if (want_order) {
    __syncthreads();
    // (the blocks of a voxel are neighbours in storage order, and the listing order only permutes them
    //  inside their voxel: a block is ranked among those)
    for (uint32_t j = tid; j < brun; j += 256) {
      const uint32_t key = s_bkey[j], vox = key >> 13;
      uint32_t lo = j;
      while (lo > 0 && (s_bkey[lo - 1] >> 13) == vox) --lo;
      uint32_t rk = lo;
      for (uint32_t i = lo; i < brun && (s_bkey[i] >> 13) == vox; ++i) rk += s_bkey[i] < key ? 1u : 0u;
      P.order_out[bbase + rk] = (int32_t)(bbase + j);
    }
  }
  BB_STAMP(8);  // finish: block order
  // the next piece of a chunked bucket starts behind this one's voxels, internal nodes and blocks
  pre_vox += nvox;
  pre_blk += cko.y ? nblk_piece : brun;
#pragma unroll
  for (int l = 0; l < BB_LEVELS; ++l) pre_lvl[l] += lvl_cnt[l];
  __syncthreads();
  }  // pieces
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
static int bucket_build_impl(octl_forest* f, const BucketBuildArgs& a, NodeTable& nt, int* done,
                             std::vector<octl_forest::LevelSeg>* segs, int64_t* n_internal, int* levels,
                             int64_t* n_voxels, int64_t* n_blocks, int64_t* pending, BucketBuildGeom* geom,
                             bool force_sync);

int forest_bucket_build(octl_forest* f, const BucketBuildArgs& a, NodeTable& nt, int* done,
                        std::vector<octl_forest::LevelSeg>* segs, int64_t* n_internal, int* levels,
                        int64_t* n_voxels, int64_t* n_blocks, int64_t* pending, BucketBuildGeom* geom) {
  return bucket_build_impl(f, a, nt, done, segs, n_internal, levels, n_voxels, n_blocks, pending, geom, false);
}

static int bucket_build_impl(octl_forest* f, const BucketBuildArgs& a, NodeTable& nt, int* done,
                             std::vector<octl_forest::LevelSeg>* segs, int64_t* n_internal, int* levels,
                             int64_t* n_voxels, int64_t* n_blocks, int64_t* pending, BucketBuildGeom* geom,
                             bool force_sync) {
  octl_ctx* ctx = f->ctx;
  hipStream_t st = ctx->stream;
  *done = 0;
  // (a rejected hint and a sparse scene make this function call itself, at most twice in a row by design: anything
  //  deeper is a bug that would otherwise end as a stack overflow)
  struct Depth {
    int& d;
    explicit Depth(int& x) : d(x) { ++d; }
    ~Depth() { --d; }
  };
  static thread_local int retry_depth = 0;
  Depth guard(retry_depth);
  if (retry_depth > 4) return octl_set_error(ctx, OCTL_E_STATE, "bucket build: the geometry retries do not terminate");
  const int64_t N = f->n_store, n_alive = f->n_alive;
  // (OCTL_NO_BUCKET_BUILD: tests compare this path with the level-synchronous one)
  if (n_alive <= 0 || !f->bbox_dev.p || ctx->opt.no_bucket_build) return OCTL_OK;
  // a single cube (bare Octree / OctreeManager) is ONE bucket: beyond what the oversize launch takes it is the
  // level loop's job from the start
  if (f->mode == 1 && n_alive > 65535) return OCTL_OK;
  const int n_poses = (int)f->pose_off.size() - 1;
  // buckets: runs of consecutive voxel keys, sized for ~2500 points on average (~1250 in a small cloud)
  // (OCTL_BUCKET_POINTS: tests force many small buckets - and with them the two-pass partition - on small clouds)
  // (a small cloud fills few workgroups and its kernels are chains of latencies, not bytes: half the bucket shortens
  //  every chain - 100 k points 0.165 -> 0.157 ms, 30 k 0.126 -> 0.117; from 1 M points on the larger bucket wins)
  const uint64_t target = ctx->opt.bucket_points > 0 ? (uint64_t)ctx->opt.bucket_points
                                                     : (n_alive <= 400000 ? 1280 : 2560);
  uint64_t want = 1;
  while (want < ((uint64_t)PT_BINS << PT_BITS) && want * target < (uint64_t)n_alive) want <<= 1;
  // One partition pass (want <= 4096 buckets): the key geometry is formed on the device (k_bucket_geom) and
  // the host does not wait for the bounding box; all 4096 buckets exist then, the ones behind the last
  // voxel key are empty.  (OCTL_SYNC_GEOM: tests run the host-side form on small clouds too.)
  const bool async_geom = !force_sync && !ctx->geom_sparse && want <= (uint64_t)PT_BINS && !ctx->opt.sync_geom;
  // voxels of slack around the true box in the geometry of the keys (geom_from_box): a scene that drifts by up to
  // that much per scan keeps fitting the previous scan's geometry.  OCTL_GEOM_MARGIN: 0 = the shipped margin of one
  // voxel, N > 0 = N voxels, negative = none (the tight box of rounds 2-5).  A single cube has no voxels to drift over.
  const int margin = f->mode != 0 ? 0
                                  : (ctx->opt.geom_margin == 0 ? 1 : (int)std::min<int64_t>(64, std::max<int64_t>(0, ctx->opt.geom_margin)));
  GeomAsk ask;
  ask.want = want;
  ask.n_alive = n_alive;
  ask.target = (uint32_t)target;
  ask.margin = margin;
  // a cloud taken in place has not been through the box pass: with the geometry of the context's previous
  // single-pass build as a hint the histogram pass finds the box itself (k_part_hist<true>, k_geom_validate);
  // without one the box pass runs now
  bool hinted = false;   // single pass: the geometry is formed on the device, the hint replaces the box pass
  bool hinted2 = false;  // two passes: the HOST forms the geometry from the hint's box instead of waiting for the box
  GeomDev hint;
  if (f->bbox_pending) {
    std::memcpy(&hint, ctx->geom_hint, sizeof(hint));
    const bool usable = ctx->geom_hint_valid && ctx->geom_hint_want == want && hint.lp.mode == f->mode &&
                        hint.lp.L == f->edge && hint.lp.c0x == f->corner[0] && hint.lp.c0y == f->corner[1] &&
                        hint.lp.c0z == f->corner[2] && !ctx->opt.no_geom_hint &&
                        hint.lp.exact_digits == (ctx->opt.no_exact_digits ? 0 : 1);
    hinted = async_geom && usable && !ctx->geom_hint_two_pass;
    hinted2 = !async_geom && !force_sync && usable && ctx->geom_hint_two_pass && f->mode == 0;
    if (!hinted && !hinted2) OCTL_TRY(store_compute_bbox(f));
  }
  // every stored point is alive (nothing was removed since the poses were added): the flags are not read
  if (f->n_alive != f->n_store) OCTL_TRY(alive_ensure(f));
  const uint8_t* alive_p = f->n_alive == f->n_store ? nullptr : f->alive.as<uint8_t>();
  int bb[6] = {0, 0, 0, 0, 0, 0};   // the box of the linear keys (padded)
  int tb[6] = {0, 0, 0, 0, 0, 0};   // the true voxel box
  uint64_t ny = 1, nz = 1;
  int s = 0;
  // (single pass with the geometry formed on the device: tables and grids for geom_bucket_cap(want) buckets -
  //  geom_from_box holds the geometry to that - instead of always 4096: a 100 k-point scan has 128)
  uint32_t nb = async_geom ? (uint32_t)geom_bucket_cap(want) : (uint32_t)PT_BINS;
  bool two_pass = false;
  if (!async_geom) {
    if (hinted2) {
      // the (padded) box of the context's previous two-pass build: validated on the device by the histogram pass itself
      std::memcpy(bb, hint.bb, sizeof(bb));
      std::memcpy(tb, hint.tb, sizeof(tb));
    } else {
      // ---- voxel bounding box (kept by the ingest kernel; the copy was enqueued with the last ingest) ------
      int32_t* bbox_host = reinterpret_cast<int32_t*>(static_cast<char*>(ctx->small_host) + MIRROR_BBOX_WORD * 4);
      HIP_TRY(ctx, hipMemcpyAsync(bbox_host, f->bbox_dev.p, 32, hipMemcpyDeviceToHost, st));
      HIP_TRY(ctx, hipStreamSynchronize(st));
      std::memcpy(tb, bbox_host, sizeof(tb));
      if (bbox_host[6])
        return octl_set_error(ctx, OCTL_E_DOMAIN,
                              "a point has a non-finite coordinate or a top-level voxel index outside +-%d",
                              OCTL_VOX_ABS_LIMIT);
      if (tb[0] > tb[3]) return OCTL_OK;
      geom_pad_box(tb, margin, bb);
    }
    if (bb[0] > bb[3]) return OCTL_OK;
    const uint64_t nx = (uint64_t)(bb[3] - bb[0] + 1);
    ny = (uint64_t)(bb[4] - bb[1] + 1);
    nz = (uint64_t)(bb[5] - bb[2] + 1);
    if (nx * ny > (1ull << 32) || nx * ny * nz >= (1ull << 32)) return OCTL_OK;  // keys would not fit 32 bits
    const uint64_t R = nx * ny * nz;
    // the host-formed geometry keeps buckets of 2^s keys (two passes split the bucket number's bits); s follows the
    // TRUE box - padding a 128^3 box crosses a power of two and would double every bucket (one rank's 125 M-point
    // shard: k_bucket_build 3.7 -> 4.9 ms, the chunk kernels 1.7 -> 4.4 ms) - the padded keys only add buckets
    const uint64_t Rt = (uint64_t)((int64_t)tb[3] - tb[0] + 1) * (uint64_t)((int64_t)tb[4] - tb[1] + 1) *
                        (uint64_t)((int64_t)tb[5] - tb[2] + 1);
    s = std::min(12, std::max(0, ceil_log2_u64(Rt) - ceil_log2_u64(want)));
    if (((R - 1) >> s) + 1 > ((uint64_t)PT_BINS << PT_BITS)) return OCTL_OK;  // a sparse scene: too many keys
    nb = (uint32_t)(((R - 1) >> s) + 1);
    two_pass = nb > (uint32_t)PT_BINS;
    if (hinted2 && !two_pass)  // (cannot happen: same box, same `want`)
      return bucket_build_impl(f, a, nt, done, segs, n_internal, levels, n_voxels, n_blocks, pending, geom, true);
    // (a SPARSE scene - more keys per wanted bucket than the 12-bit clamp on a bucket's key range allows - makes the
    //  context skip the single-pass attempt next time; judged on the true box: the padding alone never does)
    ctx->geom_sparse = ((Rt - 1) >> s) + 1 > (uint64_t)PT_BINS && want <= (uint64_t)PT_BINS;
    // Thin buckets: the 12-bit clamp on a bucket's key range left them under half their target.  A workgroup per
    // bucket is mostly set-up then, and the cost grows with the BOX (10 M points in 1024 x 1024 x 64 voxels:
    // 2.46 ms, in 8192 x 8192 x 64: 37 ms), while the level loop's depends on the points alone (1.5-1.7 ms for
    // every box measured): it takes these scenes.
    if (two_pass && (uint64_t)n_alive * 2 < (uint64_t)nb * target) return OCTL_OK;
  }

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
  lp.dshift = s;
  lp.dmask = 0xFFFFFFFFu;
  lp.raw_vp = 0;
  lp.exact_digits = ctx->opt.no_exact_digits ? 0 : 1;
  lp.width = 1u << s;   // (host-formed geometry: the shift form; geom_from_box overwrites these on the device)
  lp.winv = 0.0;
  lp.whalf = 0.0;
  uint32_t* small = ctx->small.as<uint32_t>();
  // supertiles: one round of workgroups (2 per CU) over the cloud, at most 16 tiles each
  const int cus = octl_ctx_cus(ctx);
// (A/B on the headline scene, tools/ab_build.sh: 8 records per thread 0.264 ms, 16: 0.284, 4: 0.295, 12 with 3
//  workgroups per CU: 0.329 - the kernel is latency bound at 2 waves per SIMD, 254 VGPRs with 16 records)
#define OCTL_PT_IPT 8
#define OCTL_PT_WGS 2
  constexpr int PT_IPT = OCTL_PT_IPT;  // (records per thread and tile of the partition kernels; -D for experiments)
  constexpr int tile = PT_THREADS * PT_IPT;
  auto supertiles = [&](int64_t items, int* st_tiles) {
    // (a pass with at most 256 digits keeps 3 KB of counters: three workgroups per CU instead of two)
    const int wgs = two_pass ? std::max(OCTL_PT_WGS, 3) : OCTL_PT_WGS;
    *st_tiles = (int)std::min<int64_t>(16, std::max<int64_t>(1, ceil_div(ceil_div(items, tile), (int64_t)cus * wgs)));
    return (uint32_t)ceil_div(items, (int64_t)*st_tiles * tile);
  };
  int st_tiles_a = 1, st_tiles_b = 1;
  const uint32_t nst_a = supertiles(N, &st_tiles_a);
  const uint32_t nst_b = two_pass ? supertiles(n_alive, &st_tiles_b) : 0;
  // Two passes split the bucket number's bits EVENLY (low half first, both passes stable): a tile of 4096
  // records then leaves runs of ~16 records (512 B) per digit in both passes.  (Round 2 took 12 bits first: one
  // record per digit and tile - isolated 32-byte stores at 2 TB/s, 0.61 ms per 10 M points at 125 M.)
  const int bits_a = two_pass ? std::min(PT_BITS, (ceil_log2_u64(nb) + 1) / 2) : 0;
  const uint32_t nd_a = two_pass ? (1u << bits_a) : nb;                // digits of the first pass
  const uint32_t nd_b = two_pass ? ((nb - 1) >> bits_a) + 1 : 0;       // digits of the second pass
  // ---- scratch ------------------------------------------------------------------------------------------------
  OCTL_TRY(devbuf_reserve(ctx, f->part_xyz[0], (size_t)n_alive * sizeof(PartRec)));
  if (two_pass) OCTL_TRY(devbuf_reserve(ctx, f->part_xyz[1], (size_t)n_alive * sizeof(PartRec)));
  // [supertile][digit] table | the same digit-major (scanned: the buckets' starts) | bucket bounds (two passes)
  const size_t tab_elems = (std::max((size_t)nd_a * nst_a, (size_t)nd_b * nst_b) + 1 + 15) & ~(size_t)15;
  // the partition table's transposes + scan as ONE launch (k_table_scan), the hint's validation inside it
  const bool fused_a = !ctx->opt.no_fused_tables && nst_a <= TS_MAX_ROWS;
  const bool fused_b = !ctx->opt.no_fused_tables && two_pass && nst_b <= TS_MAX_ROWS;
  OCTL_TRY(devbuf_reserve(ctx, f->bk_table, (2 * tab_elems + (two_pass ? nb + 1 : 0) + 16) * 4));
  // per-bucket totals: raw | scanned (apart: buckets beyond 4096 points add theirs later, and the scan runs again)
  const size_t tot_elems = (((size_t)BK_ROWS * nb + 8) + 15) & ~(size_t)15;
  OCTL_TRY(devbuf_reserve(ctx, f->bk_tot, 2 * tot_elems * 4));
  OCTL_TRY(devbuf_reserve(ctx, f->bk_vox, (size_t)n_alive * 12));
  OCTL_TRY(devbuf_reserve(ctx, f->bk_node, (size_t)n_alive * 12));
  OCTL_TRY(devbuf_reserve(ctx, f->leafinfo, (size_t)n_alive * 4));
  OCTL_TRY(devbuf_reserve(ctx, f->ord_idx, (size_t)n_alive * 4));
  OCTL_TRY(devbuf_reserve(ctx, f->xyz_ord, (size_t)n_alive * 24));
  OCTL_TRY(devbuf_reserve(ctx, f->pos_node, (size_t)n_alive * 4));
  OCTL_TRY(devbuf_reserve(ctx, f->blk_node, (size_t)n_alive * 4));
  OCTL_TRY(devbuf_reserve(ctx, f->blk_slot, (size_t)n_alive * 4));
  OCTL_TRY(devbuf_reserve(ctx, f->blk_start, (size_t)n_alive * 4));
  OCTL_TRY(devbuf_reserve(ctx, f->blk_size, (size_t)n_alive * 4));
  uint32_t* table = f->bk_table.as<uint32_t>();
  uint32_t* table_dm = table + tab_elems;  // digit-major
  auto transpose = [&](const uint32_t* in, uint32_t R, uint32_t Cc, uint32_t* out) {
    OCTL_LAUNCH(k_transpose_u32, dim3((Cc + 63) / 64, (R + 63) / 64), dim3(256), 0, st, in, R, Cc, out);
    return hipGetLastError();
  };
  // (SM_BK_FLAGS / SM_BK_TOTAL / SM_BK_TODO are zero: forest_build has reset the scalar block)
  // ---- partition ----------------------------------------------------------------------------------------------
  lp.dshift = s;
  lp.dmask = two_pass ? (nd_a - 1u) : 0xFFFFFFFFu;
  lp.raw_vp = two_pass ? 1 : 0;
  const LinParams lp_pass1 = lp;
  GeomDev* gdev = nullptr;
  if (async_geom) {
    gdev = reinterpret_cast<GeomDev*>(small + SM_GEOM);
    if (hinted) {
      // (normally in place already: k_build_begin copied it with the scalar block)
      if (!ctx->geom_hint_staged) OCTL_LAUNCH(k_geom_set, dim3(1), dim3(64), 0, st, hint, gdev);
    } else {
      LinParams base = lp;
      base.dshift = s;
      OCTL_LAUNCH(k_bucket_geom, dim3(1), dim3(64), 0, st, (const int32_t*)f->bbox_dev.as<int32_t>(), ask, base, gdev);
    }
    HIP_TRY(ctx, hipGetLastError());
  } else if (hinted2) {
    // the record holds the geometry the HOST formed from the hint's box (first-pass digit parameters); the kernels
    // of the first pass read it from there, everything behind it only looks at `valid`
    gdev = reinterpret_cast<GeomDev*>(small + SM_GEOM);
    GeomDev g2;
    std::memset(&g2, 0, sizeof(g2));
    g2.lp = lp_pass1;
    std::memcpy(g2.bb, bb, sizeof(bb));
    std::memcpy(g2.tb, tb, sizeof(tb));
    g2.valid = 1;
    OCTL_LAUNCH(k_geom_set, dim3(1), dim3(64), 0, st, g2, gdev);
    HIP_TRY(ctx, hipGetLastError());
  }
  auto table_scan = [&](uint32_t nst, uint32_t nd, uint32_t* bucket_start, int validate) {
    const unsigned g = (nd + TS_COLS - 1) / TS_COLS;
    uint64_t* status = nullptr;
    uint32_t epoch = 0;
    OCTL_TRY(octl_scan_status_acquire(ctx, g, &status, &epoch));
    OCTL_LAUNCH(k_table_scan, dim3(g), dim3(TS_THREADS), 0, st, table, nst, nd, bucket_start, status, epoch,
                       validate, (const int32_t*)f->bbox_dev.as<int32_t>(), ask, lp, gdev);
    HIP_TRY(ctx, hipGetLastError());
    return (int)OCTL_OK;
  };
  {
    KTimer t(ctx, "part_hist");
    if (hinted || hinted2) {
      auto kh = f->edge == 1.0 ? k_part_hist<true, true> : k_part_hist<true, false>;
      OCTL_LAUNCH(kh, dim3(nst_a), dim3(PH_THREADS), 0, st, (const double*)f->xyz.as<double>(),
                         alive_p, N, lp, (const GeomDev*)gdev, nst_a, nd_a, (int64_t)st_tiles_a * tile, table,
                         f->bbox_dev.as<int32_t>());
      HIP_TRY(ctx, hipGetLastError());
      f->bbox_pending = false;  // (the box is on the device now, whatever becomes of the hint)
      if (fused_a) {
        // (k_table_scan validates the hint on its way)
      } else if (hinted)
        OCTL_LAUNCH(k_geom_validate, dim3(1), dim3(64), 0, st, (const int32_t*)f->bbox_dev.as<int32_t>(), ask,
                           lp, gdev);
      else
        OCTL_LAUNCH(k_geom_validate2, dim3(1), dim3(64), 0, st, (const int32_t*)f->bbox_dev.as<int32_t>(), ask,
                           gdev);
    } else {
      auto kh = f->edge == 1.0 ? k_part_hist<false, true> : k_part_hist<false, false>;
      OCTL_LAUNCH(kh, dim3(nst_a), dim3(PH_THREADS), 0, st, (const double*)f->xyz.as<double>(),
                         alive_p, N, lp, (const GeomDev*)gdev, nst_a, nd_a, (int64_t)st_tiles_a * tile, table,
                         (int32_t*)nullptr);
    }
    HIP_TRY(ctx, hipGetLastError());
  }
  // (a small table: the scatter kernel's workgroups scan it themselves - ScatterOwnScan)
  const bool own_scan_a = fused_a && !two_pass && !hinted2 && nd_a <= 256u && nst_a <= PS_OWN_MAX_ROWS;
  if (!own_scan_a) {
    KTimer t(ctx, "part_scan");
    if (fused_a) {
      OCTL_TRY(table_scan(nst_a, nd_a, table_dm, hinted ? 1 : (hinted2 ? 2 : 0)));
    } else {
      HIP_TRY(ctx, transpose(table, nst_a, nd_a, table_dm));
      OCTL_TRY(octl_exclusive_scan_u32(ctx, table_dm, table_dm, (int64_t)nd_a * nst_a, nullptr));
      HIP_TRY(ctx, transpose(table_dm, nd_a, nst_a, table));
    }
  }
  {
    KTimer t(ctx, "part_scatter");
    auto ks = own_scan_a ? k_part_scatter<PT_IPT, false, 8, true>
                         : (nd_a <= 256u ? k_part_scatter<PT_IPT, false, 8> : k_part_scatter<PT_IPT, false, PT_BITS>);
    ScatterOwnScan own;
    std::memset(&own, 0, sizeof(own));
    if (own_scan_a) {
      own.bucket_start = table_dm;
      own.validate = hinted ? 1 : 0;
      own.bbox = (const int32_t*)f->bbox_dev.as<int32_t>();
      own.ask = ask;
      own.base = lp;
      own.g = gdev;
    }
    OCTL_LAUNCH(ks, dim3(nst_a), dim3(PT_THREADS), 0, st,
                       (const double*)f->xyz.as<double>(), alive_p, N, lp,
                       (const GeomDev*)gdev, nst_a, nd_a, st_tiles_a, (const uint32_t*)table,
                       (const int64_t*)f->pose_off_dev.as<int64_t>(), n_poses, a.scheme_dev,
                       f->part_xyz[0].as<PartRec>(), own);
    HIP_TRY(ctx, hipGetLastError());
  }
  const PartRec* recs = f->part_xyz[0].as<PartRec>();
  const uint32_t* bstart = table_dm;
  uint32_t bstride = fused_a ? 1u : nst_a;   // (k_table_scan leaves the buckets' starts as a plain array)
  if (two_pass) {
    // second (more significant) digit over the records of the first pass, then the bucket bounds
    lp.dshift = s + bits_a;
    lp.dmask = 0xFFFFFFFFu;
    lp.raw_vp = 0;
    {
      KTimer t(ctx, "part_hist");
      OCTL_LAUNCH(k_part_hist_rec, dim3(nst_b), dim3(PH_THREADS), 0, st,
                         (const uint4*)f->part_xyz[0].as<uint4>(), n_alive, lp, (const GeomDev*)gdev, nst_b, nd_b,
                         (int64_t)st_tiles_b * tile, table);
      HIP_TRY(ctx, hipGetLastError());
    }
    {
      KTimer t(ctx, "part_scan");
      if (fused_b) {
        OCTL_TRY(table_scan(nst_b, nd_b, table_dm, 0));
      } else {
        HIP_TRY(ctx, transpose(table, nst_b, nd_b, table_dm));
        OCTL_TRY(octl_exclusive_scan_u32(ctx, table_dm, table_dm, (int64_t)nd_b * nst_b, nullptr));
        HIP_TRY(ctx, transpose(table_dm, nd_b, nst_b, table));
      }
    }
    {
      KTimer t(ctx, "part_scatter");
      auto ks = nd_b <= 256u ? k_part_scatter<PT_IPT, true, 8> : k_part_scatter<PT_IPT, true, PT_BITS>;
      OCTL_LAUNCH(ks, dim3(nst_b), dim3(PT_THREADS), 0, st,
                         (const double*)f->part_xyz[0].as<double>(), (const uint8_t*)nullptr, n_alive, lp,
                         (const GeomDev*)gdev, nst_b, nd_b, st_tiles_b, (const uint32_t*)table, (const int64_t*)nullptr, 0,
                         (const uint8_t*)nullptr, f->part_xyz[1].as<PartRec>(), ScatterOwnScan{});
      HIP_TRY(ctx, hipGetLastError());
    }
    uint32_t* bounds = table + 2 * tab_elems;
    {
      KTimer t(ctx, "bucket_bounds");
      OCTL_LAUNCH(k_bucket_bounds, dim3((unsigned)ceil_div((int64_t)nb + 1, 256)), dim3(256), 0, st,
                         (const uint4*)f->part_xyz[1].as<uint4>(), (uint32_t)n_alive, lp, (const GeomDev*)gdev, nb, bounds);
      HIP_TRY(ctx, hipGetLastError());
    }
    recs = f->part_xyz[1].as<PartRec>();
    bstart = bounds;
    bstride = 1;
  }
  // ---- buckets ------------------------------------------------------------------------------------------------
  BkParams bp;
  bp.lp = lp;
  bp.K = a.K;
  bp.bstride = bstride;
  bp.nb = nb;
  bp.n_alive = (uint32_t)n_alive;
  bp.n_poses = n_poses;
  bp.all_scheme = a.scheme_dev ? 0 : 1;
  uint32_t* bk_tot = f->bk_tot.as<uint32_t>();
  uint32_t* bk_scan = bk_tot + tot_elems;
  // chunk plan of the buckets with more than BB_CAP points: [descriptors CK_CAP | totals CK_CAP x BK_ROWS | range per bucket]
  const size_t ck_off_tot = (size_t)CK_CAP * sizeof(ChunkDesc);
  const size_t ck_off_bkt = ck_off_tot + (size_t)CK_CAP * BK_ROWS * 4;
  OCTL_TRY(devbuf_reserve(ctx, f->bk_chunks, ck_off_bkt + (size_t)nb * 8 + 16));
  ChunkDesc* ck_desc = reinterpret_cast<ChunkDesc*>(f->bk_chunks.p);
  uint32_t* ck_tot = reinterpret_cast<uint32_t*>(static_cast<char*>(f->bk_chunks.p) + ck_off_tot);
  uint2* ck_of_bucket = reinterpret_cast<uint2*>(static_cast<char*>(f->bk_chunks.p) + ck_off_bkt);
  // The over-full buckets (plan, then one workgroup per chunk) and the normal ones (k_bucket_build) touch disjoint
  // buckets, rows of bk_tot and output ranges.  An even scene has none of the former: k_bucket_build counts them
  // (SM_BK_OVERFULL) and the host launches the two chunk kernels only when the totals say that there are some - then
  // the totals are scanned again (round 5; before: two empty launches per build).  A context whose PREVIOUS build had
  // such buckets (a skewed scene, scan after scan) launches them right away on its side stream, NEXT TO
  // k_bucket_build, as round 4 did: a chunk is one workgroup's whole bucket build, worth hiding.
  const bool chunks_beside = ctx->had_chunks;
  auto launch_chunks = [&](hipStream_t on) {
    OCTL_LAUNCH(k_bucket_plan, dim3(nb), dim3(BB_THREADS), 0, on, recs, bstart, bp, (const GeomDev*)gdev, ck_desc,
                       ck_of_bucket, bk_tot, small);
    HIP_TRY(ctx, hipGetLastError());
    // one workgroup per chunk; the list's length is on the device, the grid strides over it
    OCTL_LAUNCH(k_bucket_chunks, dim3((unsigned)std::min<int64_t>(2 * (int64_t)cus, CK_CAP)), dim3(BB_THREADS), 0, on,
                       recs, bstart, bp, (const GeomDev*)gdev, (const ChunkDesc*)ck_desc, ck_tot,
                       (const int64_t*)f->pose_off_dev.as<int64_t>(), f->ord_idx.as<uint32_t>(), f->xyz_ord.as<double>(),
                       f->leafinfo.as<uint32_t>(), f->bk_vox.as<uint32_t>(), f->bk_node.as<uint32_t>(), bk_tot, small);
    HIP_TRY(ctx, hipGetLastError());
    return (int)OCTL_OK;
  };
  {
    KTimer t(ctx, "bucket_build");
    hipStream_t side = st;
    bool on_side = false;
    if (chunks_beside && octl_ctx_side_stream(ctx) && hipEventRecord(ctx->self_gate, st) == hipSuccess &&
        hipStreamWaitEvent(ctx->self_stream, ctx->self_gate, 0) == hipSuccess) {
      side = ctx->self_stream;
      on_side = true;
    }
    if (chunks_beside) OCTL_TRY(launch_chunks(side));
    OCTL_LAUNCH(k_bucket_build, dim3(nb), dim3(BB_THREADS), 0, st, recs, bstart, bp,
                       (const GeomDev*)gdev, (const int64_t*)f->pose_off_dev.as<int64_t>(), f->ord_idx.as<uint32_t>(),
                       f->xyz_ord.as<double>(), f->leafinfo.as<uint32_t>(), f->bk_vox.as<uint32_t>(), f->bk_node.as<uint32_t>(),
                       bk_tot, small, chunks_beside ? 0 : 1);
    HIP_TRY(ctx, hipGetLastError());
    if (on_side) {
      HIP_TRY(ctx, hipEventRecord(ctx->self_done, side));
      HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->self_done, 0));
    }
  }
  static_assert(SM_GEOM == 64, "the geometry record is read back together with the 64 scalars in front of it");
  const int mirror_words = (int)(64 + (gdev ? sizeof(GeomDev) / 4 : 0));
  uint32_t sm[64];
  // k_bucket_finish, launched the ordinary way (behind the host's look at the totals) or speculatively (in front of
  // it: NodeParams::spec_geom)
  NodePtrs nd_launch;
  // (own_scan_seq != 0: the launch scans the totals itself and publishes them under that wait sequence number)
  auto launch_finish = [&](const NodeParams& np, bool with_chunk_map, uint32_t own_scan_seq = 0) {
    KTimer t(ctx, "bucket_nodes");
    const size_t lds = own_scan_seq ? ((size_t)BK_ROWS * nb + 1) * 4 : 0;
    auto kern = own_scan_seq ? k_bucket_finish<true> : k_bucket_finish<false>;
    OCTL_LAUNCH(kern, dim3(nb), dim3(256), lds, st, nd_launch, np, bstart,
                       (const uint32_t*)bk_scan, (const uint32_t*)(small + SM_BK_TOTAL),
                       (const uint32_t*)f->leafinfo.as<uint32_t>(), (const uint32_t*)f->ord_idx.as<uint32_t>(),
                       (const uint32_t*)f->bk_vox.as<uint32_t>(), (const uint32_t*)f->bk_node.as<uint32_t>(),
                       (const int64_t*)f->pose_off_dev.as<int64_t>(), (const ChunkDesc*)ck_desc, (const uint32_t*)ck_tot,
                       (const uint2*)(with_chunk_map ? ck_of_bucket : nullptr), f->pos_node.as<int32_t>(),
                       f->vlin_dev.as<uint64_t>(), f->blk_node.as<int32_t>(), f->blk_slot.as<int32_t>(),
                       f->blk_start.as<uint32_t>(), f->blk_size.as<int32_t>(), small,
                       (const uint32_t*)(own_scan_seq ? bk_tot : nullptr), bk_scan,
                       static_cast<uint32_t*>(ctx->small_host), mirror_words, own_scan_seq);
    return hipGetLastError();
  };
  NodeParams np;
  std::memset(&np, 0, sizeof(np));
  np.lp = lp;
  np.bstride = bstride;
  np.nb = nb;
  np.n_alive = (uint32_t)n_alive;
  np.n_poses = n_poses;
  np.all_scheme = bp.all_scheme;
  np.cur_epoch = a.cur_epoch;
  np.old_fc = a.old_fc;
  np.old_epoch = a.old_epoch;
  np.old_vcode = a.old_vcode;
  np.old_voxels = a.old_voxels;
  // Speculative k_bucket_finish (see NodeParams): a fresh scheme under a geometry record on the device and tables sized from the context's previous bucket build (ctx->spec_*: 25 % above what that one
  // needed).  OCTL_NO_SPEC_FINISH: never.
  bool spec_launched = false;
  int64_t spec_node_cap = 0, spec_vox_cap = 0, spec_blk_cap = 0;
  const bool spec_order = n_poses == 1 && !ctx->opt.no_fast_order;
  // (the voxel origin is only read against a previous scheme: NodeParams::org)
  const bool spec_want = async_geom && gdev && !a.old_fc && !a.old_vcode && ctx->spec_nodes > 0 &&
                         !ctx->opt.no_spec_finish;
  // (few buckets: the speculative launch scans the totals itself - no k_bucket_scan_totals in front of it)
  const bool own_scan = spec_want && !ctx->opt.no_fused_tables && (uint32_t)BK_ROWS * nb <= (uint32_t)BF_SCAN_MAX;
  auto spec_finish = [&](uint32_t own_scan_seq) {
    // (a table that held the previous build is taken as it is - growing a live buffer waits for the stream, and the
    //  kernel checks the real capacities anyway; one that did not is sized 25 % above that build)
    // (the previous build's counts, in proportion to the points: a scan ten times larger is not given tables that
    //  the ordinary launch would have to grow right away)
    const double scale = (double)n_alive / (double)std::max<int64_t>(ctx->spec_points, 1);
    auto expect = [&](int64_t prev) { return (int64_t)std::ceil((double)prev * scale); };
    const int64_t e_nodes = expect(ctx->spec_nodes), e_vox = expect(ctx->spec_vox), e_blocks = expect(ctx->spec_blocks);
    if (nt.cap < e_nodes) OCTL_TRY(nodes_reserve(ctx, nt, e_nodes + e_nodes / 4 + 64));
    if (f->vlin_dev.cap < (size_t)e_vox * 8) OCTL_TRY(devbuf_reserve(ctx, f->vlin_dev, (size_t)(e_vox + e_vox / 4 + 64) * 8));
    if (spec_order && f->fast_order.cap < (size_t)e_blocks * 4)
      OCTL_TRY(devbuf_reserve(ctx, f->fast_order, (size_t)(e_blocks + e_blocks / 4 + 64) * 4));
    spec_node_cap = nt.cap;
    spec_vox_cap = (int64_t)(f->vlin_dev.cap / 8);
    spec_blk_cap = spec_order ? (int64_t)(f->fast_order.cap / 4) : ((int64_t)1 << 40);
    nd_launch = node_ptrs(nt);
    NodeParams sp = np;
    sp.node_cap = spec_node_cap;
    sp.write_pos = 0;
    sp.order_out = spec_order ? f->fast_order.as<int32_t>() : nullptr;
    sp.org = f->vorg;
    sp.spec_geom = gdev;
    sp.spec_chunks = chunks_beside ? 1 : 0;
    sp.vox_cap = spec_vox_cap;
    sp.blk_cap = spec_blk_cap;
    HIP_TRY(ctx, launch_finish(sp, chunks_beside, own_scan_seq));
    spec_launched = true;
    return (int)OCTL_OK;
  };
  // raw totals -> scanned totals + the build's scalars in the pinned mirror; the host polls for them
  auto scan_totals = [&](bool with_spec) {
    const uint32_t wait_seq = octl_wait_next_seq(ctx);
    if (!(with_spec && own_scan)) {
      KTimer t(ctx, "bucket_scan");
      if (!ctx->opt.no_fused_tables) {
        const uint32_t n_tot = (uint32_t)BK_ROWS * nb;
        const unsigned g = (n_tot + BT_TILE - 1) / BT_TILE;
        uint64_t* status = nullptr;
        uint32_t epoch = 0;
        OCTL_TRY(octl_scan_status_acquire(ctx, g, &status, &epoch));
        OCTL_LAUNCH(k_bucket_scan_totals, dim3(g), dim3(BT_THREADS), 0, st, (const uint32_t*)bk_tot, bk_scan, n_tot,
                           nb, status, epoch, small, static_cast<uint32_t*>(ctx->small_host), mirror_words, wait_seq);
      } else {
        OCTL_TRY(octl_exclusive_scan_u32(ctx, bk_tot, bk_scan, (int64_t)BK_ROWS * nb, small + SM_BK_TOTAL));
        OCTL_LAUNCH(k_bucket_totals, dim3(1), dim3(64), 0, st, (const uint32_t*)bk_scan, nb,
                           (const uint32_t*)(small + SM_BK_TOTAL), small, static_cast<uint32_t*>(ctx->small_host),
                           mirror_words, wait_seq);
      }
      HIP_TRY(ctx, hipGetLastError());
    }
    if (with_spec) OCTL_TRY(spec_finish(own_scan ? wait_seq : 0u));
    // (the totals kernel writes the scalars and then its flag into the pinned mirror)
    const int flag = MIRROR_FLAG_BUILD;
    // (polling budget: behind an asynchronous apply_mask the previous scan's RANSAC may still be running in front of
    //  this build - 2.2 ms at 10 M points; falling back to a stream synchronisation would also wait for the
    //  speculative finish that was enqueued to run BESIDE this wait)
    OCTL_TRY(octl_wait_mirror_flags(ctx, &flag, 1, wait_seq, 500 + n_alive / 2000));
    std::memcpy(sm, ctx->small_host, sizeof(sm));
    return (int)OCTL_OK;
  };
  OCTL_TRY(scan_totals(spec_want));
  if (hinted2) {
    GeomDev g;
    std::memcpy(&g, static_cast<char*>(ctx->small_host) + SM_GEOM * 4, sizeof(g));
    if (!g.valid) {
      if (g.reason == GEOM_DOMAIN)
        return octl_set_error(ctx, OCTL_E_DOMAIN,
                              "a point has a non-finite coordinate or a top-level voxel index outside +-%d",
                              OCTL_VOX_ABS_LIMIT);
      if (g.reason == GEOM_EMPTY) return OCTL_OK;
      // the hinted box did not hold: the true one is on the device now, the host-side form builds again from it
      ctx->geom_hint_valid = false;
      return bucket_build_impl(f, a, nt, done, segs, n_internal, levels, n_voxels, n_blocks, pending, geom, true);
    }
    std::memcpy(tb, g.tb, sizeof(tb));   // the true box this cloud turned out to have
  }
  if (!async_geom && two_pass && f->mode == 0) {
    // the true box of this two-pass build, padded, is the box of the context's next one (the keys of THIS build
    // stay relative to bb; a scene that drifts is followed scan by scan)
    GeomDev g;
    std::memset(&g, 0, sizeof(g));
    g.lp = lp_pass1;
    geom_pad_box(tb, margin, g.bb);
    std::memcpy(g.tb, tb, sizeof(tb));
    g.valid = 1;
    static_assert(sizeof(GeomDev) <= sizeof(ctx->geom_hint), "hint storage");
    std::memcpy(ctx->geom_hint, &g, sizeof(g));
    ctx->geom_hint_valid = true;
    ctx->geom_hint_two_pass = true;
    ctx->geom_hint_want = want;
  }
  if (async_geom) {
    GeomDev g;
    std::memcpy(&g, static_cast<char*>(ctx->small_host) + SM_GEOM * 4, sizeof(g));
    if (!g.valid) {
      if (g.reason == GEOM_DOMAIN)
        return octl_set_error(ctx, OCTL_E_DOMAIN,
                              "a point has a non-finite coordinate or a top-level voxel index outside +-%d",
                              OCTL_VOX_ABS_LIMIT);
      if (g.reason == GEOM_EMPTY) return OCTL_OK;
      // the hinted geometry did not hold (the scene jumped by more than the margin, or its density changed): the
      // box is on the device now, the build runs again from it - one histogram pass and one round trip lost
      if (g.reason == GEOM_REHASH)
        return bucket_build_impl(f, a, nt, done, segs, n_internal, levels, n_voxels, n_blocks, pending, geom, false);
      // not a single-pass case after all (a sparse scene): the host-side form decides
      return bucket_build_impl(f, a, nt, done, segs, n_internal, levels, n_voxels, n_blocks, pending, geom, true);
    }
    // the next build's hint: the geometry of THIS cloud's true box (the same function the device evaluates) - under a
    // hint that held, the box the histogram pass found on its way; the scene is followed scan by scan
    {
      LinParams base = lp;
      GeomDev nh;
      geom_from_box(g.tb, false, ask, base, nh);
      static_assert(sizeof(GeomDev) <= sizeof(ctx->geom_hint), "hint storage");
      std::memset(ctx->geom_hint, 0, sizeof(ctx->geom_hint));
      std::memcpy(ctx->geom_hint, &nh, sizeof(nh));
      ctx->geom_hint_valid = nh.valid != 0;
      ctx->geom_hint_two_pass = false;
      ctx->geom_hint_want = want;
    }
    lp = g.lp;
    std::memcpy(bb, g.bb, sizeof(bb));
    std::memcpy(tb, g.tb, sizeof(tb));
    ny = lp.ny;
    nz = lp.nz;
  }
  // the packed voxel keys of everything that follows (k_bucket_finish's walk over the previous scheme, the
  // incremental insertion, the host's voxel list) are relative to the origin this fixes on the first build
  if (f->mode == 0) OCTL_TRY(forest_fix_origin(f, tb));
  // buckets beyond 4096 points that nobody has built yet: plan + chunks now, then the totals once more
  const bool overfull = sm[SM_BK_OVERFULL] > 0;
  if (overfull && !chunks_beside && !sm[SM_BK_FLAGS]) {
    {
      KTimer t(ctx, "bucket_build");
      OCTL_TRY(launch_chunks(st));
    }
    OCTL_TRY(scan_totals(false));
  }
  ctx->had_chunks = overfull;
  if (sm[SM_BK_FLAGS]) return OCTL_OK;  // some bucket / voxel does not fit: the caller runs the general path
  const int64_t V = sm[SM_NVOX];
  int64_t n_int = 0;
  int depth = 0;
  segs->assign(1, octl_forest::LevelSeg{0, V, 0});
  for (int l = 0; l < BB_LEVELS; ++l) {
    if (sm[SM_BK_LEVEL + l] == 0) break;
    n_int += sm[SM_BK_LEVEL + l];
    depth = l + 1;
    segs->push_back(octl_forest::LevelSeg{V + 8 * (n_int - (int64_t)sm[SM_BK_LEVEL + l]), V + 8 * n_int, l + 1});
  }
  if (depth > a.max_depth) return octl_set_error(ctx, OCTL_E_DEPTH, "maximum depth %d exceeded", a.max_depth);
  const int64_t total = V + 8 * n_int;
  if (total >= ((int64_t)1 << 31)) return octl_set_error(ctx, OCTL_E_NOMEM, "more than 2^31 scheme nodes");
  // the listing order of the blocks falls out of the same kernel for the common case: one pose, a fresh
  // scheme (one epoch), nothing left to the level loop, no bucket beyond what the kernel ranks in LDS
  const bool fast_order = n_poses == 1 && !a.old_fc && sm[SM_BK_TODO] == 0 && sm[SM_BK_NOORDER] == 0 &&
                          !ctx->opt.no_fast_order;
  // the speculative launch did the work iff the words it looked at on the device - the same ones as in `sm` - passed
  // (the geometry record is valid here: the invalid cases have returned above)
  const bool spec_held = spec_launched && sm[SM_BK_TODO] == 0 && sm[SM_BK_NOORDER] == 0 &&
                         (chunks_beside || !overfull) && V <= spec_vox_cap && (int64_t)sm[SM_NBLOCKS] <= spec_blk_cap &&
                         total <= spec_node_cap;
  ctx->spec_points = n_alive;
  ctx->spec_nodes = total;
  ctx->spec_vox = V;
  ctx->spec_blocks = sm[SM_NBLOCKS];
  if (spec_launched) (spec_held ? g_octl_spec_held : g_octl_spec_missed).fetch_add(1, std::memory_order_relaxed);
  if (!spec_held) {
    OCTL_TRY(nodes_reserve(ctx, nt, total));
    OCTL_TRY(devbuf_reserve(ctx, f->vlin_dev, (size_t)std::max<int64_t>(V, 1) * 8));
    nd_launch = node_ptrs(nt);
    // (first_child = -1 / epoch = 0 of every node: written by k_bucket_finish with the node's other fields)
    np.lp = lp;
    np.node_cap = nt.cap;
    np.write_pos = sm[SM_BK_TODO] > 0;
    np.org = f->vorg;
    np.order_out = nullptr;
    if (fast_order) {
      OCTL_TRY(devbuf_reserve(ctx, f->fast_order, (size_t)std::max<uint32_t>(sm[SM_NBLOCKS], 1) * 4));
      np.order_out = f->fast_order.as<int32_t>();
    }
    HIP_TRY(ctx, launch_finish(np, overfull || chunks_beside));
    // (block sizes: written by k_bucket_finish itself since round 5 - a block ends where the next head of its piece
    //  is, or with the piece)
  }
  if (a.old_vcode && a.old_voxels > 0) {
    // a voxel of the previous scheme that has lost all its points keeps its (empty) octree in the reference:
    // roots this path cannot make - the general path takes the build then (nothing is committed yet)
    HIP_TRY(ctx, hipMemsetAsync(small + SM_BK_MISSING, 0, 4, st));
    OCTL_LAUNCH(k_old_voxels_missing, dim3((unsigned)ceil_div(a.old_voxels, 256)), dim3(256), 0, st,
                       a.old_vcode, a.old_voxels, (const uint64_t*)f->vlin_dev.as<uint64_t>(), V, lp,
                       bb[3] - bb[0] + 1, f->vorg, small + SM_BK_MISSING);
    HIP_TRY(ctx, hipGetLastError());
    uint32_t missing = 0;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->small_host, small + SM_BK_MISSING, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    std::memcpy(&missing, ctx->small_host, 4);
    if (missing) return OCTL_OK;
  }
  nt.n = total;
  *n_internal = n_int;
  *levels = depth;
  *n_voxels = V;
  *n_blocks = sm[SM_NBLOCKS];
  *pending = sm[SM_BK_TODO];
  geom->min[0] = bb[0];
  geom->min[1] = bb[1];
  geom->min[2] = bb[2];
  geom->ny = ny;
  geom->nz = nz;
  geom->order_done = fast_order;
  *done = 1;
  return OCTL_OK;
}

// A big single cube (bare Octree / OctreeManager with millions of points) partitioned ONCE by the child digits of
// its first pm levels - the partition kernels above under mode 2: 8^pm "buckets" = the cube's depth-pm nodes in
// path order, 32-byte records (coordinates, the six digits, store index) - so that the level loop of build.hip
// starts at level pm over data that is already grouped and gathers from a 500 KB neighbourhood instead of the
// whole store (build.hip: cube_prefix_build).  Outputs stay in the forest's scratch: records in part_xyz[0], the
// first record of every bucket in *bstart (stride *bstride), a flag word that is non-zero when some point lies
// outside the cube (*bad_flag, device).
int forest_prefix_partition(octl_forest* f, int pm, const void** recs_out, const uint32_t** bstart, uint32_t* bstride,
                            const uint32_t** bad_flag) {
  octl_ctx* ctx = f->ctx;
  hipStream_t st = ctx->stream;
  const int64_t N = f->n_store;
  const int n_poses = (int)f->pose_off.size() - 1;
  LinParams lp;
  std::memset(&lp, 0, sizeof(lp));
  lp.mode = 2;
  lp.pm = pm;
  lp.L = f->edge;
  lp.c0x = f->corner[0];
  lp.c0y = f->corner[1];
  lp.c0z = f->corner[2];
  lp.ny = lp.nz = 1;
  lp.shift = 0;
  lp.dshift = 0;
  lp.dmask = 0xFFFFFFFFu;
  lp.raw_vp = 0;
  lp.exact_digits = ctx->opt.no_exact_digits ? 0 : 1;
  const uint32_t nd = 1u << (3 * pm);
  const int cus = octl_ctx_cus(ctx);
  constexpr int PT_IPT = OCTL_PT_IPT;
  constexpr int tile = PT_THREADS * PT_IPT;
  const int st_tiles = (int)std::min<int64_t>(16, std::max<int64_t>(1, ceil_div(ceil_div(N, tile), (int64_t)cus * OCTL_PT_WGS)));
  const uint32_t nst = (uint32_t)ceil_div(N, (int64_t)st_tiles * tile);
  OCTL_TRY(devbuf_reserve(ctx, f->part_xyz[0], (size_t)N * sizeof(PartRec)));
  const size_t tab_elems = (((size_t)nd * nst) + 1 + 15) & ~(size_t)15;
  OCTL_TRY(devbuf_reserve(ctx, f->bk_table, (2 * tab_elems + 16) * 4));
  uint32_t* table = f->bk_table.as<uint32_t>();
  uint32_t* table_dm = table + tab_elems;
  uint32_t* flag = table + 2 * tab_elems;
  HIP_TRY(ctx, hipMemsetAsync(flag, 0, 4, st));
  const bool fused = !ctx->opt.no_fused_tables && nst <= TS_MAX_ROWS;   // (the table in one launch: k_table_scan)
  auto transpose = [&](const uint32_t* in, uint32_t R, uint32_t Cc, uint32_t* out) {
    OCTL_LAUNCH(k_transpose_u32, dim3((Cc + 63) / 64, (R + 63) / 64), dim3(256), 0, st, in, R, Cc, out);
    return hipGetLastError();
  };
  {
    KTimer t(ctx, "prefix_hist");
    auto kh = f->edge == 1.0 ? k_part_hist<false, true> : k_part_hist<false, false>;
    OCTL_LAUNCH(kh, dim3(nst), dim3(PH_THREADS), 0, st, (const double*)f->xyz.as<double>(), (const uint8_t*)nullptr,
                       N, lp, (const GeomDev*)nullptr, nst, nd, (int64_t)st_tiles * tile, table,
                       reinterpret_cast<int32_t*>(flag));
    HIP_TRY(ctx, hipGetLastError());
    if (fused) {
      const unsigned g = (nd + TS_COLS - 1) / TS_COLS;
      uint64_t* status = nullptr;
      uint32_t epoch = 0;
      OCTL_TRY(octl_scan_status_acquire(ctx, g, &status, &epoch));
      OCTL_LAUNCH(k_table_scan, dim3(g), dim3(TS_THREADS), 0, st, table, nst, nd, table_dm, status, epoch, 0,
                         (const int32_t*)nullptr, GeomAsk{0, 0, 0, 0}, lp, (GeomDev*)nullptr);
      HIP_TRY(ctx, hipGetLastError());
    } else {
      HIP_TRY(ctx, transpose(table, nst, nd, table_dm));
      OCTL_TRY(octl_exclusive_scan_u32(ctx, table_dm, table_dm, (int64_t)nd * nst, nullptr));
      HIP_TRY(ctx, transpose(table_dm, nd, nst, table));
    }
  }
  {
    KTimer t(ctx, "prefix_scatter");
    auto ks = nd <= 256u ? k_part_scatter<PT_IPT, false, 8> : k_part_scatter<PT_IPT, false, PT_BITS>;
    OCTL_LAUNCH(ks, dim3(nst), dim3(PT_THREADS), 0, st, (const double*)f->xyz.as<double>(), (const uint8_t*)nullptr, N,
                       lp, (const GeomDev*)nullptr, nst, nd, st_tiles, (const uint32_t*)table,
                       (const int64_t*)f->pose_off_dev.as<int64_t>(), n_poses, (const uint8_t*)nullptr,
                       f->part_xyz[0].as<PartRec>(), ScatterOwnScan{});
    HIP_TRY(ctx, hipGetLastError());
  }
  *recs_out = f->part_xyz[0].p;
  *bstart = table_dm;
  *bstride = fused ? 1u : nst;   // (k_table_scan leaves the buckets' starts as a plain array)
  *bad_flag = flag;
  return OCTL_OK;
}

#ifdef BB_STAMPS
extern "C" int octl_debug_bb_stamps(octl_ctx* ctx, unsigned long long out[16], int reset) {
  if (!ctx || !out) return OCTL_E_INVALID;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bb_stamps), sizeof(unsigned long long) * 16));
  if (reset) {
    unsigned long long z[16] = {0};
    HIP_TRY(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_bb_stamps), z, sizeof(z)));
  }
  return OCTL_OK;
}
#endif


// ---- test hook: the key geometry on the host (include/octreelib_hip.h) ------------------------------------------
extern "C" int octl_debug_key_geometry(const int32_t tb[6], uint64_t want, int64_t n_alive, uint32_t target,
                                       int32_t margin, int32_t bb_out[6], uint32_t* width, uint32_t* n_buckets,
                                       int32_t* valid, int64_t* mismatches) {
  if (!tb || !bb_out || !width || !n_buckets || !valid || !mismatches || margin < 0 || margin > 64)
    return OCTL_E_INVALID;
  GeomAsk ask;
  ask.want = want;
  ask.n_alive = n_alive;
  ask.target = target;
  ask.margin = margin;
  LinParams base;
  std::memset(&base, 0, sizeof(base));
  base.L = 1.0;
  GeomDev g;
  geom_from_box(tb, false, ask, base, g);
  for (int a = 0; a < 6; ++a) bb_out[a] = g.bb[a];
  *valid = g.valid ? 1 : -(int32_t)g.reason;
  *width = 0;
  *n_buckets = 0;
  *mismatches = 0;
  if (!g.valid) return OCTL_OK;
  const uint64_t R = (uint64_t)(g.bb[3] - g.bb[0] + 1) * (uint64_t)(g.bb[4] - g.bb[1] + 1) * (uint64_t)(g.bb[5] - g.bb[2] + 1);
  *width = g.lp.width;
  *n_buckets = (uint32_t)((R + g.lp.width - 1) / g.lp.width);
  int64_t bad = 0;
  for (uint64_t lin = 0; lin < R; ++lin)
    bad += (uint32_t)std::fma((double)lin, g.lp.winv, g.lp.whalf) != (uint32_t)(lin / g.lp.width);
  *mismatches = bad;
  return OCTL_OK;
}
