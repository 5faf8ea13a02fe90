// Voxel-local build: one wavefront subdivides one top-level voxel completely.
//
// The level-synchronous path of build.hip touches every point with four kernels per tree level
// (histogram, scan, scatter, children) plus a gather at each end, and synchronises with the host
// once per level.  A top-level voxel of a Grid holds a few hundred points (1 m voxels, 10 M points:
// ~305), so its whole subtree fits one wave: 8 points per lane in registers, node counters and the
// exchange buffer in LDS.  Phase A loads the voxel's points once (the only random HBM gather of the
// build), computes their child digits with the reference's arithmetic, runs the count-driven
// recursive subdivision (OctreeNode.subdivide, octree/octree.py:20-32: split while the scheme-pose
// count exceeds K) as stable in-wave partitions, and writes the leaf-ordered permutation and
// coordinates.  A device scan turns the per-voxel, per-level internal-node counts into the SAME
// level-major node numbering the level-synchronous path produces, and phase B expands each voxel's
// leaf table into scheme nodes (octree.py:177-191).  Results are identical to build.hip's - the
// parity tests do not know which path ran.
//
// Not handled here (the caller falls back to build.hip): a voxel with more than 512 points or more
// than 128 simultaneously live nodes, leaves deeper than 7 levels, a point flagged "bad" (outside
// its cube), schemes with history (epoch inheritance) and keep_scheme re-placement.
#include "build_common.h"
#include "ref_arith.h"

namespace {

constexpr int VB_IPL = 8;               // points per lane
constexpr int VB_CAP = 64 * VB_IPL;     // points per voxel
constexpr int VB_LEVELS = 7;            // child digits kept per point (21 bits)
constexpr int VB_WPB = 4;               // wavefronts (voxels) per workgroup
constexpr int VB_MAXNODES = 128;        // live nodes (with points) per voxel and level
constexpr int VB_KEYS = 8 * VB_MAXNODES;

struct VbLds {            // per wavefront
  uint32_t w0[VB_CAP];    // exchange buffer: path word
  uint32_t w1[VB_CAP];    // exchange buffer: store index | scheme bit
  uint32_t w2[VB_CAP];    // exchange buffer: state word
  uint32_t kcnt[VB_KEYS]; // counters / offsets per (node ordinal, digit)
  uint32_t ncnt[VB_MAXNODES];  // scheme-pose points per node
};

// state word: original slot (9) | leaf depth so far (3) << 9 | final (1) << 12 | splits now (1) << 13 |
// node ordinal (7) << 14
__device__ __forceinline__ uint32_t st_orig(uint32_t s) { return s & 511u; }
__device__ __forceinline__ uint32_t st_depth(uint32_t s) { return (s >> 9) & 7u; }
__device__ __forceinline__ bool st_fin(uint32_t s) { return (s >> 12) & 1u; }
__device__ __forceinline__ bool st_split(uint32_t s) { return (s >> 13) & 1u; }
__device__ __forceinline__ uint32_t st_nid(uint32_t s) { return (s >> 14) & 127u; }
__device__ __forceinline__ uint32_t st_make(uint32_t orig, uint32_t depth, bool fin, bool split,
                                            uint32_t nid) {
  return orig | (depth << 9) | ((fin ? 1u : 0u) << 12) | ((split ? 1u : 0u) << 13) | (nid << 14);
}

__device__ __forceinline__ uint32_t digit_at(uint32_t path21, int level) {
  return (path21 >> (18 - 3 * level)) & 7u;
}

// child digits of VB_LEVELS levels below the cube (c, e), the same exact comparisons as
// compute_path in build.hip (octree/octree.py:73-75,94-97,181-191); *bad when the point is not
// inside the cube at some level
__device__ __forceinline__ uint32_t path21_of(double px, double py, double pz, double cx, double cy,
                                              double cz, double e, bool* bad) {
  uint32_t path = 0;
  double h = e / 2.0;
#pragma unroll 1
  for (int j = 0; j < VB_LEVELS; ++j) {
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

__device__ __forceinline__ uint64_t lanemask_le() {
  const unsigned lane = threadIdx.x & 63u;
  return (lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull);
}

__device__ __forceinline__ uint64_t wave_match_bits(uint32_t key, bool valid, int nbits) {
  uint64_t peers = __ballot(valid);
  for (int b = 0; b < nbits; ++b) {
    const bool bit = (key >> b) & 1u;
    const uint64_t m = __ballot(bit);
    peers &= bit ? m : ~m;
  }
  return peers;
}

// LDS hand-offs inside one wavefront: program order is enough for the hardware (one wave's LDS
// operations execute in order); the fence keeps the compiler from moving accesses across it
#define VB_SYNC()                                           \
  do {                                                      \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  \
    __builtin_amdgcn_wave_barrier();                        \
  } while (0)

__device__ __forceinline__ void vb_fallback(uint32_t* small) {
  if ((threadIdx.x & 63u) == 0) atomicOr(&small[SM_VB_FALLBACK], 1u);
}

// ---------------------------------------------------------------------------------------------
// phase A
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * VB_WPB) void k_voxel_build_a(
    NodePtrs nd, int64_t V, const uint32_t* __restrict__ val_sorted, const double* __restrict__ xyz,
    int64_t K, uint32_t* __restrict__ ord_idx, double* __restrict__ xyz_ord,
    uint32_t* __restrict__ leafinfo, uint32_t* __restrict__ lvl_cnt, uint32_t* __restrict__ small) {
  __shared__ VbLds lds[VB_WPB];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t v = (int64_t)blockIdx.x * VB_WPB + wave;
  if (v >= V) return;  // no workgroup barrier below: waves are independent
  VbLds& L = lds[wave];
  const uint32_t start = nd.start[v];
  const int n = (int)nd.count[v];
  if (n == 0) return;
  if (n > VB_CAP) {
    vb_fallback(small);
    return;
  }
  const double e0 = nd.edge[v];
  const double c0x = nd.corner[3 * v], c0y = nd.corner[3 * v + 1], c0z = nd.corner[3 * v + 2];

  uint32_t pw[VB_IPL], vw[VB_IPL], st[VB_IPL];
  double x[VB_IPL], y[VB_IPL], z[VB_IPL];
  bool bad = false;
#pragma unroll
  for (int r = 0; r < VB_IPL; ++r) {
    const int s = r * 64 + lane;
    pw[r] = 0; vw[r] = 0; x[r] = y[r] = z[r] = 0.0;
    st[r] = st_make((uint32_t)s, 0, false, false, 0);
    if (s < n) {
      vw[r] = val_sorted[start + s];
      const int64_t i = (int64_t)(vw[r] & IDX_MASK);
      x[r] = xyz[3 * i];
      y[r] = xyz[3 * i + 1];
      z[r] = xyz[3 * i + 2];
      pw[r] = path21_of(x[r], y[r], z[r], c0x, c0y, c0z, e0, &bad);
    }
  }
  if (__any(bad)) {  // such a point needs the exact slow path of build.hip
    vb_fallback(small);
    return;
  }

  int n_nodes = 1;    // live nodes (nodes that hold points)
  int depth_max = 0;
  uint32_t n_int[VB_LEVELS];
#pragma unroll
  for (int l = 0; l < VB_LEVELS; ++l) n_int[l] = 0;

#pragma unroll 1
  for (int ell = 0;; ++ell) {
    if (n_nodes > VB_MAXNODES) {  // more live nodes than the LDS counters of this path hold
      vb_fallback(small);
      return;
    }
    // scheme-pose points per live node (octree_manager.py:53-61: the union of the scheme poses)
    for (int i = lane; i < n_nodes; i += 64) L.ncnt[i] = 0;
    VB_SYNC();
#pragma unroll
    for (int r = 0; r < VB_IPL; ++r) {
      const bool valid = r * 64 + lane < n;
      if (valid && !st_fin(st[r]) && (vw[r] >> 31)) atomicAdd(&L.ncnt[st_nid(st[r])], 1u);
    }
    VB_SYNC();
    bool any_split = false;
#pragma unroll
    for (int r = 0; r < VB_IPL; ++r) {
      const bool valid = r * 64 + lane < n;
      const bool split = valid && !st_fin(st[r]) && K >= 0 && (int64_t)L.ncnt[st_nid(st[r])] > K;
      st[r] = st_make(st_orig(st[r]), st_depth(st[r]), st_fin(st[r]), split, st_nid(st[r]));
      any_split = any_split || split;
    }
    if (!__any(any_split)) break;
    if (ell == VB_LEVELS || 8 * n_nodes > VB_KEYS) {  // deeper / wider than this path handles
      vb_fallback(small);
      return;
    }
    depth_max = ell + 1;

    // stable counting sort by key = node ordinal * 8 + (digit if the node splits)
    const int nk = 8 * n_nodes;
    int kbits = 3;
    while ((1 << kbits) < nk) ++kbits;
    for (int i = lane; i < nk; i += 64) L.kcnt[i] = 0;
    VB_SYNC();
    uint32_t key[VB_IPL], rank[VB_IPL];
#pragma unroll
    for (int r = 0; r < VB_IPL; ++r) {
      const bool valid = r * 64 + lane < n;
      key[r] = st_nid(st[r]) * 8u + (st_split(st[r]) ? digit_at(pw[r], ell) : 0u);
      // rounds in order, lanes in order: ranks follow the current (stable) order
      const uint64_t peers = wave_match_bits(key[r], valid, kbits);
      const uint32_t before = __popcll(peers & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
      const int leader = __ffsll((unsigned long long)peers) - 1;
      uint32_t old = 0;
      if (valid && lane == leader) old = atomicAdd(&L.kcnt[key[r]], (uint32_t)__popcll(peers));
      old = __shfl(old, leader < 0 ? 0 : leader);
      rank[r] = old + before;
    }
    VB_SYNC();
    // exclusive prefix over the keys
    {
      const int per = (nk + 63) / 64;
      uint32_t sum = 0;
      for (int i = 0; i < per; ++i) {
        const int kk = lane * per + i;
        if (kk < nk) sum += L.kcnt[kk];
      }
      uint32_t inc = sum;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
      }
      uint32_t run = inc - sum;
      VB_SYNC();
      for (int i = 0; i < per; ++i) {
        const int kk = lane * per + i;
        if (kk < nk) {
          const uint32_t c = L.kcnt[kk];
          L.kcnt[kk] = run;
          run += c;
        }
      }
    }
    VB_SYNC();
    // move every item to its new slot (through LDS), remembering its old node ordinal
#pragma unroll
    for (int r = 0; r < VB_IPL; ++r) {
      if (r * 64 + lane < n) {
        const uint32_t dst = L.kcnt[key[r]] + rank[r];
        L.w0[dst] = pw[r];
        L.w1[dst] = vw[r];
        L.w2[dst] = st[r];
      }
    }
    VB_SYNC();
    // new node ordinals: a node = a run of equal (old ordinal, digit-if-split); internal nodes of
    // this level = runs of equal old ordinal among the splitting items
    uint32_t carry_nodes = 0, carry_int = 0;
    uint32_t prev_tail_nid = 0, prev_tail_dig = 0;
    bool prev_tail_valid = false;
#pragma unroll
    for (int r = 0; r < VB_IPL; ++r) {
      const int s = r * 64 + lane;
      const bool valid = s < n;
      uint32_t p = 0, w = 0, t = 0;
      if (valid) {
        p = L.w0[s];
        w = L.w1[s];
        t = L.w2[s];
      }
      const uint32_t onid = st_nid(t);
      const bool split = st_split(t);
      const uint32_t dig = split ? digit_at(p, ell) : 0u;
      // the item in the previous slot
      uint32_t pn = __shfl_up(onid, 1), pd = __shfl_up(dig, 1);
      bool pv = true;
      if (lane == 0) {
        pn = prev_tail_nid;
        pd = prev_tail_dig;
        pv = prev_tail_valid;
      }
      const bool head = valid && (!pv || pn != onid || pd != dig);
      const bool ihead = valid && split && (!pv || pn != onid);
      const uint64_t hb = __ballot(head), ib = __ballot(ihead);
      const uint32_t nid = carry_nodes + __popcll(hb & lanemask_le()) - 1u;
      carry_nodes += __popcll(hb);
      carry_int += __popcll(ib);
      prev_tail_nid = __shfl(onid, 63);
      prev_tail_dig = __shfl(dig, 63);
      prev_tail_valid = true;  // rounds are filled from slot 0: lane 63 of a previous round is valid
      pw[r] = p;
      vw[r] = w;
      st[r] = st_make(st_orig(t), st_depth(t) + (split ? 1u : 0u), st_fin(t) || !split, false,
                      valid ? (nid & 127u) : 0u);
    }
    n_int[ell] = carry_int;
    n_nodes = (int)carry_nodes;
    VB_SYNC();
  }

  // outputs: leaf-ordered permutation, per-point (path, depth), coordinates through the
  // orig -> final slot map
#pragma unroll
  for (int r = 0; r < VB_IPL; ++r) {
    const int s = r * 64 + lane;
    if (s < n) {
      ord_idx[start + s] = vw[r] & IDX_MASK;
      leafinfo[start + s] = pw[r] | (st_depth(st[r]) << 21);
      L.w0[st_orig(st[r])] = (uint32_t)s;
    }
  }
  VB_SYNC();
#pragma unroll
  for (int r = 0; r < VB_IPL; ++r) {
    const int s = r * 64 + lane;
    if (s < n) {
      const int64_t d = (int64_t)start + L.w0[s];
      xyz_ord[3 * d] = x[r];
      xyz_ord[3 * d + 1] = y[r];
      xyz_ord[3 * d + 2] = z[r];
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int l = 0; l < VB_LEVELS; ++l)
      if (n_int[l]) lvl_cnt[(size_t)l * V + v] = n_int[l];
    // same-address atomics serialise: only waves that raise the maximum issue one
    if ((uint32_t)depth_max > __hip_atomic_load(&small[SM_VB_DEPTH], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(&small[SM_VB_DEPTH], (uint32_t)depth_max);
  }
}

// scanned level bases -> small[32 + l] (host reads them with the flags in one go)
__global__ void k_vb_level_bases(const uint32_t* __restrict__ scanned, int64_t V,
                                 uint32_t* __restrict__ small) {
  const int l = threadIdx.x;
  if (l < VB_LEVELS) small[32 + l] = scanned[(size_t)l * V];
}

// ---------------------------------------------------------------------------------------------
// phase B: leaf table of a voxel -> scheme nodes in the level-major numbering of build.hip
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * VB_WPB) void k_voxel_build_b(
    NodePtrs nd, int64_t V, const uint32_t* __restrict__ leafinfo,
    const uint32_t* __restrict__ scanned, int cur_epoch, int32_t* __restrict__ pos_node) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t v = (int64_t)blockIdx.x * VB_WPB + wave;
  if (v >= V) return;
  const uint32_t start = nd.start[v];
  const int n = (int)nd.count[v];
  if (n == 0) return;
  const double e0 = nd.edge[v];
  const double c0x = nd.corner[3 * v], c0y = nd.corner[3 * v + 1], c0z = nd.corner[3 * v + 2];
  uint32_t pw[VB_IPL], dep[VB_IPL];
  int32_t cb_prev[VB_IPL];
#pragma unroll
  for (int r = 0; r < VB_IPL; ++r) {
    const int s = r * 64 + lane;
    pw[r] = 0; dep[r] = 0; cb_prev[r] = 0;
    if (s < n) {
      const uint32_t li = leafinfo[start + s];
      pw[r] = li & 0x1FFFFFu;
      dep[r] = li >> 21;
      if (dep[r] == 0) pos_node[start + s] = (int32_t)v;  // the root is the leaf
    }
  }
#pragma unroll 1
  for (int ell = 0; ell < VB_LEVELS; ++ell) {
    bool any_act = false;
#pragma unroll
    for (int r = 0; r < VB_IPL; ++r) any_act = any_act || (r * 64 + lane < n && dep[r] > (uint32_t)ell);
    if (!__any(any_act)) break;
    const uint32_t base = scanned[(size_t)ell * V + v];
    uint32_t carry = 0, tail_pref = 0;
    bool tail_act = false;
#pragma unroll
    for (int r = 0; r < VB_IPL; ++r) {
      const int s = r * 64 + lane;
      const bool act = s < n && dep[r] > (uint32_t)ell;  // inside an internal node of level ell
      const uint32_t pref = ell == 0 ? 0u : (pw[r] >> (21 - 3 * ell));
      uint32_t pp = __shfl_up(pref, 1);
      bool pa = __shfl_up(act ? 1 : 0, 1) != 0;
      if (lane == 0) {
        pp = tail_pref;
        pa = tail_act;
      }
      const bool head = act && (!pa || pp != pref);
      const uint64_t hb = __ballot(head);
      const uint32_t li = carry + __popcll(hb & lanemask_le()) - 1u;  // ordinal of my level-ell ancestor
      carry += __popcll(hb);
      tail_pref = __shfl(pref, 63);
      tail_act = __shfl(act ? 1 : 0, 63) != 0;
      if (act) {
        const int32_t cb = (int32_t)(V + 8 * (int64_t)(base + li));  // its 8 children
        const int32_t xid = ell == 0 ? (int32_t)v : cb_prev[r] + (int32_t)digit_at(pw[r], ell - 1);
        if (head) {
          nd.first_child[xid] = cb;
          nd.epoch[xid] = cur_epoch;
          // corner / edge of the node: descend from the root with the reference's arithmetic
          // (corner + offset, edge / 2: octree.py:181-191)
          double cx = c0x, cy = c0y, cz = c0z, e = e0;
          for (int t = 0; t < ell; ++t) {
            const uint32_t d = digit_at(pw[r], t);
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
            nd.depth[c] = ell + 1;
            nd.voxel[c] = (int32_t)v;
            nd.parent[c] = xid;
            nd.old_id[c] = -1;
            nd.edge[c] = h;
            nd.corner[3 * c + 0] = cx + ((j & 4) ? h : 0.0);
            nd.corner[3 * c + 1] = cy + ((j & 2) ? h : 0.0);
            nd.corner[3 * c + 2] = cz + ((j & 1) ? h : 0.0);
          }
        }
        if (dep[r] == (uint32_t)ell + 1u)
          pos_node[start + s] = cb + (int32_t)digit_at(pw[r], ell);
        cb_prev[r] = cb;
      }
    }
  }
}

}  // namespace

int forest_voxel_build(octl_forest* f, const VoxelBuildArgs& a, NodeTable& nt, int* done,
                       std::vector<int64_t>* level_first, int64_t* n_internal, int* levels) {
  octl_ctx* ctx = f->ctx;
  hipStream_t st = ctx->stream;
  *done = 0;
  const int64_t V = a.V, n = a.n_alive;
  if (V <= 0 || n <= 0) return OCTL_OK;
  uint32_t* small = ctx->small.as<uint32_t>();
  // scratch: per-point (path, depth) in idxbuf[0]; per-level counts in entries
  OCTL_TRY(devbuf_reserve(ctx, f->idxbuf[0], (size_t)n * 4));
  const size_t cnt_n = (size_t)VB_LEVELS * (size_t)V;
  OCTL_TRY(devbuf_reserve(ctx, f->entries, (cnt_n + 8) * 4));
  OCTL_TRY(devbuf_reserve(ctx, f->ord_idx, (size_t)n * 4));
  OCTL_TRY(devbuf_reserve(ctx, f->xyz_ord, (size_t)n * 24));
  OCTL_TRY(devbuf_reserve(ctx, f->pos_node, (size_t)n * 4));
  uint32_t* leafinfo = f->idxbuf[0].as<uint32_t>();
  uint32_t* lvl_cnt = f->entries.as<uint32_t>();
  HIP_TRY(ctx, hipMemsetAsync(lvl_cnt, 0, (cnt_n + 8) * 4, st));
  HIP_TRY(ctx, hipMemsetAsync(small + SM_VB_FALLBACK, 0, 12, st));
  NodePtrs nd = node_ptrs(nt);
  const unsigned grid = (unsigned)ceil_div(V, VB_WPB);
  {
    KTimer t(ctx, "voxel_build_a");
    hipLaunchKernelGGL(k_voxel_build_a, dim3(grid), dim3(64 * VB_WPB), 0, st, nd, V, a.val_sorted,
                       (const double*)f->xyz.as<double>(), a.K, f->ord_idx.as<uint32_t>(),
                       f->xyz_ord.as<double>(), leafinfo, lvl_cnt, small);
    HIP_TRY(ctx, hipGetLastError());
  }
  {
    KTimer t(ctx, "voxel_build_scan");
    OCTL_TRY(octl_exclusive_scan_u32(ctx, lvl_cnt, lvl_cnt, (int64_t)cnt_n, small + SM_VB_INTERNAL));
    hipLaunchKernelGGL(k_vb_level_bases, dim3(1), dim3(64), 0, st, (const uint32_t*)lvl_cnt, V, small);
    HIP_TRY(ctx, hipGetLastError());
  }
  uint32_t sm[48];
  HIP_TRY(ctx, hipMemcpyAsync(ctx->small_host, small, sizeof(sm), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  std::memcpy(sm, ctx->small_host, sizeof(sm));
  if (sm[SM_VB_FALLBACK]) return OCTL_OK;  // *done stays 0: the caller runs the general path
  const int depth = (int)sm[SM_VB_DEPTH];
  const int64_t n_int = sm[SM_VB_INTERNAL];
  if (depth > a.max_depth)
    return octl_set_error(ctx, OCTL_E_DEPTH, "maximum depth %d exceeded", a.max_depth);
  const int64_t total = V + 8 * n_int;
  if (total >= ((int64_t)1 << 31)) return octl_set_error(ctx, OCTL_E_NOMEM, "more than 2^31 scheme nodes");
  OCTL_TRY(nodes_reserve(ctx, nt, total));
  nd = node_ptrs(nt);
  if (n_int > 0) {
    HIP_TRY(ctx, hipMemsetAsync(nd.first_child + V, 0xFF, (size_t)(total - V) * 4, st));
    HIP_TRY(ctx, hipMemsetAsync(nd.epoch + V, 0, (size_t)(total - V) * 4, st));
  }
  {
    KTimer t(ctx, "voxel_build_b");
    hipLaunchKernelGGL(k_voxel_build_b, dim3(grid), dim3(64 * VB_WPB), 0, st, nd, V,
                       (const uint32_t*)leafinfo, (const uint32_t*)lvl_cnt, a.cur_epoch,
                       f->pos_node.as<int32_t>());
    HIP_TRY(ctx, hipGetLastError());
  }
  nt.n = total;
  level_first->assign({0, V});
  for (int l = 0; l < depth; ++l) {
    const int64_t next = (l + 1 < VB_LEVELS && l + 1 < depth) ? V + 8 * (int64_t)sm[32 + l + 1] : total;
    level_first->push_back(next);
  }
  *n_internal = n_int;
  *levels = depth;
  *done = 1;
  return OCTL_OK;
}
