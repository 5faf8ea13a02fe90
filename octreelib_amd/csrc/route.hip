// Multi-GPU: shard the Grid by top-level voxel and route every point to the rank that owns its
// voxel with ONE all-to-all over xGMI (RCCL grouped ncclSend/ncclRecv; RCCL has no alltoallv).
//
// The reference has no distributed code at all; what makes the path shard is that
// every top-level voxel is an independent OctreeManager (grid/grid.py:56,100-109) and
// Grid.subdivide / RANSAC have no cross-voxel dependency (grid.py:255-258).  All poses of a
// voxel go to the same rank, so synchronised subdivision stays local.
//
// RCCL is resolved with dlopen at communicator creation so the library loads (and the
// single-GPU path runs) on machines without it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>

#include "forest.h"
#include "ref_arith.h"
#include "wave_utils.h"

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  // optional (octl_comm_info: what the communicator itself says about the run)
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
};

RcclApi g_rccl;

const char* rccl_load() {
  if (g_rccl.handle) return nullptr;
  // OCTL_RCCL_LIBRARY: another build of the collective library (a site's own RCCL; the tests' stand-in that
  // lets R > 1 rank PROCESSES share one GPU, tests/rccl_stub/ - real RCCL refuses two ranks on one device)
  void* h = nullptr;
  if (const char* lib = getenv("OCTL_RCCL_LIBRARY")) {
    h = dlopen(lib, RTLD_NOW | RTLD_LOCAL);
    if (!h) return "the library named by OCTL_RCCL_LIBRARY could not be loaded";
  }
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return "librccl.so not found";
#define RCCL_SYM(field, name)                                   \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name)); \
  if (!g_rccl.field) return "missing RCCL symbol " name;
  RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
  RCCL_SYM(CommInitRank, "ncclCommInitRank")
  RCCL_SYM(CommDestroy, "ncclCommDestroy")
  RCCL_SYM(AllGather, "ncclAllGather")
  RCCL_SYM(AllReduce, "ncclAllReduce")
  RCCL_SYM(Send, "ncclSend")
  RCCL_SYM(Recv, "ncclRecv")
  RCCL_SYM(GroupStart, "ncclGroupStart")
  RCCL_SYM(GroupEnd, "ncclGroupEnd")
  RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef RCCL_SYM
  g_rccl.CommCount = reinterpret_cast<decltype(g_rccl.CommCount)>(dlsym(h, "ncclCommCount"));
  g_rccl.CommUserRank = reinterpret_cast<decltype(g_rccl.CommUserRank)>(dlsym(h, "ncclCommUserRank"));
  g_rccl.GetVersion = reinterpret_cast<decltype(g_rccl.GetVersion)>(dlsym(h, "ncclGetVersion"));
  g_rccl.handle = h;
  return nullptr;
}

#define NCCL_TRY(ctx, expr)                                                               \
  do {                                                                                    \
    ncclResult_t _r = (expr);                                                             \
    if (_r != ncclSuccess)                                                                \
      return octl_set_error((ctx), OCTL_E_COMM, "%s failed: %s", #expr,                   \
                            g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?");     \
  } while (0)

constexpr int RT_THREADS = 256;
constexpr int RT_IPT = 8;
constexpr int RT_TILE = RT_THREADS * RT_IPT;  // 2048 points per tile

// owner of the point's top-level voxel: the same voxel index the local build will compute
// (api.hip k_ingest / bucket_build.hip, grid.py:72-76)
__device__ __forceinline__ int route_dest_of(double x, double y, double z, double L, int n_ranks, bool* bad) {
  const double fx = floor_div_exact(x, L), fy = floor_div_exact(y, L), fz = floor_div_exact(z, L);
  const double lim = (double)OCTL_VOX_ABS_LIMIT;
  if ((fabs(fx) < lim) && (fabs(fy) < lim) && (fabs(fz) < lim))
    return voxel_owner_hash((int64_t)fx, (int64_t)fy, (int64_t)fz, n_ranks);
  *bad = true;
  return 0;
}

// Pass 1: points per (destination, tile): LDS histogram over 2048 points.  (The per-destination totals
// come out of the scanned table, k_route_counts: as one global atomic per (workgroup, destination) - 4883
// same-address atomics per destination for 10 M points - they were half of this kernel's time.)
__global__ __launch_bounds__(RT_THREADS) void k_route_hist(const double* __restrict__ xyz, int64_t n,
                                                           double L, int n_ranks, uint32_t ntiles,
                                                           uint32_t* __restrict__ hist,
                                                           uint32_t* __restrict__ err) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * RT_TILE;
  bool bad = false;
#pragma unroll
  for (int r = 0; r < RT_IPT; ++r) {
    const int64_t i = base + r * RT_THREADS + threadIdx.x;  // coalesced; order is irrelevant here
    if (i < n) atomicAdd(&h[route_dest_of(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], L, n_ranks, &bad)], 1u);
  }
  if (bad) atomicExch(err, 1u);
  __syncthreads();
  if ((int)threadIdx.x < n_ranks) hist[(size_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

// points per destination = differences of the row heads of the scanned (destination-major) table
__global__ void k_route_counts(const uint32_t* __restrict__ hist_scanned, uint32_t ntiles, int n_ranks,
                               const uint32_t* __restrict__ total, unsigned long long* __restrict__ counts) {
  const int d = threadIdx.x;
  if (d >= n_ranks) return;
  const uint32_t a = hist_scanned[(size_t)d * ntiles];
  const uint32_t b = d + 1 < n_ranks ? hist_scanned[(size_t)(d + 1) * ntiles] : *total;
  counts[d] = (unsigned long long)(b - a);
}

// Pass 2: the stable partition by destination WITH its payload - every point is read once and its
// coordinates + global index are written straight into the send buffers (runs of ~2048 / R points per
// destination and tile: coalesced).  The former key + index sort followed by a gather of the 24-byte
// points read 128-byte lines for 24 useful bytes in each of the R interleaved streams (0.36 ms per 10 M).
__global__ __launch_bounds__(RT_THREADS) void k_route_scatter(
    const double* __restrict__ xyz, const int64_t* __restrict__ gidx, int64_t index_base, int64_t n,
    double L, int n_ranks, uint32_t ntiles, const uint32_t* __restrict__ hist_scanned,
    double* __restrict__ out_xyz, int64_t* __restrict__ out_gidx) {
  __shared__ uint32_t cnt[RT_THREADS / 64][256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int w = 0; w < RT_THREADS / 64; ++w) cnt[w][threadIdx.x] = 0;
  __syncthreads();
  // wave w owns items [base + w*512, +512) in 8 rounds of 64 consecutive items: stream order == memory
  // order, so the partition is stable (received points stay in ascending global index per source)
  const int64_t wbase = (int64_t)blockIdx.x * RT_TILE + (int64_t)wave * (64 * RT_IPT);
  double x[RT_IPT], y[RT_IPT], z[RT_IPT];
#pragma unroll
  for (int r = 0; r < RT_IPT; ++r) {
    const int64_t i = min(wbase + r * 64 + lane, n - 1);
    x[r] = xyz[3 * i];
    y[r] = xyz[3 * i + 1];
    z[r] = xyz[3 * i + 2];
  }
  uint32_t dst_d[RT_IPT], rank[RT_IPT];
#pragma unroll
  for (int r = 0; r < RT_IPT; ++r) {
    const int64_t i = wbase + r * 64 + lane;
    const bool valid = i < n;
    bool bad = false;
    dst_d[r] = valid ? (uint32_t)route_dest_of(x[r], y[r], z[r], L, n_ranks, &bad) : 0u;
    rank[r] = wave_stable_rank<8>(dst_d[r], valid, cnt[wave]);
  }
  __syncthreads();
  {
    const uint32_t d = threadIdx.x;
    uint32_t run = (int)d < n_ranks ? hist_scanned[(size_t)d * ntiles + blockIdx.x] : 0u;
#pragma unroll
    for (int w = 0; w < RT_THREADS / 64; ++w) {
      const uint32_t c = cnt[w][d];
      cnt[w][d] = run;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RT_IPT; ++r) {
    const int64_t i = wbase + r * 64 + lane;
    if (i < n) {
      const size_t o = (size_t)cnt[wave][dst_d[r]] + rank[r];
      out_xyz[3 * o] = x[r];
      out_xyz[3 * o + 1] = y[r];
      out_xyz[3 * o + 2] = z[r];
      out_gidx[o] = gidx ? gidx[i] : index_base + i;
    }
  }
}

// Entries R and R+1 of a rank's row of the count exchange: its domain-error flag (set on the device by
// k_route_hist) and the capacity, in points, of its receive buffers - every rank then knows whether
// ANY rank failed or has to grow its buffers, and all of them take the same exit (see
// octl_route_points).
__global__ void k_route_status(unsigned long long* __restrict__ counts, int n_ranks,
                               const uint32_t* __restrict__ err, unsigned long long capacity) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    counts[n_ranks] = (unsigned long long)*err;
    counts[n_ranks + 1] = capacity;
  }
}

inline unsigned grid_for(int64_t n) { return (unsigned)ceil_div(n, 256); }

// Steps 1-2 of the routing for R destination ranks: destination of every point + counts, stable
// partition by destination (one radix pass), packed send buffers (xyz, global index) in the
// context's scratch.  Independent of the communicator (the test hook below runs it for any R).
static int route_partition(octl_ctx* ctx, const double* xyz_dev, const int64_t* gidx_dev, int64_t n,
                           int64_t index_base, double L, int R) {
  hipStream_t st = ctx->stream;
  DevBuf& hist = ctx->rt_hist;
  DevBuf& counts_d = ctx->rt_counts;
  DevBuf& matrix_d = ctx->rt_matrix;
  DevBuf& send_xyz = ctx->rt_send_xyz;
  DevBuf& send_gidx = ctx->rt_send_gidx;
  const int64_t n1 = std::max<int64_t>(n, 1);
  if (n >= ((int64_t)1 << 32)) return octl_set_error(ctx, OCTL_E_INVALID, "route: too many points");
  const uint32_t ntiles = (uint32_t)ceil_div(n1, RT_TILE);
  OCTL_TRY(devbuf_reserve(ctx, counts_d, (size_t)R * 8 + 16));
  OCTL_TRY(devbuf_reserve(ctx, matrix_d, (size_t)R * (R + 2) * 8));
  OCTL_TRY(devbuf_reserve(ctx, hist, ((size_t)R * ntiles + 8) * 4));
  OCTL_TRY(devbuf_reserve(ctx, send_xyz, (size_t)n1 * 24));
  OCTL_TRY(devbuf_reserve(ctx, send_gidx, (size_t)n1 * 8));
  uint32_t* err = ctx->small.as<uint32_t>();
  HIP_TRY(ctx, hipMemsetAsync(counts_d.p, 0, (size_t)R * 8 + 16, st));
  HIP_TRY(ctx, hipMemsetAsync(err, 0, 4, st));
  if (n > 0) {
    {
      KTimer t(ctx, "route_hist");
      OCTL_LAUNCH(k_route_hist, dim3(ntiles), dim3(RT_THREADS), 0, st, xyz_dev, n, L, R, ntiles,
                         hist.as<uint32_t>(), err);
      HIP_TRY(ctx, hipGetLastError());
      uint32_t* total = err + 1;
      OCTL_TRY(octl_exclusive_scan_u32(ctx, hist.as<uint32_t>(), hist.as<uint32_t>(), (int64_t)R * ntiles, total));
      OCTL_LAUNCH(k_route_counts, dim3(1), dim3(256), 0, st, (const uint32_t*)hist.as<uint32_t>(), ntiles, R,
                         (const uint32_t*)total, counts_d.as<unsigned long long>());
      HIP_TRY(ctx, hipGetLastError());
    }
    KTimer t(ctx, "route_scatter");
    OCTL_LAUNCH(k_route_scatter, dim3(ntiles), dim3(RT_THREADS), 0, st, xyz_dev, gidx_dev, index_base,
                       n, L, R, ntiles, (const uint32_t*)hist.as<uint32_t>(), send_xyz.as<double>(),
                       send_gidx.as<int64_t>());
    HIP_TRY(ctx, hipGetLastError());
  }
  return OCTL_OK;
}

}  // namespace

extern "C" {

int32_t octl_voxel_owner(int64_t qx, int64_t qy, int64_t qz, int32_t n_ranks) {
  return voxel_owner_hash(qx, qy, qz, n_ranks);
}

int octl_comm_unique_id(uint8_t id[OCTL_UNIQUE_ID_BYTES]) {
  static_assert(sizeof(ncclUniqueId) == OCTL_UNIQUE_ID_BYTES, "ncclUniqueId size");
  if (!id) return OCTL_E_INVALID;
  if (rccl_load()) return OCTL_E_COMM;
  ncclUniqueId uid;
  if (g_rccl.GetUniqueId(&uid) != ncclSuccess) return OCTL_E_COMM;
  std::memcpy(id, &uid, sizeof(uid));
  return OCTL_OK;
}

int octl_comm_init(octl_ctx* ctx, int32_t n_ranks, int32_t rank,
                   const uint8_t id[OCTL_UNIQUE_ID_BYTES]) {
  if (!ctx || !id || n_ranks < 1 || n_ranks > 256 || rank < 0 || rank >= n_ranks)
    return OCTL_E_INVALID;
  if (const char* e = rccl_load()) return octl_set_error(ctx, OCTL_E_COMM, "%s", e);
  if (ctx->comm) return octl_set_error(ctx, OCTL_E_STATE, "communicator already initialised");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  ncclComm_t comm;
  NCCL_TRY(ctx, g_rccl.CommInitRank(&comm, n_ranks, uid, rank));
  ctx->comm = comm;
  ctx->n_ranks = n_ranks;
  ctx->rank = rank;
  return OCTL_OK;
}

// What the COMMUNICATOR says about the run (not the launcher's environment): ranks in it, this context's rank in it,
// the collective library's version code; -1 where the library does not export the query.
int octl_comm_info(octl_ctx* ctx, int32_t* n_ranks, int32_t* user_rank, int32_t* version) {
  if (!ctx) return OCTL_E_INVALID;
  if (!ctx->comm) return octl_set_error(ctx, OCTL_E_STATE, "communicator not initialised");
  ncclComm_t comm = static_cast<ncclComm_t>(ctx->comm);
  int v = -1;
  if (n_ranks) {
    v = -1;
    if (g_rccl.CommCount) NCCL_TRY(ctx, g_rccl.CommCount(comm, &v));
    *n_ranks = v;
  }
  if (user_rank) {
    v = -1;
    if (g_rccl.CommUserRank) NCCL_TRY(ctx, g_rccl.CommUserRank(comm, &v));
    *user_rank = v;
  }
  if (version) {
    v = -1;
    if (g_rccl.GetVersion) NCCL_TRY(ctx, g_rccl.GetVersion(&v));
    *version = v;
  }
  return OCTL_OK;
}

// The device behind a context: PCI bus id ("0000:05:00.0") and the 16-byte UUID of its properties - what a
// multi-GPU run gathers to show that its ranks sit on DISTINCT devices.
int octl_device_identity(octl_ctx* ctx, char pci_bus_id[32], uint8_t uuid[16], int32_t* cus) {
  if (!ctx) return OCTL_E_INVALID;
  if (pci_bus_id) {
    pci_bus_id[0] = 0;
    HIP_TRY(ctx, hipDeviceGetPCIBusId(pci_bus_id, 32, ctx->device));
  }
  hipDeviceProp_t prop;
  HIP_TRY(ctx, hipGetDeviceProperties(&prop, ctx->device));
  if (uuid) std::memcpy(uuid, prop.uuid.bytes, 16);
  if (cus) *cus = prop.multiProcessorCount;
  return OCTL_OK;
}

int octl_comm_destroy(octl_ctx* ctx) {
  if (!ctx) return OCTL_E_INVALID;
  if (ctx->comm && g_rccl.CommDestroy) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)g_rccl.CommDestroy(static_cast<ncclComm_t>(ctx->comm));
  }
  ctx->comm = nullptr;
  ctx->n_ranks = 1;
  ctx->rank = 0;
  return OCTL_OK;
}

int octl_route_points(octl_ctx* ctx, const double* xyz_dev, const int64_t* gidx_dev, int64_t n,
                      int64_t index_base, const double corner[3], double L, int64_t* n_recv,
                      int64_t* send_counts) {
  if (!ctx || !n_recv || n < 0 || (n > 0 && !xyz_dev) || !corner)
    return OCTL_E_INVALID;
  if (corner[0] != 0.0 || corner[1] != 0.0 || corner[2] != 0.0 || !(L > 0.0) ||
      L != (double)(int64_t)L)
    return octl_set_error(ctx, OCTL_E_INVALID, "routing needs corner = 0 and an integer valued L");
  if (n >= ((int64_t)1 << 31)) return octl_set_error(ctx, OCTL_E_INVALID, "too many points");
  const int R = ctx->n_ranks, me = ctx->rank;
  if (R > 1 && !ctx->comm) return octl_set_error(ctx, OCTL_E_STATE, "communicator not initialised");
  // test hook: with a communicator present, push even the local part through RCCL (a 1-rank
  // communicator then exercises AllGather + grouped Send/Recv on a single GPU)
  const bool use_rccl = ctx->comm && (R > 1 || ctx->opt.route_self_sendrecv != 0);
  const bool self_rccl = use_rccl && ctx->opt.route_self_sendrecv != 0;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  ncclComm_t comm = static_cast<ncclComm_t>(ctx->comm);
  // (the cloud / its indices may be the target of an octl_dev_upload_async of this context that is still in flight)
  if (n > 0) {
    OCTL_TRY(ctx_wait_uploads(ctx, xyz_dev, (size_t)n * 24));
    if (gidx_dev) OCTL_TRY(ctx_wait_uploads(ctx, gidx_dev, (size_t)n * 8));
  }

  // --- 1-2. destinations, counts, stable partition, packed send buffers ---------------------------------
  DevBuf& counts_d = ctx->rt_counts;
  DevBuf& matrix_d = ctx->rt_matrix;
  DevBuf& send_xyz = ctx->rt_send_xyz;
  DevBuf& send_gidx = ctx->rt_send_gidx;
  auto cleanup = [&]() {};  // the scratch lives in the context (re-used by the next step)
  int rc = OCTL_OK;
#define RT_TRY(expr)       \
  do {                     \
    rc = (expr);           \
    if (rc != OCTL_OK) {   \
      cleanup();           \
      return rc;           \
    }                      \
  } while (0)
  RT_TRY(route_partition(ctx, xyz_dev, gidx_dev, n, index_base, L, R));
  uint32_t* err = ctx->small.as<uint32_t>();
  // --- 3. counts exchange ---------------------------------------------------------------------------------
  // A rank that left this function alone after the exchange would leave its peers waiting for ever in
  // the Send/Recv group below (RCCL has no timeout), so every exit between here and the group is
  // COLLECTIVE: each row of the exchange carries the rank's domain-error flag and the capacity of its
  // receive buffers next to its R counts; what can only fail locally (growing those buffers) is
  // agreed on with one more all-reduce, and only in the steps in which some rank has to grow.
  const int64_t cap_pts =
      (int64_t)std::min((ctx->routed_xyz.cap > 16 ? ctx->routed_xyz.cap - 16 : 0) / 24, ctx->routed_gidx.cap / 8);
  const int W = R + 2;  // words per row
  std::vector<int64_t> matrix((size_t)R * W, 0);
  if (use_rccl) {
    OCTL_LAUNCH(k_route_status, dim3(1), dim3(64), 0, st, counts_d.as<unsigned long long>(), R,
                       (const uint32_t*)err, (unsigned long long)cap_pts);
    if (hipGetLastError() != hipSuccess) return octl_set_error(ctx, OCTL_E_HIP, "route: status launch failed");
    ncclResult_t r = g_rccl.AllGather(counts_d.p, matrix_d.p, (size_t)W, ncclInt64, comm, st);
    if (r != ncclSuccess) {
      cleanup();
      return octl_set_error(ctx, OCTL_E_COMM, "ncclAllGather failed: %s", g_rccl.GetErrorString(r));
    }
    if (hipMemcpyAsync(matrix.data(), matrix_d.p, (size_t)R * W * 8, hipMemcpyDeviceToHost, st) !=
        hipSuccess) {
      cleanup();
      return octl_set_error(ctx, OCTL_E_HIP, "count download failed");
    }
  } else {
    matrix[0] = n;
    matrix[2] = cap_pts;
  }
  uint32_t err_h = 0;
  if (hipMemcpyAsync(ctx->small_host, err, 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) {
    cleanup();
    return octl_set_error(ctx, OCTL_E_HIP, "route: stream failed: %s",
                          hipGetErrorString(hipGetLastError()));
  }
  std::memcpy(&err_h, ctx->small_host, 4);
  if (!use_rccl) matrix[1] = err_h;
  auto cnt = [&](int from, int to) { return matrix[(size_t)from * W + to]; };
  bool grow_any = false;
  for (int p = 0; p < R; ++p) {  // identical on every rank: all of them return, none posts a transfer
    if (matrix[(size_t)p * W + R] != 0) {
      cleanup();
      return octl_set_error(ctx, OCTL_E_DOMAIN,
                            "route: non-finite coordinate or voxel out of range (on rank %d)", p);
    }
    int64_t recv_p = 0;
    for (int q = 0; q < R; ++q) recv_p += cnt(q, p);
    if (recv_p >= ((int64_t)1 << 31)) {
      cleanup();
      return octl_set_error(ctx, OCTL_E_INVALID, "rank %d receives more than 2^31-1 points", p);
    }
    grow_any = grow_any || std::max<int64_t>(recv_p, 1) > matrix[(size_t)p * W + R + 1];
  }
  std::vector<int64_t> soff(R + 1, 0), roff(R + 1, 0);
  for (int p = 0; p < R; ++p) {
    soff[p + 1] = soff[p] + cnt(me, p);
    roff[p + 1] = roff[p] + cnt(p, me);
    if (send_counts) send_counts[p] = cnt(me, p);
  }
  const int64_t nr = roff[R];
  {
    int rc_a = devbuf_reserve(ctx, ctx->routed_xyz, (size_t)std::max<int64_t>(nr, 1) * 24 + 16);
    if (rc_a == OCTL_OK) rc_a = devbuf_reserve(ctx, ctx->routed_gidx, (size_t)std::max<int64_t>(nr, 1) * 8);
    if (use_rccl && grow_any) {
      int64_t* flag_d = reinterpret_cast<int64_t*>(ctx->small.as<uint32_t>() + 512);
      int64_t* flag_h = static_cast<int64_t*>(ctx->small_host);
      *flag_h = rc_a == OCTL_OK ? 0 : 1;
      ncclResult_t r = ncclSuccess;
      if (hipMemcpyAsync(flag_d, flag_h, 8, hipMemcpyHostToDevice, st) != hipSuccess ||
          (r = g_rccl.AllReduce(flag_d, flag_d, 1, ncclInt64, ncclMax, comm, st)) != ncclSuccess ||
          hipMemcpyAsync(flag_h, flag_d, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
          hipStreamSynchronize(st) != hipSuccess) {
        cleanup();
        return octl_set_error(ctx, OCTL_E_COMM, "route: status all-reduce failed");
      }
      if (*flag_h != 0 && rc_a == OCTL_OK)
        rc_a = octl_set_error(ctx, OCTL_E_NOMEM, "route: another rank could not grow its receive buffers");
    }
    if (rc_a != OCTL_OK) {
      cleanup();
      return rc_a;
    }
  }
  // --- 4. the all-to-all ----------------------------------------------------------------------------------
  hipError_t self_rc = hipSuccess;
  {
    KTimer t(ctx, "route_alltoall");
    const int64_t self = cnt(me, me);
    bool self_side = false;
    if (self > 0 && !self_rccl) {
      // The rank's own part never leaves the device: two copies.  With peers to talk to they run on a stream of
      // their own, NEXT TO the transfers of the group below (on the context's stream they were queued in front
      // of it); alone (one rank) the context's stream takes them.
      hipStream_t cs = st;
      if (use_rccl) {
        // (a side stream that could not be created is not fatal: the context's stream does the copies)
        (void)octl_ctx_side_stream(ctx);
        if (ctx->self_stream && hipEventRecord(ctx->self_gate, st) == hipSuccess &&
            hipStreamWaitEvent(ctx->self_stream, ctx->self_gate, 0) == hipSuccess) {
          cs = ctx->self_stream;
          self_side = true;
        }
      }
      self_rc = hipMemcpyAsync(ctx->routed_xyz.as<double>() + 3 * roff[me],
                               send_xyz.as<double>() + 3 * soff[me], (size_t)self * 24,
                               hipMemcpyDeviceToDevice, cs);
      if (self_rc == hipSuccess)
        self_rc = hipMemcpyAsync(ctx->routed_gidx.as<int64_t>() + roff[me],
                                 send_gidx.as<int64_t>() + soff[me], (size_t)self * 8,
                                 hipMemcpyDeviceToDevice, cs);
      // (the context's stream - and with it this call's final synchronisation and the timer - waits for them)
      if (self_side && self_rc == hipSuccess) self_rc = hipEventRecord(ctx->self_done, cs);
    }
    if (use_rccl) {
      ncclResult_t r = g_rccl.GroupStart();
      for (int p = 0; p < R && r == ncclSuccess; ++p) {
        if (p == me && !self_rccl) continue;
        const int64_t sc = cnt(me, p), rcnt = cnt(p, me);
        if (sc > 0) {
          r = g_rccl.Send(send_xyz.as<double>() + 3 * soff[p], (size_t)sc * 3, ncclDouble, p, comm, st);
          if (r == ncclSuccess)
            r = g_rccl.Send(send_gidx.as<int64_t>() + soff[p], (size_t)sc, ncclInt64, p, comm, st);
        }
        if (rcnt > 0 && r == ncclSuccess) {
          r = g_rccl.Recv(ctx->routed_xyz.as<double>() + 3 * roff[p], (size_t)rcnt * 3, ncclDouble, p,
                          comm, st);
          if (r == ncclSuccess)
            r = g_rccl.Recv(ctx->routed_gidx.as<int64_t>() + roff[p], (size_t)rcnt, ncclInt64, p,
                            comm, st);
        }
      }
      ncclResult_t r2 = g_rccl.GroupEnd();
      if (r == ncclSuccess) r = r2;
      if (r != ncclSuccess) {
        cleanup();
        return octl_set_error(ctx, OCTL_E_COMM, "all-to-all failed: %s", g_rccl.GetErrorString(r));
      }
    }
    if (self_side && self_rc == hipSuccess) self_rc = hipStreamWaitEvent(st, ctx->self_done, 0);
  }
  // (the transfers of the group are posted by now on every rank: a local failure below no longer
  //  leaves a peer waiting)
  if (hipStreamSynchronize(st) != hipSuccess || self_rc != hipSuccess) {
    cleanup();
    return octl_set_error(ctx, OCTL_E_HIP, "route: all-to-all stream failed");
  }
  ctx->routed_n = nr;
  ctx->routed_taken = false;
  *n_recv = nr;
  cleanup();
#undef RT_TRY
  return OCTL_OK;
}

// The routed cloud becomes the pose: an empty forest takes the receive buffer over (and gives its own
// store buffer to the router for the next cloud), any other forest copies it behind its store.
static int add_routed(octl_forest* f, octl_ctx* rctx, int32_t* slot) {
  const int64_t n = rctx->routed_n;
  bool adopted = false;
  if (rctx->routed_taken)
    return octl_set_error(f->ctx, OCTL_E_INVALID, "the routed cloud has been handed to a forest already");
  OCTL_TRY(store_adopt(f, rctx->routed_xyz, n, &adopted));
  // a copy out of another context's buffer has to be over before that context routes again
  if (!adopted && rctx != f->ctx) HIP_TRY(f->ctx, hipStreamSynchronize(f->ctx->stream));
  rctx->routed_taken = adopted;
  if (adopted && rctx != f->ctx) {
    // the buffer handed back was the forest's store: the router's stream waits for whatever the
    // forest's stream still has queued on it
    octl_ctx* c = f->ctx;
    if (!c->handoff_event) HIP_TRY(c, hipEventCreateWithFlags(&c->handoff_event, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->handoff_event, c->stream));
    HIP_TRY(c, hipStreamWaitEvent(rctx->stream, c->handoff_event, 0));
  }
  if (slot) *slot = (int32_t)f->pose_off.size() - 1;
  f->n_store += n;
  f->n_alive += n;
  f->pose_off.push_back(f->n_store);
  f->store_dirty = true;
  return OCTL_OK;
}

int octl_forest_add_pose_routed(octl_forest* f, int32_t* slot) {
  if (!f) return OCTL_E_INVALID;
  OCTL_TRY(forest_settle(f));
  return add_routed(f, f->ctx, slot);
}

int octl_forest_add_pose_routed_from(octl_forest* f, octl_ctx* route_ctx, int32_t* slot) {
  if (!f || !route_ctx) return OCTL_E_INVALID;
  OCTL_TRY(forest_settle(f));
  if (route_ctx->device != f->ctx->device)
    return octl_set_error(f->ctx, OCTL_E_INVALID, "the routing context lives on another device");
  // octl_route_points returns after its stream has drained: the routed cloud is complete; what
  // follows runs on the forest's stream
  return add_routed(f, route_ctx, slot);
}

int octl_route_get_gidx(octl_ctx* ctx, int64_t cap, int64_t* gidx, int64_t* n) {
  if (!ctx || !n) return OCTL_E_INVALID;
  *n = ctx->routed_n;
  const int64_t m = std::min<int64_t>(cap, ctx->routed_n);
  if (m <= 0 || !gidx) return OCTL_OK;
  HIP_TRY(ctx, hipMemcpyAsync(gidx, ctx->routed_gidx.p, (size_t)m * 8, hipMemcpyDeviceToHost,
                              ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return OCTL_OK;
}

int octl_comm_allreduce_i64(octl_ctx* ctx, int64_t* inout_host, int32_t n) {
  if (!ctx || !inout_host || n < 0 || n > 256) return OCTL_E_INVALID;
  if (n == 0 || (ctx->n_ranks == 1 && !ctx->comm)) return OCTL_OK;
  if (!ctx->comm) return octl_set_error(ctx, OCTL_E_STATE, "communicator not initialised");
  hipStream_t st = ctx->stream;
  int64_t* d = reinterpret_cast<int64_t*>(ctx->small.as<uint32_t>() + 512);  // 2 KiB into the block
  HIP_TRY(ctx, hipMemcpyAsync(d, inout_host, (size_t)n * 8, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  NCCL_TRY(ctx, g_rccl.AllReduce(d, d, (size_t)n, ncclInt64, ncclSum,
                                 static_cast<ncclComm_t>(ctx->comm), st));
  HIP_TRY(ctx, hipMemcpyAsync(inout_host, d, (size_t)n * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  return OCTL_OK;
}

int octl_debug_route_partition(octl_ctx* ctx, const double* xyz, int64_t n, int64_t index_base,
                               double L, int32_t n_ranks, int64_t* counts, double* xyz_out,
                               int64_t* gidx_out) {
  if (!ctx || n < 0 || n_ranks < 1 || n_ranks > 256 || (n > 0 && (!xyz || !xyz_out || !gidx_out)) ||
      !counts)
    return OCTL_E_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  DevBuf in;
  OCTL_TRY(devbuf_reserve(ctx, in, (size_t)std::max<int64_t>(n, 1) * 24));
  int rc = OCTL_OK;
  if (n > 0 && hipMemcpyAsync(in.p, xyz, (size_t)n * 24, hipMemcpyHostToDevice, st) != hipSuccess)
    rc = OCTL_E_HIP;
  if (rc == OCTL_OK) rc = route_partition(ctx, in.as<double>(), nullptr, n, index_base, L, n_ranks);
  if (rc == OCTL_OK) {
    if (hipMemcpyAsync(counts, ctx->rt_counts.p, (size_t)n_ranks * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        (n > 0 && (hipMemcpyAsync(xyz_out, ctx->rt_send_xyz.p, (size_t)n * 24, hipMemcpyDeviceToHost, st) != hipSuccess ||
                   hipMemcpyAsync(gidx_out, ctx->rt_send_gidx.p, (size_t)n * 8, hipMemcpyDeviceToHost, st) != hipSuccess)))
      rc = OCTL_E_HIP;
  }
  if (hipStreamSynchronize(st) != hipSuccess) rc = rc == OCTL_OK ? OCTL_E_HIP : rc;
  devbuf_free(in);
  return rc == OCTL_E_HIP ? octl_set_error(ctx, rc, "octl_debug_route_partition failed") : rc;
}

}  // extern "C"
