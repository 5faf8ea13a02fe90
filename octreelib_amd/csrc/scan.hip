// Device-wide exclusive prefix sum of uint32, wave64.  Tables: ONE launch per scan (single pass
// with decoupled look-back; the step runs ~10 scans of small tables, so launches matter more
// than bytes).  Each block handles one 2048-item tile with 32 B per thread (two dwordx4), publishes
// its aggregate, and adds up its predecessors' published aggregates / inclusive prefixes.
//   * tile = blockIdx.x.  The look-back spin cannot deadlock because workgroups are dispatched in
//     blockIdx order within each XCD's queue: the lowest unfinished tile is always resident (every
//     workgroup resident on its XCD has a lower index, hence would be unfinished and lower), and it
//     only waits for finished tiles.  (A ticket counter instead costs one same-address atomic per
//     tile, ~10 ns each, serialised: +45 us on a 10 M-item scan - measured.)
//   * a tile's status is ONE 64-bit word {scan epoch : 30, flag : 2, value : 32} written and read
//     with single 8-byte relaxed atomics at agent scope - no separate flag/value ordering - and
//     the epoch makes words of earlier scans read as "not yet published": no reset launch.
// Point-sized inputs use the three-kernel version (reduce / recurse / down-sweep), see
// octl_exclusive_scan_u32.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "lookback.h"

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_IPT = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_IPT;

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t t = __shfl_up(v, off);
    if (lane >= off) v += t;
  }
  return v;
}

// exclusive scan of one value per thread across the 256-thread block; returns the exclusive
// prefix and the block total through *total
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* total,
                                                         uint32_t* lds /* >= 4 */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = wave_inclusive_scan(v);
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < SCAN_THREADS / 64; ++w) {
    uint32_t s = lds[w];
    if (w < wave) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__device__ __forceinline__ void load_tile(const uint32_t* in, int64_t n, int64_t base,
                                          uint32_t (&x)[SCAN_IPT]) {
  const int64_t i0 = base + (int64_t)threadIdx.x * SCAN_IPT;
  if (i0 + SCAN_IPT <= n) {
    const uint4* p = reinterpret_cast<const uint4*>(in + i0);
    uint4 a = p[0], b = p[1];
    x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w;
    x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
  } else {
#pragma unroll
    for (int j = 0; j < SCAN_IPT; ++j) x[j] = (i0 + j < n) ? in[i0 + j] : 0u;
  }
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_reduce(const uint32_t* __restrict__ in,
                                                              int64_t n,
                                                              uint32_t* __restrict__ sums) {
  __shared__ uint32_t lds[4];
  uint32_t x[SCAN_IPT];
  load_tile(in, n, (int64_t)blockIdx.x * SCAN_TILE, x);
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_IPT; ++j) s += x[j];
  uint32_t total;
  (void)block_exclusive_scan(s, &total, lds);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// offsets == nullptr: single tile (block offset 0); total_out written by block 0 only then
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_down(const uint32_t* in, uint32_t* out,
                                                            int64_t n,
                                                            const uint32_t* __restrict__ offsets,
                                                            uint32_t* total_out) {
  __shared__ uint32_t lds[4];
  uint32_t x[SCAN_IPT];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
  load_tile(in, n, base, x);
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_IPT; ++j) s += x[j];
  uint32_t total;
  uint32_t pre = block_exclusive_scan(s, &total, lds);
  pre += offsets ? offsets[blockIdx.x] : 0u;
  if (total_out && !offsets && threadIdx.x == 0) *total_out = total;
  const int64_t i0 = base + (int64_t)threadIdx.x * SCAN_IPT;
  uint32_t y[SCAN_IPT];
#pragma unroll
  for (int j = 0; j < SCAN_IPT; ++j) {
    y[j] = pre;
    pre += x[j];
  }
  if (i0 + SCAN_IPT <= n) {
    uint4* p = reinterpret_cast<uint4*>(out + i0);
    p[0] = make_uint4(y[0], y[1], y[2], y[3]);
    p[1] = make_uint4(y[4], y[5], y[6], y[7]);
  } else {
#pragma unroll
    for (int j = 0; j < SCAN_IPT; ++j)
      if (i0 + j < n) out[i0 + j] = y[j];
  }
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_lookback(
    const uint32_t* in, uint32_t* out, int64_t n, uint64_t* __restrict__ status, uint32_t epoch,
    uint32_t* total_out) {
  __shared__ uint32_t lds[4];
  __shared__ uint32_t s_excl;
  const uint32_t tile = blockIdx.x;
  uint32_t x[SCAN_IPT];
  const int64_t base = (int64_t)tile * SCAN_TILE;
  load_tile(in, n, base, x);
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_IPT; ++j) s += x[j];
  uint32_t total;
  uint32_t pre = block_exclusive_scan(s, &total, lds);
  const uint32_t excl = lookback_exclusive(status, epoch, tile, total, &s_excl);
  if (total_out && threadIdx.x == 0 && tile == gridDim.x - 1) *total_out = excl + total;
  pre += excl;
  const int64_t i0 = base + (int64_t)threadIdx.x * SCAN_IPT;
  uint32_t y[SCAN_IPT];
#pragma unroll
  for (int j = 0; j < SCAN_IPT; ++j) {
    y[j] = pre;
    pre += x[j];
  }
  if (i0 + SCAN_IPT <= n) {
    uint4* p = reinterpret_cast<uint4*>(out + i0);
    p[0] = make_uint4(y[0], y[1], y[2], y[3]);
    p[1] = make_uint4(y[4], y[5], y[6], y[7]);
  } else {
#pragma unroll
    for (int j = 0; j < SCAN_IPT; ++j)
      if (i0 + j < n) out[i0 + j] = y[j];
  }
}

int scan_single_pass(octl_ctx* ctx, const uint32_t* in, uint32_t* out, int64_t n,
                     uint32_t* total_dev) {
  const int64_t nb = ceil_div(n, SCAN_TILE);
  if (nb >= ((int64_t)1 << 31)) return octl_set_error(ctx, OCTL_E_INVALID, "scan: input too large");
  uint64_t* status = nullptr;
  uint32_t epoch = 0;
  OCTL_TRY(octl_scan_status_acquire(ctx, nb, &status, &epoch));
  OCTL_LAUNCH(k_scan_lookback, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, ctx->stream, in,
                     out, n, status, epoch, total_dev);
  HIP_TRY(ctx, hipGetLastError());
  return OCTL_OK;
}

int scan_rec(octl_ctx* ctx, const uint32_t* in, uint32_t* out, int64_t n, uint32_t* total_dev,
             int level) {
  const int64_t nb = ceil_div(n, SCAN_TILE);
  if (nb <= 1) {
    OCTL_LAUNCH(k_scan_down, dim3(1), dim3(SCAN_THREADS), 0, ctx->stream, in, out, n,
                       (const uint32_t*)nullptr, total_dev);
    HIP_TRY(ctx, hipGetLastError());
    return OCTL_OK;
  }
  if (level >= 3) return octl_set_error(ctx, OCTL_E_INVALID, "scan: input too large");
  OCTL_TRY(devbuf_reserve(ctx, ctx->scan_tmp[level], (size_t)nb * sizeof(uint32_t)));
  uint32_t* sums = ctx->scan_tmp[level].as<uint32_t>();
  OCTL_LAUNCH(k_scan_reduce, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, ctx->stream, in, n,
                     sums);
  HIP_TRY(ctx, hipGetLastError());
  OCTL_TRY(scan_rec(ctx, sums, sums, nb, total_dev, level + 1));
  OCTL_LAUNCH(k_scan_down, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, ctx->stream, in, out,
                     n, (const uint32_t*)sums, (uint32_t*)nullptr);
  HIP_TRY(ctx, hipGetLastError());
  return OCTL_OK;
}

}  // namespace

// Status words for ONE look-back chain of up to `tiles` tiles (scan.hip's own kernel and the fused table kernels of
// bucket_build.hip / ransac.hip / api.hip that carry a look-back of their own): a fresh epoch of the context's status
// array - words of earlier chains read as "not yet published", so nothing is reset between launches.
int octl_scan_status_acquire(octl_ctx* ctx, int64_t tiles, uint64_t** status, uint32_t* epoch) {
  const size_t need = (size_t)std::max<int64_t>(tiles, 1) * 8;  // status u64[tiles]
  if (ctx->scan_status.cap < need || ctx->scan_epoch >= (1u << 30) - 1) {
    if (ctx->scan_status.cap < need) OCTL_TRY(devbuf_reserve(ctx, ctx->scan_status, need + need / 2));
    HIP_TRY(ctx, hipMemsetAsync(ctx->scan_status.p, 0, ctx->scan_status.cap, ctx->stream));
    ctx->scan_epoch = 0;
  }
  *epoch = ++ctx->scan_epoch;  // 0 = "never published"
  *status = ctx->scan_status.as<uint64_t>();
  return OCTL_OK;
}

int octl_exclusive_scan_u32(octl_ctx* ctx, const uint32_t* in, uint32_t* out, int64_t n,
                            uint32_t* total_dev) {
  if (n <= 0) {
    if (total_dev) HIP_TRY(ctx, hipMemsetAsync(total_dev, 0, sizeof(uint32_t), ctx->stream));
    return OCTL_OK;
  }
  // 16-byte alignment is needed by the dwordx4 paths
  if ((reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(out) & 15))
    return octl_set_error(ctx, OCTL_E_INVALID, "scan: buffers must be 16-byte aligned");
  // Tables (up to 1024 tiles = 2 M items: block tables, histograms, node counts) take the single
  // pass: one launch instead of three to five.  Point-sized inputs keep the three-kernel version:
  // with thousands of tiles in flight the look-back chain at the start of the grid costs more
  // than the second read of the input (10 M items: 49 vs 37 us - measured).
  // (OCTL_SCAN=1pass / 3pass forces one form: A/B only)
  const bool single = ctx->opt.scan_mode ? ctx->opt.scan_mode == 1 : ceil_div(n, SCAN_TILE) <= 1024;
  if (!single) return scan_rec(ctx, in, out, n, total_dev, 0);
  return scan_single_pass(ctx, in, out, n, total_dev);
}
