// Device-wide exclusive prefix sum of uint32 (reduce / recurse / down-sweep), wave64.
// HBM-bound streaming kernels: each block handles one 2048-item tile with 32 B per thread
// (two dwordx4) so a wave touches 2 KiB of contiguous memory per pass.
#include "common.h"

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

int scan_rec(octl_ctx* ctx, const uint32_t* in, uint32_t* out, int64_t n, uint32_t* total_dev,
             int level) {
  const int64_t nb = ceil_div(n, SCAN_TILE);
  if (nb <= 1) {
    hipLaunchKernelGGL(k_scan_down, dim3(1), dim3(SCAN_THREADS), 0, ctx->stream, in, out, n,
                       (const uint32_t*)nullptr, total_dev);
    HIP_TRY(ctx, hipGetLastError());
    return OCTL_OK;
  }
  if (level >= 3) return octl_set_error(ctx, OCTL_E_INVALID, "scan: input too large");
  OCTL_TRY(devbuf_reserve(ctx, ctx->scan_tmp[level], (size_t)nb * sizeof(uint32_t)));
  uint32_t* sums = ctx->scan_tmp[level].as<uint32_t>();
  hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, ctx->stream, in, n,
                     sums);
  HIP_TRY(ctx, hipGetLastError());
  OCTL_TRY(scan_rec(ctx, sums, sums, nb, total_dev, level + 1));
  hipLaunchKernelGGL(k_scan_down, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, ctx->stream, in, out,
                     n, (const uint32_t*)sums, (uint32_t*)nullptr);
  HIP_TRY(ctx, hipGetLastError());
  return OCTL_OK;
}

}  // namespace

int octl_exclusive_scan_u32(octl_ctx* ctx, const uint32_t* in, uint32_t* out, int64_t n,
                            uint32_t* total_dev) {
  if (n <= 0) {
    if (total_dev) HIP_TRY(ctx, hipMemsetAsync(total_dev, 0, sizeof(uint32_t), ctx->stream));
    return OCTL_OK;
  }
  // 16-byte alignment is needed by the dwordx4 paths
  if ((reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(out) & 15))
    return octl_set_error(ctx, OCTL_E_INVALID, "scan: buffers must be 16-byte aligned");
  return scan_rec(ctx, in, out, n, total_dev, 0);
}
