// Stable LSD radix sort of (u64 key, u32 value) pairs, 8 bits per pass — the top-level-voxel
// bucketing step of Grid.insert_points (reference: np.unique(axis=0) + argsort at
// grid/grid.py:79-90, an O(N log N) comparison sort of (x,y,z) rows on one CPU thread).
//
// Per pass: (1) per-tile 256-bin histograms staged in LDS, (2) one device-wide exclusive scan
// of the digit-major table hist[digit][tile], (3) stable scatter: ranks inside a tile come from
// wave64 ballot matching + per-wave LDS counters, so no global atomics and a deterministic,
// stable result.  HBM traffic per pass: 12 B/pt read twice + 12 B/pt written.
#include "common.h"
#include "wave_utils.h"

namespace {

constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_IPT = 8;                       // items per thread
constexpr int RS_TILE = RS_THREADS * RS_IPT;    // 2048 items per tile
constexpr int RS_WAVE_ITEMS = 64 * RS_IPT;      // 512 consecutive items per wave
constexpr int RS_BINS = 256;

__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const uint64_t* __restrict__ keys,
                                                        int64_t n, int shift, uint32_t ntiles,
                                                        uint32_t* __restrict__ hist) {
  __shared__ uint32_t cnt[RS_BINS];
  cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
  for (int r = 0; r < RS_IPT; ++r) {
    const int64_t i = base + r * RS_THREADS + threadIdx.x;  // coalesced; order is irrelevant here
    if (i < n) atomicAdd(&cnt[(uint32_t)(keys[i] >> shift) & 0xFFu], 1u);
  }
  __syncthreads();
  hist[(size_t)threadIdx.x * ntiles + blockIdx.x] = cnt[threadIdx.x];
}

__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(
    const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
    uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, int64_t n, int shift,
    uint32_t ntiles, const uint32_t* __restrict__ hist_scanned) {
  __shared__ uint32_t cnt[RS_WAVES][RS_BINS];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int w = 0; w < RS_WAVES; ++w) cnt[w][threadIdx.x] = 0;
  __syncthreads();

  // wave w owns items [base + w*512, base + (w+1)*512), processed in 8 rounds of 64
  // consecutive items: stream order == memory order, so the sort is stable
  const int64_t wbase = (int64_t)blockIdx.x * RS_TILE + (int64_t)wave * RS_WAVE_ITEMS;
  uint64_t key[RS_IPT];
  uint32_t val[RS_IPT], rank[RS_IPT];
#pragma unroll
  for (int r = 0; r < RS_IPT; ++r) {
    const int64_t i = wbase + r * 64 + lane;
    const bool valid = i < n;
    key[r] = valid ? keys_in[i] : 0ull;
    val[r] = valid ? vals_in[i] : 0u;
    const uint32_t d = (uint32_t)(key[r] >> shift) & 0xFFu;
    rank[r] = wave_stable_rank<8>(d, valid, cnt[wave]);
  }
  __syncthreads();
  // per digit: exclusive prefix over the waves of this tile + global offset of (digit, tile)
  {
    const uint32_t d = threadIdx.x;
    uint32_t run = hist_scanned[(size_t)d * ntiles + blockIdx.x];
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) {
      const uint32_t c = cnt[w][d];
      cnt[w][d] = run;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RS_IPT; ++r) {
    const int64_t i = wbase + r * 64 + lane;
    if (i < n) {
      const uint32_t d = (uint32_t)(key[r] >> shift) & 0xFFu;
      const uint32_t dst = cnt[wave][d] + rank[r];
      keys_out[dst] = key[r];
      vals_out[dst] = val[r];
    }
  }
}

}  // namespace

int octl_radix_sort_u64_u32(octl_ctx* ctx, uint64_t* keys[2], uint32_t* vals[2], int64_t n,
                            int key_bits, DevBuf& hist_scratch, int* result) {
  *result = 0;
  if (n <= 1 || key_bits <= 0) return OCTL_OK;
  if (n >= (int64_t)1 << 32) return octl_set_error(ctx, OCTL_E_INVALID, "sort: n >= 2^32");
  const uint32_t ntiles = (uint32_t)ceil_div(n, RS_TILE);
  const int64_t nh = (int64_t)RS_BINS * ntiles;
  // +4 keeps the scan's dwordx4 tail inside the allocation
  OCTL_TRY(devbuf_reserve(ctx, hist_scratch, (size_t)(nh + 4) * sizeof(uint32_t)));
  uint32_t* hist = hist_scratch.as<uint32_t>();
  const int passes = (key_bits + 7) / 8;
  int cur = 0;
  for (int p = 0; p < passes; ++p) {
    const int shift = 8 * p;
    {
      KTimer t(ctx, "sort_hist");
      OCTL_LAUNCH(k_rs_hist, dim3(ntiles), dim3(RS_THREADS), 0, ctx->stream,
                         (const uint64_t*)keys[cur], n, shift, ntiles, hist);
      HIP_TRY(ctx, hipGetLastError());
    }
    {
      KTimer t(ctx, "sort_scan");
      OCTL_TRY(octl_exclusive_scan_u32(ctx, hist, hist, nh, nullptr));
    }
    {
      KTimer t(ctx, "sort_scatter");
      OCTL_LAUNCH(k_rs_scatter, dim3(ntiles), dim3(RS_THREADS), 0, ctx->stream,
                         (const uint64_t*)keys[cur], (const uint32_t*)vals[cur], keys[cur ^ 1],
                         vals[cur ^ 1], n, shift, ntiles, (const uint32_t*)hist);
      HIP_TRY(ctx, hipGetLastError());
    }
    cur ^= 1;
  }
  *result = cur;
  return OCTL_OK;
}
