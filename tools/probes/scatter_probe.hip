// Probe (not part of the library): what bounds isolated 32-byte record stores on gfx950 - DRAM, or the path to it?
// 10 M records of 32 bytes (two 16-byte stores per lane, like k_part_scatter) are written to pseudo-random slots of a
// window of W megabytes: every slot of the window is written about 320 MB / W times.  If the time does not fall once
// the window fits the 256 MB memory-side cache (or the 32 MB of L2), the limit is not DRAM row activations.
// Also: the same records written in order (coalesced), and in runs of 2 / 4 / 8 consecutive records per destination.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// RUN consecutive records go to consecutive slots (a run starts at a random slot that is a multiple of RUN)
template <int RUN>
__global__ __launch_bounds__(256) void k_scatter(uint4* __restrict__ out, uint32_t slots, long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const uint32_t run = (uint32_t)(i / RUN), in = (uint32_t)(i % RUN);
    const uint32_t slot = (uint32_t)(((uint64_t)mix(run) * (slots / RUN)) >> 32) * RUN + in;
    const uint4 v = uint4{(uint32_t)i, 1u, 2u, 3u};
    out[2 * (size_t)slot] = v;
    out[2 * (size_t)slot + 1] = v;
  }
}
__global__ __launch_bounds__(256) void k_linear(uint4* __restrict__ out, long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const uint4 v = uint4{(uint32_t)i, 1u, 2u, 3u};
    out[2 * i] = v;
    out[2 * i + 1] = v;
  }
}
int main() {
  const long n = 10000000;
  uint4* buf;
  hipMalloc(&buf, 1024L << 20);
  hipMemset(buf, 0, 1024L << 20);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](auto launch) {
    float best = 1e9f;
    for (int r = 0; r < 4; ++r) {
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
  };
  float t = timeit([&]() { hipLaunchKernelGGL(k_linear, dim3(4096), dim3(256), 0, 0, buf, n); });
  printf("in order (coalesced)                         %7.1f us  %5.1f ps per record  %5.2f TB/s\n", t * 1e3, t * 1e9 / n, n * 32 / t / 1e9);
  for (int w : {8, 32, 128, 320, 1024}) {
    const uint32_t slots = (uint32_t)(((long)w << 20) / 32);
    t = timeit([&]() { hipLaunchKernelGGL(k_scatter<1>, dim3(4096), dim3(256), 0, 0, buf, slots, n); });
    printf("isolated records, window %5d MB            %7.1f us  %5.1f ps per record  %5.2f TB/s\n", w, t * 1e3, t * 1e9 / n, n * 32 / t / 1e9);
  }
  // the same isolated records from fewer waves per CU (k_part_scatter stores from the four waves of one workgroup at a time)
  for (int g : {256, 512, 1024, 2048}) {
    const uint32_t sl = (uint32_t)((320L << 20) / 32);
    t = timeit([&]() { hipLaunchKernelGGL(k_scatter<1>, dim3(g), dim3(256), 0, 0, buf, sl, n); });
    printf("isolated records, 320 MB, %4d workgroups     %7.1f us  %5.1f ps per record  (%d waves per CU)\n", g, t * 1e3, t * 1e9 / n, g / 64);
  }
  const uint32_t slots = (uint32_t)((320L << 20) / 32);
  t = timeit([&]() { hipLaunchKernelGGL(k_scatter<2>, dim3(4096), dim3(256), 0, 0, buf, slots, n); });
  printf("runs of 2 records (64 B), window 320 MB      %7.1f us  %5.1f ps per record\n", t * 1e3, t * 1e9 / n);
  t = timeit([&]() { hipLaunchKernelGGL(k_scatter<4>, dim3(4096), dim3(256), 0, 0, buf, slots, n); });
  printf("runs of 4 records (128 B), window 320 MB     %7.1f us  %5.1f ps per record\n", t * 1e3, t * 1e9 / n);
  t = timeit([&]() { hipLaunchKernelGGL(k_scatter<8>, dim3(4096), dim3(256), 0, 0, buf, slots, n); });
  printf("runs of 8 records (256 B), window 320 MB     %7.1f us  %5.1f ps per record\n", t * 1e3, t * 1e9 / n);
  t = timeit([&]() { hipLaunchKernelGGL(k_scatter<16>, dim3(4096), dim3(256), 0, 0, buf, slots, n); });
  printf("runs of 16 records (512 B), window 320 MB    %7.1f us  %5.1f ps per record\n", t * 1e3, t * 1e9 / n);
  return 0;
}
