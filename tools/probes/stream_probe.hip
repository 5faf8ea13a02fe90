// Probe (not part of the library): how fast does gfx950 read a COLD 240 MB cloud of f64 xyz records, by access pattern?
//   A  the partition kernels' pattern: lane i loads x, y, z of point i (three 8-byte loads, stride 24 bytes)
//   B  the same bytes as contiguous 16-byte loads (lane i loads bytes [16 i, 16 i + 16) of the wave's 1536-byte span)
//   C  B with two loads in flight per lane
// Between timed launches another 768 MB buffer is read, so that neither L2 nor the 256 MB memory-side cache holds
// the cloud ("cold"); "warm" = the same launch again right away.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
typedef double d2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_a(const double* __restrict__ xyz, long n, double* out) {
  double s = 0;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    s += xyz[3 * i] + xyz[3 * i + 1] + xyz[3 * i + 2];
  }
  if (s == 1.2345) out[0] = s;
}
template <int U>
__global__ __launch_bounds__(256) void k_b(const d2* __restrict__ p, long n2, double* out) {
  double s = 0;
  long i = blockIdx.x * 256L * U + threadIdx.x;
  const long step = (long)gridDim.x * 256 * U;
  for (; i + (U - 1) * 256 < n2; i += step) {
    d2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = p[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) s += v[u].x + v[u].y;
  }
  if (s == 1.2345) out[0] = s;
}
__global__ __launch_bounds__(256) void k_flush(const d2* __restrict__ p, long n2, double* out) {
  double s = 0;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n2; i += (long)gridDim.x * 256) s += p[i].x;
  if (s == 1.2345) out[0] = s;
}
// H: the partition histogram's skeleton - `nst` workgroups of 1024 threads, each over its own contiguous chunk of
// st_items points, two points (48 contiguous bytes, three 16-byte loads) per thread and step, the next pair's loads
// issued before the current pair is used.  WORK 0: sum only; 1: + floor and a 4096-bin LDS histogram; 2: + box min/max
template <int WORK>
__global__ __launch_bounds__(1024) void k_h(const double* __restrict__ xyz, long N, long st_items, uint32_t* __restrict__ table,
                                             double* out) {
  __shared__ uint32_t hist[4096];
  for (int d = threadIdx.x; d < 4096; d += 1024) hist[d] = 0;
  __syncthreads();
  const long base = blockIdx.x * st_items, lim = min(N, base + st_items), stride = 2 * 1024;
  double s = 0;
  int mn = 1 << 30, mx = -(1 << 30);
  auto count = [&](double x, double y, double z) {
    if (WORK == 0) { s += x + y + z; return; }
    const int qx = (int)floor(x), qy = (int)floor(y), qz = (int)floor(z);
    if (WORK >= 2) { mn = min(mn, min(qx, min(qy, qz))); mx = max(mx, max(qx, max(qy, qz))); }
    atomicAdd(&hist[((qx * 32 + qy) * 32 + qz) >> 3 & 4095], 1u);
  };
  long i = base + 2 * (long)threadIdx.x;
  bool have = i + 1 < lim;
  d2 a = {0, 0}, b = a, c = a, na = a, nb = a, nc = a;
  auto load = [&](long j, d2& A, d2& B, d2& Cc) { const d2* p = (const d2*)(xyz + 3 * j); A = p[0]; B = p[1]; Cc = p[2]; };
  if (have) load(i, a, b, c);
  while (have) {
    const long j = i + stride;
    const bool hn = j + 1 < lim;
    if (hn) load(j, na, nb, nc);
    count(a.x, a.y, b.x);
    count(b.y, c.x, c.y);
    a = na; b = nb; c = nc; i = j; have = hn;
  }
  __syncthreads();
  if (WORK) for (int d = threadIdx.x; d < 4096; d += 1024) table[(size_t)blockIdx.x * 4096 + d] = hist[d];
  if (s == 1.2345 || mn == 12345 || mx == 54321) out[0] = s;
}

__global__ __launch_bounds__(256) void k_dirty(d2* __restrict__ p, long n2) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n2; i += (long)gridDim.x * 256) p[i] = d2{1.0, 2.0};
}
int main() {
  const long n = 10000000;
  double *cloud, *other, *out;
  hipMalloc(&cloud, n * 24);
  hipMalloc(&other, 768L << 20);
  hipMalloc(&out, 8);
  hipMemset(cloud, 0, n * 24);
  hipMemset(other, 0, 768L << 20);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto flush = [&]() { hipLaunchKernelGGL(k_flush, dim3(4096), dim3(256), 0, 0, (const d2*)other, (768L << 20) / 16, out); };
  auto timeit = [&](const char* name, auto launch) {
    float cold = 1e9f, warm = 1e9f;
    for (int r = 0; r < 5; ++r) {
      flush();
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < cold) cold = ms;
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1); if (ms < warm) warm = ms;
    }
    printf("%-44s cold %6.1f us  %5.2f TB/s    warm %6.1f us  %5.2f TB/s\n", name, cold * 1e3, n * 24 / cold / 1e9, warm * 1e3, n * 24 / warm / 1e9);
  };
  for (int g : {1024, 2048, 4096, 8192}) {
    char nm[96];
    snprintf(nm, sizeof nm, "A x,y,z per lane            grid %5d", g);
    timeit(nm, [&]() { hipLaunchKernelGGL(k_a, dim3(g), dim3(256), 0, 0, cloud, n, out); });
    snprintf(nm, sizeof nm, "B 16-byte contiguous        grid %5d", g);
    timeit(nm, [&]() { hipLaunchKernelGGL(k_b<1>, dim3(g), dim3(256), 0, 0, (const d2*)cloud, n * 3 / 2, out); });
    snprintf(nm, sizeof nm, "C 16-byte contiguous x 2    grid %5d", g);
    timeit(nm, [&]() { hipLaunchKernelGGL(k_b<2>, dim3(g), dim3(256), 0, 0, (const d2*)cloud, n * 3 / 2, out); });
    snprintf(nm, sizeof nm, "D 16-byte contiguous x 4    grid %5d", g);
    timeit(nm, [&]() { hipLaunchKernelGGL(k_b<4>, dim3(g), dim3(256), 0, 0, (const d2*)cloud, n * 3 / 2, out); });
  }
  // the same reads behind a kernel that WRITES the other buffer: the caches hold dirty lines that the stream evicts
  auto timeit_dirty = [&](const char* name, auto launch) {
    float cold = 1e9f;
    for (int r = 0; r < 5; ++r) {
      hipLaunchKernelGGL(k_dirty, dim3(4096), dim3(256), 0, 0, (d2*)other, (768L << 20) / 16);
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < cold) cold = ms;
    }
    printf("%-44s behind 768 MB of writes: %6.1f us  %5.2f TB/s\n", name, cold * 1e3, n * 24 / cold / 1e9);
  };
  timeit_dirty("A x,y,z per lane            grid  4096", [&]() { hipLaunchKernelGGL(k_a, dim3(4096), dim3(256), 0, 0, cloud, n, out); });
  uint32_t* table;
  hipMalloc(&table, 2048L * 4096 * 4);
  for (int nst : {489, 978, 1956}) {
    const long st_items = ((n + nst - 1) / nst + 1) & ~1L;
    char nm[96];
    snprintf(nm, sizeof nm, "H0 chunks, loads only       %4d wgs x 1024", nst);
    timeit(nm, [&]() { hipLaunchKernelGGL(k_h<0>, dim3(nst), dim3(1024), 0, 0, cloud, n, st_items, table, out); });
    snprintf(nm, sizeof nm, "H1 + floor + LDS histogram  %4d wgs x 1024", nst);
    timeit(nm, [&]() { hipLaunchKernelGGL(k_h<1>, dim3(nst), dim3(1024), 0, 0, cloud, n, st_items, table, out); });
    snprintf(nm, sizeof nm, "H2 + box                    %4d wgs x 1024", nst);
    timeit(nm, [&]() { hipLaunchKernelGGL(k_h<2>, dim3(nst), dim3(1024), 0, 0, cloud, n, st_items, table, out); });
    timeit_dirty(nm, [&]() { hipLaunchKernelGGL(k_h<2>, dim3(nst), dim3(1024), 0, 0, cloud, n, st_items, table, out); });
  }
  return 0;
}
