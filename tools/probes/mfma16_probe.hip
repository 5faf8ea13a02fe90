// Probe (not part of the library): would the f32 screen of k_ransac gain from v_mfma_f32_16x16x4_f32 when it runs
// beside the plane fits' f64 VALU work?  k_ransac is VALU-ISSUE bound (round 5: its time is its VALU instruction count x
// 4 cycles); the screen's three dot-product FMAs per (point, hypothesis) are 19 % of those instructions.  This probe
// times, per 256 (point, hypothesis) pairs: FILL f64 FMAs (the fits' share of the stream: ~40 per 256 pairs) plus
//   MODE 0  the scalar screen: 3 v_fma_f32 + square + alignbit + half a min3 per pair and lane  (22 VALU per 256 pairs)
//   MODE 1  one 16x16x4 MFMA for the 256 dot products + square + alignbit + half a min3 per output (10 VALU + 1 MFMA)
// Layout and ownership of the outputs are NOT what the kernel would need (no cross-lane regrouping here): this is an
// upper bound of what the matrix pipe can take off the VALU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float fma32(float a, float b, float c) {
  float r;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float min3abs(float m, float a, float b) {
  float r;
  asm("v_min3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(m), "v"(a), "v"(b));
  return r;
}

template <int MODE, int FILL>
__global__ __launch_bounds__(64, 4) void k_probe(const f4* __restrict__ planes, const f4* __restrict__ pts, int n,
                                                  int iters, float nthr2, unsigned* __restrict__ out_cnt,
                                                  double* __restrict__ out_d) {
  __shared__ f4 s_loc[64];
  for (int i = threadIdx.x; i < n; i += blockDim.x) s_loc[i] = pts[i];
  __syncthreads();
  const int gid = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
  f4 P[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) P[q] = planes[(gid * 4 + q) % 4096];
  unsigned cnt = 0;
  float margin = __int_as_float(0x7f800000);
  double d0 = 1.0 + gid * 1e-9, d1 = 1.5, d2 = 0.75, d3 = 1.25;
  const double m0 = 1.0000001, m1 = 0.9999999;
  for (int it = 0; it < iters; ++it) {
    for (int base = 0; base + 16 <= n; base += 16) {
      unsigned hist[4] = {0, 0, 0, 0};
      // the fits' share: FILL independent-ish f64 FMAs
#pragma unroll
      for (int f = 0; f < FILL / 4; ++f) {
        d0 = fma(d0, m0, 1e-12);
        d1 = fma(d1, m1, 1e-12);
        d2 = fma(d2, m0, -1e-12);
        d3 = fma(d3, m1, -1e-12);
      }
      if (MODE == 0) {
        // 16 points x 4 hypotheses of the lane = 256 pairs per wave... per LANE 64 pairs: keep the per-wave total at 256
        // pairs per 4 points x 1 hypothesis-group: 4 points x (the lane's 4 hypotheses)  -> do 4 points here
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f4 L = s_loc[base + r];
          float e[4];
#pragma unroll
          for (int q = 0; q < 1; ++q) {
            const float sv = fma32(P[q].x, L.x, fma32(P[q].y, L.y, fma32(P[q].z, L.z, P[q].w)));
            e[q] = fma32(sv, sv, nthr2);
            hist[q] = __builtin_amdgcn_alignbit(hist[q], __float_as_uint(e[q]), 31);
          }
          if (r & 1) margin = min3abs(margin, e[0], e[0]);
        }
      } else {
        // A[m = lane % 16][k = lane / 16]: component k of point base + m; B[k][n = lane % 16]: component k of "hypothesis" n
        const f4 Lp = s_loc[base + (lane & 15)];
        const int kc = lane >> 4;
        const float av = kc == 0 ? Lp.x : (kc == 1 ? Lp.y : (kc == 2 ? Lp.z : 1.0f));
        const float bv = kc == 0 ? P[0].x : (kc == 1 ? P[0].y : (kc == 2 ? P[0].z : P[0].w));
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
        float e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          e[r] = fma32(acc[r], acc[r], nthr2);
          hist[0] = __builtin_amdgcn_alignbit(hist[0], __float_as_uint(e[r]), 31);
        }
        margin = min3abs(margin, e[0], e[1]);
        margin = min3abs(margin, e[2], e[3]);
      }
      cnt += __popc(hist[0]);
    }
  }
  out_cnt[gid] = cnt + (unsigned)(margin > 0.f);
  out_d[gid] = d0 + d1 + d2 + d3;
}

template <int MODE, int FILL>
float run(const f4* pl, const f4* pt, int n, int iters, unsigned* cnt, double* dd, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_probe<MODE, FILL>), dim3(blocks), dim3(64), 0, 0, pl, pt, n, iters, -0.0001f, cnt, dd);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  return best;
}

int main() {
  const int n = 32, blocks = 256 * 64, iters = 400;
  std::vector<float> planes(4096 * 4), pts(64 * 4);
  srand(1);
  for (auto& v : planes) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (int i = 0; i < 64; ++i) {
    for (int c = 0; c < 3; ++c) pts[4 * i + c] = (float)rand() / RAND_MAX * 0.3f;
    pts[4 * i + 3] = 1.0f;
  }
  f4 *d_pl, *d_pt;
  unsigned* d_cnt;
  double* d_d;
  hipMalloc(&d_pl, planes.size() * 4);
  hipMalloc(&d_pt, pts.size() * 4);
  hipMemcpy(d_pl, planes.data(), planes.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_pt, pts.data(), pts.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&d_cnt, (size_t)blocks * 64 * 4);
  hipMalloc(&d_d, (size_t)blocks * 64 * 8);
  // per inner step and wave: MODE 0 = 4 points x 64 lanes = 256 pairs; MODE 1 = one 16x16 tile = 256 pairs
  printf("FILL  0: scalar %.3f ms   mfma16 %.3f ms\n", run<0, 0>(d_pl, d_pt, n, iters, d_cnt, d_d, blocks), run<1, 0>(d_pl, d_pt, n, iters, d_cnt, d_d, blocks));
  printf("FILL 20: scalar %.3f ms   mfma16 %.3f ms\n", run<0, 20>(d_pl, d_pt, n, iters, d_cnt, d_d, blocks), run<1, 20>(d_pl, d_pt, n, iters, d_cnt, d_d, blocks));
  printf("FILL 40: scalar %.3f ms   mfma16 %.3f ms\n", run<0, 40>(d_pl, d_pt, n, iters, d_cnt, d_d, blocks), run<1, 40>(d_pl, d_pt, n, iters, d_cnt, d_d, blocks));
  printf("FILL 80: scalar %.3f ms   mfma16 %.3f ms\n", run<0, 80>(d_pl, d_pt, n, iters, d_cnt, d_d, blocks), run<1, 80>(d_pl, d_pt, n, iters, d_cnt, d_d, blocks));
  return 0;
}
