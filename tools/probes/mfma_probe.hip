// Probe (not part of the library): can the f32 screen of the RANSAC scoring loop go through
// v_mfma_f32_4x4x1_16b_f32 (own hypothesis per lane, 4 points per instruction)?
//  1. layout + bitwise equality with the scalar fma chain
//  2. issue cost: time of the MFMA formulation vs the scalar one at 4 waves/SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

// planes: per lane 4 floats (a,b,c,to); pts: n x float4 (u,v,w,1)
template <int MODE>
__global__ __launch_bounds__(256, 4) void k_probe(const f4* __restrict__ planes, const f4* __restrict__ pts,
                                                   int n, int iters, float nthr2, unsigned* __restrict__ out_cnt,
                                                   float* __restrict__ out_margin, float* __restrict__ out_s) {
  __shared__ f4 s_loc[256];
  for (int i = threadIdx.x; i < n; i += blockDim.x) s_loc[i] = pts[i];
  __syncthreads();
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  f4 P[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) P[q] = planes[(gid * 4 + q) % 4096];
  unsigned cnt[4] = {0, 0, 0, 0};
  float margin[2] = {__int_as_float(0x7f800000), __int_as_float(0x7f800000)};
  const int lane4 = threadIdx.x & 3;
  for (int it = 0; it < iters; ++it) {
    unsigned hist[4] = {0, 0, 0, 0};
    for (int base = 0; base < n; base += 4) {
      if (MODE == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f4 L = s_loc[base + r];
          float e[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float sv = fma32(P[q].x, L.x, fma32(P[q].y, L.y, fma32(P[q].z, L.z, P[q].w)));
            if (out_s && it == 0 && q == 0) out_s[(size_t)gid * n + base + r] = sv;
            e[q] = fma32(sv, sv, nthr2);
            hist[q] = __builtin_amdgcn_alignbit(hist[q], __float_as_uint(e[q]), 31);
          }
          margin[0] = min3abs(margin[0], e[0], e[1]);
          margin[1] = min3abs(margin[1], e[2], e[3]);
        }
      } else {
        // lane l supplies row i = l % 4 of its block: point base + i
        const f4 A = s_loc[base + lane4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f4 acc = {0.f, 0.f, 0.f, 0.f};
          acc = __builtin_amdgcn_mfma_f32_4x4x1f32(A.w, P[q].w, acc, 0, 0, 0);  // to * 1
          acc = __builtin_amdgcn_mfma_f32_4x4x1f32(A.z, P[q].z, acc, 0, 0, 0);  // + c w
          acc = __builtin_amdgcn_mfma_f32_4x4x1f32(A.y, P[q].y, acc, 0, 0, 0);  // + b v
          acc = __builtin_amdgcn_mfma_f32_4x4x1f32(A.x, P[q].x, acc, 0, 0, 0);  // + a u
          float e[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (out_s && it == 0 && q == 0) out_s[(size_t)gid * n + base + r] = acc[r];
            e[r] = fma32(acc[r], acc[r], nthr2);
            hist[q] = __builtin_amdgcn_alignbit(hist[q], __float_as_uint(e[r]), 31);
          }
          margin[q >> 1] = min3abs(margin[q >> 1], e[0], e[1]);
          margin[q >> 1] = min3abs(margin[q >> 1], e[2], e[3]);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) cnt[q] += __popc(hist[q]);
  }
  out_cnt[gid] = cnt[0] + 3 * cnt[1] + 5 * cnt[2] + 7 * cnt[3];
  out_margin[gid] = fminf(margin[0], margin[1]);
}

int main() {
  const int n = 24, blocks = 256 * 16, threads = 256, iters = 200;
  std::vector<float> planes(4096 * 4), pts(n * 4);
  srand(1);
  for (auto& v : planes) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (int i = 0; i < n; ++i) {
    for (int c = 0; c < 3; ++c) pts[4 * i + c] = (float)rand() / RAND_MAX * 0.3f;
    pts[4 * i + 3] = 1.0f;
  }
  f4 *d_pl, *d_pt; unsigned* d_cnt[2]; float* d_m[2]; float* d_s[2];
  hipMalloc(&d_pl, planes.size() * 4); hipMalloc(&d_pt, pts.size() * 4);
  hipMemcpy(d_pl, planes.data(), planes.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_pt, pts.data(), pts.size() * 4, hipMemcpyHostToDevice);
  const size_t total = (size_t)blocks * threads;
  for (int m = 0; m < 2; ++m) { hipMalloc(&d_cnt[m], total * 4); hipMalloc(&d_m[m], total * 4); hipMalloc(&d_s[m], total * n * 4); }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms[2];
  for (int rep = 0; rep < 2; ++rep) {
    for (int m = 0; m < 2; ++m) {
      float* s_out = rep == 0 ? d_s[m] : nullptr;
      hipEventRecord(e0);
      if (m == 0) hipLaunchKernelGGL(k_probe<0>, dim3(blocks), dim3(threads), 0, 0, d_pl, d_pt, n, iters, -0.0001f, d_cnt[0], d_m[0], s_out);
      else hipLaunchKernelGGL(k_probe<1>, dim3(blocks), dim3(threads), 0, 0, d_pl, d_pt, n, iters, -0.0001f, d_cnt[1], d_m[1], s_out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms[m], e0, e1);
    }
  }
  std::vector<unsigned> c0(total), c1(total); std::vector<float> s0(total * n), s1(total * n), m0(total), m1(total);
  hipMemcpy(c0.data(), d_cnt[0], total * 4, hipMemcpyDeviceToHost); hipMemcpy(c1.data(), d_cnt[1], total * 4, hipMemcpyDeviceToHost);
  hipMemcpy(s0.data(), d_s[0], total * n * 4, hipMemcpyDeviceToHost); hipMemcpy(s1.data(), d_s[1], total * n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(m0.data(), d_m[0], total * 4, hipMemcpyDeviceToHost); hipMemcpy(m1.data(), d_m[1], total * 4, hipMemcpyDeviceToHost);
  size_t bad_s = 0, bad_c = 0, bad_m = 0;
  for (size_t i = 0; i < total * n; ++i) bad_s += memcmp(&s0[i], &s1[i], 4) != 0;
  for (size_t i = 0; i < total; ++i) { bad_c += c0[i] != c1[i]; bad_m += memcmp(&m0[i], &m1[i], 4) != 0; }
  const double pairs = (double)total * 4 * n * iters;
  printf("scalar %.3f ms  mfma %.3f ms   (%.2f / %.2f Gpairs/s)   mismatches: s %zu of %zu, counts %zu, margins %zu\n", ms[0], ms[1],
         pairs / ms[0] / 1e6, pairs / ms[1] / 1e6, bad_s, total * n, bad_c, bad_m);
  if (bad_s) for (size_t i = 0, shown = 0; i < total * n && shown < 8; ++i) if (memcmp(&s0[i], &s1[i], 4)) { printf("  [%zu] scalar %.9g mfma %.9g\n", i, s0[i], s1[i]); ++shown; }
  return 0;
}
