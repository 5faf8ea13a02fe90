// Probe (not part of the library): issue cost of v_pk_fma_f32 against v_fma_f32 on gfx950, in isolation.
// 8 independent accumulators per lane, ITER x 8 instructions per wave; reports cycles per instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float s[8];
  f2 p[8];
  for (int i = 0; i < 8; ++i) { s[i] = threadIdx.x * 1e-3f + i; p[i] = f2{s[i], s[i] + 1.f}; }
  const f2 av = {a, a}, bv = {b, b};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(a), "v"(b));
      if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(av), "v"(bv));
      if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(p[i]) : "v"(av), "v"(bv));
      if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(av));
      if (MODE == 4) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(p[i]) : "v"(av), "v"(bv));
    }
  }
  float r = 0;
  for (int i = 0; i < 8; ++i) r += s[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE>
void run(const char* name, float* out, int wg_per_cu) {
  const int iters = 20000, blocks = 256 * wg_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001f, 1e-9f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  // per SIMD: wg_per_cu waves (256 threads = 4 waves = one per SIMD), each iters*8 instructions
  const double instr = (double)wg_per_cu * iters * 8;
  printf("%-28s waves/SIMD %d  %.3f ms  %.2f cycles/instr at 2.4 GHz\n", name, wg_per_cu, best, best * 1e-3 * 2.4e9 / instr);
}
int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  for (int w : {1, 2, 4, 8}) {
    if (w == 1) { run<0>("v_fma_f32", out, 1); run<1>("v_pk_fma_f32", out, 1); run<2>("v_pk_fma_f32 op_sel bcast", out, 1); run<3>("v_pk_mul_f32", out, 1); run<4>("v_fma_f64", out, 1); }
    if (w == 2) { run<0>("v_fma_f32", out, 2); run<1>("v_pk_fma_f32", out, 2); run<2>("v_pk_fma_f32 op_sel bcast", out, 2); run<3>("v_pk_mul_f32", out, 2); run<4>("v_fma_f64", out, 2); }
    if (w == 4) { run<0>("v_fma_f32", out, 4); run<1>("v_pk_fma_f32", out, 4); run<2>("v_pk_fma_f32 op_sel bcast", out, 4); run<3>("v_pk_mul_f32", out, 4); run<4>("v_fma_f64", out, 4); }
    if (w == 8) { run<0>("v_fma_f32", out, 8); run<1>("v_pk_fma_f32", out, 8); run<2>("v_pk_fma_f32 op_sel bcast", out, 8); run<3>("v_pk_mul_f32", out, 8); run<4>("v_fma_f64", out, 8); }
  }
  return 0;
}
