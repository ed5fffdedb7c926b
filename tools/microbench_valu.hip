// tools/microbench_valu.hip -- issue-rate probe for the VALU instructions the NN
// kernels are made of (gfx950).  Not part of the product; numbers quoted in
// DESIGN.md come from here.   hipcc -O3 --offload-arch=gfx950 tools/microbench_valu.hip -o /tmp/mb && /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float seed) {
  float a0 = threadIdx.x * 1e-3f + seed, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
  float a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  const float b = 1.0001f, c = 1e-7f;
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
  float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  const float2v pb = {b, b}, pc = {c, c};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (MODE == 0) {  // v_fma_f32 x8 independent
        a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c);
        a4 = __builtin_fmaf(a4, b, c); a5 = __builtin_fmaf(a5, b, c); a6 = __builtin_fmaf(a6, b, c); a7 = __builtin_fmaf(a7, b, c);
      } else if (MODE == 1) {  // v_pk_fma_f32 x4 (8 lanes-worth of fma)
        p0 = __builtin_elementwise_fma(p0, pb, pc); p1 = __builtin_elementwise_fma(p1, pb, pc);
        p2 = __builtin_elementwise_fma(p2, pb, pc); p3 = __builtin_elementwise_fma(p3, pb, pc);
      } else if (MODE == 2) {  // v_fma_f64 x4
        d0 = __builtin_fma(d0, 1.0000001, 1e-9); d1 = __builtin_fma(d1, 1.0000001, 1e-9);
        d2 = __builtin_fma(d2, 1.0000001, 1e-9); d3 = __builtin_fma(d3, 1.0000001, 1e-9);
      } else if (MODE == 3) {  // v_min3_f32 x8
        a0 = __builtin_fminf(__builtin_fminf(a0, a1), c); a1 = __builtin_fminf(__builtin_fminf(a1, a2), b);
        a2 = __builtin_fminf(__builtin_fminf(a2, a3), c); a3 = __builtin_fminf(__builtin_fminf(a3, a4), b);
        a4 = __builtin_fminf(__builtin_fminf(a4, a5), c); a5 = __builtin_fminf(__builtin_fminf(a5, a6), b);
        a6 = __builtin_fminf(__builtin_fminf(a6, a7), c); a7 = __builtin_fminf(__builtin_fminf(a7, a0), b);
      } else if (MODE == 4) {  // cvt f32->f64->f32 pairs x4 (2 instr each)
        a0 = (float)((double)a0 * 1.0000001); a1 = (float)((double)a1 * 1.0000001);
        a2 = (float)((double)a2 * 1.0000001); a3 = (float)((double)a3 * 1.0000001);
      } else if (MODE == 5) {  // v_sqrt_f32 (approx) x8
        a0 = __builtin_amdgcn_sqrtf(a0) + b; a1 = __builtin_amdgcn_sqrtf(a1) + b; a2 = __builtin_amdgcn_sqrtf(a2) + b; a3 = __builtin_amdgcn_sqrtf(a3) + b;
        a4 = __builtin_amdgcn_sqrtf(a4) + b; a5 = __builtin_amdgcn_sqrtf(a5) + b; a6 = __builtin_amdgcn_sqrtf(a6) + b; a7 = __builtin_amdgcn_sqrtf(a7) + b;
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3) + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
double run(const char* name, double instr_per_iter, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd;  // 4 waves per block, 4 SIMDs per CU
  float* out;
  hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE><<<blocks, 256>>>(out, 100, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<MODE><<<blocks, 256>>>(out, iters, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double wave_instr = (double)blocks * 4 * iters * instr_per_iter;
  const double per_simd_per_s = wave_instr / 1024.0 / (ms * 1e-3);
  printf("%-34s waves/SIMD=%d  %.3f ms  %.1f G wave-instr/s/SIMD  -> %.2f cycles/instr @2.4GHz\n", name, waves_per_simd, ms,
         per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
  hipFree(out);
  return ms;
}

int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_fma_f32", 64, w);
    run<1>("v_pk_fma_f32", 32, w);
    run<2>("v_fma_f64", 32, w);
    run<3>("v_min3_f32", 64, w);
    run<4>("cvt_f64_f32+mul_f64+cvt_f32_f64", 32 * 3, w);
    run<5>("v_sqrt_f32 + v_add_f32", 64 * 2, w);
  }
  return 0;
}
