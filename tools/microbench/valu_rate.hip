// Microbenchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 vs v_sqrt_f32 on gfx950, by waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, unsigned long long *cyc) {
  float a[16];
  float2v p[16];
  for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = float2v{a[i], a[i] + 0.5f}; }
  const float m = 1.0001f, c = 0.0001f;
  const float2v pm = {m, m}, pc = {c, c};
  unsigned long long t0 = __builtin_readcyclecounter();
  t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) a[i] = __builtin_fmaf(a[i], m, c);
      if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], pm, pc);
      if (MODE == 2) a[i] = __builtin_amdgcn_sqrtf(a[i]);
      if (MODE == 3) asm volatile("v_cmp_le_f32 vcc, 0, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  float *out; unsigned long long *cyc;
  hipMalloc(&out, 4 * 256 * 4096); hipMalloc(&cyc, 8 * 4096);
  const int iters = 20000;
  const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_sqrt_f32", "v_cmp+v_cndmask"};
  for (int mode = 0; mode < 4; ++mode)
    for (int wps = 1; wps <= 8; wps *= 2) {  // waves per SIMD: blocks of 256 threads = 1 wave per SIMD each
      int blocks = 256 * wps;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> h(blocks);
      hipMemcpy(h.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
      double avg = 0; for (auto v : h) avg += v; avg /= blocks;
      double instr_per_wave = 16.0 * iters * (mode == 3 ? 2 : 1);
      // s_memtime ticks at 100 MHz? report both wall-derived and tick-derived
      double wave_instr_total = instr_per_wave * blocks * 4;  // 4 waves per block
      double per_simd = wave_instr_total / 1024.0;
      printf("%-16s waves/SIMD %d: %.3f ms, %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz), memtime ticks/instr %.3f\n",
             names[mode], wps, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4, avg / instr_per_wave);
    }
  return 0;
}
