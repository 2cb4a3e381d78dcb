// global_load_lds_dwordx4 on gfx950: where does lane i's 16 bytes land, with a non-zero LDS base and masked lanes?
//   hipcc --offload-arch=gfx950 -O3 -o lds_dma tools/microbench/lds_dma.hip && ./lds_dma
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ void k(const float4 *src, float4 *dst, unsigned n_active) {
  __shared__ float4 pad[37];    // push the window off LDS address 0
  __shared__ float4 win[2][64];
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  pad[lane % 37] = make_float4(-1, -1, -1, -1);
  win[wave][lane] = make_float4(-2, -2, -2, -2);
  __syncthreads();
  // a scattered source address per lane: record (lane * 7) % 64 of the block's 64
  if (lane < n_active) __builtin_amdgcn_global_load_lds(src + (blockIdx.x * 2 + wave) * 64 + (lane * 7u) % 64u, &win[wave][0], 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
  __syncthreads();
  dst[(blockIdx.x * 2 + wave) * 64 + lane] = win[wave][lane];
}

int main() {
  const unsigned blocks = 4, n = blocks * 128;
  std::vector<float4> h(n), out(n);
  for (unsigned i = 0; i < n; ++i) h[i] = make_float4(float(i), float(i) + 0.25f, float(i) + 0.5f, float(i) + 0.75f);
  float4 *d_src, *d_dst;
  hipMalloc(&d_src, n * sizeof(float4));
  hipMalloc(&d_dst, n * sizeof(float4));
  hipMemcpy(d_src, h.data(), n * sizeof(float4), hipMemcpyHostToDevice);
  int bad = 0;
  for (unsigned n_active : {64u, 32u, 5u}) {
    hipLaunchKernelGGL(k, dim3(blocks), dim3(128), 0, 0, d_src, d_dst, n_active);
    hipMemcpy(out.data(), d_dst, n * sizeof(float4), hipMemcpyDeviceToHost);
    for (unsigned w = 0; w < blocks * 2; ++w)
      for (unsigned lane = 0; lane < 64; ++lane) {
        const float4 got = out[w * 64 + lane];
        const float want = lane < n_active ? float(w * 64 + (lane * 7u) % 64u) : -2.0f;
        if (got.x != want) {
          if (bad < 8) printf("n_active %u wave %u lane %u: got %.2f want %.2f\n", n_active, w, lane, got.x, want);
          ++bad;
        }
      }
  }
  printf("lds_dma: %d mismatches (lane i's 16 bytes land at window[i]; masked lanes write nothing)\n", bad);
  return bad != 0;
}
