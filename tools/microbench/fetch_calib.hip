// Calibration of rocprofv3 FETCH_SIZE for the access pattern of the render kernel: 16-byte texel
// gathers at random addresses of a large table (the skybox), one per lane.  MI355X_MICROARCH.md:
// FETCH_SIZE reads 1/2 of the bytes of wide coalesced streams and is uncalibrated for other widths.
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib
// prints the number of gathers; each one touches one previously unseen 64-B sector / 128-B line.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void gather16(const float4* __restrict__ tab, size_t n_tex, float4* out, unsigned seed) {
  unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long z = (i + seed) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
  out[i] = tab[z % n_tex];
}
__global__ void stream16(const float4* __restrict__ tab, size_t n_tex, size_t offset, float4* out) {
  unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  out[i] = tab[(offset + i) % n_tex];   // never leaves the table
}
int main(int argc, char** argv) {
  const size_t table_bytes = (argc > 1 ? atol(argv[1]) : 512) << 20;   // MiB
  const size_t n_tex = table_bytes / 16, n = 4u << 20;                  // 4 Mi gathers
  float4 *tab, *out;
  hipMalloc(&tab, table_bytes); hipMalloc(&out, n * 16);
  hipMemset(tab, 1, table_bytes);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(gather16, dim3(n / 256), dim3(256), 0, 0, tab, n_tex, out, 1234567u * rep);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(stream16, dim3(n / 256), dim3(256), 0, 0, tab, n_tex, (size_t)rep * n, out);
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { printf("error %d\n", (int)e); return 1; }
  printf("table %zu MiB, %zu gathers/streamed float4 per launch: 64-B sectors = %zu bytes, useful = %zu bytes\n",
         table_bytes >> 20, n, n * 64, n * 16);
  return 0;
}
