// Does s_barrier ignore waves of the workgroup that have already ended?  (needed for block-level
// ray compaction where surplus waves exit between bounces)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  __shared__ int x[4];
  const int wave = threadIdx.x >> 6;
  if (threadIdx.x < 4) x[threadIdx.x] = 0;
  __syncthreads();
  for (int round = 0; round < 4; ++round) {
    if (wave >= 4 - round) return;            // one more wave leaves every round
    if ((threadIdx.x & 63) == 0) atomicAdd(&x[round], 1);
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[blockIdx.x * 4 + 0] = x[0]; out[blockIdx.x * 4 + 1] = x[1]; out[blockIdx.x * 4 + 2] = x[2]; out[blockIdx.x * 4 + 3] = x[3]; }
}
int main() {
  int* d; hipMalloc(&d, 1024 * 16); hipMemset(d, 0xff, 1024 * 16);
  hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, d);
  hipError_t e = hipDeviceSynchronize();
  int h[8]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("sync=%d counts %d %d %d %d | %d %d %d %d (expect 4 3 2 1)\n", (int)e, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
  return 0;
}
