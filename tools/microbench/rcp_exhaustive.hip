// Which FMA refinement of v_rcp_f32 equals the IEEE reciprocal 1.0f / x for which inputs?  Every float, on the device.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o rcp_exhaustive tools/microbench/rcp_exhaustive.hip && ./rcp_exhaustive
#include <hip/hip_runtime.h>

#include <cstdio>

__device__ __forceinline__ float cand_a(float s) {  // one Newton step
  const float r = __builtin_amdgcn_rcpf(s);
  return fmaf(fmaf(-s, r, 1.0f), r, r);
}
__device__ __forceinline__ float cand_b(float s) {  // two
  float r = __builtin_amdgcn_rcpf(s);
  r = fmaf(fmaf(-s, r, 1.0f), r, r);
  return fmaf(fmaf(-s, r, 1.0f), r, r);
}

// out[c][0] mismatches, out[c][1] / [2] smallest / largest |x| (bits) among the mismatching NORMAL inputs whose
// reciprocal is normal too (2^-126 <= |x| <= 2^126), out[c][3] mismatches outside that range
__global__ void sweep(unsigned long long *out) {
  const unsigned stride = gridDim.x * blockDim.x;
  unsigned long long bad[2] = {0, 0}, outside[2] = {0, 0};
  unsigned lo[2] = {0xffffffffu, 0xffffffffu}, hi[2] = {0, 0};
  for (unsigned long long u = blockIdx.x * blockDim.x + threadIdx.x; u <= 0xffffffffull; u += stride) {
    const float x = __uint_as_float(static_cast<unsigned>(u));
    const unsigned want = __float_as_uint(1.0f / x);
    const unsigned got[2] = {__float_as_uint(cand_a(x)), __float_as_uint(cand_b(x))};
    const unsigned mag = static_cast<unsigned>(u) & 0x7fffffffu;
    const bool inside = mag >= 0x00800000u && mag <= 0x7e800000u;  // 2^-126 .. 2^126
    const bool nan = mag > 0x7f800000u;
    for (int c = 0; c < 2; ++c) {
      const bool same = got[c] == want || (nan && (got[c] & 0x7fffffffu) > 0x7f800000u);
      if (!same) {
        if (inside) {
          ++bad[c];
          lo[c] = min(lo[c], mag);
          hi[c] = max(hi[c], mag);
        } else {
          ++outside[c];
        }
      }
    }
  }
  for (int c = 0; c < 2; ++c) {
    atomicAdd(&out[c * 4 + 0], bad[c]);
    atomicMin(&out[c * 4 + 1], static_cast<unsigned long long>(lo[c]));
    atomicMax(&out[c * 4 + 2], static_cast<unsigned long long>(hi[c]));
    atomicAdd(&out[c * 4 + 3], outside[c]);
  }
}

int main() {
  unsigned long long h[8] = {0, ~0ull, 0, 0, 0, ~0ull, 0, 0}, *d;
  hipMalloc(&d, sizeof h);
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(sweep, dim3(8192), dim3(256), 0, 0, d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char *names[2] = {"rcp + 1 Newton step (3 instructions)", "rcp + 2 Newton steps (5 instructions)"};
  for (int c = 0; c < 2; ++c)
    printf("%-40s: %llu mismatches among 2^-126 <= |x| <= 2^126 (|x| bits %08llx .. %08llx), %llu outside\n", names[c], h[c * 4],
           h[c * 4 + 1], h[c * 4 + 2], h[c * 4 + 3]);
  return 0;
}
