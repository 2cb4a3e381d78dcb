#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4v __attribute__((ext_vector_type(4)));
struct S { const float4* p; unsigned n; };
__global__ void k(S s, float* out) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((address_space(4))) const f4v* cptr;
  const cptr gp = (cptr)(const void*)s.p;
  float acc = 0;
  for (unsigned i = 0; i < s.n; i += 4) {
    const auto g0 = gp[i], g1 = gp[i+1], g2 = gp[i+2], g3 = gp[i+3];
    acc += (threadIdx.x - g0.x) * g0.w + (threadIdx.x - g1.y) * g1.w + g2.z + g3.w;
  }
  out[threadIdx.x] = acc;
#endif
}
int main() {
  float4 h[16]; for (int i = 0; i < 16; ++i) h[i] = make_float4(i, i + 0.25f, i + 0.5f, i + 0.75f);
  float4* d; float* o; hipMalloc(&d, sizeof h); hipMalloc(&o, 64 * 4);
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  S s{d, 8};
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, o);
  float r[64]; hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
  double e0 = 0; for (unsigned i = 0; i < 8; i += 4) e0 += (0 - h[i].x) * h[i].w + (0 - h[i+1].y) * h[i+1].w + h[i+2].z + h[i+3].w;
  printf("lane0 %f expected %f ; lane5 %f\n", r[0], e0, r[5]);
  return 0;
}
