// Device -> host rate of one MI355X box for a frame-sized buffer (8.3 MB = 1920 x 1080 RGBA8): pinned and pageable destinations,
// the copy split over 1 / 2 / 4 streams (SDMA engines), and a kernel that stores straight into mapped pinned host memory.
//   hipcc --offload-arch=gfx950 -O2 -o tools/microbench/d2h_rate tools/microbench/d2h_rate.hip && tools/microbench/d2h_rate
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));                  \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

__global__ void store_to_host(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

int main() {
  const size_t bytes = 1920ull * 1080 * 4;
  void *d = nullptr, *pinned = nullptr, *mapped = nullptr;
  CK(hipMalloc(&d, bytes));
  CK(hipMemset(d, 0x5a, bytes));
  CK(hipHostMalloc(&pinned, bytes, hipHostMallocDefault));
  CK(hipHostMalloc(&mapped, bytes, hipHostMallocMapped));
  void *pageable = std::malloc(bytes);
  hipStream_t st[4];
  for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  auto time = [&](const char *what, auto &&fn) {
    for (int i = 0; i < 5; ++i) fn();
    CK(hipDeviceSynchronize());
    const int reps = 200;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) fn();
    CK(hipDeviceSynchronize());
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
    std::printf("%-52s %.3f ms per frame  %.1f GB/s\n", what, s * 1e3, bytes / s / 1e9);
    return 0;
  };
  for (int n = 1; n <= 4; n *= 2) {
    char name[96];
    std::snprintf(name, sizeof name, "hipMemcpyAsync -> pinned, %d stream(s)", n);
    time(name, [&] {
      const size_t part = bytes / n;
      for (int k = 0; k < n; ++k)
        (void)hipMemcpyAsync(static_cast<char *>(pinned) + k * part, static_cast<char *>(d) + k * part, part, hipMemcpyDeviceToHost, st[k]);
      for (int k = 0; k < n; ++k) (void)hipStreamSynchronize(st[k]);
    });
  }
  time("hipMemcpyAsync -> pinned, back to back (no sync between)", [&] { (void)hipMemcpyAsync(pinned, d, bytes, hipMemcpyDeviceToHost, st[0]); });
  time("two streams back to back", [&] {
    (void)hipMemcpyAsync(pinned, d, bytes / 2, hipMemcpyDeviceToHost, st[0]);
    (void)hipMemcpyAsync(static_cast<char *>(pinned) + bytes / 2, static_cast<char *>(d) + bytes / 2, bytes / 2, hipMemcpyDeviceToHost, st[1]);
  });
  time("hipMemcpy -> pageable", [&] { (void)hipMemcpy(pageable, d, bytes, hipMemcpyDeviceToHost); });
  void *mapped_dev = nullptr;
  CK(hipHostGetDevicePointer(&mapped_dev, mapped, 0));
  for (int blocks : {64, 256, 1024}) {
    char name[96];
    std::snprintf(name, sizeof name, "kernel stores into mapped host memory, %d blocks", blocks);
    time(name, [&] {
      hipLaunchKernelGGL(store_to_host, dim3(blocks), dim3(256), 0, st[0], static_cast<const uint4 *>(d), static_cast<uint4 *>(mapped_dev), bytes / 16);
      (void)hipStreamSynchronize(st[0]);
    });
  }
  return 0;
}
