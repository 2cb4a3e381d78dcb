// Host-side cost of the HIP calls a single-frame launch is made of (rt_shim.hip render_frames): microseconds per call, the queue
// kept short (a synchronisation every 64 iterations) so that the figures are the runtime's call costs, not back-pressure.
//   hipcc --offload-arch=gfx950 -O2 -o tools/microbench/host_calls tools/microbench/host_calls.hip && tools/microbench/host_calls
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

__global__ void tiny(int *p) {
  if (p && threadIdx.x == 1234567) *p = 1;
}
struct Big {
  char bytes[1400];
};
__global__ void tiny_big_args(Big a, Big b) {
  if (a.bytes[0] == 77 && b.bytes[5] == 99 && threadIdx.x == 1234567) __builtin_trap();
}

int main() {
  hipStream_t s0, s1;
  (void)hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
  (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  hipEvent_t timed[64], plain[64];
  for (auto &e : timed) (void)hipEventCreate(&e);
  for (auto &e : plain) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
  Big a{}, b{};
  auto run = [&](const char *what, auto &&fn) {
    for (int i = 0; i < 64; ++i) fn(i);
    (void)hipDeviceSynchronize();
    const int reps = 4096;
    double total = 0.0;
    for (int base = 0; base < reps; base += 64) {
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 64; ++i) fn(i);
      total += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      (void)hipDeviceSynchronize();
    }
    std::printf("%-64s %.2f us\n", what, total / reps * 1e6);
  };
  run("kernel launch, 8-byte arguments", [&](int) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s0, nullptr); });
  run("kernel launch, 2.8 KB of arguments (DeviceScene + FrameParams)", [&](int) { hipLaunchKernelGGL(tiny_big_args, dim3(1), dim3(64), 0, s0, a, b); });
  run("hipEventRecord, timing event", [&](int i) { (void)hipEventRecord(timed[i & 63], s0); });
  run("hipEventRecord, hipEventDisableTiming", [&](int i) { (void)hipEventRecord(plain[i & 63], s0); });
  run("kernel + hipEventRecord (plain) + hipStreamWaitEvent on another stream", [&](int i) {
    hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s0, nullptr);
    (void)hipEventRecord(plain[i & 63], s0);
    (void)hipStreamWaitEvent(s1, plain[i & 63], 0);
  });
  run("the single-frame sequence: wait, record, kernel, record, wait, record, kernel, record (timing events)", [&](int i) {
    (void)hipStreamWaitEvent(s0, timed[(i + 32) & 63], 0);
    (void)hipEventRecord(timed[i & 63], s0);
    hipLaunchKernelGGL(tiny_big_args, dim3(1), dim3(64), 0, s0, a, b);
    (void)hipEventRecord(plain[i & 63], s0);
    (void)hipStreamWaitEvent(s1, plain[i & 63], 0);
    (void)hipEventRecord(plain[(i + 1) & 63], s1);
    hipLaunchKernelGGL(tiny_big_args, dim3(1), dim3(64), 0, s1, a, b);
    (void)hipEventRecord(timed[(i + 32) & 63], s1);
  });
  run("the same without the timing-only records: wait, kernel, record, wait, kernel, record (plain events)", [&](int i) {
    (void)hipStreamWaitEvent(s0, plain[(i + 32) & 63], 0);
    hipLaunchKernelGGL(tiny_big_args, dim3(1), dim3(64), 0, s0, a, b);
    (void)hipEventRecord(plain[i & 63], s0);
    (void)hipStreamWaitEvent(s1, plain[i & 63], 0);
    hipLaunchKernelGGL(tiny_big_args, dim3(1), dim3(64), 0, s1, a, b);
    (void)hipEventRecord(plain[(i + 32) & 63], s1);
  });
  run("both kernels on one stream, one record", [&](int i) {
    hipLaunchKernelGGL(tiny_big_args, dim3(1), dim3(64), 0, s0, a, b);
    hipLaunchKernelGGL(tiny_big_args, dim3(1), dim3(64), 0, s0, a, b);
    (void)hipEventRecord(plain[i & 63], s0);
  });
  run("hipSetDevice", [&](int) { (void)hipSetDevice(0); });
  return 0;
}
