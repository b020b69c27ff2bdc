// event_cost_probe.hip -- what one timing point between two kernels of a stream costs in device time:
// nothing, a hipEventRecord (timing event), a hipEventRecord (no-timing event), a one-thread stamp kernel
// writing wall_clock64() to memory.  Prints microseconds per link of a chain of short kernels.
// Build: hipcc -O2 --offload-arch=gfx950 -o scripts/bin/event_cost_probe scripts/probes/event_cost_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(int cycles) { const unsigned long long t0 = clock64(); while (clock64() - t0 < (unsigned long long)cycles) {} }
__global__ void stamp(unsigned long long *slot) { *slot = wall_clock64(); }
int main(int argc, char **argv) {
  const int n = 64, cycles = argc > 1 ? atoi(argv[1]) : 20000, reps = 20;
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t s; CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi));
  std::vector<hipEvent_t> ev(n), evn(n);
  for (auto &e : ev) CK(hipEventCreate(&e));
  for (auto &e : evn) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  unsigned long long *slots; CK(hipMalloc(&slots, n * 8));
  const char *names[] = {"nothing", "hipEventRecord (timing)", "hipEventRecord (hipEventDisableTiming)", "stamp kernel", "two hipEventRecords", "two stamp kernels"};
  double base = 0;
  for (int mode = 0; mode < 6; ++mode) {
    auto chain = [&]() {
      for (int i = 0; i < n; ++i) {
        hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, cycles);
        if (mode == 1 || mode == 4) hipEventRecord(ev[i], s);
        if (mode == 4) hipEventRecord(evn[i], s);
        if (mode == 2) hipEventRecord(evn[i], s);
        if (mode == 3 || mode == 5) hipLaunchKernelGGL(stamp, dim3(1), dim3(1), 0, s, slots + i);
        if (mode == 5) hipLaunchKernelGGL(stamp, dim3(1), dim3(1), 0, s, slots + i);
      }
    };
    chain(); CK(hipStreamSynchronize(s));
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) chain();
    CK(hipStreamSynchronize(s));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps / n;
    if (mode == 0) base = us;
    printf("%-42s %.2f us per link (+%.2f)\n", names[mode], us, us - base);
  }
  int khz = 0; CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0));
  unsigned long long h[2]; CK(hipMemcpy(h, slots, 16, hipMemcpyDeviceToHost));
  printf("wall clock %d kHz; two consecutive links' stamps differ by %.2f us\n", khz, (double)(h[1] - h[0]) / khz * 1e3);
  return 0;
}
