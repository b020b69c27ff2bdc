// graph_probe.hip -- does a captured hipGraph shorten the gap at a CROSS-STREAM dependency?
// A chain of n short kernels alternating between two streams (each waits for the previous one through an event),
// run eagerly and as an instantiated graph; and the same chain on one stream.  Prints microseconds per link.
// Build: hipcc -O2 --offload-arch=gfx950 -o scripts/bin/graph_probe scripts/probes/graph_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin(unsigned long long *sink, int cycles) {
  const unsigned long long t0 = clock64();
  while (clock64() - t0 < (unsigned long long)cycles) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) { sink[0] = t0; sink[4] = (unsigned long long)cycles; }
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 40, cycles = argc > 2 ? atoi(argv[2]) : 2000, reps = 20;
  unsigned long long *sink;
  CK(hipMalloc(&sink, 128));
  hipStream_t s[2];
  int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithPriority(&s[0], hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithPriority(&s[1], hipStreamNonBlocking, lo));
  std::vector<hipEvent_t> ev(n);
  for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  auto chain = [&](bool cross) -> int {
    for (int i = 0; i < n; ++i) {
      hipStream_t st = s[cross ? i & 1 : 0];
      if (cross && i) CK(hipStreamWaitEvent(st, ev[i - 1], 0));
      hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, sink, cycles);
      if (cross) CK(hipEventRecord(ev[i], st));
    }
    if (cross) CK(hipStreamWaitEvent(s[0], ev[n - 1], 0));
    return 0;
  };
  auto time_it = [&](auto &&f) -> double {
    f();
    hipStreamSynchronize(s[0]); hipStreamSynchronize(s[1]);
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) f();
    hipStreamSynchronize(s[0]); hipStreamSynchronize(s[1]);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps / n;
  };
  for (int cross = 0; cross < 2; ++cross) {
    const double eager = time_it([&] { chain(cross); });
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeGlobal));
    if (chain(cross)) return 1;
    CK(hipStreamEndCapture(s[0], &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    const double graph = time_it([&] { hipGraphLaunch(ge, s[0]); });
    printf("%s chain of %d kernels (%d cycles each): eager %.2f us per link, graph %.2f us per link\n", cross ? "two-stream" : "one-stream", n, cycles, eager, graph);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  // fork-join, the sweep's shape: A on s0, then B (long) on s1 beside C (short) on s0, joined before the next A
  {
    const int m = n / 4 > 0 ? n / 4 : 1;
    hipEvent_t ef, ej;
    CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    auto fj = [&]() -> int {
      for (int i = 0; i < m; ++i) {
        hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s[0], sink, cycles);          // A
        CK(hipEventRecord(ef, s[0]));
        CK(hipStreamWaitEvent(s[1], ef, 0));
        hipLaunchKernelGGL(spin, dim3(128), dim3(256), 0, s[1], sink + 1, 2 * cycles);  // B: half the machine, twice as long
        hipLaunchKernelGGL(spin, dim3(128), dim3(256), 0, s[0], sink + 2, cycles);      // C
        CK(hipEventRecord(ej, s[1]));
        CK(hipStreamWaitEvent(s[0], ej, 0));
      }
      return 0;
    };
    auto time_fj = [&](auto &&f) -> double {
      f();
      hipStreamSynchronize(s[0]); hipStreamSynchronize(s[1]);
      const auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < reps; ++r) f();
      hipStreamSynchronize(s[0]); hipStreamSynchronize(s[1]);
      return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps / m;
    };
    const double eager = time_fj([&] { fj(); });
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeGlobal));
    if (fj()) return 1;
    CK(hipStreamEndCapture(s[0], &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    const double graph = time_fj([&] { hipGraphLaunch(ge, s[0]); });
    printf("fork-join x%d (A %d cycles, then B %d beside C %d): eager %.2f us per round, graph %.2f us per round (kernels alone: 3x the one-stream link time above)\n", m, cycles, 2 * cycles, cycles, eager, graph);
  }
  // the form a sweep would take: re-captured every round (new kernel arguments, other timing events), the executable
  // graph updated in place (hipGraphExecUpdate), launched; host cost per round and whether the update took
  {
    const int m = 200;
    hipEvent_t ef, ej, t0e[2], t1e[2];
    CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    for (int i = 0; i < 2; ++i) { CK(hipEventCreate(&t0e[i])); CK(hipEventCreate(&t1e[i])); }
    unsigned long long *marks;
    CK(hipMalloc(&marks, 8 * 8));
    auto round = [&](int i) -> int {
      CK(hipEventRecord(t0e[i & 1], s[0]));
      CK(hipMemsetAsync(marks + 4, 0, 8, s[0]));
      hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s[0], sink, cycles + i);
      CK(hipEventRecord(ef, s[0]));
      CK(hipStreamWaitEvent(s[1], ef, 0));
      hipLaunchKernelGGL(spin, dim3(128), dim3(256), 0, s[1], sink + 1, 2 * cycles);
      hipLaunchKernelGGL(spin, dim3(128), dim3(256), 0, s[0], sink + 2, cycles);
      CK(hipEventRecord(ej, s[1]));
      CK(hipStreamWaitEvent(s[0], ej, 0));
      CK(hipEventRecord(t1e[i & 1], s[0]));
      return 0;
    };
    hipGraphExec_t ge = nullptr;
    int updates = 0, reinst = 0;
    double host_us = 0;
    hipStreamSynchronize(s[0]);
    const auto w0 = std::chrono::steady_clock::now();
    for (int i = 0; i < m; ++i) {
      const auto h0 = std::chrono::steady_clock::now();
      hipGraph_t g;
      CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
      if (round(i)) return 1;
      CK(hipStreamEndCapture(s[0], &g));
      bool ok = false;
      if (ge) {
        hipGraphNode_t bad;
        hipGraphExecUpdateResult res;
        ok = hipGraphExecUpdate(ge, g, &bad, &res) == hipSuccess;
        if (ok) ++updates; else { (void)hipGetLastError(); hipGraphExecDestroy(ge); ge = nullptr; }
      }
      if (!ge) { CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0)); ++reinst; }
      CK(hipGraphDestroy(g));
      CK(hipGraphLaunch(ge, s[0]));
      host_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
    }
    hipStreamSynchronize(s[0]);
    const double wall = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count() / m;
    float e0 = -1, e1 = -1;
    const hipError_t r0 = hipEventElapsedTime(&e0, t0e[0], t1e[0]), r1 = hipEventElapsedTime(&e1, t0e[1], t1e[1]);
    unsigned long long seen = 0;
    CK(hipMemcpy(&seen, sink + 4, 8, hipMemcpyDeviceToHost));
    printf("re-captured + updated every round x%d: %.2f us per round (wall), host %.2f us per round; %d updates, %d instantiations; "
           "event pair 0: %s %.2f us, pair 1: %s %.2f us; last kernel argument seen %llu (expected %d)\n",
           m, wall, host_us / m, updates, reinst, hipGetErrorName(r0), e0 * 1e3, hipGetErrorName(r1), e1 * 1e3, seen, cycles + m - 1);
  }
  return 0;
}
