// tools/host_call_cost_probe.hip — HOST time of the HIP calls a training step is made of (no profiler): kernel launches with small /
// large argument blocks, event record, stream wait, memset; back to back on busy streams.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/host_call_cost_probe tools/host_call_cost_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Small { void* p; long n; };
struct Big { char b[1024]; };
__global__ void k_small(Small a) { if (a.n < 0) ((int*)a.p)[0] = 1; }
__global__ void k_big(Big a) { if (a.b[5] == 77 && a.b[900] == 3) ((int*)nullptr)[0] = 1; }
__global__ void k_spin(long cycles) { long t0 = clock64(); while (clock64() - t0 < cycles) {} }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s[3];
  for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
  hipEvent_t ev[64];
  for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  void* buf; hipMalloc(&buf, 1 << 20);
  Small sa{buf, 1}; Big ba{};
  const int N = 2000;
  for (int busy = 0; busy < 2; ++busy) {
    for (int rep = 0; rep < 2; ++rep) {
      if (busy) hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s[0], 40000000L);   // ~20 ms of GPU work in front: the calls below only enqueue
      double t0 = now();
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s[0], sa);
      double t1 = now();
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_big, dim3(64), dim3(256), 0, s[0], ba);
      double t2 = now();
      for (int i = 0; i < N; ++i) hipEventRecord(ev[i & 63], s[0]);
      double t3 = now();
      for (int i = 0; i < N; ++i) { hipEventRecord(ev[i & 63], s[0]); hipStreamWaitEvent(s[1 + (i & 1)], ev[i & 63], 0); }
      double t4 = now();
      for (int i = 0; i < N; ++i) hipMemsetAsync(buf, 0, 4096, s[0]);
      double t5 = now();
      for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s[i % 3], sa); }
      double t6 = now();
      hipDeviceSynchronize();
      if (rep) printf("%s: launch(16 B args) %.2f us | launch(1 KB args) %.2f | event record %.2f | record + wait on another stream %.2f | memset %.2f | launch round-robin on 3 streams %.2f\n",
                      busy ? "device busy" : "device idle", (t1 - t0) / N, (t2 - t1) / N, (t3 - t2) / N, (t4 - t3) / N, (t5 - t4) / N, (t6 - t5) / N);
    }
  }
  return 0;
}
