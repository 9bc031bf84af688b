// tools/event_cost_probe.hip — what a hipEventRecord / hipStreamWaitEvent pair costs ON THE GPU TIMELINE of the recording stream:
// a chain of N dependent ~10-us kernels on stream A with (a) nothing, (b) an event record, (c) a record + a wait on stream B,
// (d) a record + a wait on B + a tiny kernel on B, between consecutive kernels.  hipcc --offload-arch=gfx950 -O3 -o tools/event_cost_probe tools/event_cost_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(float* p, int iters) {
  float v = p[threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  p[threadIdx.x] = v;
}
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e__)); return 1; } } while (0)
int main() {
  float *p, *q;
  CK(hipMalloc(&p, 4096)); CK(hipMalloc(&q, 4096));
  hipStream_t A, B;
  CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  const int N = 200, EV = 64;
  hipEvent_t ev[EV], t0, t1;
  for (int i = 0; i < EV; ++i) CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
  CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  for (int iters : {2000, 8000}) {
    for (int mode = 0; mode < 5; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(t0, A));
        for (int i = 0; i < N; ++i) {
          hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, A, p, iters);
          if (mode >= 1) CK(hipEventRecord(ev[i % EV], A));
          if (mode >= 2) CK(hipStreamWaitEvent(B, ev[i % EV], 0));
          if (mode >= 3) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, B, q, 10);
          if (mode >= 4) { CK(hipEventRecord(ev[(i + 32) % EV], A)); CK(hipStreamWaitEvent(B, ev[(i + 32) % EV], 0)); CK(hipEventRecord(ev[(i + 16) % EV], A)); }
        }
        CK(hipEventRecord(t1, A));
        CK(hipEventSynchronize(t1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, t0, t1));
        if (rep) printf("iters %5d mode %d (%s): %.2f us per kernel step\n", iters, mode,
                        mode == 0 ? "kernels only" : mode == 1 ? "+ record" : mode == 2 ? "+ record + wait on B" : mode == 3 ? "+ record + wait + kernel on B" : "+ 3 records + 2 waits + kernel on B", 1e3 * ms / N);
      }
    }
  }
  return 0;
}
