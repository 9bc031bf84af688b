// tools/graph_probe.hip — what a kernel launch, an event fork/join and a hipGraph replay cost on the HOST (ROCm 7.2, gfx950).
// hipcc --offload-arch=gfx950 -O2 -o tools/graph_probe tools/graph_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e__), __LINE__); return 1; } } while (0)
__global__ void k_tiny(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
__global__ void k_spin(float* p, int iters) { float a = p[threadIdx.x]; for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f; p[threadIdx.x] = a; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const int NK = argc > 1 ? atoi(argv[1]) : 300;
  float* d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
  hipStream_t s0, s1, s2; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  std::vector<hipEvent_t> ev(4 * NK); for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (int i = 0; i < 100; ++i) k_tiny<<<1, 64, 0, s0>>>(d, 64);
  CK(hipStreamSynchronize(s0));
  const int REP = 20;
  // 1. plain launches on one stream
  { double t0 = now(); for (int r = 0; r < REP; ++r) for (int i = 0; i < NK; ++i) k_tiny<<<4, 256, 0, s0>>>(d, 1024); double t1 = now(); CK(hipStreamSynchronize(s0)); double t2 = now();
    printf("launch only: %.2f us/launch host, %.2f us/launch incl. drain\n", 1e6 * (t1 - t0) / REP / NK, 1e6 * (t2 - t0) / REP / NK); }
  // 1b. launches of ~20us kernels (GPU is the limiter: does the host block on a full queue?)
  { double t0 = now(); for (int r = 0; r < 4; ++r) for (int i = 0; i < NK; ++i) k_spin<<<256, 256, 0, s0>>>(d, 4000); double t1 = now(); CK(hipStreamSynchronize(s0)); double t2 = now();
    printf("launch of long kernels: %.2f us/launch host, %.2f us/kernel incl. drain\n", 1e6 * (t1 - t0) / 4 / NK, 1e6 * (t2 - t0) / 4 / NK); }
  // 2. every 4th launch forks to s1 and joins back (2 records + 2 waits)
  { double t0 = now(); int e = 0; for (int r = 0; r < REP; ++r) { e = 0; for (int i = 0; i < NK; ++i) { if (i % 4 == 0) { hipEventRecord(ev[e], s0); hipStreamWaitEvent(s1, ev[e], 0); ++e; k_tiny<<<4, 256, 0, s1>>>(d + 4096, 1024); hipEventRecord(ev[e], s1); hipStreamWaitEvent(s0, ev[e], 0); ++e; } else k_tiny<<<4, 256, 0, s0>>>(d, 1024); } }
    double t1 = now(); CK(hipDeviceSynchronize()); double t2 = now();
    printf("launch + fork/join every 4th: %.2f us/launch host, %.2f incl. drain (%d event ops per %d launches)\n", 1e6 * (t1 - t0) / REP / NK, 1e6 * (t2 - t0) / REP / NK, 2 * e, NK); }
  // 3. the same DAG captured into a hipGraph and replayed
  for (int variant = 0; variant < 2; ++variant) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    int e = 0;
    for (int i = 0; i < NK; ++i) {
      if (variant == 1 && i % 4 == 0) { hipEventRecord(ev[e], s0); hipStreamWaitEvent(s1, ev[e], 0); ++e; k_tiny<<<4, 256, 0, s1>>>(d + 4096, 1024); hipEventRecord(ev[e], s1); hipStreamWaitEvent(s0, ev[e], 0); ++e; }
      else k_tiny<<<4, 256, 0, s0>>>(d, 1024);
    }
    CK(hipStreamEndCapture(s0, &g));
    double ti = now(); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0)); double ti1 = now();
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
    for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s0));
    CK(hipStreamSynchronize(s0));
    double t0 = now(); for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ge, s0)); double t1 = now(); CK(hipStreamSynchronize(s0)); double t2 = now();
    printf("hipGraph (%s, %zu nodes, instantiate %.2f ms): launch %.1f us/graph host = %.2f us/node, %.1f us/graph incl. drain = %.2f us/node\n", variant ? "forked every 4th" : "linear", nn, 1e3 * (ti1 - ti),
           1e6 * (t1 - t0) / REP, 1e6 * (t1 - t0) / REP / NK, 1e6 * (t2 - t0) / REP, 1e6 * (t2 - t0) / REP / NK);
    // per-node parameter update cost (hipGraphExecKernelNodeSetParams): what a shape change per replay would cost
    std::vector<hipGraphNode_t> nodes(nn); CK(hipGraphGetNodes(g, nodes.data(), &nn));
    int upd = 0; double tu0 = now();
    for (size_t i = 0; i < nn; ++i) { hipGraphNodeType ty; CK(hipGraphNodeGetType(nodes[i], &ty)); if (ty != hipGraphNodeTypeKernel) continue; hipKernelNodeParams p; CK(hipGraphKernelNodeGetParams(nodes[i], &p)); p.gridDim.x = 5; CK(hipGraphExecKernelNodeSetParams(ge, nodes[i], &p)); ++upd; }
    double tu1 = now();
    printf("  exec-node param update: %.2f us/node (%d nodes)\n", 1e6 * (tu1 - tu0) / (upd ? upd : 1), upd);
    CK(hipGraphLaunch(ge, s0)); CK(hipStreamSynchronize(s0));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  printf("ok\n");
  return 0;
}
