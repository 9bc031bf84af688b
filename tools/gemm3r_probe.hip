// tools/gemm3r_probe.hip — k_lin3r alone (forward form, fp32 rows in and out) with compile-time ablations:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DW3R_ABL=<mask> -I graphtrans_amd/csrc -I include -o tools/gemm3_probe_r<mask> tools/gemm3r_probe.hip
// masks: 1 no DMA, 2 no row loads, 4 no MFMA, 8 no stores (linear3r.h)
#include <mutex>
#include <vector>
#include "gt_common.h"
#include "mfma_frag.h"
void gt_set_error(const char*, ...) {}
unsigned gt_prof_mask() { return 0; }
int64_t gt_prof_begin(const char*, hipStream_t, const int64_t*, int) { return -1; }
void gt_prof_end(int64_t, hipStream_t) {}
namespace {
using namespace gtf;
__device__ __forceinline__ uint32_t lin_hash(uint32_t s0, uint32_t s1, uint32_t row, uint32_t col) {
  uint32_t x = (row * 0x9E3779B1u + s0) ^ (col * 0x85EBCA77u + s1);
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
#include "linear32.h"
#include "linear3x.h"
#include "linear3r.h"
__global__ void k_empty(int) {}
}  // namespace
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e__)); return 1; } } while (0)
int main(int argc, char** argv) {
  int64_t shapes[][3] = {{31598, 300, 300}, {131072, 256, 256}, {12800, 300, 300}, {128, 300, 300}, {31598, 300, 3200}};
  for (auto& sh : shapes) {
    const int64_t M = sh[0], N = sh[1], K = sh[2];
    float *x, *w, *y, *b; void* img;
    CK(hipMalloc(&x, M * K * 4)); CK(hipMalloc(&w, N * K * 4)); CK(hipMalloc(&y, M * N * 4)); CK(hipMalloc(&b, N * 4));
    std::vector<float> h(M * K); for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f; CK(hipMemcpy(x, h.data(), M * K * 4, hipMemcpyHostToDevice));
    h.resize(N * K); for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f; CK(hipMemcpy(w, h.data(), N * K * 4, hipMemcpyHostToDevice));
    CK(hipMemset(b, 0, N * 4));
    const size_t ib = w3_image_bytes(N, K); CK(hipMalloc(&img, ib + 1024));
    W3Jobs jobs{}; W3Job& J = jobs.j[0];
    J.w = w; J.img = (unsigned char*)img; J.R = (int)N; J.C = (int)K; J.ldw = (int)K; J.transposed = 0; J.ntp = (int)w3_ntp(N); J.ksteps = (int)gt_cdiv(K, 32); J.block0 = 0; jobs.n = 1;
    hipLaunchKernelGGL(k_w3_image, dim3(J.ksteps * ((J.ntp + 3) / 4)), dim3(256), 0, 0, jobs);
    L32Args a{}; a.a = x; a.bias = b; a.out = y; a.M = M; a.Nout = N; a.Kc = K; a.lda = K; a.ldo = N; a.act = 1; a.inv_keep = 1.f; a.w3 = img;
    for (int i = 0; i < 5; ++i) w3r_launch(0, a);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    const int R = 50;
    for (int i = 0; i < R; ++i) w3r_launch(0, a);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("abl %d  %6lld x %4lld x %4lld : %7.1f us  (%6.1f TF fp32-equivalent, %.3f of the bf16x6 ceiling)\n", W3R_ABL, (long long)M, (long long)N, (long long)K, 1e3 * ms / R, 2.0 * M * N * K / (1e3 * ms / R) / 1e6, 2.0 * M * N * K / (1e3 * ms / R) / 1e6 / 416.7);
    CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(y)); CK(hipFree(b)); CK(hipFree(img));
  }
  // launch floor: an empty kernel of the same grid / block / LDS
  CK(hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_empty, dim3(247), dim3(512), 120 * 1024, 0, 0);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_empty, dim3(247), dim3(512), 120 * 1024, 0, 0);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("empty kernel, 247 x 512 threads, 120 KB LDS: %.1f us per launch\n", 1e3 * ms / 50);
  return 0;
}
