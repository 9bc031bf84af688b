// tools/gemm3r_dw_probe.hip — k_lin3r_dw alone (no reduce) with compile-time ablations:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DW3RD_ABL=<mask> -I graphtrans_amd/csrc -I include -o tools/gemm3_probe_dw<mask> tools/gemm3r_dw_probe.hip
// masks: 1 no MFMA, 2 no fragment reads, 4 no split / plane stores, 8 no row loads, 16 no partial stores (linear3r.h)
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
__device__ __forceinline__ uint32_t lin_hash(uint32_t s0, uint32_t s1, uint32_t row, uint32_t col) { return row ^ col ^ s0 ^ s1; }
#include "linear32.h"
#include "linear3x.h"
#include "linear3r.h"
}  // namespace
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e__)); return 1; } } while (0)
int main() {
  int64_t shapes[][3] = {{31598, 300, 300}, {131072, 256, 256}, {31598, 160, 160}, {320000, 300, 300}};
  for (auto& sh : shapes) {
    const int64_t M = sh[0], N = sh[1], K = sh[2];
    float *x, *dy, *part;
    CK(hipMalloc(&x, M * K * 4)); CK(hipMalloc(&dy, M * N * 4));
    std::vector<float> h(M * (K > N ? K : N)); for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
    CK(hipMemcpy(x, h.data(), M * K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, h.data(), M * N * 4, hipMemcpyHostToDevice));
    const int nkb = (int)gt_cdiv(K, W3D_T), nnb = (int)gt_cdiv(N, W3D_T);
    const int s3 = w3_dw_splits(M, nkb * nnb);
    CK(hipMalloc(&part, (size_t)s3 * (N * K + N) * 4));
    L32DwArgs d{}; d.dy = dy; d.x = x; d.M = M; d.N = N; d.K = K; d.ldy = N; d.ldx = K; d.inv_keep = 1.f;
    d.part = part; d.dbpart = part + (size_t)s3 * N * K; d.splits = s3; d.nkb = nkb; d.nnb = nnb; d.m_per_split = gt_cdiv(gt_cdiv(M, s3), 32) * 32;
    dim3 grid((unsigned)(gt_cdiv(s3, 8) * 8 * nkb * nnb));
    for (int i = 0; i < 5; ++i) w3r_launch_dw(grid, 0, d);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    const int R = 50;
    for (int i = 0; i < R; ++i) w3r_launch_dw(grid, 0, d);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / R;
    printf("abl %2d  %6lld x %4lld x %4lld : %7.1f us  (%6.1f TF fp32-equivalent, %.3f of the bf16x6 ceiling; %d splits, %lld stages per block)\n", W3RD_ABL, (long long)M, (long long)N,
           (long long)K, us, 2.0 * M * N * K / us / 1e6, 2.0 * M * N * K / us / 1e6 / 416.7, s3, (long long)(d.m_per_split / 32));
    CK(hipFree(x)); CK(hipFree(dy)); CK(hipFree(part));
  }
  return 0;
}
