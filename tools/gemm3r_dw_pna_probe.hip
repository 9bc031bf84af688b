// tools/gemm3r_dw_pna_probe.hip — k_lin3r_dw on the Code2-PNA post stack (4 towers, 15 945 x 204 x 340) with compile-time ablations, in
// three layouts: the fused layer's (column slices of [N][816] / [N][1360], blockIdx.y = tower), one tower dense, one tower on slices.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DW3RD_ABL=<mask> -I graphtrans_amd/csrc -I include -o tools/gemm3_probe_dwpna<mask> tools/gemm3r_dw_pna_probe.hip
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
  const int64_t M = 15945, N = 204, K = 340, T = 4;
  float *x, *dy, *part;
  CK(hipMalloc(&x, M * K * T * 4)); CK(hipMalloc(&dy, M * N * T * 4));
  std::vector<float> h(M * K * T); for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
  CK(hipMemcpy(x, h.data(), M * K * T * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, h.data(), M * N * T * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&part, (size_t)64 * T * (N * K + N) * 4));
  struct Cfg { const char* name; int groups; int64_t ldy, ldx; int shape; int splits; };
  const Cfg cfgs[] = {{"4 towers on slices, 224x128", 4, N * T, K * T, 1, 0}, {"4 towers on slices, 160x160", 4, N * T, K * T, 0, 0},
                      {"1 tower dense, 224x128", 1, N, K, 1, 21}, {"1 tower on slices, 224x128", 1, N * T, K * T, 1, 21},
                      {"4 towers dense-per-tower pitch, 224x128", 4, N, K, 1, 0}};
  for (const Cfg& c : cfgs) {
    const int nkb = (int)gt_cdiv(K, w3r_dw_xt(c.shape)), nnb = (int)gt_cdiv(N, w3r_dw_zt(c.shape));
    const int s3 = c.splits ? c.splits : w3_dw_splits(M, nkb * nnb * c.groups);
    L32DwArgs d{}; d.dy = dy; d.x = x; d.M = M; d.N = N; d.K = K; d.ldy = c.ldy; d.ldx = c.ldx; d.inv_keep = 1.f;
    d.groups = c.groups; d.g_y = c.ldy == N ? M * N : N; d.g_x = c.ldx == K ? M * K : K; d.g_part = (int64_t)s3 * (N * K + N);
    d.part = part; d.dbpart = part + (size_t)s3 * N * K; d.splits = s3; d.nkb = nkb; d.nnb = nnb; d.m_per_split = gt_cdiv(gt_cdiv(M, s3), 32) * 32;
    dim3 grid((unsigned)(gt_cdiv(s3, 8) * 8 * nkb * nnb), (unsigned)c.groups);
    for (int i = 0; i < 5; ++i) w3r_launch_dw(grid, 0, d, c.shape);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    const int R = 50;
    for (int i = 0; i < R; ++i) w3r_launch_dw(grid, 0, d, c.shape);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / R;
    printf("abl %2d  %-42s: %7.1f us  (%d splits, %lld stages per block = %.2f us per stage; %d blocks)\n", W3RD_ABL, c.name, us, s3, (long long)(d.m_per_split / 32),
           us / (d.m_per_split / 32), s3 * nkb * nnb * c.groups);
  }
  return 0;
}
