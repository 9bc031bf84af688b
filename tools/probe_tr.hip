// probe_tr.hip — one-off hardware probe (run on the GPU box): prints what ds_read_b64_tr_b16
// returns for a known LDS image, to pin the lane/element mapping assumed by attention.hip.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 40];
  int l = threadIdx.x;
  for (int i = l; i < 64 * 40; i += 64) lds[i] = (short)i;  // element value = its index; row stride 40
  __syncthreads();
  int n = l & 15, g = l >> 4;
  const short* p = lds + (g * 8 + (n >> 2)) * 40 + (n & 3) * 4;
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  for (int i = 0; i < 4; ++i) out[l * 4 + i] = a[i];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
    int n = l & 15, g = l >> 4; int want = (g * 8 + i) * 40 + n;  // X[k = g*8+i][col = n]
    if (h[l * 4 + i] != want) ++bad;
  }
  printf("tr_probe mismatches: %d\n", bad);
  for (int l = 0; l < 64; l += 5) printf("lane %2d: %d %d %d %d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  return 0;
}
