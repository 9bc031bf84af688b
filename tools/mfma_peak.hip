// f32-input MFMA issue-rate probe: waves of pure v_mfma_f32_16x16x4_f32 / 32x32x2 chains (no memory traffic).
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC>
__global__ void __launch_bounds__(256) k16(float* out, int iters, float a, float b) {
  f32x4 acc[NACC];
  for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ void __launch_bounds__(256) k32(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][5];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
float run(F launch, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  float* out;
  hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 4000;
  for (int blocks : {256, 512, 1024, 2048}) {
    float ms = run([&] { hipLaunchKernelGGL(k16<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 2.0f); }, 5);
    double fl = (double)blocks * 4 * iters * 8 * 2048.0;
    printf("16x16x4  acc 8  blocks %4d: %8.3f ms  %7.1f TF\n", blocks, ms, fl / ms / 1e9);
    ms = run([&] { hipLaunchKernelGGL(k16<19>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 2.0f); }, 5);
    fl = (double)blocks * 4 * iters * 19 * 2048.0;
    printf("16x16x4  acc 19 blocks %4d: %8.3f ms  %7.1f TF\n", blocks, ms, fl / ms / 1e9);
    ms = run([&] { hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 2.0f); }, 5);
    fl = (double)blocks * 4 * iters * 4 * 4096.0;
    printf("32x32x2  acc 4  blocks %4d: %8.3f ms  %7.1f TF\n", blocks, ms, fl / ms / 1e9);
  }
  return 0;
}
