// optim.hip — AdamW over every parameter tensor in ONE launch.
//
// Reference (paths under /root/reference): optim.AdamW(model.parameters(), lr, weight_decay) at
// main.py:178, stepped once per batch in trainers/base_trainer.py:36.  torch's fused implementation
// groups ~150 tensors into 6 multi_tensor_apply launches and spends ~0.8 ms of host time per step on
// grouping and bookkeeping (tools/host_phases.py); here a device-resident table built once describes
// every tensor, a static chunk map assigns 2048-element chunks to blocks, and only the gradient
// pointers (fresh allocations every step) travel as kernel arguments.
//
//   p *= 1 - lr * wd;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// (torch/optim/adamw.py, _single_tensor_adam with decoupled weight decay, amsgrad = False).
// HBM-bound: reads p, g, m, v and writes p, m, v once: 28 bytes per parameter.
#include "gt_common.h"

namespace {

constexpr int OT = 256;
constexpr int CHUNK = 2048;  // elements per block: 2 float4 per thread

struct GradPtrs {
  const float* g[GT_ADAMW_MAX_TENSORS];
};

__global__ void __launch_bounds__(OT) k_adamw(const gt_adamw_tensor* __restrict__ table, const int32_t* __restrict__ chunk_tensor,
                                              const int32_t* __restrict__ chunk_local, int64_t chunk_begin, int tensor_begin,
                                              GradPtrs gp, float lr, float b1, float b2, float eps, float wd, float inv_c1,
                                              float inv_sqrt_c2) {
  const int64_t c = chunk_begin + blockIdx.x;
  const int t = chunk_tensor[c];
  const float* g = gp.g[t - tensor_begin];
  if (!g) return;  // no gradient this step: torch skips the parameter
  const gt_adamw_tensor T = table[t];
  const int64_t base = (int64_t)chunk_local[c] * CHUNK;
  const float decay = 1.0f - lr * wd, step = lr * inv_c1;
  const bool vec = ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(T.param) |
                     reinterpret_cast<uintptr_t>(T.exp_avg) | reinterpret_cast<uintptr_t>(T.exp_avg_sq)) & 15) == 0;
#pragma unroll
  for (int u = 0; u < CHUNK / (OT * 4); ++u) {
    const int64_t i = base + ((int64_t)u * OT + threadIdx.x) * 4;
    if (i >= T.numel) break;
    if (vec && i + 4 <= T.numel) {
      float4 p = *reinterpret_cast<const float4*>(T.param + i), gg = *reinterpret_cast<const float4*>(g + i);
      float4 m = *reinterpret_cast<const float4*>(T.exp_avg + i), v = *reinterpret_cast<const float4*>(T.exp_avg_sq + i);
      float* pp = reinterpret_cast<float*>(&p); float* pg = reinterpret_cast<float*>(&gg);
      float* pm = reinterpret_cast<float*>(&m); float* pv = reinterpret_cast<float*>(&v);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pp[e] *= decay;
        pm[e] = b1 * pm[e] + (1.0f - b1) * pg[e];
        pv[e] = b2 * pv[e] + (1.0f - b2) * pg[e] * pg[e];
        pp[e] -= step * pm[e] / (sqrtf(pv[e]) * inv_sqrt_c2 + eps);
      }
      *reinterpret_cast<float4*>(T.param + i) = p;
      *reinterpret_cast<float4*>(T.exp_avg + i) = m;
      *reinterpret_cast<float4*>(T.exp_avg_sq + i) = v;
    } else {
      for (int64_t j = i; j < i + 4 && j < T.numel; ++j) {
        float p = T.param[j] * decay;
        const float gj = g[j];
        const float m = b1 * T.exp_avg[j] + (1.0f - b1) * gj;
        const float v = b2 * T.exp_avg_sq[j] + (1.0f - b2) * gj * gj;
        p -= step * m / (sqrtf(v) * inv_sqrt_c2 + eps);
        T.param[j] = p; T.exp_avg[j] = m; T.exp_avg_sq[j] = v;
      }
    }
  }
}

}  // namespace

extern "C" int gt_adamw_chunk_elems(void) { return CHUNK; }

extern "C" int gt_adamw_step(const gt_adamw_tensor* table, const int32_t* chunk_tensor, const int32_t* chunk_local,
                             int64_t chunk_begin, int64_t num_chunks, int tensor_begin, int num_tensors,
                             const float* const* grads_host, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int64_t step, gt_stream_t stream_) {
  GT_CHECK_ARG(table && chunk_tensor && chunk_local && grads_host, "null buffer");
  GT_CHECK_ARG(num_tensors >= 0 && num_tensors <= GT_ADAMW_MAX_TENSORS, "at most GT_ADAMW_MAX_TENSORS tensors per call");
  GT_CHECK_ARG(step >= 1, "step counts from 1");
  GT_CHECK_ARG(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "betas must be in [0, 1)");
  if (num_chunks <= 0 || num_tensors == 0) return GT_OK;
  GradPtrs gp{};
  for (int t = 0; t < num_tensors; ++t) gp.g[t] = grads_host[t];
  const double c1 = 1.0 - pow((double)beta1, (double)step), c2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(k_adamw, dim3((unsigned)num_chunks), dim3(OT), 0, (hipStream_t)stream_, table, chunk_tensor, chunk_local,
                     chunk_begin, tensor_begin, gp, lr, beta1, beta2, eps, weight_decay, (float)(1.0 / c1),
                     (float)(1.0 / sqrt(c2)));
  GT_CHECK_LAUNCH();
  return GT_OK;
}
