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
                                              float inv_sqrt_c2, const float* __restrict__ grad_scale) {
  const int64_t c = chunk_begin + blockIdx.x;
  const int t = chunk_tensor[c];
  const float* g = gp.g[t - tensor_begin];
  if (!g) return;  // no gradient this step: torch skips the parameter
  const gt_adamw_tensor T = table[t];
  const int64_t base = (int64_t)chunk_local[c] * CHUNK;
  const float decay = 1.0f - lr * wd, step = lr * inv_c1;
  const float gs = grad_scale ? *grad_scale : 1.0f;  // gradient clipping coefficient (gt_grad_clip_coef)
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
        pg[e] *= gs;
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
        const float gj = g[j] * gs;
        const float m = b1 * T.exp_avg[j] + (1.0f - b1) * gj;
        const float v = b2 * T.exp_avg_sq[j] + (1.0f - b2) * gj * gj;
        p -= step * m / (sqrtf(v) * inv_sqrt_c2 + eps);
        T.param[j] = p; T.exp_avg[j] = m; T.exp_avg_sq[j] = v;
      }
    }
  }
}

// ---- global gradient norm for clipping (torch.nn.utils.clip_grad_norm_, trainers/base_trainer.py:34-35) ----
// partial[c] = sum of squares of chunk c (same chunk map as k_adamw); fixed-order sums -> deterministic
__global__ void __launch_bounds__(OT) k_grad_sqnorm(const gt_adamw_tensor* __restrict__ table, const int32_t* __restrict__ chunk_tensor,
                                                    const int32_t* __restrict__ chunk_local, int64_t chunk_begin, int tensor_begin,
                                                    GradPtrs gp, float* __restrict__ partial) {
  __shared__ float red[OT / GT_WAVE];
  const int64_t c = chunk_begin + blockIdx.x;
  const int t = chunk_tensor[c];
  const float* g = gp.g[t - tensor_begin];
  float s = 0.f;
  if (g) {
    const int64_t numel = table[t].numel, base = (int64_t)chunk_local[c] * CHUNK;
    for (int64_t i = base + threadIdx.x; i < base + CHUNK && i < numel; i += OT) s += g[i] * g[i];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[c] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out2 = { total_norm, clip_coef = min(1, max_norm / (total_norm + 1e-6)) }
__global__ void __launch_bounds__(OT) k_grad_clip_coef(const float* __restrict__ partial, int64_t n, float max_norm,
                                                       float* __restrict__ out2) {
  __shared__ float red[OT / GT_WAVE];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += OT) s += partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    out2[0] = norm;
    out2[1] = fminf(1.0f, max_norm / (norm + 1e-6f));
  }
}

}  // namespace

extern "C" int gt_adamw_chunk_elems(void) { return CHUNK; }

extern "C" int gt_grad_sqnorm(const gt_adamw_tensor* table, const int32_t* chunk_tensor, const int32_t* chunk_local,
                              int64_t chunk_begin, int64_t num_chunks, int tensor_begin, int num_tensors,
                              const float* const* grads_host, float* partial, gt_stream_t stream_) {
  GT_CHECK_ARG(table && chunk_tensor && chunk_local && grads_host && partial, "null buffer");
  GT_CHECK_ARG(num_tensors >= 0 && num_tensors <= GT_ADAMW_MAX_TENSORS, "at most GT_ADAMW_MAX_TENSORS tensors per call");
  if (num_chunks <= 0 || num_tensors == 0) return GT_OK;
  GradPtrs gp{};
  for (int t = 0; t < num_tensors; ++t) gp.g[t] = grads_host[t];
  hipLaunchKernelGGL(k_grad_sqnorm, dim3((unsigned)num_chunks), dim3(OT), 0, (hipStream_t)stream_, table, chunk_tensor,
                     chunk_local, chunk_begin, tensor_begin, gp, partial);
  GT_CHECK_LAUNCH();
  return GT_OK;
}

extern "C" int gt_grad_clip_coef(const float* partial, int64_t num_partials, float max_norm, float* out2, gt_stream_t stream_) {
  GT_CHECK_ARG(partial && out2 && num_partials > 0 && max_norm > 0.f, "bad arguments");
  hipLaunchKernelGGL(k_grad_clip_coef, dim3(1), dim3(OT), 0, (hipStream_t)stream_, partial, num_partials, max_norm, out2);
  GT_CHECK_LAUNCH();
  return GT_OK;
}

extern "C" int gt_adamw_step(const gt_adamw_tensor* table, const int32_t* chunk_tensor, const int32_t* chunk_local,
                             int64_t chunk_begin, int64_t num_chunks, int tensor_begin, int num_tensors,
                             const float* const* grads_host, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int64_t step, const float* grad_scale, gt_stream_t stream_) {
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
                     (float)(1.0 / sqrt(c2)), grad_scale);
  GT_CHECK_LAUNCH();
  return GT_OK;
}
