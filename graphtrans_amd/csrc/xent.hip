// xent.hip — softmax cross-entropy over the stacked prediction heads, forward and backward.
//
// Reference (paths under /root/reference): the Code2 training loss, dataset/code.py:39-45:
//   loss = (1 / max_seq_len) * sum_l CrossEntropyLoss()(pred_list[l], y_arr[:, l])
// with pred_list[l] = graph_pred_linear_list[l](h_graph) (models/gnn_transformer.py:124-126); torch
// runs log_softmax + nll_loss (+ their backwards) per head.  Here the L heads are one GEMM into a
// [B][ld] buffer (head l at columns l*C .. l*C+C), and this file reduces it to the scalar loss:
//   k_xent_row   : one block per (b, l) row: lse = logsumexp(x), row_loss = lse - x[target]
//   k_xent_mean  : per-head mean over the non-ignored rows (ignore_index = -100, torch's default),
//                  then the mean over heads; fixed summation order
//   k_xent_bwd   : dlogits = (exp(x - lse) - onehot) * grad / (L * count_l), pad columns zeroed
// HBM-bound: the forward reads B*L*C floats once (the second sweep of a 20 KB row hits L2), the
// backward reads and writes them once.
#include "gt_common.h"

namespace {

constexpr int XT = 256;
constexpr int64_t IGNORE = -100;

__device__ __forceinline__ float block_max(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  v = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  return v;
}
__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  v = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  return v;
}

__global__ void __launch_bounds__(XT) k_xent_row(const float* __restrict__ logits, int64_t L, int64_t C, int64_t ld,
                                                 const int64_t* __restrict__ target, int64_t tstride,
                                                 float* __restrict__ lse, float* __restrict__ row_loss) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x, b = row / L, l = row % L;
  const float* x = logits + b * ld + l * C;
  float m = -INFINITY;
  for (int64_t c = threadIdx.x; c < C; c += XT) m = fmaxf(m, x[c]);
  m = block_max(m, red);
  float s = 0.f;
  for (int64_t c = threadIdx.x; c < C; c += XT) s += expf(x[c] - m);
  s = block_sum(s, red);
  if (threadIdx.x == 0) {
    const float ls = m + logf(s);
    lse[row] = ls;
    const int64_t t = target[b * tstride + l];
    row_loss[row] = (t == IGNORE || t < 0 || t >= C) ? 0.f : ls - x[t];
  }
}

// head_scale[l] = 1 / (L * count_l) (0 when every row of the head is ignored); loss = sum_l mean_l / L
__global__ void __launch_bounds__(XT) k_xent_mean(const float* __restrict__ row_loss, const int64_t* __restrict__ target,
                                                  int64_t tstride, int64_t B, int64_t L, int64_t C,
                                                  float* __restrict__ head_scale, float* __restrict__ loss) {
  __shared__ float red[4];
  float total = 0.f, bad = 0.f;
  for (int64_t l = 0; l < L; ++l) {
    float s = 0.f, n = 0.f;
    for (int64_t b = threadIdx.x; b < B; b += XT) {
      const int64_t t = target[b * tstride + l];
      if (t != IGNORE && t >= 0 && t < C) { s += row_loss[b * L + l]; n += 1.f; }
      else if (t != IGNORE) bad += 1.f;   // out of range and not ignore_index: torch raises a device-side assert
    }
    s = block_sum(s, red);
    n = block_sum(n, red);
    const float sc = n > 0.f ? 1.0f / ((float)L * n) : 0.f;
    if (threadIdx.x == 0) head_scale[l] = sc;
    total += n > 0.f ? s * sc : nanf("");  // torch: mean over zero rows is NaN
  }
  bad = block_sum(bad, red);
  if (threadIdx.x == 0) {
    loss[0] = total;
    loss[1] = bad;   // status word: number of out-of-range targets (treated as ignored above); the host raises on it when validating
  }
}

__global__ void __launch_bounds__(XT) k_xent_bwd(const float* __restrict__ logits, const float* __restrict__ lse,
                                                 const int64_t* __restrict__ target, int64_t tstride,
                                                 const float* __restrict__ head_scale, const float* __restrict__ grad_loss,
                                                 int64_t L, int64_t C, int64_t ld, float* __restrict__ dlogits) {
  const int64_t row = blockIdx.x, b = row / L, l = row % L;
  const float* x = logits + b * ld + l * C;
  float* d = dlogits + b * ld + l * C;
  const int64_t t = target[b * tstride + l];
  const bool live = t != IGNORE && t >= 0 && t < C;
  const float sc = live ? head_scale[l] * *grad_loss : 0.f;
  const float ls = lse[row];
  for (int64_t c = threadIdx.x; c < C; c += XT) {
    const float p = expf(x[c] - ls);
    d[c] = live ? (p - (c == t ? 1.f : 0.f)) * sc : 0.f;
  }
  if (l == L - 1)
    for (int64_t c = L * C + threadIdx.x; c < ld; c += XT) dlogits[b * ld + c] = 0.f;
}

// ---- masked binary cross-entropy with logits (the Molpcba loss, dataset/mol.py:24-31) ---------------------
//   is_labeled = y == y;  loss = BCEWithLogitsLoss()(pred[is_labeled], y[is_labeled])   (mean over labelled entries)
//   k_bce_row : per graph row: sum and count of the labelled entries
//   k_bce_mean: fixed-order totals; loss = S / den with den = the local count, or a caller-supplied
//               denominator (data parallel: the global count / world size keeps the averaged gradient exact)
//   k_bce_bwd : dlogits = labelled ? (sigmoid(x) - y) * grad / den : 0
__device__ __forceinline__ float bce_logits(float x, float y) {
  return fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
}

__global__ void __launch_bounds__(XT) k_bce_row(const float* __restrict__ logits, const float* __restrict__ target,
                                                int64_t T, int64_t ld, int64_t tld, float* __restrict__ row_sum,
                                                float* __restrict__ row_cnt) {
  __shared__ float red[4];
  const int64_t b = blockIdx.x;
  float s = 0.f, n = 0.f;
  for (int64_t t = threadIdx.x; t < T; t += XT) {
    const float y = target[b * tld + t];
    if (y == y) { s += bce_logits(logits[b * ld + t], y); n += 1.f; }
  }
  s = block_sum(s, red);
  n = block_sum(n, red);
  if (threadIdx.x == 0) { row_sum[b] = s; row_cnt[b] = n; }
}

__global__ void __launch_bounds__(XT) k_bce_mean(const float* __restrict__ row_sum, const float* __restrict__ row_cnt,
                                                 int64_t B, const float* __restrict__ den_in, float* __restrict__ out2) {
  __shared__ float red[4];
  float s = 0.f, n = 0.f;
  for (int64_t b = threadIdx.x; b < B; b += XT) { s += row_sum[b]; n += row_cnt[b]; }
  s = block_sum(s, red);
  n = block_sum(n, red);
  if (threadIdx.x == 0) {
    const float den = den_in ? *den_in : n;
    out2[0] = s / den;  // 0 / 0 = NaN like torch's mean over an empty selection
    out2[1] = den;
  }
}

__global__ void __launch_bounds__(XT) k_bce_bwd(const float* __restrict__ logits, const float* __restrict__ target,
                                                const float* __restrict__ out2, const float* __restrict__ grad_loss,
                                                int64_t T, int64_t ld, int64_t tld, float* __restrict__ dlogits) {
  const int64_t b = blockIdx.x;
  const float sc = *grad_loss / out2[1];
  for (int64_t t = threadIdx.x; t < ld; t += XT) {
    float d = 0.f;
    if (t < T) {
      const float y = target[b * tld + t];
      if (y == y) d = (1.f / (1.f + expf(-logits[b * ld + t])) - y) * sc;
    }
    dlogits[b * ld + t] = d;
  }
}

}  // namespace

extern "C" int gt_bce_masked_fwd(const float* logits, const float* target, int64_t B, int64_t T, int64_t ld,
                                 int64_t target_ld, const float* den_in, float* row_sum, float* row_cnt, float* out2,
                                 gt_stream_t stream_) {
  GT_CHECK_ARG(B > 0 && T > 0 && ld >= T && target_ld >= T, "bad sizes");
  GT_CHECK_ARG(logits && target && row_sum && row_cnt && out2, "null buffer");
  hipStream_t stream = (hipStream_t)stream_;
  GtProfScope prof__(GT_PROF_NORM, "gt_bce_masked_fwd", stream_, {B, T});
  hipLaunchKernelGGL(k_bce_row, dim3((unsigned)B), dim3(XT), 0, stream, logits, target, T, ld, target_ld, row_sum, row_cnt);
  hipLaunchKernelGGL(k_bce_mean, dim3(1), dim3(XT), 0, stream, (const float*)row_sum, (const float*)row_cnt, B, den_in, out2);
  GT_CHECK_LAUNCH();
  return GT_OK;
}

extern "C" int gt_bce_masked_bwd(const float* logits, const float* target, const float* out2, const float* grad_loss,
                                 int64_t B, int64_t T, int64_t ld, int64_t target_ld, float* dlogits, gt_stream_t stream_) {
  GT_CHECK_ARG(B > 0 && T > 0 && ld >= T && target_ld >= T, "bad sizes");
  GT_CHECK_ARG(logits && target && out2 && grad_loss && dlogits, "null buffer");
  GtProfScope prof__(GT_PROF_NORM, "gt_bce_masked_bwd", stream_, {B, T});
  hipLaunchKernelGGL(k_bce_bwd, dim3((unsigned)B), dim3(XT), 0, (hipStream_t)stream_, logits, target, out2, grad_loss, T, ld,
                     target_ld, dlogits);
  GT_CHECK_LAUNCH();
  return GT_OK;
}

extern "C" int gt_xent_fwd(const float* logits, int64_t B, int64_t L, int64_t C, int64_t ld, const int64_t* target,
                           int64_t target_stride, float* lse, float* row_loss, float* head_scale, float* loss,
                           gt_stream_t stream_) {
  GT_CHECK_ARG(B > 0 && L > 0 && C > 0 && ld >= L * C, "bad sizes");
  GT_CHECK_ARG(logits && target && lse && row_loss && head_scale && loss, "null buffer");
  GT_CHECK_ARG(target_stride >= L, "target_stride < L");
  hipStream_t stream = (hipStream_t)stream_;
  GtProfScope prof__(GT_PROF_NORM, "gt_xent_fwd", stream_, {B, L, C});
  hipLaunchKernelGGL(k_xent_row, dim3((unsigned)(B * L)), dim3(XT), 0, stream, logits, L, C, ld, target, target_stride, lse,
                     row_loss);
  hipLaunchKernelGGL(k_xent_mean, dim3(1), dim3(XT), 0, stream, (const float*)row_loss, target, target_stride, B, L, C,
                     head_scale, loss);
  GT_CHECK_LAUNCH();
  return GT_OK;
}

extern "C" int gt_xent_bwd(const float* logits, const float* lse, const float* head_scale, const int64_t* target,
                           int64_t target_stride, const float* grad_loss, int64_t B, int64_t L, int64_t C, int64_t ld,
                           float* dlogits, gt_stream_t stream_) {
  GT_CHECK_ARG(B > 0 && L > 0 && C > 0 && ld >= L * C, "bad sizes");
  GT_CHECK_ARG(logits && lse && head_scale && target && grad_loss && dlogits, "null buffer");
  GtProfScope prof__(GT_PROF_NORM, "gt_xent_bwd", stream_, {B, L, C});
  hipLaunchKernelGGL(k_xent_bwd, dim3((unsigned)(B * L)), dim3(XT), 0, (hipStream_t)stream_, logits, lse, target,
                     target_stride, head_scale, grad_loss, L, C, ld, dlogits);
  GT_CHECK_LAUNCH();
  return GT_OK;
}
