// norm.hip — BatchNorm1d (train/eval, optional fused ReLU) and residual + dropout + LayerNorm,
// forward and backward, for the node/token row matrices of the GraphTrans hot path.
//
// Reference call sites (paths under /root/reference):
//   modules/gnn_module.py:84,204  h = batch_norms[layer](h) ; :86-90,205-209  F.relu / dropout
//   modules/gnn_module.py:161-170 virtual-node MLP BatchNorm1d(2D) / BatchNorm1d(D) + ReLU
//   modules/conv.py:18-20         GIN mlp BatchNorm1d(2D) + ReLU
//   modules/transformer_encoder.py:28-32,56-59  nn.TransformerEncoderLayer: x = LN(x + dropout(sub(x)));
//                                 norm_input / final LayerNorm
// torch's BatchNorm kernels dominated the first profile (25 % of GPU time at N = 32 k rows x 300
// channels, profiles/r01a): column statistics here are one streaming pass (shifted sums: pivot =
// row 0 of every column, so sum / sum-of-squares do not cancel), deterministic block partials, a
// latency-hidden fixed-order finish, then one apply pass with ReLU fused.
#include <stdlib.h>

#include "gt_common.h"

namespace {

constexpr int NT = 256;
constexpr int MAX_PART = 256;  // column-partial blocks

// thread -> (chunk column, row lane) mapping for an [rows][D] matrix walked by a 256-thread block
struct ColMap {
  int C, CW, R, tc, tr;
  __device__ __forceinline__ ColMap(int64_t D) {
    C = (int)(D / 4);
    CW = C < NT ? C : NT;
    R = NT / CW;
    tc = threadIdx.x % CW;
    tr = threadIdx.x / CW;  // >= R for the idle tail threads
  }
};

// reduce `nv` float4 accumulators over the R row-lanes of a block through LDS and hand the result
// to row-lane 0 (fixed order)
template <int NV>
__device__ __forceinline__ void block_rowlane_reduce(float4 (&v)[NV], const ColMap& m, float4* sm) {
  // sm: [R][CW][NV]
  if (m.tr < m.R) {
#pragma unroll
    for (int i = 0; i < NV; ++i) sm[(m.tr * m.CW + m.tc) * NV + i] = v[i];
  }
  __syncthreads();
  if (m.tr == 0) {
    for (int r = 1; r < m.R; ++r)
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] = gt_add4(v[i], sm[(r * m.CW + m.tc) * NV + i]);
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// BatchNorm
// ------------------------------------------------------------------------------------------------
// pass 1: per-block shifted sums  part[blk][0][D] = sum(x - pivot), part[blk][1][D] = sum((x-pivot)^2)
template <typename T>
__global__ void __launch_bounds__(NT) k_bn_stats_partial(const T* __restrict__ x, int64_t N, int64_t D,
                                                         float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float4 sm4[];
  ColMap m(D);
  const int64_t rows_per = (N + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per;
  const int64_t r1 = r0 + rows_per < N ? r0 + rows_per : N;
  for (int c0 = 0; c0 < m.C; c0 += m.CW) {  // uniform trip count (block-wide barriers inside)
    const int c = c0 + m.tc;
    const bool cact = c < m.C;
    float4 acc[2] = {gt_zero4(), gt_zero4()};
    if (m.tr < m.R && cact) {
      const float4 piv = gt_load4<T>(x + (int64_t)c * 4);  // row 0
      auto add_row = [&](float4 v) {
        v = make_float4(v.x - piv.x, v.y - piv.y, v.z - piv.z, v.w - piv.w);
        acc[0] = gt_add4(acc[0], v);
        acc[1] = make_float4(fmaf(v.x, v.x, acc[1].x), fmaf(v.y, v.y, acc[1].y), fmaf(v.z, v.z, acc[1].z),
                             fmaf(v.w, v.w, acc[1].w));
      };
      // 8 independent row loads in flight per thread (one load per trip is a chain of memory round trips)
      int64_t r = r0 + m.tr;
      for (; r + 7 * m.R < r1; r += 8 * m.R) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = gt_load4<T>(x + (r + (int64_t)u * m.R) * D + (int64_t)c * 4);
#pragma unroll
        for (int u = 0; u < 8; ++u) add_row(v[u]);
      }
      for (; r < r1; r += m.R) add_row(gt_load4<T>(x + r * D + (int64_t)c * 4));
    }
    block_rowlane_reduce<2>(acc, m, sm4);
    if (m.tr == 0 && cact) {
      float* p = part + (int64_t)blockIdx.x * 2 * D + (int64_t)c * 4;
      *reinterpret_cast<float4*>(p) = acc[0];
      *reinterpret_cast<float4*>(p + D) = acc[1];
    }
  }
}

// Fixed-order column sums over `nblk` partial rows.  Block = 32 columns x 8 partial-lanes: lane p
// sums partial rows p, p+8, ... (8 independent loads in flight), then the 8 lanes are combined in a
// fixed order through LDS.  Returns the total to the threads with p == 0 (others get garbage).
constexpr int FIN_COLS = 32, FIN_LANES = 8;
// the finish kernels sum up to 512 partial rows per column: 32 lanes per column (1024-thread blocks) keep the
// dependent-load chain at 2 batches of 8 -- with 8 lanes these 4..10-block kernels took 7-10 us each
constexpr int FINK_LANES = 32;
template <int LANES>
__device__ __forceinline__ float finish_sum(const float* __restrict__ part, int nblk, int64_t stride, int64_t off,
                                            bool active, float* sm /* [LANES][FIN_COLS] */) {
  constexpr int FIN_LANES = LANES;
  const int cl = threadIdx.x % FIN_COLS, p = threadIdx.x / FIN_COLS;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (active) {
    int b = p;
    for (; b + 7 * FIN_LANES < nblk; b += 8 * FIN_LANES) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += part[(int64_t)(b + u * FIN_LANES) * stride + off];
    }
    for (; b < nblk; b += FIN_LANES) a[0] += part[(int64_t)b * stride + off];
  }
  sm[p * FIN_COLS + cl] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  float t = 0.f;
  if (p == 0) {
#pragma unroll
    for (int q = 0; q < FIN_LANES; ++q) t += sm[q * FIN_COLS + cl];
  }
  __syncthreads();
  return t;
}

// pass 2: mean / rstd per column (+ running-stat update, torch semantics: unbiased running_var)
template <typename T>
__global__ void k_bn_stats_finish(const T* __restrict__ x, const float* __restrict__ part, int nblk, int64_t N,
                                  int64_t D, float eps, float momentum, float* __restrict__ mean,
                                  float* __restrict__ rstd, float* __restrict__ running_mean,
                                  float* __restrict__ running_var, int64_t* __restrict__ num_batches_tracked) {
  __shared__ float sm[FINK_LANES * FIN_COLS];
  const int64_t c = (int64_t)blockIdx.x * FIN_COLS + threadIdx.x % FIN_COLS;
  if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches_tracked) num_batches_tracked[0] += 1;
  const float s1 = finish_sum<FINK_LANES>(part, nblk, 2 * D, c, c < D, sm);
  const float s2 = finish_sum<FINK_LANES>(part, nblk, 2 * D, D + c, c < D, sm);
  if (c >= D || threadIdx.x >= FIN_COLS) return;
  const float piv = sizeof(T) == 4 ? (float)reinterpret_cast<const float*>(x)[c]
                                   : gt_bf16_to_f32(reinterpret_cast<const gt_bf16*>(x)[c]);
  const float inv_n = 1.0f / (float)N;
  const float m1 = s1 * inv_n;
  float var = s2 * inv_n - m1 * m1;
  var = var < 0.f ? 0.f : var;
  const float mu = piv + m1;
  mean[c] = mu;
  rstd[c] = 1.0f / sqrtf(var + eps);
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    const float unbiased = N > 1 ? var * ((float)N / (float)(N - 1)) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  }
}

// eval mode: mean = running_mean, rstd = (running_var + eps)^-1/2
__global__ void k_bn_eval_stats(const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                int64_t D, float eps, float* __restrict__ mean, float* __restrict__ rstd) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= D) return;
  mean[c] = running_mean[c];
  rstd[c] = 1.0f / sqrtf(running_var[c] + eps);
}

// dropout fused behind BatchNorm(+ReLU): F.dropout(h, drop_ratio) of the GNN layers
// (modules/gnn_module.py:88-90,209-212,222).  Counter hash of (seed, row, column), replayed by the backward.
struct BnDrop {
  uint32_t thr, s0, s1;  // keep iff hash >= thr ; thr == 0: no dropout
  float inv_keep;
};
__device__ __forceinline__ uint32_t bn_hash(uint32_t s0, uint32_t s1, uint32_t row, uint32_t col) {
  uint32_t x = (row * 0x9E3779B1u + s0) ^ (col * 0x85EBCA77u + s1);
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float4 bn_drop4(float4 v, const BnDrop& d, uint32_t row, uint32_t col) {
  return make_float4(bn_hash(d.s0, d.s1, row, col) >= d.thr ? v.x * d.inv_keep : 0.f,
                     bn_hash(d.s0, d.s1, row, col + 1) >= d.thr ? v.y * d.inv_keep : 0.f,
                     bn_hash(d.s0, d.s1, row, col + 2) >= d.thr ? v.z * d.inv_keep : 0.f,
                     bn_hash(d.s0, d.s1, row, col + 3) >= d.thr ? v.w * d.inv_keep : 0.f);
}

// pass 3: y = drop((x - mean) * rstd * w + b  [relu]) [+ resid] [+ bcast[bidx[row]]]
// (the last term is the NEXT layer's virtual-node add h_list[l+1] + vn[batch], gnn_module.py:199, folded into the
// producer of h_list[l+1]: no separate N x D read-modify-write pass)
template <typename T>
__global__ void __launch_bounds__(NT) k_bn_apply(const T* __restrict__ x, const float* __restrict__ mean,
                                                 const float* __restrict__ rstd, const float* __restrict__ w,
                                                 const float* __restrict__ b, const T* __restrict__ resid, int relu,
                                                 BnDrop drop, int64_t N, int64_t D, T* __restrict__ y,
                                                 const T* __restrict__ bcast, const int32_t* __restrict__ bidx) {
  const int64_t C = D / 4, total = N * C;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    const int64_t c = (i % C) * 4;
    float4 v = gt_load4<T>(x + i * 4);
    const float4 mu = *reinterpret_cast<const float4*>(mean + c), rs = *reinterpret_cast<const float4*>(rstd + c);
    const float4 ww = *reinterpret_cast<const float4*>(w + c), bb = *reinterpret_cast<const float4*>(b + c);
    v = make_float4((v.x - mu.x) * rs.x * ww.x + bb.x, (v.y - mu.y) * rs.y * ww.y + bb.y,
                    (v.z - mu.z) * rs.z * ww.z + bb.z, (v.w - mu.w) * rs.w * ww.w + bb.w);
    if (relu) v = gt_relu4(v);
    if (drop.thr) v = bn_drop4(v, drop, (uint32_t)(i / C), (uint32_t)c);
    if (resid) v = gt_add4(v, gt_load4<T>(resid + i * 4));
    if (bcast) v = gt_add4(v, gt_load4<T>(bcast + (int64_t)bidx[i / C] * D + c));
    gt_store4<T>(y + i * 4, v);
  }
}

// relu gate recomputed from the BN input: 1[(x - mean) * rstd * w + b > 0]
__device__ __forceinline__ float4 bn_gate(float4 g, float4 v, float4 mu, float4 rs, float4 w, float4 b) {
  return make_float4((v.x - mu.x) * rs.x * w.x + b.x > 0.f ? g.x : 0.f, (v.y - mu.y) * rs.y * w.y + b.y > 0.f ? g.y : 0.f,
                     (v.z - mu.z) * rs.z * w.z + b.z > 0.f ? g.z : 0.f, (v.w - mu.w) * rs.w * w.w + b.w > 0.f ? g.w : 0.f);
}

// backward pass 1: part[blk][0][D] = sum(dy'), part[blk][1][D] = sum(dy' * xhat),  dy' = dy * 1[y > 0] if relu
template <typename T>
__global__ void __launch_bounds__(NT) k_bn_bwd_partial(const T* __restrict__ x, const T* __restrict__ dy,
                                                       const float* __restrict__ w, const float* __restrict__ b,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       int relu, BnDrop drop, int64_t N, int64_t D, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float4 sm4[];
  ColMap m(D);
  const int64_t rows_per = (N + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per;
  const int64_t r1 = r0 + rows_per < N ? r0 + rows_per : N;
  for (int c0 = 0; c0 < m.C; c0 += m.CW) {
    const int c = c0 + m.tc;
    const bool cact = c < m.C;
    float4 acc[2] = {gt_zero4(), gt_zero4()};
    if (m.tr < m.R && cact) {
      const float4 mu = *reinterpret_cast<const float4*>(mean + c * 4), rs = *reinterpret_cast<const float4*>(rstd + c * 4);
      const float4 ww = *reinterpret_cast<const float4*>(w + c * 4), bb = *reinterpret_cast<const float4*>(b + c * 4);
      auto add_row = [&](float4 g, const float4 v, int64_t r) {
        if (drop.thr) g = bn_drop4(g, drop, (uint32_t)r, (uint32_t)(c * 4));
        if (relu) g = bn_gate(g, v, mu, rs, ww, bb);
        acc[0] = gt_add4(acc[0], g);
        acc[1] = make_float4(fmaf(g.x, (v.x - mu.x) * rs.x, acc[1].x), fmaf(g.y, (v.y - mu.y) * rs.y, acc[1].y),
                             fmaf(g.z, (v.z - mu.z) * rs.z, acc[1].z), fmaf(g.w, (v.w - mu.w) * rs.w, acc[1].w));
      };
      int64_t r = r0 + m.tr;
      for (; r + 3 * m.R < r1; r += 4 * m.R) {  // 8 independent loads in flight per thread
        float4 g[4], v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int64_t o = (r + (int64_t)u * m.R) * D + (int64_t)c * 4;
          g[u] = gt_load4<T>(dy + o);
          v[u] = gt_load4<T>(x + o);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) add_row(g[u], v[u], r + (int64_t)u * m.R);
      }
      for (; r < r1; r += m.R) {
        const int64_t o = r * D + (int64_t)c * 4;
        add_row(gt_load4<T>(dy + o), gt_load4<T>(x + o), r);
      }
    }
    block_rowlane_reduce<2>(acc, m, sm4);
    if (m.tr == 0 && cact) {
      float* p = part + (int64_t)blockIdx.x * 2 * D + (int64_t)c * 4;
      *reinterpret_cast<float4*>(p) = acc[0];
      *reinterpret_cast<float4*>(p + D) = acc[1];
    }
  }
}

__global__ void k_bn_bwd_finish(const float* __restrict__ part, int nblk, int64_t D, float* __restrict__ dbias,
                                float* __restrict__ dweight) {
  __shared__ float sm[FINK_LANES * FIN_COLS];
  const int64_t c = (int64_t)blockIdx.x * FIN_COLS + threadIdx.x % FIN_COLS;
  const float s1 = finish_sum<FINK_LANES>(part, nblk, 2 * D, c, c < D, sm);
  const float s2 = finish_sum<FINK_LANES>(part, nblk, 2 * D, D + c, c < D, sm);
  if (c >= D || threadIdx.x >= FIN_COLS) return;
  dbias[c] = s1;
  dweight[c] = s2;
}

// backward pass 3: train: dx = w rstd (dy' - dbias/N - xhat dweight/N) ; eval: dx = w rstd dy'
template <typename T>
__global__ void __launch_bounds__(NT) k_bn_bwd_apply(const T* __restrict__ x, const T* __restrict__ dy,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ w, const float* __restrict__ b,
                                                     const float* __restrict__ dbias, const float* __restrict__ dweight,
                                                     int relu, float inv_n, BnDrop drop, int64_t N, int64_t D,
                                                     T* __restrict__ dx, const float* __restrict__ count_dev) {
  const int64_t C = D / 4, total = N * C;   // inv_n: 1 / (rows the statistics were taken over) in training, 0 in eval
  if (count_dev) inv_n = 1.0f / count_dev[0];   // synchronised statistics: the all-reduced row count of every rank
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    const int64_t c = (i % C) * 4;
    float4 g = gt_load4<T>(dy + i * 4);
    const float4 v = gt_load4<T>(x + i * 4);
    const float4 mu = *reinterpret_cast<const float4*>(mean + c), rs = *reinterpret_cast<const float4*>(rstd + c);
    const float4 ww = *reinterpret_cast<const float4*>(w + c);
    if (drop.thr) g = bn_drop4(g, drop, (uint32_t)(i / C), (uint32_t)c);
    if (relu) g = bn_gate(g, v, mu, rs, ww, *reinterpret_cast<const float4*>(b + c));
    const float4 db = *reinterpret_cast<const float4*>(dbias + c), dw = *reinterpret_cast<const float4*>(dweight + c);
    float4 r;
    r.x = ww.x * rs.x * (g.x - db.x * inv_n - (v.x - mu.x) * rs.x * dw.x * inv_n);
    r.y = ww.y * rs.y * (g.y - db.y * inv_n - (v.y - mu.y) * rs.y * dw.y * inv_n);
    r.z = ww.z * rs.z * (g.z - db.z * inv_n - (v.z - mu.z) * rs.z * dw.z * inv_n);
    r.w = ww.w * rs.w * (g.w - db.w * inv_n - (v.w - mu.w) * rs.w * dw.w * inv_n);
    gt_store4<T>(dx + i * 4, r);
  }
}


// ---- few rows (the virtual-node MLP normalises B = 256 graph rows, modules/gnn_module.py:161-170): ONE launch
// per direction instead of three.  A block owns 32 columns x all rows (8 row lanes); statistics and apply in
// the same kernel, the second sweep over the <= 1024 x 32 slab comes from L1 / L2.
constexpr int SMALL_ROWS = 1024;

template <typename T>
__device__ __forceinline__ float ld1(const T* p) {
  if constexpr (sizeof(T) == 4) return (float)*reinterpret_cast<const float*>(p);
  else return gt_bf16_to_f32(*reinterpret_cast<const gt_bf16*>(p));
}
template <typename T>
__device__ __forceinline__ void st1(T* p, float v) {
  if constexpr (sizeof(T) == 4) *reinterpret_cast<float*>(p) = v;
  else *reinterpret_cast<gt_bf16*>(p) = gt_f32_to_bf16(v);
}
constexpr int SM_COLS = 8;    // columns per block: 256-thread blocks find a slot on a busy chip at once (the virtual-node chain runs beside
                              // chip-filling kernels; 1024-thread blocks waited up to 70 us for a CU with 16 free wave slots)
constexpr int SM_LANES = 32;  // row lanes per column: 8 rows per thread at B = 256, 4 loads in flight
// sum over the SM_LANES row lanes of a column, valid in row lane 0
__device__ __forceinline__ float lanes_sum(float v, float* sm) {
  const int cl = threadIdx.x % SM_COLS, p = threadIdx.x / SM_COLS;
  sm[p * SM_COLS + cl] = v;
  __syncthreads();
  float t = 0.f;
  if (p == 0) {
#pragma unroll
    for (int q = 0; q < SM_LANES; ++q) t += sm[q * SM_COLS + cl];
  }
  __syncthreads();
  return t;
}

template <typename T>
__global__ void __launch_bounds__(SM_COLS * SM_LANES) k_bn_small_fwd(
    const T* __restrict__ x, int64_t N, int64_t D, float eps, float momentum, const float* __restrict__ w,
    const float* __restrict__ b, const T* __restrict__ resid, int relu, BnDrop drop, float* __restrict__ mean,
    float* __restrict__ rstd, float* __restrict__ running_mean, float* __restrict__ running_var,
    int64_t* __restrict__ num_batches_tracked, T* __restrict__ y, const T* __restrict__ bcast, const int32_t* __restrict__ bidx) {
  __shared__ float sm[SM_LANES * SM_COLS];
  __shared__ float s_mu[SM_COLS], s_rs[SM_COLS];
  const int cl = threadIdx.x % SM_COLS, p = threadIdx.x / SM_COLS;
  const int64_t c = (int64_t)blockIdx.x * SM_COLS + cl;
  const bool act = c < D;
  if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches_tracked) num_batches_tracked[0] += 1;
  const float piv = act ? ld1<T>(x + c) : 0.f;  // shifted sums: no cancellation for large means
  float a1[2] = {0.f, 0.f}, a2[2] = {0.f, 0.f};
  if (act) {
    int64_t r = p;
    for (; r + 3 * SM_LANES < N; r += 4 * SM_LANES) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = ld1<T>(x + (r + u * SM_LANES) * D + c) - piv;
#pragma unroll
      for (int u = 0; u < 4; ++u) { a1[u & 1] += v[u]; a2[u & 1] = fmaf(v[u], v[u], a2[u & 1]); }
    }
    for (; r < N; r += SM_LANES) {
      const float v0 = ld1<T>(x + r * D + c) - piv;
      a1[0] += v0; a2[0] = fmaf(v0, v0, a2[0]);
    }
  }
  const float s1 = lanes_sum(a1[0] + a1[1], sm), s2 = lanes_sum(a2[0] + a2[1], sm);
  if (p == 0 && act) {
    const float inv_n = 1.0f / (float)N;
    const float m1 = s1 * inv_n;
    float var = s2 * inv_n - m1 * m1;
    var = var < 0.f ? 0.f : var;
    const float mu = piv + m1, rs = 1.0f / sqrtf(var + eps);
    mean[c] = mu;
    rstd[c] = rs;
    s_mu[cl] = mu;
    s_rs[cl] = rs;
    if (running_mean) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
      const float unbiased = N > 1 ? var * ((float)N / (float)(N - 1)) : var;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
  }
  __syncthreads();
  if (!act) return;
  const float mu = s_mu[cl], rs = s_rs[cl], ww = w[c], bb = b[c];
  for (int64_t r = p; r < N; r += SM_LANES) {
    float v = (ld1<T>(x + r * D + c) - mu) * rs * ww + bb;
    if (relu) v = fmaxf(v, 0.f);
    if (drop.thr) v = bn_hash(drop.s0, drop.s1, (uint32_t)r, (uint32_t)c) >= drop.thr ? v * drop.inv_keep : 0.f;
    if (resid) v += ld1<T>(resid + r * D + c);
    if (bcast) v += ld1<T>(bcast + (int64_t)bidx[r] * D + c);
    st1<T>(y + r * D + c, v);
  }
}

template <typename T>
__global__ void __launch_bounds__(SM_COLS * SM_LANES) k_bn_small_bwd(
    const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ w, const float* __restrict__ b, int relu, int training, BnDrop drop, int64_t N, int64_t D,
    float* __restrict__ dbias, float* __restrict__ dweight, T* __restrict__ dx) {
  __shared__ float sm[SM_LANES * SM_COLS];
  __shared__ float s_db[SM_COLS], s_dw[SM_COLS];
  const int cl = threadIdx.x % SM_COLS, p = threadIdx.x / SM_COLS;
  const int64_t c = (int64_t)blockIdx.x * SM_COLS + cl;
  const bool act = c < D;
  const float mu = act ? mean[c] : 0.f, rs = act ? rstd[c] : 0.f, ww = act ? w[c] : 0.f, bb = act ? b[c] : 0.f;
  auto grad_at = [&](int64_t r, float& xh) {
    float g = ld1<T>(dy + r * D + c);
    const float v = ld1<T>(x + r * D + c);
    xh = (v - mu) * rs;
    if (drop.thr) g = bn_hash(drop.s0, drop.s1, (uint32_t)r, (uint32_t)c) >= drop.thr ? g * drop.inv_keep : 0.f;
    if (relu) g = (v - mu) * rs * ww + bb > 0.f ? g : 0.f;
    return g;
  };
  float a0[2] = {0.f, 0.f}, a1[2] = {0.f, 0.f};
  if (act) {
    int64_t r = p;
    for (; r + SM_LANES < N; r += 2 * SM_LANES) {
      float xh0, xh1;
      const float g0 = grad_at(r, xh0), g1 = grad_at(r + SM_LANES, xh1);
      a0[0] += g0; a1[0] = fmaf(g0, xh0, a1[0]);
      a0[1] += g1; a1[1] = fmaf(g1, xh1, a1[1]);
    }
    if (r < N) {
      float xh0;
      const float g0 = grad_at(r, xh0);
      a0[0] += g0; a1[0] = fmaf(g0, xh0, a1[0]);
    }
  }
  const float s0 = lanes_sum(a0[0] + a0[1], sm), s1 = lanes_sum(a1[0] + a1[1], sm);
  if (p == 0 && act) {
    dbias[c] = s0;
    dweight[c] = s1;
    s_db[cl] = s0;
    s_dw[cl] = s1;
  }
  __syncthreads();
  if (!act) return;
  const float inv_n = training ? 1.0f / (float)N : 0.f;
  const float db = s_db[cl] * inv_n, dw = s_dw[cl] * inv_n;
  for (int64_t r = p; r < N; r += SM_LANES) {
    float xh;
    const float g = grad_at(r, xh);
    st1<T>(dx + r * D + c, ww * rs * (g - db - xh * dw));
  }
}

// ---- synchronised statistics across data-parallel ranks (modules/gnn_module.py:204,164,167 normalise over the
// single-device batch, which graph sharding splits).  The collective belongs to the caller (torch.distributed / RCCL): while a hook
// is set for the calling HOST THREAD every training-mode BatchNorm of this library -- also the ones inside the composite layer
// entry points -- hands its local statistics to the hook between its two passes:
//   kind 0 (forward):  buf[0 .. n)           = {rows, mean[D], biased var[D]} of the local rows, n = 2 D + 1
//                      buf[n .. n + world n) <- the hook all-gathers every rank's n floats here, in rank order
//   kind 1 (backward): buf[0 .. 2 D)         = {sum dy', sum dy' xhat} over the local rows  <- the hook all-reduces (sum) in place
// `stream` is the stream the BatchNorm runs on (the collective has to be ordered on it); the hook returns 0 or an error code.
// Forward then merges the ranks' statistics in rank order (Chan et al.) -- identical on every rank -- and updates the running
// statistics with the GLOBAL batch; backward applies with the global sums and the global row count.  Parameter gradients stay the
// local sums (the gradient all-reduce averages them like every other parameter).
constexpr int BN_SYNC_MAX_WORLD = 64;
struct BnSync {
  gt_bn_sync_fn fn = nullptr;
  void* user = nullptr;
  int world = 1;
};
thread_local BnSync g_bn_sync;

__global__ void k_bn_sync_pack(const float* __restrict__ mean, const float* __restrict__ rstd, float rows, float eps, int64_t D,
                               float* __restrict__ buf) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0) buf[0] = rows;
  if (c < D) {
    const float rs = rstd[c];
    const float var = 1.0f / (rs * rs) - eps;
    buf[1 + c] = mean[c];
    buf[1 + D + c] = var > 0.f ? var : 0.f;
  }
}
// one thread per column: ranks merged in rank order, in double
__global__ void k_bn_sync_merge(const float* __restrict__ gathered, int world, int64_t D, float eps, float momentum,
                                float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ running_mean,
                                float* __restrict__ running_var, int64_t* __restrict__ nbt) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = 2 * D + 1;
  if (c == 0 && nbt) nbt[0] += 1;
  if (c >= D) return;
  double cnt = 0.0, mu = 0.0, m2 = 0.0;
  for (int r = 0; r < world; ++r) {
    const double nr = (double)gathered[r * n];
    if (nr <= 0.0) continue;
    const double mr = (double)gathered[r * n + 1 + c], vr = (double)gathered[r * n + 1 + D + c];
    const double tot = cnt + nr, delta = mr - mu;
    m2 += vr * nr + delta * delta * cnt * nr / tot;
    mu += delta * nr / tot;
    cnt = tot;
  }
  const double var = cnt > 0.0 ? m2 / cnt : 0.0;
  mean[c] = (float)mu;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unbiased = cnt > 1.0 ? var * (cnt / (cnt - 1.0)) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}
__global__ void k_bn_sync_copy2(const float* __restrict__ a, const float* __restrict__ b, float rows, int64_t D, float* __restrict__ out) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0) out[2 * D] = rows;
  if (c < D) {
    out[c] = a[c];
    out[D + c] = b[c];
  }
}
// the exchange area of a BatchNorm workspace: behind the column partials (gt_batchnorm_workspace_bytes)
float* bn_sync_buf(void* workspace, int64_t rows, int64_t dim);

int part_blocks(int64_t N);
float* bn_sync_buf(void* workspace, int64_t rows, int64_t dim) {
  uintptr_t p = (uintptr_t)workspace + (size_t)part_blocks(rows) * 2 * dim * sizeof(float);
  return (float*)((p + 255) & ~(uintptr_t)255);
}

int part_blocks(int64_t N) {
  int64_t b = gt_cdiv(N, 32);
  return (int)(b < 1 ? 1 : (b > MAX_PART ? MAX_PART : b));
}
int flat_blocks(int64_t items) {
  int64_t g = gt_cdiv(items, NT * 2);
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}
size_t rowlane_lds(int64_t D, int nv) {
  int C = (int)(D / 4);
  int CW = C < NT ? C : NT;
  int R = NT / CW;
  return (size_t)R * CW * nv * sizeof(float4);
}

int check_norm(const char* fn, int dtype, int64_t rows, int64_t D) {
  if (dtype != GT_F32 && dtype != GT_BF16) { gt_set_error("%s: bad dtype", fn); return GT_ERR_INVALID_ARG; }
  if (rows < 0 || D <= 0) { gt_set_error("%s: bad sizes", fn); return GT_ERR_INVALID_ARG; }
  if (D % 4 != 0 || D > 4096) { gt_set_error("%s: dim %lld unsupported (dim %% 4 == 0, dim <= 4096)", fn, (long long)D); return GT_ERR_UNSUPPORTED; }
  return GT_OK;
}

// ------------------------------------------------------------------------------------------------
// residual + dropout + LayerNorm   (one 16/32/64-lane group per row)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ln_hash(uint32_t s0, uint32_t s1, uint32_t row, uint32_t col) {
  uint32_t x = (row * 0x9E3779B1u + s0) ^ (col * 0x85EBCA77u + s1);
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

struct LnArgs {
  const void* x;      // sub-layer output (dropout applies to it) or the only input
  const void* resid;  // optional residual stream
  const float* w;
  const float* b;
  void* y;
  float* mean;
  float* rstd;
  const void* dy;     // bwd
  void* dx;           // bwd: grad wrt x (through dropout); may be null when x has no grad
  void* dresid;       // bwd: grad wrt resid (= dz); null when no residual
  float* part;        // bwd: [blocks][2][D]
  int64_t rows, D;
  float eps, inv_keep;
  uint32_t thr, s0, s1;
};

template <typename T, int LPN, int NCH>
__device__ __forceinline__ void ln_load_z(const LnArgs& a, int64_t row, int sl, float4 (&z)[NCH], bool (&keep)[NCH][4]) {
  const T* x = reinterpret_cast<const T*>(a.x);
  const T* rs = reinterpret_cast<const T*>(a.resid);
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int col = (sl + j * 64) * 4;
    keep[j][0] = keep[j][1] = keep[j][2] = keep[j][3] = true;
    if (col >= a.D) {
      z[j] = gt_zero4();
      continue;
    }
    float4 v = gt_load4<T>(x + row * a.D + col);
    if (a.thr) {
      float* vv = reinterpret_cast<float*>(&v);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        keep[j][e] = ln_hash(a.s0, a.s1, (uint32_t)row, (uint32_t)(col + e)) >= a.thr;
        vv[e] = keep[j][e] ? vv[e] * a.inv_keep : 0.f;
      }
    }
    if (rs) v = gt_add4(v, gt_load4<T>(rs + row * a.D + col));
    z[j] = v;
  }
}

template <int LPN>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = 1; o < LPN; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename T, int LPN, int NCH>
__global__ void __launch_bounds__(NT) k_ln_fwd(LnArgs a) {
  constexpr int NPW = 64 / LPN;
  const int lane = threadIdx.x & 63;
  const int sub = lane / LPN, sl = lane % LPN;
  const int64_t row = ((int64_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6)) * NPW + sub;
  if (row >= a.rows) return;
  float4 z[NCH];
  bool keep[NCH][4];
  ln_load_z<T, LPN, NCH>(a, row, sl, z, keep);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NCH; ++j) s += (z[j].x + z[j].y) + (z[j].z + z[j].w);
  const float mu = group_sum<LPN>(s) / (float)a.D;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int col = (sl + j * 64) * 4;
    if (col < a.D) {
      const float dx = z[j].x - mu, dy = z[j].y - mu, dz = z[j].z - mu, dw = z[j].w - mu;
      q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  }
  const float rs = 1.0f / sqrtf(group_sum<LPN>(q) / (float)a.D + a.eps);
  T* y = reinterpret_cast<T*>(a.y);
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int col = (sl + j * 64) * 4;
    if (col >= a.D) continue;
    const float4 w = *reinterpret_cast<const float4*>(a.w + col), b = *reinterpret_cast<const float4*>(a.b + col);
    gt_store4<T>(y + row * a.D + col, make_float4((z[j].x - mu) * rs * w.x + b.x, (z[j].y - mu) * rs * w.y + b.y,
                                                  (z[j].z - mu) * rs * w.z + b.z, (z[j].w - mu) * rs * w.w + b.w));
  }
  if (sl == 0) {
    a.mean[row] = mu;
    a.rstd[row] = rs;
  }
}

// backward: persistent grid; per-lane column accumulators for dw/db, block partials, fixed-order finish
// NTB threads per block: the column partials are reduced per block, so big blocks (1024 threads) put 6-8
// waves on every SIMD without multiplying the partial rows the finish kernel has to sum.
template <typename T, int LPN, int NCH, int NTB>
__global__ void __launch_bounds__(NTB) k_ln_bwd(LnArgs a) {
  constexpr int NPW = 64 / LPN;
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [4 waves][2][D]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int sub = lane / LPN, sl = lane % LPN;
  float4 aw[NCH], ab[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) aw[j] = ab[j] = gt_zero4();
  const T* dy = reinterpret_cast<const T*>(a.dy);
  const int64_t total_waves = (int64_t)gridDim.x * (NTB / 64);
  // the next row's operands are in flight while this row is reduced (a row is only 3 short loads per
  // lane followed by two cross-lane reductions: without the prefetch every trip exposes a round trip)
  struct Raw {
    float4 x[NCH], r[NCH], d[NCH];
    float mu, rs;
  };
  const T* xin = reinterpret_cast<const T*>(a.x);
  const T* rin = reinterpret_cast<const T*>(a.resid);
  auto load_raw = [&](int64_t row, Raw& q) {
    if (row >= a.rows) return;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int col = (sl + j * 64) * 4;
      if (col >= a.D) continue;
      q.x[j] = gt_load4<T>(xin + row * a.D + col);
      q.r[j] = rin ? gt_load4<T>(rin + row * a.D + col) : gt_zero4();
      q.d[j] = gt_load4<T>(dy + row * a.D + col);
    }
    q.mu = a.mean[row];
    q.rs = a.rstd[row];
  };
  Raw cur, nxt;
  const int64_t base0 = ((int64_t)blockIdx.x * (NTB / 64) + wid) * NPW;
  load_raw(base0 + sub, cur);
  for (int64_t base = base0; base < a.rows; base += total_waves * NPW) {
    const int64_t row = base + sub;
    load_raw(row + total_waves * NPW, nxt);
    if (row < a.rows) {
    float4 z[NCH];
    bool keep[NCH][4];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int col = (sl + j * 64) * 4;
      keep[j][0] = keep[j][1] = keep[j][2] = keep[j][3] = true;
      if (col >= a.D) {
        z[j] = gt_zero4();
        continue;
      }
      float4 v = cur.x[j];
      if (a.thr) {
        float* vv = reinterpret_cast<float*>(&v);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          keep[j][e] = ln_hash(a.s0, a.s1, (uint32_t)row, (uint32_t)(col + e)) >= a.thr;
          vv[e] = keep[j][e] ? vv[e] * a.inv_keep : 0.f;
        }
      }
      if (rin) v = gt_add4(v, cur.r[j]);
      z[j] = v;
    }
    const float mu = cur.mu, rs = cur.rs;
    float4 g[NCH], xh[NCH];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int col = (sl + j * 64) * 4;
      if (col >= a.D) {
        g[j] = xh[j] = gt_zero4();
        continue;
      }
      const float4 d = cur.d[j];
      const float4 w = *reinterpret_cast<const float4*>(a.w + col);
      xh[j] = make_float4((z[j].x - mu) * rs, (z[j].y - mu) * rs, (z[j].z - mu) * rs, (z[j].w - mu) * rs);
      ab[j] = gt_add4(ab[j], d);
      aw[j] = make_float4(fmaf(d.x, xh[j].x, aw[j].x), fmaf(d.y, xh[j].y, aw[j].y), fmaf(d.z, xh[j].z, aw[j].z),
                          fmaf(d.w, xh[j].w, aw[j].w));
      g[j] = make_float4(d.x * w.x, d.y * w.y, d.z * w.z, d.w * w.w);
      s1 += (g[j].x + g[j].y) + (g[j].z + g[j].w);
      s2 += (g[j].x * xh[j].x + g[j].y * xh[j].y) + (g[j].z * xh[j].z + g[j].w * xh[j].w);
    }
    const float m1 = group_sum<LPN>(s1) / (float)a.D, m2 = group_sum<LPN>(s2) / (float)a.D;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int col = (sl + j * 64) * 4;
      if (col >= a.D) continue;
      float4 dz = make_float4(rs * (g[j].x - m1 - xh[j].x * m2), rs * (g[j].y - m1 - xh[j].y * m2),
                              rs * (g[j].z - m1 - xh[j].z * m2), rs * (g[j].w - m1 - xh[j].w * m2));
      if (a.dresid) gt_store4<T>(reinterpret_cast<T*>(a.dresid) + row * a.D + col, dz);
      if (a.dx) {
        if (a.thr) {
          dz.x = keep[j][0] ? dz.x * a.inv_keep : 0.f;
          dz.y = keep[j][1] ? dz.y * a.inv_keep : 0.f;
          dz.z = keep[j][2] ? dz.z * a.inv_keep : 0.f;
          dz.w = keep[j][3] ? dz.w * a.inv_keep : 0.f;
        }
        gt_store4<T>(reinterpret_cast<T*>(a.dx) + row * a.D + col, dz);
      }
    }
    }
    cur = nxt;
  }
  // reduce (sub-groups -> waves -> block partial)
  float* part = a.part + (int64_t)blockIdx.x * 2 * a.D;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int col = (sl + j * 64) * 4;
    float4 w4 = aw[j], b4 = ab[j];
#pragma unroll
    for (int o = LPN; o < 64; o <<= 1) {
      w4.x += __shfl_xor(w4.x, o, 64); w4.y += __shfl_xor(w4.y, o, 64); w4.z += __shfl_xor(w4.z, o, 64); w4.w += __shfl_xor(w4.w, o, 64);
      b4.x += __shfl_xor(b4.x, o, 64); b4.y += __shfl_xor(b4.y, o, 64); b4.z += __shfl_xor(b4.z, o, 64); b4.w += __shfl_xor(b4.w, o, 64);
    }
    if (sub == 0 && col < a.D) {
      *reinterpret_cast<float4*>(lds + ((int64_t)wid * 2 + 0) * a.D + col) = w4;
      *reinterpret_cast<float4*>(lds + ((int64_t)wid * 2 + 1) * a.D + col) = b4;
    }
  }
  __syncthreads();
  for (int64_t i = threadIdx.x; i < 2 * a.D; i += NTB) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NTB / 64; ++w) t += lds[(int64_t)w * 2 * a.D + i];
    part[i] = t;  // [0][D] = dweight partial, [1][D] = dbias partial
  }
}

// LayerNorm backward for bf16 rows of EXACTLY 128 columns (the encoder layers' d_model): 16 lanes per row, 8 columns per lane in one
// 16-byte load per operand -- four rows per wave and trip instead of two, half the trips and half the dependent shuffle steps of the
// generic kernel (19 us for 40 MB at Code2's 32 k token rows: a chain of round trips, not bandwidth).  Same arithmetic, same dropout
// hash, same partial layout ([block][2][D]) for k_ln_bwd_finish.
// eight consecutive row elements as they travel in registers: bf16 = one 16-byte load, fp32 (r6: the fp32 contract mode's token rows ran the
// generic kernel, 37 us per call against 12 for this one) = two
template <typename T>
struct Row8;
template <>
struct Row8<gt_bf16> {
  uint4 v;
  __device__ __forceinline__ void load(const gt_bf16* p) { v = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void zero() { v = make_uint4(0, 0, 0, 0); }
  __device__ __forceinline__ void get(float* f) const {
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[2 * e] = __uint_as_float(u[e] << 16); f[2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u); }
  }
  static __device__ __forceinline__ void store(gt_bf16* p, const float* z) {
    *reinterpret_cast<uint4*>(p) = make_uint4(gt_pack_bf16(z[0], z[1]), gt_pack_bf16(z[2], z[3]), gt_pack_bf16(z[4], z[5]), gt_pack_bf16(z[6], z[7]));
  }
};
template <>
struct Row8<float> {
  float4 lo, hi;
  __device__ __forceinline__ void load(const float* p) { lo = *reinterpret_cast<const float4*>(p); hi = *reinterpret_cast<const float4*>(p + 4); }
  __device__ __forceinline__ void zero() { lo = gt_zero4(); hi = gt_zero4(); }
  __device__ __forceinline__ void get(float* f) const {
    f[0] = lo.x; f[1] = lo.y; f[2] = lo.z; f[3] = lo.w; f[4] = hi.x; f[5] = hi.y; f[6] = hi.z; f[7] = hi.w;
  }
  static __device__ __forceinline__ void store(float* p, const float* z) {
    *reinterpret_cast<float4*>(p) = make_float4(z[0], z[1], z[2], z[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(z[4], z[5], z[6], z[7]);
  }
};

template <typename T, int NTB, int DD>
__global__ void __launch_bounds__(NTB) k_ln_bwd_d128(LnArgs a) {
  static_assert(DD == 128 || DD == 256, "8 columns per lane, 16 or 32 lanes per row");
  constexpr int LPR = DD / 8, RPW = 64 / LPR;   // lanes per row, rows per wave and trip
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [waves][2][DD]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int sub = lane / LPR, sl = lane % LPR, col = sl * 8;
  const T* dy = reinterpret_cast<const T*>(a.dy);
  const T* xin = reinterpret_cast<const T*>(a.x);
  const T* rin = reinterpret_cast<const T*>(a.resid);
  float gw[8], aw[8], ab[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { gw[e] = a.w[col + e]; aw[e] = 0.f; ab[e] = 0.f; }
  const int64_t stride = (int64_t)gridDim.x * (NTB / 64) * RPW;
  int64_t row = ((int64_t)blockIdx.x * (NTB / 64) + wid) * RPW + sub;
  // the next trip's operands are in flight while this trip is reduced
  Row8<T> cx, cr, cd, nx, nr, nd;
  float cmu, crs, nmu, nrs;
  auto clampr = [&](int64_t r) { return r < a.rows ? r : a.rows - 1; };
  cr.zero();
  nr.zero();
  {
    const int64_t q = clampr(row) * DD + col;
    cx.load(xin + q);
    if (rin) cr.load(rin + q);
    cd.load(dy + q);
    cmu = a.mean[clampr(row)];
    crs = a.rstd[clampr(row)];
  }
  for (; row - sub < a.rows; row += stride) {   // (uniform per wave: its four rows start at row - sub)
    {
      const int64_t rn = clampr(row + stride), q = rn * DD + col;
      nx.load(xin + q);
      if (rin) nr.load(rin + q);
      nd.load(dy + q);
      nmu = a.mean[rn];
      nrs = a.rstd[rn];
    }
    const bool live = row < a.rows;
    const float cnt = live ? 1.f : 0.f;
    float d[8], x[8], r[8];
    cd.get(d);
    cx.get(x);
    cr.get(r);
    bool keep[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      keep[e] = true;
      if (a.thr) {
        keep[e] = ln_hash(a.s0, a.s1, (uint32_t)(live ? row : a.rows - 1), (uint32_t)(col + e)) >= a.thr;
        x[e] = keep[e] ? x[e] * a.inv_keep : 0.f;
      }
    }
    float g[8], xh[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float z = rin ? x[e] + r[e] : x[e];
      xh[e] = (z - cmu) * crs;
      g[e] = d[e] * gw[e];
      ab[e] = fmaf(d[e], cnt, ab[e]);
      aw[e] = fmaf(d[e] * cnt, xh[e], aw[e]);
    }
    s1 = ((g[0] + g[1]) + (g[2] + g[3])) + ((g[4] + g[5]) + (g[6] + g[7]));
    s2 = ((g[0] * xh[0] + g[1] * xh[1]) + (g[2] * xh[2] + g[3] * xh[3])) + ((g[4] * xh[4] + g[5] * xh[5]) + (g[6] * xh[6] + g[7] * xh[7]));
    const float m1 = group_sum<LPR>(s1) * (1.0f / (float)DD), m2 = group_sum<LPR>(s2) * (1.0f / (float)DD);
    float dz[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) dz[e] = crs * (g[e] - m1 - xh[e] * m2);
    if (live) {
      if (a.dresid) Row8<T>::store(reinterpret_cast<T*>(a.dresid) + row * DD + col, dz);
      if (a.dx) {
        if (a.thr) {
#pragma unroll
          for (int e = 0; e < 8; ++e) dz[e] = keep[e] ? dz[e] * a.inv_keep : 0.f;
        }
        Row8<T>::store(reinterpret_cast<T*>(a.dx) + row * DD + col, dz);
      }
    }
    cx = nx; cr = nr; cd = nd; cmu = nmu; crs = nrs;
  }
  // the four row groups of a wave -> the waves of the block -> one partial row per block
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if (LPR == 16) { aw[e] += __shfl_xor(aw[e], 16, 64); ab[e] += __shfl_xor(ab[e], 16, 64); }
    aw[e] += __shfl_xor(aw[e], 32, 64);
    ab[e] += __shfl_xor(ab[e], 32, 64);
  }
  if (sub == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      lds[(wid * 2 + 0) * DD + col + e] = aw[e];
      lds[(wid * 2 + 1) * DD + col + e] = ab[e];
    }
  }
  __syncthreads();
  float* part = a.part + (int64_t)blockIdx.x * 2 * DD;
  for (int i = threadIdx.x; i < 2 * DD; i += NTB) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NTB / 64; ++w) t += lds[w * 2 * DD + i];
    part[i] = t;   // [0][DD] = dweight partial, [1][DD] = dbias partial
  }
}
__global__ void k_ln_bwd_finish(const float* __restrict__ part, int nblk, int64_t D, float* __restrict__ dweight,
                                float* __restrict__ dbias) {
  __shared__ float sm[FINK_LANES * FIN_COLS];
  const int64_t c = (int64_t)blockIdx.x * FIN_COLS + threadIdx.x % FIN_COLS;
  const float s1 = finish_sum<FINK_LANES>(part, nblk, 2 * D, c, c < D, sm);
  const float s2 = finish_sum<FINK_LANES>(part, nblk, 2 * D, D + c, c < D, sm);
  if (c >= D || threadIdx.x >= FIN_COLS) return;
  dweight[c] = s1;
  dbias[c] = s2;
}

constexpr int LN_BWD_BLOCKS = 256;   // (512 blocks for the d = 128 kernel -- one trip per wave -- measured 1.2 % slower end to end)

template <typename T, bool BWD>
void ln_launch(const LnArgs& a, int grid_bwd, hipStream_t stream) {
  const int64_t D = a.D;
  if constexpr (BWD) {
    if (D == 128) {
      hipLaunchKernelGGL((k_ln_bwd_d128<T, 1024, 128>), dim3(grid_bwd), dim3(1024), (size_t)16 * 2 * 128 * 4, stream, a);
      return;
    }
    if (D == 256) {   // the same scheme at 32 lanes per row (the Erdos-Renyi stress: d_model 256, 131 k token rows)
      hipLaunchKernelGGL((k_ln_bwd_d128<T, 1024, 256>), dim3(grid_bwd), dim3(1024), (size_t)16 * 2 * 256 * 4, stream, a);
      return;
    }
  }
#define GT_LN(LPN, NCH)                                                                                      \
  do {                                                                                                       \
    if constexpr (BWD) {                                                                                     \
      if (D <= 256)                                                                                        \
        hipLaunchKernelGGL((k_ln_bwd<T, LPN, NCH, 1024>), dim3(grid_bwd), dim3(1024), (size_t)16 * 2 * D * 4, \
                           stream, a);                                                                       \
      else                                                                                                   \
        hipLaunchKernelGGL((k_ln_bwd<T, LPN, NCH, NT>), dim3(grid_bwd), dim3(NT), (size_t)(NT / 64) * 2 * D * 4, \
                           stream, a);                                                                       \
    } else {                                                                                                 \
      int64_t waves = gt_cdiv(a.rows, 64 / (LPN));                                                           \
      hipLaunchKernelGGL((k_ln_fwd<T, LPN, NCH>), dim3((unsigned)gt_cdiv(waves, NT / 64)), dim3(NT), 0,      \
                         stream, a);                                                                         \
    }                                                                                                        \
  } while (0)
  if (D <= 64) GT_LN(16, 1);
  else if (D <= 128) GT_LN(32, 1);
  else if (D <= 256) GT_LN(64, 1);
  else if (D <= 512) GT_LN(64, 2);
  else if (D <= 768) GT_LN(64, 3);
  else GT_LN(64, 4);
#undef GT_LN
}

void fill_drop(LnArgs& a, float dropout_p, uint64_t seed) {
  a.inv_keep = 1.0f / (1.0f - dropout_p);
  double thr = (double)dropout_p * 4294967296.0;
  a.thr = dropout_p > 0.f ? (uint32_t)(thr > 4294967295.0 ? 4294967295.0 : (thr < 1.0 ? 1.0 : thr)) : 0u;
  a.s0 = (uint32_t)seed;
  a.s1 = (uint32_t)(seed >> 32);
}

}  // namespace

// ---- C ABI -----------------------------------------------------------------------------------------
static BnDrop make_bn_drop(float dropout_p, uint64_t seed) {
  BnDrop d{};
  d.inv_keep = 1.0f / (1.0f - dropout_p);
  double thr = (double)dropout_p * 4294967296.0;
  d.thr = dropout_p > 0.f ? (uint32_t)(thr > 4294967295.0 ? 4294967295.0 : (thr < 1.0 ? 1.0 : thr)) : 0u;
  d.s0 = (uint32_t)seed;
  d.s1 = (uint32_t)(seed >> 32);
  return d;
}

extern "C" size_t gt_batchnorm_workspace_bytes(int64_t rows, int64_t dim) {
  // + the exchange buffers of synchronised statistics (gt_bn_sync_set): packed local statistics, the all-gathered copies of up to
  // BN_SYNC_MAX_WORLD ranks and the all-reduced gradient sums
  return (size_t)part_blocks(rows) * 2 * dim * sizeof(float) + 256 + (size_t)(BN_SYNC_MAX_WORLD + 2) * (2 * dim + 4) * sizeof(float);
}

extern "C" int gt_batchnorm_fwd(int dtype, const void* x, const float* weight, const float* bias,
                                float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                float momentum, float eps, int training, int relu, const void* resid, int64_t rows,
                                int64_t dim, void* y, float* save_mean, float* save_rstd, float dropout_p,
                                uint64_t seed, void* workspace, size_t workspace_bytes, gt_stream_t stream_) {
  return gt_batchnorm_fwd_bcast(dtype, x, weight, bias, running_mean, running_var, num_batches_tracked, momentum, eps, training,
                                relu, resid, nullptr, nullptr, nullptr, rows, dim, y, save_mean, save_rstd, dropout_p, seed, workspace,
                                workspace_bytes, stream_);
}

extern "C" int gt_batchnorm_fwd_bcast(int dtype, const void* x, const float* weight, const float* bias,
                                      float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                      float momentum, float eps, int training, int relu, const void* resid,
                                      const void* bcast, const int32_t* bcast_index, void* ev_bcast_ready, int64_t rows,
                                      int64_t dim, void* y, float* save_mean, float* save_rstd, float dropout_p,
                                      uint64_t seed, void* workspace, size_t workspace_bytes, gt_stream_t stream_) {
  int rc = check_norm("gt_batchnorm_fwd", dtype, rows, dim);
  GT_CHECK_ARG(!bcast || bcast_index, "bcast needs bcast_index");

  if (rc) return rc;
  GT_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "dropout_p must be in [0,1)");
  const BnDrop drop = make_bn_drop(training ? dropout_p : 0.f, seed);
  GT_CHECK_ARG(x && weight && bias && y && save_mean && save_rstd, "null buffer");
  GT_CHECK_ARG(training || (running_mean && running_var), "eval mode needs running statistics");
  const bool sync = training && g_bn_sync.fn && g_bn_sync.world > 1;   // statistics over every rank's rows (gt_bn_sync_set)
  if (rows == 0 && !sync) return GT_OK;
  hipStream_t stream = (hipStream_t)stream_;
  const int cgrid = (int)gt_cdiv(dim, FIN_COLS);
  if (training) {
    GT_CHECK_ARG(rows > 1 || sync, "BatchNorm in training mode needs more than 1 row");  // torch raises too
    if (rows <= SMALL_ROWS && !sync) {
      if (bcast && ev_bcast_ready) {
        rc = gt_stream_wait_event(stream_, ev_bcast_ready);
        if (rc) return rc;
      }
      if (dtype == GT_F32)
        hipLaunchKernelGGL(k_bn_small_fwd<float>, dim3((unsigned)gt_cdiv(dim, SM_COLS)), dim3(SM_COLS * SM_LANES), 0, stream, (const float*)x, rows, dim,
                           eps, momentum, weight, bias, (const float*)resid, relu, drop, save_mean, save_rstd, running_mean,
                           running_var, num_batches_tracked, (float*)y, (const float*)bcast, bcast_index);
      else
        hipLaunchKernelGGL(k_bn_small_fwd<gt_bf16>, dim3((unsigned)gt_cdiv(dim, SM_COLS)), dim3(SM_COLS * SM_LANES), 0, stream, (const gt_bf16*)x, rows,
                           dim, eps, momentum, weight, bias, (const gt_bf16*)resid, relu, drop, save_mean, save_rstd,
                           running_mean, running_var, num_batches_tracked, (gt_bf16*)y, (const gt_bf16*)bcast, bcast_index);
      GT_CHECK_LAUNCH();
      return GT_OK;
    }
    const int nb = part_blocks(rows);
    if (!workspace || workspace_bytes < gt_batchnorm_workspace_bytes(rows, dim)) {
      gt_set_error("gt_batchnorm_fwd: workspace too small");
      return GT_ERR_WORKSPACE;
    }
    float* part = (float*)workspace;
    size_t lds = rowlane_lds(dim, 2);
    // synchronised: the local pass leaves the running statistics alone -- the merge below updates them with the global batch
    float* rm_l = sync ? nullptr : running_mean;
    float* rv_l = sync ? nullptr : running_var;
    int64_t* nbt_l = sync ? nullptr : num_batches_tracked;
    if (rows > 0) {
      if (dtype == GT_F32) {
        hipLaunchKernelGGL(k_bn_stats_partial<float>, dim3(nb), dim3(NT), lds, stream, (const float*)x, rows, dim, part);
        hipLaunchKernelGGL(k_bn_stats_finish<float>, dim3(cgrid), dim3(FIN_COLS * FINK_LANES), 0, stream, (const float*)x, part, nb, rows,
                           dim, eps, momentum, save_mean, save_rstd, rm_l, rv_l, nbt_l);
      } else {
        hipLaunchKernelGGL(k_bn_stats_partial<gt_bf16>, dim3(nb), dim3(NT), lds, stream, (const gt_bf16*)x, rows, dim, part);
        hipLaunchKernelGGL(k_bn_stats_finish<gt_bf16>, dim3(cgrid), dim3(FIN_COLS * FINK_LANES), 0, stream, (const gt_bf16*)x, part, nb,
                           rows, dim, eps, momentum, save_mean, save_rstd, rm_l, rv_l, nbt_l);
      }
    }
    if (sync) {
      float* sb = bn_sync_buf(workspace, rows, dim);
      const int64_t n = 2 * dim + 1;
      const unsigned cg = (unsigned)gt_cdiv(dim > 1 ? dim : 1, 256);
      if (rows > 0) hipLaunchKernelGGL(k_bn_sync_pack, dim3(cg), dim3(256), 0, stream, save_mean, save_rstd, (float)rows, eps, dim, sb);
      else (void)hipMemsetAsync(sb, 0, (size_t)n * sizeof(float), stream);
      GT_CHECK_LAUNCH();
      const int hrc = g_bn_sync.fn(g_bn_sync.user, 0, sb, n, stream_);
      if (hrc) { gt_set_error("gt_batchnorm_fwd: the statistics hook failed (%d)", hrc); return GT_ERR_LAUNCH; }
      hipLaunchKernelGGL(k_bn_sync_merge, dim3(cg), dim3(256), 0, stream, sb + n, g_bn_sync.world, dim, eps, momentum, save_mean, save_rstd,
                         running_mean, running_var, num_batches_tracked);
      if (rows == 0) { GT_CHECK_LAUNCH(); return GT_OK; }
    }
  } else {
    hipLaunchKernelGGL(k_bn_eval_stats, dim3(cgrid), dim3(256), 0, stream, running_mean, running_var, dim, eps, save_mean,
                       save_rstd);
  }
  const int g = flat_blocks(rows * (dim / 4));
  if (bcast && ev_bcast_ready) {   // the rows to add come from another stream (the virtual-node update): join it here
    rc = gt_stream_wait_event(stream_, ev_bcast_ready);
    if (rc) return rc;
  }
  if (dtype == GT_F32)
    hipLaunchKernelGGL(k_bn_apply<float>, dim3(g), dim3(NT), 0, stream, (const float*)x, save_mean, save_rstd, weight,
                       bias, (const float*)resid, relu, drop, rows, dim, (float*)y, (const float*)bcast, bcast_index);
  else
    hipLaunchKernelGGL(k_bn_apply<gt_bf16>, dim3(g), dim3(NT), 0, stream, (const gt_bf16*)x, save_mean, save_rstd, weight,
                       bias, (const gt_bf16*)resid, relu, drop, rows, dim, (gt_bf16*)y, (const gt_bf16*)bcast, bcast_index);
  GT_CHECK_LAUNCH();
  return GT_OK;
}

extern "C" int gt_batchnorm_bwd(int dtype, const void* x, const void* dy, const float* weight, const float* bias,
                                const float* save_mean, const float* save_rstd, int training, int relu, int64_t rows,
                                int64_t dim, void* dx, float* dweight, float* dbias, float dropout_p, uint64_t seed,
                                void* workspace, size_t workspace_bytes, gt_stream_t stream_) {
  int rc = check_norm("gt_batchnorm_bwd", dtype, rows, dim);
  if (rc) return rc;
  GT_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "dropout_p must be in [0,1)");
  const BnDrop drop = make_bn_drop(training ? dropout_p : 0.f, seed);
  GT_CHECK_ARG(x && dy && weight && bias && save_mean && save_rstd && dx && dweight && dbias, "null buffer");
  const bool sync = training && g_bn_sync.fn && g_bn_sync.world > 1;
  if (rows == 0 && !sync) return GT_OK;
  if (!workspace || workspace_bytes < gt_batchnorm_workspace_bytes(rows, dim)) {
    gt_set_error("gt_batchnorm_bwd: workspace too small");
    return GT_ERR_WORKSPACE;
  }
  hipStream_t stream = (hipStream_t)stream_;
  const int nb = part_blocks(rows);
  float* part = (float*)workspace;
  const size_t lds = rowlane_lds(dim, 2);
  const int cgrid = (int)gt_cdiv(dim, FIN_COLS);
  const int g = flat_blocks(rows * (dim / 4));
  if (sync) {   // local sums -> all-reduce (sums and row count) -> apply with the global ones; dweight / dbias stay local
    const unsigned cg = (unsigned)gt_cdiv(dim > 1 ? dim : 1, 256);
    float* sb = bn_sync_buf(workspace, rows, dim);
    if (rows > 0) {
      if (dtype == GT_F32)
        hipLaunchKernelGGL(k_bn_bwd_partial<float>, dim3(nb), dim3(NT), lds, stream, (const float*)x, (const float*)dy, weight, bias,
                           save_mean, save_rstd, relu, drop, rows, dim, part);
      else
        hipLaunchKernelGGL(k_bn_bwd_partial<gt_bf16>, dim3(nb), dim3(NT), lds, stream, (const gt_bf16*)x, (const gt_bf16*)dy, weight,
                           bias, save_mean, save_rstd, relu, drop, rows, dim, part);
      hipLaunchKernelGGL(k_bn_bwd_finish, dim3(cgrid), dim3(FIN_COLS * FINK_LANES), 0, stream, part, nb, dim, dbias, dweight);
      hipLaunchKernelGGL(k_bn_sync_copy2, dim3(cg), dim3(256), 0, stream, dbias, dweight, (float)rows, dim, sb);
    } else {
      (void)hipMemsetAsync(dbias, 0, (size_t)dim * sizeof(float), stream);
      (void)hipMemsetAsync(dweight, 0, (size_t)dim * sizeof(float), stream);
      (void)hipMemsetAsync(sb, 0, (size_t)(2 * dim + 1) * sizeof(float), stream);
    }
    GT_CHECK_LAUNCH();
    const int hrc = g_bn_sync.fn(g_bn_sync.user, 1, sb, 2 * dim + 1, stream_);
    if (hrc) { gt_set_error("gt_batchnorm_bwd: the statistics hook failed (%d)", hrc); return GT_ERR_LAUNCH; }
    if (rows > 0) {
      if (dtype == GT_F32)
        hipLaunchKernelGGL(k_bn_bwd_apply<float>, dim3(g), dim3(NT), 0, stream, (const float*)x, (const float*)dy, save_mean, save_rstd,
                           weight, bias, sb, sb + dim, relu, 1.0f, drop, rows, dim, (float*)dx, (const float*)(sb + 2 * dim));
      else
        hipLaunchKernelGGL(k_bn_bwd_apply<gt_bf16>, dim3(g), dim3(NT), 0, stream, (const gt_bf16*)x, (const gt_bf16*)dy, save_mean,
                           save_rstd, weight, bias, sb, sb + dim, relu, 1.0f, drop, rows, dim, (gt_bf16*)dx, (const float*)(sb + 2 * dim));
    }
    GT_CHECK_LAUNCH();
    return GT_OK;
  }
  if (rows <= SMALL_ROWS) {
    if (dtype == GT_F32)
      hipLaunchKernelGGL(k_bn_small_bwd<float>, dim3((unsigned)gt_cdiv(dim, SM_COLS)), dim3(SM_COLS * SM_LANES), 0, stream, (const float*)x,
                         (const float*)dy, save_mean, save_rstd, weight, bias, relu, training, drop, rows, dim, dbias, dweight,
                         (float*)dx);
    else
      hipLaunchKernelGGL(k_bn_small_bwd<gt_bf16>, dim3((unsigned)gt_cdiv(dim, SM_COLS)), dim3(SM_COLS * SM_LANES), 0, stream, (const gt_bf16*)x,
                         (const gt_bf16*)dy, save_mean, save_rstd, weight, bias, relu, training, drop, rows, dim, dbias,
                         dweight, (gt_bf16*)dx);
    GT_CHECK_LAUNCH();
    return GT_OK;
  }
  if (dtype == GT_F32) {
    hipLaunchKernelGGL(k_bn_bwd_partial<float>, dim3(nb), dim3(NT), lds, stream, (const float*)x, (const float*)dy,
                       weight, bias, save_mean, save_rstd, relu, drop, rows, dim, part);
    hipLaunchKernelGGL(k_bn_bwd_finish, dim3(cgrid), dim3(FIN_COLS * FINK_LANES), 0, stream, part, nb, dim, dbias, dweight);
    hipLaunchKernelGGL(k_bn_bwd_apply<float>, dim3(g), dim3(NT), 0, stream, (const float*)x, (const float*)dy,
                       save_mean, save_rstd, weight, bias, dbias, dweight, relu, training ? 1.0f / (float)rows : 0.f, drop, rows, dim, (float*)dx, (const float*)nullptr);
  } else {
    hipLaunchKernelGGL(k_bn_bwd_partial<gt_bf16>, dim3(nb), dim3(NT), lds, stream, (const gt_bf16*)x, (const gt_bf16*)dy,
                       weight, bias, save_mean, save_rstd, relu, drop, rows, dim, part);
    hipLaunchKernelGGL(k_bn_bwd_finish, dim3(cgrid), dim3(FIN_COLS * FINK_LANES), 0, stream, part, nb, dim, dbias, dweight);
    hipLaunchKernelGGL(k_bn_bwd_apply<gt_bf16>, dim3(g), dim3(NT), 0, stream, (const gt_bf16*)x, (const gt_bf16*)dy,
                       save_mean, save_rstd, weight, bias, dbias, dweight, relu, training ? 1.0f / (float)rows : 0.f, drop, rows, dim, (gt_bf16*)dx, (const float*)nullptr);
  }
  GT_CHECK_LAUNCH();
  return GT_OK;
}

// gt_batchnorm_bwd with the pass-1 partial sums supplied by the caller: part[nparts][2][dim] (sum dy', sum dy' * xhat over any
// partition of the rows, e.g. the 64-row tiles of the GEMM whose dX epilogue produced them: gt_linear_bwd_bnstats).  Runs the
// fixed-order finish and the apply pass only.  dropout_p must be 0 (the producers do not replay the dropout mask).
extern "C" int gt_batchnorm_bwd_parts(int dtype, const void* x, const void* dy, const float* weight, const float* bias,
                                      const float* save_mean, const float* save_rstd, int training, int relu, int64_t rows,
                                      int64_t dim, void* dx, float* dweight, float* dbias, const float* part, int64_t nparts,
                                      gt_stream_t stream_) {
  int rc = check_norm("gt_batchnorm_bwd_parts", dtype, rows, dim);
  if (rc) return rc;
  GT_CHECK_ARG(x && dy && weight && bias && save_mean && save_rstd && dx && dweight && dbias && part, "null buffer");
  GT_CHECK_ARG(nparts >= 1 && nparts <= 0x7fffffff, "bad partial count");
  if (rows == 0) return GT_OK;
  hipStream_t stream = (hipStream_t)stream_;
  const BnDrop drop = make_bn_drop(0.f, 0);
  const int cgrid = (int)gt_cdiv(dim, FIN_COLS);
  const int g = flat_blocks(rows * (dim / 4));
  hipLaunchKernelGGL(k_bn_bwd_finish, dim3(cgrid), dim3(FIN_COLS * FINK_LANES), 0, stream, part, (int)nparts, dim, dbias, dweight);
  if (dtype == GT_F32)
    hipLaunchKernelGGL(k_bn_bwd_apply<float>, dim3(g), dim3(NT), 0, stream, (const float*)x, (const float*)dy, save_mean, save_rstd,
                       weight, bias, dbias, dweight, relu, training ? 1.0f / (float)rows : 0.f, drop, rows, dim, (float*)dx, (const float*)nullptr);
  else
    hipLaunchKernelGGL(k_bn_bwd_apply<gt_bf16>, dim3(g), dim3(NT), 0, stream, (const gt_bf16*)x, (const gt_bf16*)dy, save_mean,
                       save_rstd, weight, bias, dbias, dweight, relu, training ? 1.0f / (float)rows : 0.f, drop, rows, dim, (gt_bf16*)dx, (const float*)nullptr);
  GT_CHECK_LAUNCH();
  return GT_OK;
}

// ---- the apply passes on their own: synchronised BatchNorm across data-parallel ranks --------------------
// The statistics (forward) and the two gradient sums (backward) are reduced over the ranks by the caller between the
// library's local pass and these apply passes (graphtrans_amd/ops.py:sync_batch_norm).
extern "C" int gt_batchnorm_apply(int dtype, const void* x, const float* mean, const float* rstd, const float* weight,
                                  const float* bias, int relu, const void* resid, int64_t rows, int64_t dim, void* y,
                                  float dropout_p, uint64_t seed, gt_stream_t stream_) {
  int rc = check_norm("gt_batchnorm_apply", dtype, rows, dim);
  if (rc) return rc;
  GT_CHECK_ARG(x && mean && rstd && weight && bias && y, "null buffer");
  GT_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "dropout_p must be in [0,1)");
  if (rows == 0) return GT_OK;
  const BnDrop drop = make_bn_drop(dropout_p, seed);
  hipStream_t stream = (hipStream_t)stream_;
  const int g = flat_blocks(rows * (dim / 4));
  if (dtype == GT_F32)
    hipLaunchKernelGGL(k_bn_apply<float>, dim3(g), dim3(NT), 0, stream, (const float*)x, mean, rstd, weight, bias,
                       (const float*)resid, relu, drop, rows, dim, (float*)y, (const float*)nullptr, (const int32_t*)nullptr);
  else
    hipLaunchKernelGGL(k_bn_apply<gt_bf16>, dim3(g), dim3(NT), 0, stream, (const gt_bf16*)x, mean, rstd, weight, bias,
                       (const gt_bf16*)resid, relu, drop, rows, dim, (gt_bf16*)y, (const gt_bf16*)nullptr, (const int32_t*)nullptr);
  GT_CHECK_LAUNCH();
  return GT_OK;
}

extern "C" int gt_batchnorm_bwd_apply(int dtype, const void* x, const void* dy, const float* weight, const float* bias,
                                      const float* mean, const float* rstd, const float* sum_dy, const float* sum_dy_xhat,
                                      double count, int relu, int64_t rows, int64_t dim, void* dx, float dropout_p,
                                      uint64_t seed, gt_stream_t stream_) {
  int rc = check_norm("gt_batchnorm_bwd_apply", dtype, rows, dim);
  if (rc) return rc;
  GT_CHECK_ARG(x && dy && weight && bias && mean && rstd && sum_dy && sum_dy_xhat && dx, "null buffer");
  GT_CHECK_ARG(count >= 1.0, "count = rows the statistics were taken over (all ranks)");
  if (rows == 0) return GT_OK;
  const BnDrop drop = make_bn_drop(dropout_p, seed);
  hipStream_t stream = (hipStream_t)stream_;
  const int g = flat_blocks(rows * (dim / 4));
  const float inv_n = (float)(1.0 / count);
  if (dtype == GT_F32)
    hipLaunchKernelGGL(k_bn_bwd_apply<float>, dim3(g), dim3(NT), 0, stream, (const float*)x, (const float*)dy, mean, rstd, weight,
                       bias, sum_dy, sum_dy_xhat, relu, inv_n, drop, rows, dim, (float*)dx, (const float*)nullptr);
  else
    hipLaunchKernelGGL(k_bn_bwd_apply<gt_bf16>, dim3(g), dim3(NT), 0, stream, (const gt_bf16*)x, (const gt_bf16*)dy, mean, rstd,
                       weight, bias, sum_dy, sum_dy_xhat, relu, inv_n, drop, rows, dim, (gt_bf16*)dx, (const float*)nullptr);
  GT_CHECK_LAUNCH();
  return GT_OK;
}

// Per host thread: every training-mode BatchNorm call of this thread exchanges its statistics through `fn` (contract above);
// world = ranks in the exchange (2..64); fn == NULL or world <= 1 switches it off again.
extern "C" int gt_bn_sync_set(gt_bn_sync_fn fn, void* user, int world) {
  GT_CHECK_ARG(world >= 0 && world <= BN_SYNC_MAX_WORLD, "at most 64 ranks");
  g_bn_sync.fn = (fn && world > 1) ? fn : nullptr;
  g_bn_sync.user = user;
  g_bn_sync.world = g_bn_sync.fn ? world : 1;
  return GT_OK;
}

extern "C" int gt_layernorm_fwd(int dtype, const void* x, const void* resid, const float* weight, const float* bias,
                                float eps, float dropout_p, uint64_t seed, int64_t rows, int64_t dim, void* y,
                                float* save_mean, float* save_rstd, gt_stream_t stream_) {
  int rc = check_norm("gt_layernorm_fwd", dtype, rows, dim);
  if (rc) return rc;
  GT_CHECK_ARG(dim <= 1024, "LayerNorm dim > 1024 unsupported");
  GT_CHECK_ARG(x && weight && bias && y && save_mean && save_rstd, "null buffer");
  GT_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "dropout_p must be in [0,1)");
  if (rows == 0) return GT_OK;
  LnArgs a{};
  a.x = x; a.resid = resid; a.w = weight; a.b = bias; a.y = y; a.mean = save_mean; a.rstd = save_rstd;
  a.rows = rows; a.D = dim; a.eps = eps;
  fill_drop(a, dropout_p, seed);
  hipStream_t stream = (hipStream_t)stream_;
  if (dtype == GT_F32) ln_launch<float, false>(a, 0, stream);
  else ln_launch<gt_bf16, false>(a, 0, stream);
  GT_CHECK_LAUNCH();
  return GT_OK;
}

extern "C" size_t gt_layernorm_bwd_workspace_bytes(int64_t rows, int64_t dim) {
  (void)rows;
  return (size_t)LN_BWD_BLOCKS * 2 * dim * sizeof(float) + 256;
}

// dweight[c] = sum_b part[b][0][c], dbias[c] = sum_b part[b][1][c] over `nblk` block partials [2][dim] (k_ln_bwd_finish: the fixed
// order of gt_layernorm_bwd's own finish); for producers of such partials outside this file (linear1.h's LayerNorm-backward epilogue)
extern "C" int gt_layernorm_bwd_finish(const float* part, int nblk, int64_t dim, float* dweight, float* dbias, gt_stream_t stream_) {
  GT_CHECK_ARG(part && nblk >= 1 && dim >= 1 && dweight && dbias, "bad arguments");
  hipLaunchKernelGGL(k_ln_bwd_finish, dim3((unsigned)gt_cdiv(dim, FIN_COLS)), dim3(FIN_COLS * FINK_LANES), 0, (hipStream_t)stream_, part, nblk, dim,
                     dweight, dbias);
  GT_CHECK_LAUNCH();
  return GT_OK;
}

extern "C" int gt_layernorm_bwd(int dtype, const void* x, const void* resid, const void* dy, const float* weight,
                                const float* save_mean, const float* save_rstd, float dropout_p, uint64_t seed,
                                int64_t rows, int64_t dim, void* dx, void* dresid, float* dweight, float* dbias,
                                void* workspace, size_t workspace_bytes, gt_stream_t stream_) {
  int rc = check_norm("gt_layernorm_bwd", dtype, rows, dim);
  if (rc) return rc;
  GT_CHECK_ARG(dim <= 1024, "LayerNorm dim > 1024 unsupported");
  GT_CHECK_ARG(x && dy && weight && save_mean && save_rstd && dweight && dbias, "null buffer");
  GT_CHECK_ARG(dx || dresid, "nothing to compute");
  if (!workspace || workspace_bytes < gt_layernorm_bwd_workspace_bytes(rows, dim)) {
    gt_set_error("gt_layernorm_bwd: workspace too small");
    return GT_ERR_WORKSPACE;
  }
  hipStream_t stream = (hipStream_t)stream_;
  LnArgs a{};
  a.x = x; a.resid = resid; a.w = weight; a.dy = dy; a.dx = dx; a.dresid = dresid; a.mean = const_cast<float*>(save_mean);
  a.rstd = const_cast<float*>(save_rstd); a.rows = rows; a.D = dim; a.part = (float*)workspace;
  fill_drop(a, dropout_p, seed);
  const bool d128 = dim == 128;   // k_ln_bwd_d128: four rows per wave and trip
  const int64_t npw = d128 ? 4 : (dim <= 64 ? 4 : (dim <= 128 ? 2 : 1));
  const int bwd_waves = dim <= 256 ? 16 : NT / 64;   // waves per block of the launch below
  int64_t want = gt_cdiv(gt_cdiv(rows > 0 ? rows : 1, npw), bwd_waves);
  const int grid = (int)(want < LN_BWD_BLOCKS ? want : LN_BWD_BLOCKS);
  // inside a deferred-reduce section (gt_defer_begin: the whole-model backward) the block partials go to the section's arena and their
  // column sums join the section's one reduce launch: no 5-us finish kernel between two GEMMs of the caller's stream
  void* dpart = gt_defer_take((size_t)grid * 2 * dim * sizeof(float));
  if (dpart) a.part = (float*)dpart;
  if (dtype == GT_F32) ln_launch<float, true>(a, grid, stream);
  else ln_launch<gt_bf16, true>(a, grid, stream);
  if (dpart) {
    GT_CHECK_LAUNCH();
    return gt_defer_push(a.part, grid, dim, 2 * dim, dweight, a.part + dim, dim, 2 * dim, dbias);
  }
  // d gamma / d beta are parameter gradients, nothing on the critical path reads them -- but their 5-us column finish stays on the
  // caller's stream: forking it onto the overlap stream costs the caller's queue an event record now and a wait when the workspace is
  // handed on, two queue packets for one, and measured 0.5 % slower end to end on Code2 (profiles/LOG.md, round 5)
  hipLaunchKernelGGL(k_ln_bwd_finish, dim3((unsigned)gt_cdiv(dim, FIN_COLS)), dim3(FIN_COLS * FINK_LANES), 0, stream, (const float*)workspace, grid,
                     dim, dweight, dbias);
  GT_CHECK_LAUNCH();
  return GT_OK;
}
