// linear_bn_slab.h — Linear + BatchNorm over FEW rows as ONE launch per direction (included by norm.hip).
//
// The virtual-node update (modules/gnn_module.py:161-170,217-229) is Linear(D, 2D) -> BatchNorm -> ReLU -> Linear(2D, D) -> BatchNorm ->
// ReLU on ONE ROW PER GRAPH (B = 256 rows): ~0.1 GFLOP behind six launches forward and six backward on its own stream, a chain that
// the layer's BatchNorm apply (forward) and dX GEMM (backward) wait for.  A BatchNorm over B rows needs every row of a column and
// nothing else -- so a block that owns a SLAB of 16 output columns x all rows runs the GEMM for its slab and the whole BatchNorm on
// the accumulators, with no cross-block step (an in-launch hand-over between blocks costs an agent-scope release + acquire, 1.7 us
// each: MI355X_MICROARCH.md, as much as the launch boundary it would replace):
//   k_slab_lin_bn_fwd   z = x W^T + b (saved) ; y = [drop(relu(BN(z)))] (+ resid) ; batch statistics saved, running statistics updated
//   k_slab_dx_bn_bwd    da = dz_up W_up (the upper Linear's dX, slab = 16 of ITS input columns) ; dz = BN_bwd(da; z, mean, rstd [, relu
//                       gate]) of the BatchNorm below ; d gamma, d beta
// 8 waves per block, wave w owns the 16-row tiles w, w + 8, ...; operands straight from L2 in MFMA fragment shape (the forward's W
// rows, both kernels' activation rows), the backward's 16 W columns through one transposed LDS slab; column sums by DPP shuffles
// over the 16 row lanes and a [8 waves][16] LDS table in wave order -> bitwise reproducible.  Exact fp32 MFMA or bf16 MFMA (TC) like
// linear_small.h.  Fragment convention of mfma_frag.h.
#pragma once
#include "mfma_frag.h"

namespace slab {
using namespace gtf;

constexpr int SL_WAVES = 8, SL_THREADS = SL_WAVES * 64, SL_COLS = 16;
constexpr int SL_MAX_ROWS = 512, SL_MT = SL_MAX_ROWS / 16 / SL_WAVES;   // <= 4 row tiles per wave

struct SlabArgs {
  // forward                                  | backward
  const float* x;      // [M][K] input rows   | dz_up [M][N]: gradient of the UPPER Linear's output
  const float* w;      // [N][K]              | W_up [N][K]
  const float* bias;   // [N] or null         | -
  float* z;            // [M][N] pre-norm out | da scratch: not used (kept in registers)
  const float* zin;    // -                   | z [M][K] of the BatchNorm below (its saved input)
  float* y;            // [M][N]              | dz [M][K]
  const float* resid;  // [M][N] or null      | -
  const float *bn_w, *bn_b;
  float *mean, *rstd;            // saved statistics: written (fwd) / read (bwd)
  float *rmean, *rvar;           // running statistics or null
  int64_t* nbt;
  float *dgamma, *dbeta;         // backward
  int64_t M, N, K;
  float momentum, eps;
  int relu;
  BnDrop drop;
};

__device__ __forceinline__ void sl_load8(const float* p, int64_t rem, float* f) {   // 8 floats, zero beyond `rem` (a multiple of 4)
  const float4 a = rem > 0 ? *reinterpret_cast<const float4*>(p) : gt_zero4();
  const float4 b = rem > 4 ? *reinterpret_cast<const float4*>(p + 4) : gt_zero4();
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w;
  f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
// sum over the 16 row lanes (n) that share a lane group g: afterwards every lane of the group holds it
__device__ __forceinline__ float sl_rows16(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}
// block-wide column sums: v[r] = this wave's sum for column g * 4 + r -> the sum over the 8 waves in wave order, in every lane
__device__ __forceinline__ void sl_block_cols(float (&v)[4], float (*tab)[SL_COLS], int wave, int n, int g) {
  __syncthreads();   // (the table's previous use is over)
  if (n == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) tab[wave][g * 4 + r] = v[r];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < SL_WAVES; ++w) s += tab[w][g * 4 + r];
    v[r] = s;
  }
}

template <typename TC>
__global__ void __launch_bounds__(SL_THREADS) k_slab_lin_bn_fwd(SlabArgs a) {
  __shared__ float tab[SL_WAVES][SL_COLS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
  const int64_t c0 = (int64_t)blockIdx.x * SL_COLS;
  const int ntiles = (int)((a.M + 15) / 16);
  // B-side rows of the MFMA = output columns: lane n <-> column c0 + n (clamped: its products are never stored)
  const int64_t wc = c0 + n < a.N ? c0 + n : a.N - 1;
  const float* wr = a.w + wc * a.K + g * 8;
  const float* xr[SL_MT];
  bool live[SL_MT];
#pragma unroll
  for (int t = 0; t < SL_MT; ++t) {
    const int tile = wave + t * SL_WAVES;
    live[t] = tile < ntiles;
    const int64_t m = (int64_t)tile * 16 + n;
    xr[t] = a.x + (live[t] && m < a.M ? m : 0) * a.K + g * 8;
  }
  f32x4 acc[SL_MT];
#pragma unroll
  for (int t = 0; t < SL_MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float fw[8], fx[SL_MT][8];
  sl_load8(wr, a.K - g * 8, fw);
#pragma unroll
  for (int t = 0; t < SL_MT; ++t)
    if (live[t]) sl_load8(xr[t], a.K - g * 8, fx[t]);
  for (int64_t k0 = 0; k0 < a.K; k0 += 32) {
    const Frag<TC> aw = frag_from_f32<TC>(fw);
    Frag<TC> ax[SL_MT];
#pragma unroll
    for (int t = 0; t < SL_MT; ++t)
      if (live[t]) ax[t] = frag_from_f32<TC>(fx[t]);
    if (k0 + 32 < a.K) {   // the next step's operands fly during this step's MFMAs
      const int64_t rem = a.K - (k0 + 32) - g * 8;
      sl_load8(wr + k0 + 32, rem, fw);
#pragma unroll
      for (int t = 0; t < SL_MT; ++t)
        if (live[t]) sl_load8(xr[t] + k0 + 32, rem, fx[t]);
    }
#pragma unroll
    for (int t = 0; t < SL_MT; ++t)
      if (live[t]) acc[t] = mma(aw, ax[t], acc[t]);
  }
  // acc[t][r] = z[row (wave + 8 t) * 16 + n][column c0 + g * 4 + r]
  const int64_t col = c0 + g * 4;
  const bool cok = col < a.N;   // (N % 4 == 0: a lane's four columns exist together)
  float4 bs = gt_zero4();
  if (a.bias && cok) bs = *reinterpret_cast<const float4*>(a.bias + col);
  bool rok[SL_MT];
#pragma unroll
  for (int t = 0; t < SL_MT; ++t) {
    const int64_t row = (int64_t)(wave + t * SL_WAVES) * 16 + n;
    rok[t] = live[t] && row < a.M;
    acc[t][0] += bs.x; acc[t][1] += bs.y; acc[t][2] += bs.z; acc[t][3] += bs.w;
    if (rok[t] && cok && a.z) *reinterpret_cast<float4*>(a.z + row * a.N + col) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
  }
  // ---- batch statistics of the slab's columns: mean, then the centred second moment (two passes over registers)
  const float inv_m = 1.0f / (float)a.M;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < SL_MT; ++t)
    if (rok[t]) {
#pragma unroll
      for (int r = 0; r < 4; ++r) s[r] += acc[t][r];
    }
#pragma unroll
  for (int r = 0; r < 4; ++r) s[r] = sl_rows16(s[r]);
  sl_block_cols(s, tab, wave, n, g);
  float mu[4], q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 4; ++r) mu[r] = s[r] * inv_m;
#pragma unroll
  for (int t = 0; t < SL_MT; ++t)
    if (rok[t]) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = acc[t][r] - mu[r]; q[r] = fmaf(d, d, q[r]); }
    }
#pragma unroll
  for (int r = 0; r < 4; ++r) q[r] = sl_rows16(q[r]);
  sl_block_cols(q, tab, wave, n, g);
  float rs[4], ww[4] = {0.f, 0.f, 0.f, 0.f}, bb[4] = {0.f, 0.f, 0.f, 0.f};
  if (cok) {
    const float4 w4 = *reinterpret_cast<const float4*>(a.bn_w + col), b4 = *reinterpret_cast<const float4*>(a.bn_b + col);
    ww[0] = w4.x; ww[1] = w4.y; ww[2] = w4.z; ww[3] = w4.w;
    bb[0] = b4.x; bb[1] = b4.y; bb[2] = b4.z; bb[3] = b4.w;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) rs[r] = 1.0f / sqrtf(q[r] * inv_m + a.eps);
  if (wave == 0 && n == 0 && cok) {
    *reinterpret_cast<float4*>(a.mean + col) = make_float4(mu[0], mu[1], mu[2], mu[3]);
    *reinterpret_cast<float4*>(a.rstd + col) = make_float4(rs[0], rs[1], rs[2], rs[3]);
    if (a.rmean) {
      const float ub = a.M > 1 ? (float)a.M / (float)(a.M - 1) : 1.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a.rmean[col + r] = (1.f - a.momentum) * a.rmean[col + r] + a.momentum * mu[r];
        a.rvar[col + r] = (1.f - a.momentum) * a.rvar[col + r] + a.momentum * (q[r] * inv_m * ub);
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.nbt) a.nbt[0] += 1;
  // ---- apply
#pragma unroll
  for (int t = 0; t < SL_MT; ++t) {
    if (!rok[t] || !cok) continue;
    const int64_t row = (int64_t)(wave + t * SL_WAVES) * 16 + n;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = (acc[t][r] - mu[r]) * rs[r] * ww[r] + bb[r];
      if (a.relu) v[r] = fmaxf(v[r], 0.f);
      if (a.drop.thr) v[r] = bn_hash(a.drop.s0, a.drop.s1, (uint32_t)row, (uint32_t)(col + r)) >= a.drop.thr ? v[r] * a.drop.inv_keep : 0.f;
    }
    if (a.resid) {
      const float4 rr = *reinterpret_cast<const float4*>(a.resid + row * a.N + col);
      v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
    }
    *reinterpret_cast<float4*>(a.y + row * a.N + col) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// da[m][k] = sum_n dz_up[m][n] W_up[n][k] for the slab's 16 columns k, then the BatchNorm-backward of the layer below on them
template <typename TC>
__global__ void __launch_bounds__(SL_THREADS) k_slab_dx_bn_bwd(SlabArgs a) {
  extern __shared__ __attribute__((aligned(16))) float s_wt[];   // [16 slab columns][NP]: W_up[:, slab] transposed (NP = N rounded to 32, + 4)
  __shared__ float tab[SL_WAVES][SL_COLS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
  const int64_t c0 = (int64_t)blockIdx.x * SL_COLS;   // first of this block's columns of K
  const int ntiles = (int)((a.M + 15) / 16);
  const int NP = (int)((a.N + 31) / 32 * 32) + 4;
  for (int i = threadIdx.x; i < SL_COLS * NP; i += SL_THREADS) s_wt[i] = 0.f;
  __syncthreads();
  for (int64_t i = threadIdx.x; i < a.N * 4; i += SL_THREADS) {   // row i / 4 of W_up, 16-byte chunk i % 4 of the slab's 64 bytes
    const int64_t row = i >> 2;
    const int ch = (int)(i & 3);
    if (c0 + ch * 4 < a.K) {
      const float4 v = *reinterpret_cast<const float4*>(a.w + row * a.K + c0 + ch * 4);
      s_wt[(ch * 4 + 0) * NP + row] = v.x; s_wt[(ch * 4 + 1) * NP + row] = v.y;
      s_wt[(ch * 4 + 2) * NP + row] = v.z; s_wt[(ch * 4 + 3) * NP + row] = v.w;
    }
  }
  __syncthreads();
  const float* dr[SL_MT];
  bool live[SL_MT];
#pragma unroll
  for (int t = 0; t < SL_MT; ++t) {
    const int tile = wave + t * SL_WAVES;
    live[t] = tile < ntiles;
    const int64_t m = (int64_t)tile * 16 + n;
    dr[t] = a.x + (live[t] && m < a.M ? m : 0) * a.N + g * 8;
  }
  f32x4 acc[SL_MT];
#pragma unroll
  for (int t = 0; t < SL_MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float fd[SL_MT][8];
#pragma unroll
  for (int t = 0; t < SL_MT; ++t)
    if (live[t]) sl_load8(dr[t], a.N - g * 8, fd[t]);
  for (int64_t n0 = 0; n0 < a.N; n0 += 32) {
    Frag<TC> ad[SL_MT];
#pragma unroll
    for (int t = 0; t < SL_MT; ++t)
      if (live[t]) ad[t] = frag_from_f32<TC>(fd[t]);
    if (n0 + 32 < a.N) {
      const int64_t rem = a.N - (n0 + 32) - g * 8;
#pragma unroll
      for (int t = 0; t < SL_MT; ++t)
        if (live[t]) sl_load8(dr[t] + n0 + 32, rem, fd[t]);
    }
    float fw[8];   // lane n <-> slab column n: 8 consecutive contraction slots from the transposed slab (zero beyond N)
    const float* ws = s_wt + n * NP + n0 + g * 8;
    const float4 w0 = *reinterpret_cast<const float4*>(ws), w1 = *reinterpret_cast<const float4*>(ws + 4);
    fw[0] = w0.x; fw[1] = w0.y; fw[2] = w0.z; fw[3] = w0.w; fw[4] = w1.x; fw[5] = w1.y; fw[6] = w1.z; fw[7] = w1.w;
    const Frag<TC> aw = frag_from_f32<TC>(fw);
#pragma unroll
    for (int t = 0; t < SL_MT; ++t)
      if (live[t]) acc[t] = mma(aw, ad[t], acc[t]);
  }
  // acc[t][r] = da[row (wave + 8 t) * 16 + n][column c0 + g * 4 + r]
  const int64_t col = c0 + g * 4;
  const bool cok = col < a.K;
  float mu[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {0.f, 0.f, 0.f, 0.f}, ww[4] = {0.f, 0.f, 0.f, 0.f}, bb[4] = {0.f, 0.f, 0.f, 0.f};
  if (cok) {
    const float4 m4 = *reinterpret_cast<const float4*>(a.mean + col), r4 = *reinterpret_cast<const float4*>(a.rstd + col);
    const float4 w4 = *reinterpret_cast<const float4*>(a.bn_w + col), b4 = *reinterpret_cast<const float4*>(a.bn_b + col);
    mu[0] = m4.x; mu[1] = m4.y; mu[2] = m4.z; mu[3] = m4.w; rs[0] = r4.x; rs[1] = r4.y; rs[2] = r4.z; rs[3] = r4.w;
    ww[0] = w4.x; ww[1] = w4.y; ww[2] = w4.z; ww[3] = w4.w; bb[0] = b4.x; bb[1] = b4.y; bb[2] = b4.z; bb[3] = b4.w;
  }
  float xh[SL_MT][4];
  bool rok[SL_MT];
  float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < SL_MT; ++t) {
    const int64_t row = (int64_t)(wave + t * SL_WAVES) * 16 + n;
    rok[t] = live[t] && row < a.M && cok;
    if (!rok[t]) continue;
    const float4 z4 = *reinterpret_cast<const float4*>(a.zin + row * a.K + col);
    const float zz[4] = {z4.x, z4.y, z4.z, z4.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xh[t][r] = (zz[r] - mu[r]) * rs[r];
      float gq = acc[t][r];
      if (a.relu) gq = xh[t][r] * ww[r] + bb[r] > 0.f ? gq : 0.f;
      acc[t][r] = gq;
      s0[r] += gq;
      s1[r] = fmaf(gq, xh[t][r], s1[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) { s0[r] = sl_rows16(s0[r]); s1[r] = sl_rows16(s1[r]); }
  sl_block_cols(s0, tab, wave, n, g);
  sl_block_cols(s1, tab, wave, n, g);
  if (wave == 0 && n == 0 && cok) {
    *reinterpret_cast<float4*>(a.dbeta + col) = make_float4(s0[0], s0[1], s0[2], s0[3]);
    *reinterpret_cast<float4*>(a.dgamma + col) = make_float4(s1[0], s1[1], s1[2], s1[3]);
  }
  const float inv_m = 1.0f / (float)a.M;
#pragma unroll
  for (int t = 0; t < SL_MT; ++t) {
    if (!rok[t]) continue;
    const int64_t row = (int64_t)(wave + t * SL_WAVES) * 16 + n;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = ww[r] * rs[r] * (acc[t][r] - s0[r] * inv_m - xh[t][r] * s1[r] * inv_m);
    *reinterpret_cast<float4*>(a.y + row * a.K + col) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// (N <= 960: the backward's transposed 16-column slab of W_up, 16 x (N + pad) floats, stays below 64 KB of LDS)
static inline bool slab_shape_ok(int64_t M, int64_t N, int64_t K) {
  return M >= 2 && M <= SL_MAX_ROWS && N >= 4 && K >= 4 && N % 4 == 0 && K % 4 == 0 && N <= 960 && K <= 4096;
}

}  // namespace slab
