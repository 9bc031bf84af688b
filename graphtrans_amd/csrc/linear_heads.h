// linear_heads.h — the prediction heads: a short-M GEMM with a VERY wide output.  Included by linear.hip (anonymous namespace).
//
// Reference: models/gnn_transformer.py:120-126 -- max_seq_len Linear(128, 5002) heads on the pooled rows: here ONE
// (B, 128) x (128, 25 010) GEMM over the stacked weights (256 rows, 1.64 GFLOP per direction, 12.8 MB of weights, 25.6 MB of logits).
//
// The one-wave-per-tile kernels of linear_small.h give every wave a 16 x 32 output tile: each pair of MFMA tiles re-fetches its
// operand fragments (41 us forward), and the dX form -- whose contraction is the 25 010 columns -- stayed on the tiled split-N
// kernel of round 1 (75 us, the first kernel of every backward: nothing to overlap it with).  Here:
//   forward : block = ALL rows (256) x 128 columns, 8 waves of 64 x 64 (16 accumulator tiles each): a wave's 8 operand fragments
//             per 32-deep step feed 16 tile products, straight from global / L2 (X is 128 KB, shared by every block); no LDS.
//   dX      : dX[M][128] = dZ[M][N] W[N][128]: block = a range of 32-deep steps of N (3-4 of 782), all rows x all 128 output columns;
//             W's 32 x 128 slab of a step goes through LDS (read transposed), dZ fragments straight from global; fp32 partials
//             [block][M][128], one fixed-order reduce launch (+ the addends) -> bitwise reproducible.
// Exact fp32 (eight v_mfma_f32_16x16x4_f32 per step) or bf16 operands, as the kernels they replace.
#pragma once

struct HeadsArgs {
  const float* x;      // fwd: X [M][ldx]
  const float* w;      // [N][128]
  const float* bias;   // fwd: [N] or null
  const float* dy;     // dx: dZ [M][ldy]
  float* out;          // fwd: Y [M][ldy]; dx: partials [nblk][Mp][128]
  int64_t M, N, ldx, ldy;
  int nblk, steps;     // dx: blocks along N, 32-deep steps of N
};
constexpr int HD_K = 128;          // d_model of the heads' input
constexpr int HD_LDW = HD_K + 4;   // LDS row pitch of a W slab (2-way conflicts on the transposed reads: 32 reads against 128 MFMAs)

static inline bool heads_shape_ok(int x_dtype, int y_dtype, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldy, int groups) {
  return groups == 1 && x_dtype == GT_F32 && y_dtype == GT_F32 && K == HD_K && N >= 4096 && M > 0 && M <= 4096 && ldx % 4 == 0 &&
         ldy % 4 == 0;
}
static inline int heads_dx_blocks(int64_t N) {
  const int steps = (int)gt_cdiv(N, 32);
  return steps < 256 ? steps : 256;
}
static inline size_t heads_dx_workspace_bytes(int64_t M, int64_t N) {
  return (size_t)heads_dx_blocks(N) * (size_t)(gt_cdiv(M, 256) * 256) * HD_K * sizeof(float);
}

// 4 x 4 tile products of one 32-deep step.  fp32: slot by slot over the 16 INDEPENDENT accumulators (eight chained
// v_mfma_f32_16x16x4_f32 on one accumulator wait for each other: measured 42 us against ~41 us of the one-wave-per-tile kernel)
__device__ __forceinline__ void heads_mma(const Frag<float> (&fa)[4], const Frag<float> (&fb)[4], f32x4 (&acc)[4][4]) {
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j].v[e], fb[i].v[e], acc[i][j], 0, 0, 0);
}
__device__ __forceinline__ void heads_mma(const Frag<gt_bf16> (&fa)[4], const Frag<gt_bf16> (&fb)[4], f32x4 (&acc)[4][4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i][j] = mma(fa[j], fb[i], acc[i][j]);
}

template <typename TC>
__global__ void __launch_bounds__(512) k_heads_fwd(HeadsArgs a) {
  const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6, wm = wv >> 1, wk = wv & 1;
  const int64_t m0 = (int64_t)blockIdx.y * 256 + wm * 64, c0 = (int64_t)blockIdx.x * 128 + wk * 64;
  if (m0 >= a.M || c0 >= a.N) return;   // (wave-uniform; this kernel has no barrier)
  const float* xr[4];
  const float* wr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t row = m0 + i * 16 + n, col = c0 + i * 16 + n;
    xr[i] = a.x + (row < a.M ? row : a.M - 1) * a.ldx + g * 8;     // clamped rows / columns are computed and never stored
    wr[i] = a.w + (col < a.N ? col : a.N - 1) * HD_K + g * 8;
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < HD_K / 32; ++s) {
    Frag<TC> fx[4], fw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float t[8];
      *reinterpret_cast<float4*>(t) = *reinterpret_cast<const float4*>(xr[i] + s * 32);
      *reinterpret_cast<float4*>(t + 4) = *reinterpret_cast<const float4*>(xr[i] + s * 32 + 4);
      fx[i] = frag_from_f32<TC>(t);
      *reinterpret_cast<float4*>(t) = *reinterpret_cast<const float4*>(wr[i] + s * 32);
      *reinterpret_cast<float4*>(t + 4) = *reinterpret_cast<const float4*>(wr[i] + s * 32 + 4);
      fw[i] = frag_from_f32<TC>(t);
    }
    heads_mma(fw, fx, acc);   // c[r] = Y[m0 + i*16 + n][c0 + j*16 + g*4 + r]
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t row = m0 + i * 16 + n;
    if (row >= a.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t col = c0 + j * 16 + g * 4;
      if (col >= a.N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      float* o = a.out + row * a.ldy + col;
      if (col + 3 < a.N) {
        if (a.bias) {
          const float4 b = *reinterpret_cast<const float4*>(a.bias + col);
          v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        for (int e = 0; e < 4 && col + e < a.N; ++e) o[e] = v[e] + (a.bias ? a.bias[col + e] : 0.f);
      }
    }
  }
}

template <typename TC>
__global__ void __launch_bounds__(512) k_heads_dx(HeadsArgs a) {
  __shared__ __attribute__((aligned(16))) float sw[2][32 * HD_LDW];
  const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6, wm = wv >> 1, wk = wv & 1;
  const int b = blockIdx.x;
  const int s_lo = (int)((int64_t)b * a.steps / a.nblk), s_hi = (int)((int64_t)(b + 1) * a.steps / a.nblk);
  const int64_t m0 = (int64_t)blockIdx.y * 256 + wm * 64;
  const float* zr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t row = m0 + i * 16 + n;
    zr[i] = a.dy + (row < a.M ? row : a.M - 1) * a.ldy + g * 8;
  }
  // staging of a W slab (32 rows of N x 128): thread -> 2 x 16 bytes, p = tid + q * 512 -> row p / 32, 16-byte column p % 32
  const int sr0 = threadIdx.x >> 5, sr1 = sr0 + 16, sc = (threadIdx.x & 31) * 4;
  float4 w0, w1;
  {
    const int64_t r0 = (int64_t)s_lo * 32 + sr0, r1 = (int64_t)s_lo * 32 + sr1;
    w0 = *reinterpret_cast<const float4*>(a.w + (r0 < a.N ? r0 : a.N - 1) * HD_K + sc);
    w1 = *reinterpret_cast<const float4*>(a.w + (r1 < a.N ? r1 : a.N - 1) * HD_K + sc);
    if (r0 >= a.N) w0 = gt_zero4();
    if (r1 >= a.N) w1 = gt_zero4();
  }
  *reinterpret_cast<float4*>(&sw[0][sr0 * HD_LDW + sc]) = w0;
  *reinterpret_cast<float4*>(&sw[0][sr1 * HD_LDW + sc]) = w1;
  __syncthreads();
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int s = s_lo; s < s_hi; ++s) {
    const int cur = (s - s_lo) & 1;
    {   // the next slab (clamped: the last iteration reloads its own) while this one is multiplied
      const int sn = s + 1 < s_hi ? s + 1 : s;
      const int64_t r0 = (int64_t)sn * 32 + sr0, r1 = (int64_t)sn * 32 + sr1;
      w0 = *reinterpret_cast<const float4*>(a.w + (r0 < a.N ? r0 : a.N - 1) * HD_K + sc);
      w1 = *reinterpret_cast<const float4*>(a.w + (r1 < a.N ? r1 : a.N - 1) * HD_K + sc);
      if (r0 >= a.N) w0 = gt_zero4();
      if (r1 >= a.N) w1 = gt_zero4();
    }
    Frag<TC> fz[4], fw[4];
    const int64_t rem = a.N - (int64_t)s * 32 - g * 8;   // valid length of this lane's 8 columns (a multiple of 2: zero beyond)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float t[8];
      const float* p = zr[i] + (int64_t)s * 32;
      const float4 u0 = rem > 0 ? *reinterpret_cast<const float4*>(p) : gt_zero4();
      const float4 u1 = rem > 4 ? *reinterpret_cast<const float4*>(p + 4) : gt_zero4();
      t[0] = u0.x; t[1] = rem > 1 ? u0.y : 0.f; t[2] = rem > 2 ? u0.z : 0.f; t[3] = rem > 3 ? u0.w : 0.f;
      t[4] = u1.x; t[5] = rem > 5 ? u1.y : 0.f; t[6] = rem > 6 ? u1.z : 0.f; t[7] = rem > 7 ? u1.w : 0.f;
      fz[i] = frag_from_f32<TC>(t);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = sw[cur][(g * 8 + e) * HD_LDW + wk * 64 + j * 16 + n];
      fw[j] = frag_from_f32<TC>(t);
    }
    heads_mma(fw, fz, acc);   // c[r] = dX[m0 + i*16 + n][wk*64 + j*16 + g*4 + r]
    *reinterpret_cast<float4*>(&sw[cur ^ 1][sr0 * HD_LDW + sc]) = w0;
    *reinterpret_cast<float4*>(&sw[cur ^ 1][sr1 * HD_LDW + sc]) = w1;
    __syncthreads();
  }
  const int64_t Mp = (int64_t)gridDim.y * 256;
  float* part = a.out + ((int64_t)b * Mp + m0) * HD_K;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (m0 + i * 16 + n >= a.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<float4*>(part + (int64_t)(i * 16 + n) * HD_K + wk * 64 + j * 16 + g * 4) =
          make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
  }
}

// dX[m][0..128) = sum over the N-range blocks, in block order (+ addends): one block per row, 32 x 16-byte columns x 8 block lanes
__global__ void __launch_bounds__(256) k_heads_dx_reduce(const float* __restrict__ part, int nblk, int64_t Mp, int64_t ldx,
                                                         const float* __restrict__ add1, const float* __restrict__ add2,
                                                         float* __restrict__ out) {
  __shared__ float4 sm[8][32];
  const int c4 = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int64_t m = blockIdx.x;
  float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int b0 = pl; b0 < nblk; b0 += 32) {
    float4 u[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int b = b0 + q * 8;
      u[q] = b < nblk ? *reinterpret_cast<const float4*>(part + ((int64_t)b * Mp + m) * HD_K + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) t = gt_add4(t, u[q]);
  }
  sm[pl][c4] = t;
  __syncthreads();
  if (pl) return;
#pragma unroll
  for (int q = 1; q < 8; ++q) t = gt_add4(t, sm[q][c4]);
  if (add1) t = gt_add4(t, *reinterpret_cast<const float4*>(add1 + m * ldx + c4 * 4));
  if (add2) t = gt_add4(t, *reinterpret_cast<const float4*>(add2 + m * ldx + c4 * 4));
  *reinterpret_cast<float4*>(out + m * ldx + c4 * 4) = t;
}

static inline void heads_launch_fwd(int compute, hipStream_t stream, const HeadsArgs& a) {
  const dim3 grid((unsigned)gt_cdiv(a.N, 128), (unsigned)gt_cdiv(a.M, 256));
  if (compute == GT_F32) hipLaunchKernelGGL(k_heads_fwd<float>, grid, dim3(512), 0, stream, a);
  else hipLaunchKernelGGL(k_heads_fwd<gt_bf16>, grid, dim3(512), 0, stream, a);
}
// dx[M][ldx] = dZ W (+ addends); workspace >= heads_dx_workspace_bytes(M, N)
static inline void heads_launch_dx(int compute, hipStream_t stream, HeadsArgs a, void* workspace, float* dx, const float* add1,
                                   const float* add2) {
  a.steps = (int)gt_cdiv(a.N, 32);
  a.nblk = heads_dx_blocks(a.N);
  a.out = (float*)workspace;
  const dim3 grid((unsigned)a.nblk, (unsigned)gt_cdiv(a.M, 256));
  if (compute == GT_F32) hipLaunchKernelGGL(k_heads_dx<float>, grid, dim3(512), 0, stream, a);
  else hipLaunchKernelGGL(k_heads_dx<gt_bf16>, grid, dim3(512), 0, stream, a);
  hipLaunchKernelGGL(k_heads_dx_reduce, dim3((unsigned)a.M), dim3(256), 0, stream, (const float*)workspace, a.nblk,
                     (int64_t)gt_cdiv(a.M, 256) * 256, a.ldx, add1, add2, dx);
}
