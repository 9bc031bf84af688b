// linear_small.h — nn.Linear forward / backward for SHORT row counts (M <= 512: the virtual-node MLPs work on one row per
// graph, modules/gnn_module.py:161-170), fp32 storage, exact-fp32 or bf16 MFMA.  Included by linear.hip.
//
// The tiled kernels give such a GEMM a handful of 64 x 128 blocks that each walk the whole contraction behind a barrier
// per k-step: 16-32 us of latency for 0.1 GFLOP.  Here the problem is cut the other way: ONE WAVE PER 16 x 32 OUTPUT TILE
// PAIR, no LDS, no barriers, no split partials -- a wave streams its operand rows straight from L2 (everything is a few
// hundred KB) in MFMA fragment shape and finishes in ~2k cycles of matrix-pipe time; a 256 x 600 x 300 layer is ~300
// independent waves spread over the chip.  dW writes its final tile directly (the contraction is only M long): no
// partial buffers and no reduce launch.
//   fwd : Y[M][N]  = act(X W^T + b) (* dropout)        k_small_fwd
//   dX  : dX[M][K] = dZ W (+ addends), dZ = dY 1[Y>0]   k_small_dx
//   dW  : dW[N][K] = dZ^T X, db[N] = colsum(dZ)         k_small_dw
// Fragment convention of mfma_frag.h: a 32-deep step, lane (n = lane & 15, g = lane >> 4) owns slots g*8 .. g*8+7.
#pragma once

struct SmallArgs {
  const float* x;      // [M][ldx]
  const float* w;      // [N][K]
  const float* bias;   // [N] or null
  const float* dy;     // [M][ldy]
  const float* ymask;  // [M][ldy] or null
  const float* add1;   // [M][ldx] or null (dX addends)
  const float* add2;
  float* out;          // fwd: Y [M][ldy]; dx: dX [M][ldx]; dw: dW [N][K]
  float* gout;         // fwd, act == 2 (gelu): [M][ldy] multiplier for the backward, or null
  float* db;           // dw: [N] or null
  int64_t M, N, K, ldx, ldy;
  int act;
  float inv_keep;
  uint32_t thr, s0, s1;
};

// 8 consecutive floats p[0..7] of a row whose valid length from p is `rem` (a multiple of 4); zero beyond
__device__ __forceinline__ void load8(const float* p, int64_t rem, float* f) {
  const float4 a = rem > 0 ? *reinterpret_cast<const float4*>(p) : gt_zero4();
  const float4 b = rem > 4 ? *reinterpret_cast<const float4*>(p + 4) : gt_zero4();
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w;
  f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

// Split-K (KS = 4): the four waves of a block share ONE output tile pair and take every fourth 32-deep step of the contraction;
// their accumulators meet in LDS and wave 0 runs the epilogue.  The chain of dependent loads is what these launches cost
// (a 256 x 600 x 300 layer is 10-19 steps of ~1 us): four times shorter with KS = 4, used when the tile count alone cannot fill
// the chip; KS = 1 (a wave per tile pair, no LDS) for wide outputs such as the prediction heads.
template <int KS>
__device__ __forceinline__ bool small_combine(f32x4 (&acc)[2], float* sred /* [3][2][64][4] */) {
  if constexpr (KS == 1) return true;
  const int lane = threadIdx.x & 63, wq = threadIdx.x >> 6;
  if (wq > 0) {
    *reinterpret_cast<float4*>(sred + (((wq - 1) * 2 + 0) * 64 + lane) * 4) = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
    *reinterpret_cast<float4*>(sred + (((wq - 1) * 2 + 1) * 64 + lane) * 4) = make_float4(acc[1][0], acc[1][1], acc[1][2], acc[1][3]);
  }
  __syncthreads();
  if (wq > 0) return false;
#pragma unroll
  for (int w = 0; w < 3; ++w)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float4 t = *reinterpret_cast<const float4*>(sred + ((w * 2 + q) * 64 + lane) * 4);
      acc[q][0] += t.x; acc[q][1] += t.y; acc[q][2] += t.z; acc[q][3] += t.w;
    }
  return true;
}

// ---- forward: wave = 16 rows x 32 columns ---------------------------------------------------------------------------
template <typename TC, int KS>
__global__ void __launch_bounds__(256) k_small_fwd(SmallArgs a) {
  __shared__ __attribute__((aligned(16))) float sred[KS == 1 ? 4 : 3 * 2 * 64 * 4];
  const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
  const int wq = threadIdx.x >> 6;
  const int64_t wave = KS == 1 ? (int64_t)blockIdx.x * 4 + wq : (int64_t)blockIdx.x;
  const int64_t ntp = (a.N + 31) / 32;
  // KS == 1 (wide outputs: the prediction heads, 256 x 25 010 x 128): the row tiles of one column pair are NEIGHBOURS -- the four
  // waves of a block read the same 32 weight rows (L1 hits) instead of re-reading them from L2 once per row tile
  const int64_t mtp = (a.M + 15) / 16;
  const int64_t mt = KS == 1 ? wave % mtp : wave / ntp, np = KS == 1 ? wave / mtp : wave % ntp;
  const bool live = mt * 16 < a.M && np < ntp;
  if (KS == 1 && !live) return;
  const int64_t mrow = live ? mt * 16 : 0;
  const int64_t m = mrow + n < a.M ? mrow + n : a.M - 1;          // clamped rows are computed and never stored
  const float* xr = a.x + m * a.ldx + g * 8;
  const float* wr[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int64_t r = np * 32 + q * 16 + n;
    wr[q] = a.w + (r < a.N ? r : a.N - 1) * a.K + g * 8;
  }
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  float fx[8], fw[2][8];
  const int64_t kbeg = KS == 1 ? 0 : (int64_t)wq * 32, kstride = 32 * KS;
  if (kbeg < a.K) {
    load8(xr + kbeg, a.K - kbeg - g * 8, fx);
#pragma unroll
    for (int q = 0; q < 2; ++q) load8(wr[q] + kbeg, a.K - kbeg - g * 8, fw[q]);
  }
  for (int64_t k0 = kbeg; k0 < a.K; k0 += kstride) {
    const Frag<TC> ax = frag_from_f32<TC>(fx);
    const Frag<TC> aw0 = frag_from_f32<TC>(fw[0]), aw1 = frag_from_f32<TC>(fw[1]);
    if (k0 + kstride < a.K) {   // next step's operands fly during this step's MFMAs
      const int64_t rem = a.K - (k0 + kstride) - g * 8;
      load8(xr + k0 + kstride, rem, fx);
#pragma unroll
      for (int q = 0; q < 2; ++q) load8(wr[q] + k0 + kstride, rem, fw[q]);
    }
    acc[0] = mma(aw0, ax, acc[0]);
    acc[1] = mma(aw1, ax, acc[1]);
  }
  if (!small_combine<KS>(acc, sred)) return;
  // acc[q][r] = C[column np*32 + q*16 + g*4 + r][row mt*16 + n]
  const int64_t row = mrow + n;
  if (!live || row >= a.M) return;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int64_t col = np * 32 + q * 16 + g * 4;
    if (col >= a.N) continue;
    float4 v = make_float4(acc[q][0], acc[q][1], acc[q][2], acc[q][3]);
    if (a.bias) v = gt_add4(v, bias_chunk(a.bias, col, a.N));
    float4 gm = gt_zero4();
    float* vv = reinterpret_cast<float*>(&v);
    float* gg = reinterpret_cast<float*>(&gm);
    if (a.act == 1) v = gt_relu4(v);
    else if (a.act == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) gt_gelu(vv[e], vv[e], gg[e]);
    }
    if (a.thr) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool keep = lin_hash(a.s0, a.s1, (uint32_t)row, (uint32_t)(col + e)) >= a.thr;
        vv[e] = keep ? vv[e] * a.inv_keep : 0.f;
        gg[e] = keep ? gg[e] * a.inv_keep : 0.f;
      }
    }
    *reinterpret_cast<float4*>(a.out + row * a.ldy + col) = v;
    if (a.act == 2 && a.gout) *reinterpret_cast<float4*>(a.gout + row * a.ldy + col) = gm;
  }
}

// ---- dX[M][K] = dZ W: wave = 16 rows x 32 output columns (k); W is read transposed (8 strided dwords per step) ----------
template <typename TC, int KS>
__global__ void __launch_bounds__(256) k_small_dx(SmallArgs a) {
  __shared__ __attribute__((aligned(16))) float sred[KS == 1 ? 4 : 3 * 2 * 64 * 4];
  const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
  const int wq = threadIdx.x >> 6;
  const int64_t wave = KS == 1 ? (int64_t)blockIdx.x * 4 + wq : (int64_t)blockIdx.x;
  const int64_t ktp = (a.K + 31) / 32;
  const int64_t mt = wave / ktp, kp = wave % ktp;
  const bool live = mt * 16 < a.M;
  if (KS == 1 && !live) return;
  const int64_t mrow = live ? mt * 16 : 0;
  const int64_t m = mrow + n < a.M ? mrow + n : a.M - 1;
  const float* zr = a.dy + m * a.ldy + g * 8;
  const float* yr = a.ymask ? a.ymask + m * a.ldy + g * 8 : nullptr;
  int64_t kc[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int64_t c = kp * 32 + q * 16 + n;
    kc[q] = c < a.K ? c : a.K - 1;
  }
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  for (int64_t n0 = KS == 1 ? 0 : (int64_t)wq * 32; n0 < a.N; n0 += 32 * KS) {
    float fz[8], fy[8], fw[2][8];
    const int64_t rem = a.N - n0 - g * 8;
    load8(zr + n0, rem, fz);
    if (yr) {
      load8(yr + n0, rem, fy);
#pragma unroll
      for (int e = 0; e < 8; ++e) fz[e] = gt_gate(fz[e], fy[e], a.inv_keep);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int64_t nn = n0 + g * 8 + e;
        fw[q][e] = nn < a.N ? a.w[nn * a.K + kc[q]] : 0.f;
      }
    const Frag<TC> az = frag_from_f32<TC>(fz);
    acc[0] = mma(frag_from_f32<TC>(fw[0]), az, acc[0]);
    acc[1] = mma(frag_from_f32<TC>(fw[1]), az, acc[1]);
  }
  if (!small_combine<KS>(acc, sred)) return;
  const int64_t row = mrow + n;
  if (!live || row >= a.M) return;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int64_t col = kp * 32 + q * 16 + g * 4;
    if (col >= a.K) continue;
    float4 v = make_float4(acc[q][0], acc[q][1], acc[q][2], acc[q][3]);
    if (a.add1) v = gt_add4(v, *reinterpret_cast<const float4*>(a.add1 + row * a.ldx + col));
    if (a.add2) v = gt_add4(v, *reinterpret_cast<const float4*>(a.add2 + row * a.ldx + col));
    *reinterpret_cast<float4*>(a.out + row * a.ldx + col) = v;
  }
}

// ---- dW[N][K] = dZ^T X, db = colsum(dZ): wave = 32 rows (n) x 16 columns (k), the whole contraction (M <= 512) ----------
// C is produced as [k][n] (MFMA rows = k from X, columns = n from dZ): a lane ends up with 4 consecutive k of one n.
template <typename TC, int KS>
__global__ void __launch_bounds__(256) k_small_dw(SmallArgs a) {
  __shared__ __attribute__((aligned(16))) float sred[KS == 1 ? 4 : 3 * 2 * 64 * 4];
  __shared__ float sdb[KS == 1 ? 1 : 3 * 2 * 64];
  const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
  const int wq = threadIdx.x >> 6;
  const int64_t wave = KS == 1 ? (int64_t)blockIdx.x * 4 + wq : (int64_t)blockIdx.x;
  const int64_t kt = (a.K + 15) / 16;
  const int64_t np = wave / kt, kti = wave % kt;
  const bool live = np * 32 < a.N;
  if (KS == 1 && !live) return;
  const int64_t nrow = live ? np * 32 : 0;
  const int64_t kcol = kti * 16 + n < a.K ? kti * 16 + n : a.K - 1;
  int64_t nc[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int64_t c = nrow + q * 16 + n;
    nc[q] = c < a.N ? c : a.N - 1;
  }
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  float dbs[2] = {0.f, 0.f};
  for (int64_t m0 = KS == 1 ? 0 : (int64_t)wq * 32; m0 < a.M; m0 += 32 * KS) {
    float fx[8], fz[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int64_t mm = m0 + g * 8 + e;
      const bool ok = mm < a.M;
      fx[e] = ok ? a.x[mm * a.ldx + kcol] : 0.f;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float z = ok ? a.dy[mm * a.ldy + nc[q]] : 0.f;
        if (a.ymask && ok) z = gt_gate(z, a.ymask[mm * a.ldy + nc[q]], a.inv_keep);
        fz[q][e] = z;
        dbs[q] += z;
      }
    }
    const Frag<TC> ax = frag_from_f32<TC>(fx);
    acc[0] = mma(ax, frag_from_f32<TC>(fz[0]), acc[0]);
    acc[1] = mma(ax, frag_from_f32<TC>(fz[1]), acc[1]);
  }
  if constexpr (KS > 1) {   // the waves' column sums of dZ travel beside their accumulators
    if (wq > 0) { sdb[((wq - 1) * 2 + 0) * 64 + lane] = dbs[0]; sdb[((wq - 1) * 2 + 1) * 64 + lane] = dbs[1]; }
  }
  if (!small_combine<KS>(acc, sred)) return;
  if constexpr (KS > 1) {
#pragma unroll
    for (int w = 0; w < 3; ++w) { dbs[0] += sdb[(w * 2 + 0) * 64 + lane]; dbs[1] += sdb[(w * 2 + 1) * 64 + lane]; }
  }
  if (!live) return;
  // acc[q][r] = C[k = kti*16 + g*4 + r][n = np*32 + q*16 + n]
  const int64_t kc = kti * 16 + g * 4;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int64_t row = nrow + q * 16 + n;
    if (row < a.N && kc < a.K) *reinterpret_cast<float4*>(a.out + row * a.K + kc) = make_float4(acc[q][0], acc[q][1], acc[q][2], acc[q][3]);
    if (a.db && kti == 0) {   // the four slot groups of a column hold disjoint m: fold them (fixed order)
      float v = dbs[q];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (g == 0 && row < a.N) a.db[row] = v;
    }
  }
}
