// linear32.h — exact-fp32 MFMA GEMMs (v_mfma_f32_16x16x4_f32) for the big-M linears of the message-passing side
// (included by linear.hip inside its anonymous namespace; shares LinArgs-independent helpers: lin_hash, bias_chunk).
//
// At 1/16 of the bf16 MFMA rate these GEMMs are MFMA-bound on every shape of this path (M ~ 3e4, K, N in 128..600:
// 75 flop/B against a ridge of 20), so the tile is chosen for the matrix pipe, not for HBM:
//   * a block owns 64 rows x ALL output columns (up to 19 n-tiles = 304 columns per column block): N = 300 costs 304
//     columns of MFMA work, not the 384 of 128-wide tiles (-21 % MFMAs), and the activation rows are read once;
//   * a wave owns 16 rows x NT n-tiles = NT independent accumulators: the 40-cycle dependent-accumulator latency of
//     the 32-cycle 16x16x4 MFMA never stalls the pipe, one wave per SIMD already saturates it;
//   * 16-deep k-stages (K = 300 -> 19 stages = 304, not 320), two LDS stages, one barrier per stage; 47 KB of LDS per
//     block (NT = 19): three blocks per CU take turns on the matrix pipe while the others stage / wait at their barrier
//     (with 24-float padded rows = 70 KB only ONE block fitted: 55 % MFMA-busy, PMC in profiles/r02*_pmc_lin32.txt).
// fwd:  Y[M][N]  = act(X W^T + b)                      k_lin32<TX, TY, NT, false>(X, W)
// dX :  dX[M][K] = dZ W  = dZ (W^T)^T (+ addends)       k_lin32<TY, TX, NT, true >(dY|mask, WT)   WT = k_transpose32(W)
// dW :  dW[N][K] = dZ^T X, db = colsum(dZ)             k_lin32_dw (split over M, fixed-order reduce by k_split_reduce)
#pragma once

constexpr int W32_BM = 64, W32_BK = 16, W32_LD = 16;   // unpadded rows: 2-way conflicts on the ds_read_b128 fragment reads (8 instead of 4 cycles, against 128 cycles of MFMA per read) buy 3 blocks per CU instead of 1

struct L32Args {
  const void* a;       // [M][lda]  fwd: X, dx: dY
  const void* amask;   // [M][lda]  dx: forward output Y (dZ = dY * (Y > 0) * inv_keep) or null
  const float* w;      // [Nout][ldw] fp32, contraction index contiguous (fwd: W[N][K]; dx: WT[K][N])
  const float* bias;   // [Nout] or null
  const void* add1;    // [M][ldo] storage type of out, or null
  const void* add2;
  void* out;           // [M][ldo]
  void* gout;          // fwd, act == 2 (gelu): [M][ldo] multiplier for the backward, or null
  // BNS (dX form): `out` is the dy of a BatchNorm further down the backward pass; its two column statistics are summed here
  //   bn_part[row tile][0][Nout] = sum_rows dy', [1][Nout] = sum_rows dy' * xhat,  dy' = dy * 1[xhat * w + b > 0] if bn_relu
  // (the row-tile partials k_bn_bwd_partial would produce from a second pass over dy and the BatchNorm input bn_x)
  const float* bn_x;   // [M][bn_ldx] the BatchNorm's input rows
  const float *bn_mean, *bn_rstd, *bn_w, *bn_b;   // [Nout]
  float* bn_part;
  int64_t bn_ldx;
  int bn_relu;
  int64_t M, Nout, Kc;
  int64_t lda, ldw, ldo;
  int act;
  float inv_keep;
  uint32_t thr, s0, s1;
  int ncb;             // column blocks of NT n-tiles
  const void* w3;      // linear3x.h: the bf16x3 image of the weight (k_lin3), w is unused then
  int w3_ntp;          // its 16-row tiles per plane and k-step
  int ntb;             // k_lin3r (linear3r.h): n-tiles per column block
  // k_lin3r dX form only -- a BROADCAST addend: out[m] += add_bc[add_bidx[m]] (the virtual-node update's gradient d_t0[graph of node m],
  // modules/gnn_module.py:219 backward, without materialising it per node): rows [.][ldo] fp32, int32 index per GEMM row
  const float* add_bc;
  const int32_t* add_bidx;
  // k_lin3 only -- a GROUPED launch (blockIdx.y = group, e.g. the towers of PNAConv, modules/pna/pna_module.py:116-133): group g reads
  // a / amask + g * g_a elements, the image w3 + g * g_img bytes, bias + g * g_b, and writes out (add1 / add2 alike) + g * g_o elements
  int groups;          // 0 / 1 = one GEMM
  int g_b;
  int64_t g_a, g_o, g_img;
  // k_lin3 only -- the JK = "cat" concatenation without a copy (torch.cat([h_list[0], h_list[-1]], 1), modules/gnn_module.py:104-105):
  const void* a2;      // contraction columns [a_split, Kc) of the row operand come from this matrix (pitch lda2); null = none
  int64_t a_split, lda2;
  void* out2;          // output columns [out_split, Nout) go to this matrix (pitch ldo2); null = none (then no addends)
  int64_t out_split, ldo2;
  // k_lin3 only -- a ROW MAP between the GEMM's rows (graph nodes) and the rows of a token matrix (gnn2transformer writing / reading
  // the Transformer's token rows in place, models/gnn_transformer.py:92-96): int32 [M], -1 = the node has no token row (truncated)
  const int32_t* out_rows;   // forward: output row m is stored at row out_rows[m] of `out` (skipped when < 0)
  const int32_t* a_rows;     // dX form: row m of the row operand is row a_rows[m] of `a` (zeros when < 0)
  // k_lin3 forward only, one column block = the whole row (Nout = NT x 16): LayerNorm of the stored (TO-rounded) output row in the
  // same epilogue (transformer_encoder.py:53-57 norm_input behind gnn2transformer): ln_out[row] = LN(out[row]) * ln_w + ln_b,
  // ln_mean / ln_rstd[row] = its statistics; rows are the OUTPUT rows (through out_rows when set)
  const float *ln_w, *ln_b;
  void* ln_out;              // [.][ldo], TO
  float *ln_mean, *ln_rstd;
  float ln_eps;
};

// 16-byte chunk of TA -> up to 8 floats
template <typename TA>
struct Chunk32 {
  static constexpr int E = 16 / sizeof(TA);
};
template <typename TA>
__device__ __forceinline__ void chunk_to_f32(const uint4& v, float* f) {
  if constexpr (sizeof(TA) == 4) {
    f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
  } else {
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f[2 * q] = __uint_as_float(u[q] << 16);
      f[2 * q + 1] = __uint_as_float(u[q] & 0xffff0000u);
    }
  }
}

template <typename TA, typename TO, int NT, bool MASK, bool GELU = false, bool BNS = false>   // GELU / BNS: their own instantiations
__global__ void __launch_bounds__(256) k_lin32(L32Args a) {   // (an epilogue compiled into the common one costs every launch its occupancy)
  constexpr int BM = W32_BM, BK = W32_BK, LD = W32_LD;
  constexpr int EA = Chunk32<TA>::E;            // elements per 16-byte chunk of the row operand
  constexpr int ACH = BK / EA;                  // chunks per row of the A tile (4 fp32 / 2 bf16)
  constexpr int WROWS = NT * 16;
  constexpr int WIT = (WROWS * 4 + 255) / 256;  // W-tile chunks per thread
  constexpr int STAGE = (BM + WROWS) * LD;
  constexpr int EPI = 4 * 16 * (128 + 4) + WROWS + (BNS ? 8 * WROWS : 0);   // epilogue: four per-wave patches + the bias row (+ BNS: [4 waves][2][WROWS])
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE > EPI ? 2 * STAGE : EPI];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  int64_t mt;
  int cb;
  {  // XCD-aware order: the column blocks of a row tile get ids 8 apart (same XCD, same L2)
    const int64_t b = blockIdx.x;
    const int64_t group = b / (8 * a.ncb);
    const int r = (int)(b % (8 * a.ncb));
    cb = r / 8;
    mt = group * 8 + r % 8;
  }
  const int64_t m0 = mt * BM;
  if (m0 >= a.M) return;
  const int64_t n0 = (int64_t)cb * WROWS;
  const TA* A = reinterpret_cast<const TA*>(a.a);
  const TA* Am = reinterpret_cast<const TA*>(a.amask);
  const bool has_mask = MASK && Am != nullptr;

  // ---- register staging (branch-free: out-of-range rows are clamped to a valid row -- their results are never stored --
  // and the K tail is clamped to the last chunk of the row and zeroed by a select) --------------------------------------
  const bool a_thr = tid < BM * ACH;
  const int ar = tid / ACH, ac = (tid % ACH) * EA;
  const int64_t a_row = m0 + ar < a.M ? m0 + ar : a.M - 1;
  const TA* a_src = A + a_row * a.lda;
  const TA* m_src = has_mask ? Am + a_row * a.lda : nullptr;
  uint4 va = make_uint4(0, 0, 0, 0), vm = make_uint4(0, 0, 0, 0);
  uint4 vw[WIT];
  const float* w_src[WIT];
  int w_lds[WIT];
#pragma unroll
  for (int i = 0; i < WIT; ++i) {
    const int c = tid + i * 256;
    const int r = c >> 2;
    const int64_t wr = n0 + r < a.Nout ? n0 + r : a.Nout - 1;
    w_src[i] = a.w + wr * a.ldw;
    w_lds[i] = r < WROWS ? (BM + r) * LD + (c & 3) * 4 : -1;
  }
  const int wcc = (tid & 3) * 4;   // (tid + i*256) & 3 == tid & 3
  // full stages load unconditionally (the loads stay in flight behind the MFMAs of the current stage: any select on
  // their result would make the compiler wait for them on the spot); only the LAST stage of a K that is not a multiple
  // of 16 is partial: its chunks are clamped to the last chunk of the row and zeroed when they are written to LDS
  bool a_zero = false, w_zero = false;
  auto load = [&](int64_t k0) {
    int64_t ka = k0 + ac, kw = k0 + wcc;
    a_zero = false;
    w_zero = false;
    if (k0 + BK > a.Kc) {   // wave-uniform: the tail stage
      a_zero = ka >= a.Kc;
      w_zero = kw >= a.Kc;
      ka = a_zero ? a.Kc - EA : ka;
      kw = w_zero ? a.Kc - 4 : kw;
    }
    va = *reinterpret_cast<const uint4*>(a_src + ka);
    if constexpr (MASK) {
      if (has_mask) vm = *reinterpret_cast<const uint4*>(m_src + ka);
    }
#pragma unroll
    for (int i = 0; i < WIT; ++i) vw[i] = *reinterpret_cast<const uint4*>(w_src[i] + kw);
  };
  auto store = [&](float* st) {
    if (a_thr) {
      float f[EA];
      if (a_zero) va = make_uint4(0, 0, 0, 0);
      chunk_to_f32<TA>(va, f);
      if constexpr (MASK) {
        if (has_mask) {
          float y[EA];
          chunk_to_f32<TA>(vm, y);
#pragma unroll
          for (int e = 0; e < EA; ++e) f[e] = gt_gate(f[e], y[e], a.inv_keep);
        }
      }
      float* dst = st + ar * LD + ac;
#pragma unroll
      for (int e = 0; e < EA; e += 4) *reinterpret_cast<float4*>(dst + e) = make_float4(f[e], f[e + 1], f[e + 2], f[e + 3]);
    }
#pragma unroll
    for (int i = 0; i < WIT; ++i)
      if (w_lds[i] >= 0) *reinterpret_cast<uint4*>(st + w_lds[i]) = w_zero ? make_uint4(0, 0, 0, 0) : vw[i];
  };

  f32x4 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  load(0);
  store(smem);
  __syncthreads();
  int cur = 0;
  for (int64_t k0 = 0; k0 < a.Kc; k0 += BK, cur ^= 1) {
    const bool more = k0 + BK < a.Kc;
    if (more) load(k0 + BK);
    const float* sX = smem + cur * STAGE;
    const float* sW = sX + BM * LD;
    const float4 xa = *reinterpret_cast<const float4*>(sX + (wid * 16 + n) * LD + g * 4);
    const float xv[4] = {xa.x, xa.y, xa.z, xa.w};
#pragma unroll
    for (int jj = 0; jj < NT; jj += 4) {
      float4 wb[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (jj + q < NT) wb[q] = *reinterpret_cast<const float4*>(sW + ((jj + q) * 16 + n) * LD + g * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (jj + q < NT) {
            const float wv = i == 0 ? wb[q].x : (i == 1 ? wb[q].y : (i == 2 ? wb[q].z : wb[q].w));
            acc[jj + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, xv[i], acc[jj + q], 0, 0, 0);
          }
        }
      }
    }
    if (more) store(smem + (cur ^ 1) * STAGE);
    __syncthreads();
  }

  // ---- epilogue: acc[j][r] = C[column n0 + j*16 + g*4 + r][row m0 + wid*16 + n] ------------------------------------
  // Eight n-tiles (128 columns) at a time through a per-wave LDS patch [16 rows][128 + 4]: the accumulator layout alone
  // gives store instructions that touch 16 rows x 64 bytes; from the patch a store instruction covers 2 rows x 512
  // contiguous bytes.  The bias goes through LDS too (one global
  // round trip for the block), the addends are fetched per batch BEFORE its stores.
  constexpr int PLD = 128 + 4;
  float* sB = smem + 4 * 16 * PLD;   // the main loop ended with a barrier: the stage buffers are free
  if (a.bias) {
    for (int c = tid; c < WROWS; c += 256) sB[c] = n0 + c < a.Nout ? a.bias[n0 + c] : 0.f;
    __syncthreads();
  }
  float* patch = smem + wid * 16 * PLD;
  const int64_t mrow0 = m0 + wid * 16;
#pragma unroll
  for (int jb = 0; jb < NT; jb += 8) {
    const int nj = NT - jb < 8 ? NT - jb : 8;          // n-tiles in this batch (compile-time after unrolling)
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (jb + q < NT)
        *reinterpret_cast<float4*>(patch + n * PLD + q * 16 + g * 4) = make_float4(acc[jb + q][0], acc[jb + q][1], acc[jb + q][2], acc[jb + q][3]);
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // 16 rows x nj*16 columns: chunk c -> row c / (nj*4), column chunk c % (nj*4)
    const int cpr = nj * 4;                          // 16-byte chunks per row
    float4 v[8], e1[8], e2[8];
    bool ok[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int c = lane + t * 64;
      const int r = c / cpr, c4 = (c % cpr) * 4;
      const int64_t m = mrow0 + r, col = n0 + jb * 16 + c4;
      ok[t] = c < 16 * cpr && m < a.M && col < a.Nout;
      if (ok[t]) {
        e1[t] = a.add1 ? gt_load4<TO>(reinterpret_cast<const TO*>(a.add1) + m * a.ldo + col) : gt_zero4();
        e2[t] = a.add2 ? gt_load4<TO>(reinterpret_cast<const TO*>(a.add2) + m * a.ldo + col) : gt_zero4();
      }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int c = lane + t * 64;
      const int r = c / cpr, c4 = (c % cpr) * 4;
      const int64_t m = mrow0 + r, col = n0 + jb * 16 + c4;
      if (ok[t]) {
        v[t] = *reinterpret_cast<const float4*>(patch + r * PLD + c4);
        if (a.bias) v[t] = gt_add4(v[t], *reinterpret_cast<const float4*>(sB + jb * 16 + c4));
        float* vv = reinterpret_cast<float*>(&v[t]);
        if (a.act == 1) v[t] = gt_relu4(v[t]);
        if constexpr (GELU) {
          {
            float4 gm;
            float* gg = reinterpret_cast<float*>(&gm);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              gt_gelu(vv[e], vv[e], gg[e]);
              if (a.thr && lin_hash(a.s0, a.s1, (uint32_t)m, (uint32_t)(col + e)) < a.thr) gg[e] = 0.f;
              else if (a.thr) gg[e] *= a.inv_keep;
            }
            if (a.gout) gt_store4<TO>(reinterpret_cast<TO*>(a.gout) + m * a.ldo + col, gm);
          }
        }
        if (a.thr) {
#pragma unroll
          for (int e = 0; e < 4; ++e) vv[e] = lin_hash(a.s0, a.s1, (uint32_t)m, (uint32_t)(col + e)) >= a.thr ? vv[e] * a.inv_keep : 0.f;
        }
        v[t] = gt_add4(gt_add4(v[t], e1[t]), e2[t]);
        if constexpr (BNS) *reinterpret_cast<float4*>(patch + r * PLD + c4) = v[t];   // the final dy, for the column pass below
      }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int c = lane + t * 64;
      const int r = c / cpr, c4 = (c % cpr) * 4;
      if (ok[t]) gt_store4<TO>(reinterpret_cast<TO*>(a.out) + (mrow0 + r) * a.ldo + n0 + jb * 16 + c4, v[t]);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (BNS) {
      // column pass: lane = column (two per lane), the wave's 16 rows top to bottom -> per-wave sums, fixed order
      float* red = smem + 4 * 16 * PLD + WROWS + wid * 2 * WROWS;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int colp = half * 64 + lane;            // column inside this batch of nj n-tiles
        const int64_t col = n0 + jb * 16 + colp;
        if (colp < nj * 16) {
          float s1 = 0.f, s2 = 0.f;
          if (col < a.Nout) {
            const float mu = a.bn_mean[col], rs = a.bn_rstd[col], ww = a.bn_w[col], bb = a.bn_b[col];
            float xv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) xv[r] = mrow0 + r < a.M ? a.bn_x[(mrow0 + r) * a.bn_ldx + col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              if (mrow0 + r < a.M) {
                const float xh = (xv[r] - mu) * rs;
                float dyv = patch[r * PLD + colp];
                if (a.bn_relu && !(xh * ww + bb > 0.f)) dyv = 0.f;
                s1 += dyv;
                s2 = fmaf(dyv, xh, s2);
              }
            }
          }
          red[jb * 16 + colp] = s1;
          red[WROWS + jb * 16 + colp] = s2;
        }
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  if constexpr (BNS) {   // the four waves' sums -> this row tile's partial row
    __syncthreads();
    const float* red = smem + 4 * 16 * PLD + WROWS;
    for (int c = tid; c < 2 * WROWS; c += 256) {
      const int which = c / WROWS, cc = c % WROWS;
      if (n0 + cc < a.Nout) {
        const float t = (red[c] + red[2 * WROWS + c]) + (red[4 * WROWS + c] + red[6 * WROWS + c]);
        a.bn_part[(mt * 2 + which) * a.Nout + n0 + cc] = t;
      }
    }
  }
}

// WT[K][N] = W[N][K]^T  (32 x 32 tiles through LDS)
__global__ void __launch_bounds__(256) k_transpose32(const float* __restrict__ W, float* __restrict__ WT, int64_t N, int64_t K) {
  __shared__ float t[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const int64_t k0 = (int64_t)blockIdx.x * 32, n0 = (int64_t)blockIdx.y * 32;
#pragma unroll
  for (int r = ty; r < 32; r += 8)
    t[r][tx] = (n0 + r < N && k0 + tx < K) ? W[(n0 + r) * K + k0 + tx] : 0.f;
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8)
    if (k0 + r < K && n0 + tx < N) WT[(k0 + r) * N + n0 + tx] = t[tx][r];
}

// ---- dW[N][K] = dZ^T X, db = colsum(dZ): contraction over M, split over blocks -------------------------------------
// A block owns NT n-tiles x 4 k-tiles (one k-tile per wave) and an M-range; operands are staged as they lie in memory
// ([m][n] and [m][k], 16 rows per stage) and read transposed with scalar ds_read_b32 (row pitches = 4 mod 8 floats: the
// four k-slot groups of a wave hit disjoint bank halves).  C is produced as [k-column][n]: a lane holds 4 consecutive
// k of one n -> 16-byte stores into the partial buffer.
struct L32DwArgs {
  const void* dy;      // [M][ldy]
  const void* ymask;   // [M][ldy] or null
  const void* x;       // [M][ldx]
  float* part;         // [splits][N][K]
  float* dbpart;       // [splits][N] or null
  int64_t M, N, K, ldy, ldx;
  float inv_keep;
  int splits, nkb, nnb;   // k-blocks of 64 columns, n-blocks of NT n-tiles
  int64_t m_per_split;
  const void* x2;      // columns [x_split, K) of X come from this matrix (pitch ldx2); null = none
  int64_t x_split, ldx2;
  const int32_t* dy_rows;   // k_lin3_dw only: row m of dY is row dy_rows[m] of `dy` (zeros when < 0); see L32Args
  // k_lin3r_dw only -- a GROUPED launch (blockIdx.y = group): group g reads dy / ymask + g * g_y and x + g * g_x elements and writes
  // its partials g * g_part floats behind part / dbpart
  int groups;               // 0 / 1 = one GEMM
  int64_t g_y, g_x, g_part;
};

template <typename TY, typename TX, int NT, bool MASK>
__global__ void __launch_bounds__(256) k_lin32_dw(L32DwArgs a) {
  constexpr int BMc = 16;
  constexpr int NC = NT * 16;
  constexpr int LDZ = NC + 4, LDX = 64 + 4;
  constexpr int EY = Chunk32<TY>::E, EX = Chunk32<TX>::E;
  constexpr int ZCH = NC / EY;                       // chunks per dZ row
  constexpr int ZIT = (BMc * ZCH + 255) / 256;
  constexpr int XCH = 64 / EX;
  constexpr int STAGE = BMc * (LDZ + LDX);
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  int64_t split_;
  int tile_;
  {
    const int nt = a.nkb * a.nnb;
    const int64_t b = blockIdx.x;
    const int64_t group = b / (8 * nt);
    const int r = (int)(b % (8 * nt));
    tile_ = r / 8;
    split_ = group * 8 + r % 8;
  }
  if (split_ >= a.splits) return;
  const int kb = tile_ % a.nkb, nb = tile_ / a.nkb;
  const int64_t n0 = (int64_t)nb * NC, k0 = (int64_t)kb * 64;
  const int64_t mb = split_ * a.m_per_split;
  const int64_t me = mb + a.m_per_split < a.M ? mb + a.m_per_split : a.M;
  const TY* dY = reinterpret_cast<const TY*>(a.dy);
  const TY* Ym = reinterpret_cast<const TY*>(a.ymask);
  const bool has_mask = MASK && Ym != nullptr;
  const TX* X = reinterpret_cast<const TX*>(a.x);

  // branch-free staging: a chunk's column validity is fixed for the whole split (clamped base pointer + a zero flag);
  // rows run out only in the LAST stage of the split (wave-uniform test), where they are clamped to the split's last row
  // and zeroed when written to LDS.  No select touches a load result before the MFMAs of the current stage are issued.
  uint4 vz[ZIT], vmk[MASK ? ZIT : 1], vx = make_uint4(0, 0, 0, 0);
  const TY* z_src[ZIT];
  const TY* zm_src[MASK ? ZIT : 1];
  int z_r[ZIT];
  bool z_colok[ZIT];
#pragma unroll
  for (int i = 0; i < ZIT; ++i) {
    const int c = tid + i * 256;
    const int r = c / ZCH, cc = (c % ZCH) * EY;
    z_r[i] = r < BMc ? r : -1;
    z_colok[i] = r < BMc && n0 + cc < a.N;
    const int64_t col = n0 + cc < a.N ? n0 + cc : 0;
    z_src[i] = dY + col;
    if constexpr (MASK) zm_src[i] = has_mask ? Ym + col : nullptr;
  }
  const bool x_thr = tid < BMc * XCH;
  const int xr = x_thr ? tid / XCH : 0, xc = (tid % XCH) * EX;
  const bool x_colok = x_thr && k0 + xc < a.K;
  const TX* x_src = X + (k0 + xc < a.K ? k0 + xc : 0);
  int64_t x_ld = a.ldx;
  if (a.x2 && k0 + xc >= a.x_split && k0 + xc < a.K) {   // this thread's chunk column lies in the second matrix (x_split % 4 == 0)
    x_src = reinterpret_cast<const TX*>(a.x2) + (k0 + xc - a.x_split);
    x_ld = a.ldx2;
  }
  int64_t tail_rows = BMc;   // valid rows of the stage being staged
  auto load = [&](int64_t m0) {
    tail_rows = me - m0 < BMc ? me - m0 : BMc;
    const int64_t last = me - 1;
#pragma unroll
    for (int i = 0; i < ZIT; ++i) {
      int64_t row = m0 + (z_r[i] < 0 ? 0 : z_r[i]);
      row = row < me ? row : last;
      vz[i] = *reinterpret_cast<const uint4*>(z_src[i] + row * a.ldy);
      if constexpr (MASK) {
        if (has_mask) vmk[i] = *reinterpret_cast<const uint4*>(zm_src[i] + row * a.ldy);
      }
    }
    int64_t row = m0 + xr;
    row = row < me ? row : last;
    vx = *reinterpret_cast<const uint4*>(x_src + row * x_ld);
  };
  auto store = [&](float* st) {
    float* sZ = st;
    float* sX = st + BMc * LDZ;
#pragma unroll
    for (int i = 0; i < ZIT; ++i) {
      const int c = tid + i * 256;
      const int r = c / ZCH, cc = (c % ZCH) * EY;
      if (r < BMc) {
        float f[EY];
        if (!z_colok[i] || r >= tail_rows) vz[i] = make_uint4(0, 0, 0, 0);
        chunk_to_f32<TY>(vz[i], f);
        if constexpr (MASK) {
          if (has_mask) {
            float y[EY];
            chunk_to_f32<TY>(vmk[i], y);
#pragma unroll
            for (int e = 0; e < EY; ++e) f[e] = gt_gate(f[e], y[e], a.inv_keep);
          }
        }
#pragma unroll
        for (int e = 0; e < EY; e += 4) *reinterpret_cast<float4*>(sZ + r * LDZ + cc + e) = make_float4(f[e], f[e + 1], f[e + 2], f[e + 3]);
      }
    }
    if (x_thr) {
      float f[EX];
      if (!x_colok || xr >= tail_rows) vx = make_uint4(0, 0, 0, 0);
      chunk_to_f32<TX>(vx, f);
#pragma unroll
      for (int e = 0; e < EX; e += 4) *reinterpret_cast<float4*>(sX + xr * LDX + xc + e) = make_float4(f[e], f[e + 1], f[e + 2], f[e + 3]);
    }
  };

  f32x4 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbacc[2] = {0.f, 0.f};   // kb == 0: columns n0 + tid and n0 + tid + 256 (NC <= 304)
  if (mb < me) {
    load(mb);
    store(smem);
  }
  __syncthreads();
  int cur = 0;
  for (int64_t m0 = mb; m0 < me; m0 += BMc, cur ^= 1) {
    const bool more = m0 + BMc < me;
    if (more) load(m0 + BMc);
    const float* sZ = smem + cur * STAGE;
    const float* sX = sZ + BMc * LDZ;
    if (kb == 0) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if (tid + h * 256 < NC) {
#pragma unroll
          for (int r = 0; r < BMc; ++r) dbacc[h] += sZ[r * LDZ + tid + h * 256];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float xv = sX[(g * 4 + i) * LDX + wid * 16 + n];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float zv = sZ[(g * 4 + i) * LDZ + j * 16 + n];
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv, zv, acc[j], 0, 0, 0);
      }
    }
    if (more) store(smem + (cur ^ 1) * STAGE);
    __syncthreads();
  }
  // acc[j][r] = C[k-column k0 + wid*16 + g*4 + r][n = n0 + j*16 + n]
  float* part = a.part + (int64_t)split_ * a.N * a.K;
  const int64_t kc = k0 + wid * 16 + g * 4;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int64_t row = n0 + j * 16 + n;
    if (row < a.N && kc < a.K) *reinterpret_cast<float4*>(part + row * a.K + kc) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
  }
  if (kb == 0 && a.dbpart) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
      if (tid + h * 256 < NC && n0 + tid + h * 256 < a.N) a.dbpart[(int64_t)split_ * a.N + n0 + tid + h * 256] = dbacc[h];
  }
}

// n-tiles per column block: the candidate with the fewest padded tiles (ties -> the wider one)
static inline int w32_pick_nt(int64_t N) {
  const int64_t tiles = gt_cdiv(N, 16);
  const int cand[4] = {19, 16, 12, 8};
  int best = 8;
  int64_t best_waste = 1 << 30;
  for (int c : cand) {
    const int64_t waste = gt_cdiv(tiles, c) * c - tiles;
    if (waste < best_waste) { best_waste = waste; best = c; }
  }
  return best;
}
