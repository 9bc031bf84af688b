// linear1.h — the bf16 GEMMs of the Transformer encoder layers (in_proj / out_proj / linear1 / linear2, forward and dX) with the
// WEIGHT STATIONARY IN REGISTERS.  Included by linear.hip inside its anonymous namespace, after linear3x.h.
//
// Reference: nn.TransformerEncoderLayer in modules/transformer_encoder.py:28-32 (self_attn in/out projections, linear1 + activation +
// dropout, linear2): cuBLAS GEMMs + separate bias / activation / dropout launches there.
//
// Why: these GEMMs are M = tokens (3e4 .. 1.3e5) x {128, 512} x {128, 384, 512}: a few MB of bf16 rows against <= 128 KB of weights,
// 1.7 us of MFMA time.  The tiled kernel (k_linear_fwd / k_linear_dx) re-stages a 128-column slice of the fp32 master weight through
// LDS in every 64-row block (fp32 -> bf16 conversion included) with a barrier per 32-deep k-step: its LDS write / barrier / fragment
// read skeleton, not HBM or the matrix cores, set its 21 / 24 us (0.17-0.20 of HBM on algorithmic bytes).
// Here the weight is converted ONCE per step into an "image" in MFMA fragment order (k_w1_image, all encoder weights in one
// launch): n-tile t, k-step s, lane l holds W[t*16 + (l & 15)][s*32 + (l >> 4)*8 .. +8] as 16 bytes, so a wave fetches a
// fragment with ONE coalesced 1-KB load and keeps the fragments of its NTW n-tiles x all K in <= 128 VGPRs for the lifetime of the
// block.  Blocks are persistent (one per CU, 8 waves = 2 row halves x 4 column quarters); the only thing that moves per row tile is
// the activation tile (64 rows x 128 k = 16 KB, register-staged into a swizzled LDS stage, two stages, ONE barrier per 128-deep
// chunk instead of one per 32), and the epilogue runs per wave through a private LDS patch (no block barrier) with bias / ReLU /
// dropout / the gradient gate / two residual addends fused, storing 16 bytes per lane in row segments of NTW x 32 bytes.
// The dX form is the same kernel on the image of W^T (built by the same launch).  A gradient gate is applied to the OUTPUT
// columns here (dX of linear2 writes dZ1 = (dF2 W2) * 1[f1 > 0] / keep directly), not to the dY operand while it is staged as the
// tiled kernels do: the gated tensor is what BOTH the dX and the dW GEMM of linear1 read next.
#pragma once

constexpr int W1_MAX_JOBS = 32;
struct W1Job {
  const float* w;   // fp32 master weight [rows][ldw]
  unsigned char* img;
  int R, C;         // image rows (output columns of the GEMM, multiple of 16) and contraction length (multiple of 32)
  int ldw;
  int transposed;   // 0: element (r, c) = w[r * ldw + c]; 1: = w[c * ldw + r] (the image of W^T: the dX form)
  int block0;       // first block of this job
};
struct W1Jobs {
  W1Job j[W1_MAX_JOBS];
  int n;
};
static inline size_t w1_image_bytes(int64_t R, int64_t C) { return (size_t)(R / 16) * (size_t)(C / 32) * 1024; }

// one wave per fragment (n-tile t, k-step s); 4 fragments per block
__global__ void __launch_bounds__(256) k_w1_image(W1Jobs jobs) {
  int ji = 0;
  for (int i = 1; i < jobs.n; ++i)
    if ((int)blockIdx.x >= jobs.j[i].block0) ji = i;
  const W1Job& J = jobs.j[ji];
  const int ks = J.C / 32;
  const int f = ((int)blockIdx.x - J.block0) * 4 + (threadIdx.x >> 6);
  if (f >= (J.R / 16) * ks) return;
  const int t = f / ks, s = f % ks, l = threadIdx.x & 63;
  const int row = t * 16 + (l & 15), k0 = s * 32 + (l >> 4) * 8;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = J.transposed ? J.w[(int64_t)(k0 + e) * J.ldw + row] : J.w[(int64_t)row * J.ldw + k0 + e];
  *reinterpret_cast<uint4*>(J.img + (int64_t)f * 1024 + l * 16) =
      make_uint4(gt_pack_bf16(v[0], v[1]), gt_pack_bf16(v[2], v[3]), gt_pack_bf16(v[4], v[5]), gt_pack_bf16(v[6], v[7]));
}

struct L1Args {
  const gt_bf16* a;           // rows [M][lda]: X (forward) or dY (dX form)
  const unsigned char* img;   // image of W (forward) or W^T (dX form)
  const float* bias;          // [N] or null
  const gt_bf16* gate;        // [M][ldo] or null: out = gt_gate(acc, gate, gate_inv_keep) (a forward output, or a multiplier when 0)
  const gt_bf16* add1;        // [M][ldo] or null: residual addends, added after the gate
  const gt_bf16* add2;
  gt_bf16* out;               // [M][ldo]
  int64_t M, lda, ldo;
  int N, K;
  int act;                    // 0 none, 1 relu
  float inv_keep;             // forward dropout behind the activation
  uint32_t thr, s0, s1;
  float gate_inv_keep;
  // LN instantiations (one column block = the whole row): out = a = bf16(acc + bias) is SAVED (the LayerNorm backward re-reads it),
  // ln_out = LayerNorm(ln_resid + dropout(a)) * ln_w + ln_b, ln_mean / ln_rstd = the row statistics it saves
  const gt_bf16* ln_resid;    // [M][ldo] or null
  const float* ln_w;
  const float* ln_b;
  gt_bf16* ln_out;            // [M][ldo]
  float* ln_mean;
  float* ln_rstd;
  float ln_eps, ln_inv_keep;
  uint32_t ln_thr, ln_s0, ln_s1;
  // LNB instantiations (dX form, one column block = the whole row): g = acc (+ add1 + add2), rounded to bf16, is the gradient of
  // LayerNorm(ln_resid + dropout(lnb_x)) (statistics ln_mean / ln_rstd, weight ln_w, dropout ln_thr / ln_s0 / ln_s1 / ln_inv_keep): the
  // epilogue writes lnb_dresid = dz and lnb_dsub = dz * dropout mask instead of g, and the block's partial column sums of g * xhat and
  // g into lnb_part[block][2][N] (k_ln_bwd_finish / the deferred reduce sum them: the LayerNorm's weight / bias gradient)
  const gt_bf16* lnb_x;       // [M][ldo] the sub-layer output the forward normalised (before dropout)
  gt_bf16* lnb_dsub;          // [M][ldo] or null
  gt_bf16* lnb_dresid;        // [M][ldo] or null
  float* lnb_part;
  int ncb;                    // column blocks of 64 * NTW columns
  int sgroups;                // row-tile groups in flight: grid = 8 * ncb * sgroups
  int row_tiles;              // ceil(M / 64)
  int tm;                     // k_lin2 (linear2.h): rows per work item
};

constexpr int W1_TM = 64;            // rows per tile
constexpr int W1_STAGE = 64 * 256;   // 64 rows x 128 k of bf16
constexpr int W1_THREADS = 512;
constexpr int64_t W1_MIN_M = 1;

template <int NTW>
constexpr int w1_patch_ld() { return NTW * 16 + 4; }
template <int NTW>
constexpr size_t w1_lds_bytes() { return 2 * W1_STAGE + 8 * 16 * w1_patch_ld<NTW>() * sizeof(float) + 64 * 4 * 2 * sizeof(float); }

// Chan's merge of two (mean, M2) partials over EQUAL counts n each
__device__ __forceinline__ void w1_merge(float& mean, float& m2, float mean_b, float m2_b, float n) {
  const float dlt = mean_b - mean;
  mean += 0.5f * dlt;
  m2 += m2_b + dlt * dlt * (0.5f * n);
}

// KS = K / 32 k-steps (K % 128 == 0), NTW = n-tiles per wave (column block = 4 * NTW * 16 columns)
template <int KS, int NTW, int LN = 0>   // LN: 0 plain epilogue, 1 LayerNorm forward, 2 LayerNorm backward (LNB)
__global__ void __launch_bounds__(W1_THREADS, 1) k_lin1(L1Args a) {
  constexpr int KCH = KS / 4;                 // 128-deep chunks per row tile
  constexpr int PLD = w1_patch_ld<NTW>();
  static_assert(KS % 4 == 0 && NTW * KS * 4 <= 128, "weight fragments must fit 128 VGPRs");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem1[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n = lane & 15, g = lane >> 4, wm = wid >> 2, wn = wid & 3;
  float* patch = reinterpret_cast<float*>(smem1 + 2 * W1_STAGE) + wid * 16 * PLD;
  float2* rowstat = reinterpret_cast<float2*>(smem1 + 2 * W1_STAGE + 8 * 16 * PLD * sizeof(float));   // LN: [64 rows][4 column waves] (mean, M2)
  // block -> (XCD, column block, row-tile group): the ncb column blocks of a row tile share an XCD (one L2) and run together
  const int xcd = blockIdx.x & 7, cb = (blockIdx.x >> 3) % a.ncb, sg = (blockIdx.x >> 3) / a.ncb;
  const int col0 = (cb * 4 + wn) * NTW * 16;   // first output column of this wave

  // ---- the wave's weight fragments: NTW x KS coalesced 1-KB loads, resident from here on
  Frag<gt_bf16> wf[NTW][KS];
  {
    const uint4* wi = reinterpret_cast<const uint4*>(a.img) + (int64_t)(col0 / 16) * KS * 64 + lane;
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int s = 0; s < KS; ++s) wf[j][s].v = wi[(j * KS + s) * 64];
  }
  // my row tiles: (sg + it * sgroups) * 8 + xcd
  const int tile_stride = a.sgroups * 8;
  const int first_tile = sg * 8 + xcd;
  const int my_tiles = first_tile < a.row_tiles ? (a.row_tiles - first_tile + tile_stride - 1) / tile_stride : 0;
  const int nchunks = my_tiles * KCH;
  if (nchunks == 0) {
    if constexpr (LN == 2) {   // (its partial row is summed with the others)
      if (tid < 2 * NTW * 64) a.lnb_part[(int64_t)blockIdx.x * 2 * NTW * 64 + tid] = 0.f;
    }
    return;
  }
  // LNB: this thread's running column sums of g * xhat and g (its 8 columns are the same in every tile) live in a private LDS column
  // [16][512] behind the patches (as registers they spilled beside the 128 weight registers of KS = 16)
  float* lacc = reinterpret_cast<float*>(smem1 + w1_lds_bytes<NTW>()) + tid;
  if constexpr (LN == 2) {
#pragma unroll
    for (int e = 0; e < 16; ++e) lacc[e * W1_THREADS] = 0.f;
  }

  // staging: thread -> 2 x 16 bytes of a chunk: p = tid + q * 512 -> row p / 16, 16-byte column p % 16 (a wave reads 4 whole rows).
  // (plain values, no lambdas writing captured registers: those end up in scratch)
  const int sr0 = tid >> 4, sr1 = sr0 + 32, sc = tid & 15;
  const int lds0 = sr0 * 256 + ((sc ^ (sr0 & 15)) << 4), lds1 = sr1 * 256 + ((sc ^ (sr1 & 15)) << 4);
  const gt_bf16* abase = a.a + sc * 8;
#define W1_CHUNK_PTR(c_, sr_) \
  (reinterpret_cast<const uint4*>(abase + min((int64_t)(first_tile + ((c_) / KCH) * tile_stride) * W1_TM + (sr_), a.M - 1) * a.lda + ((c_) % KCH) * 128))
  uint4 ra0, ra1;
  f32x4 acc[2][NTW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  ra0 = *W1_CHUNK_PTR(0, sr0);
  ra1 = *W1_CHUNK_PTR(0, sr1);
  *reinterpret_cast<uint4*>(smem1 + lds0) = ra0;
  *reinterpret_cast<uint4*>(smem1 + lds1) = ra1;
  {
    const int c1 = nchunks > 1 ? 1 : 0;
    ra0 = *W1_CHUNK_PTR(c1, sr0);
    ra1 = *W1_CHUNK_PTR(c1, sr1);
  }
  __syncthreads();
  int c = 0;   // chunk counter: row tile c / KCH, 128-deep chunk c % KCH
  for (int t = 0; t < my_tiles; ++t) {
    // (the k-step index into the resident fragments must be a compile-time constant: the chunks of a tile are unrolled)
#pragma unroll
    for (int kc = 0; kc < KCH; ++kc, ++c) {
    const int cur = c & 1;
    // unconditional (conditionally written staging registers end up in scratch): the last iterations rewrite a stage nobody
    // reads / reload the last chunk
    *reinterpret_cast<uint4*>(smem1 + (cur ^ 1) * W1_STAGE + lds0) = ra0;
    *reinterpret_cast<uint4*>(smem1 + (cur ^ 1) * W1_STAGE + lds1) = ra1;
    {
      const int c2 = c + 2 < nchunks ? c + 2 : nchunks - 1;
      ra0 = *W1_CHUNK_PTR(c2, sr0);
      ra1 = *W1_CHUNK_PTR(c2, sr1);
    }
    // (the loads are ISSUED here: hipcc's scheduler otherwise sinks them below the chunk's MFMAs, and the next chunk's wait at the top of
    // the loop then sees their whole latency -- every chunk of a K = 384 / 512 tile)
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* st = smem1 + cur * W1_STAGE;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      Frag<gt_bf16> fx[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = wm * 32 + i * 16 + n;
        fx[i].v = *reinterpret_cast<const uint4*>(st + r * 256 + (((s4 * 4 + g) ^ (r & 15)) << 4));
      }
#pragma unroll
      for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][j] = mma(wf[j][kc * 4 + s4], fx[i], acc[i][j]);
    }
    if (kc == KCH - 1) {
      // ---- epilogue of this row tile: per wave, 2 m-tiles x (NTW x 16) columns through the wave's patch
      const int tile = first_tile + t * tile_stride;
      if constexpr (LN == 2) {
        // ---- LayerNorm-BACKWARD epilogue (k_ln_bwd_d128's arithmetic on the rows this block holds): phase A per wave -- g = bf16(acc +
        // addends), xhat recomputed from the saved sub-layer output / residual / statistics, the wave's row sums of g * w and g * w * xhat
        // over its NTW x 16 columns -> LDS --, block barrier, phase B: dz = rstd * (g w - mean(g w) - xhat * mean(g w xhat))
        constexpr int CPR = NTW * 2, NCH = 16 * CPR, NQ = NCH / 64;
        static_assert(NCH % 64 == 0, "LNB epilogue: whole waves of chunks");
        constexpr float INV_D = 1.0f / (float)(NTW * 64);
        // (one m-tile at a time, a block barrier each: holding both m-tiles' g / xhat rows across ONE barrier spilled at KS = 16)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float gk[NQ][8], xk[NQ][8], rk[NQ];
          uint32_t kp[NQ];
#pragma unroll
          for (int j = 0; j < NTW; ++j) {
            *reinterpret_cast<float4*>(patch + n * PLD + j * 16 + g * 4) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          __builtin_amdgcn_wave_barrier();
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            const int ch = lane + q * 64;
            const int r = ch / CPR, c8 = (ch % CPR) * 8;
            int64_t m = (int64_t)tile * W1_TM + wm * 32 + i * 16 + r;
            const bool live = m < a.M;
            if (!live) m = a.M - 1;   // (tail rows compute on a valid row, store nothing and add nothing to the column sums)
            const int col = col0 + c8;
            const int64_t o = m * a.ldo + col;
            const uint4 sx = *reinterpret_cast<const uint4*>(a.lnb_x + o);
            const uint4 sr = a.ln_resid ? *reinterpret_cast<const uint4*>(a.ln_resid + o) : make_uint4(0, 0, 0, 0);
            const float mu = a.ln_mean[m], rs = a.ln_rstd[m];
            float v[8];
            *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(patch + r * PLD + c8);
            *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(patch + r * PLD + c8 + 4);
            if (a.add1) {
              const uint4 ad = *reinterpret_cast<const uint4*>(a.add1 + o);
              const uint32_t u[4] = {ad.x, ad.y, ad.z, ad.w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v[2 * e] += __uint_as_float(u[e] << 16);
                v[2 * e + 1] += __uint_as_float(u[e] & 0xffff0000u);
              }
            }
            if (a.add2) {
              const uint4 ad = *reinterpret_cast<const uint4*>(a.add2 + o);
              const uint32_t u[4] = {ad.x, ad.y, ad.z, ad.w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v[2 * e] += __uint_as_float(u[e] << 16);
                v[2 * e + 1] += __uint_as_float(u[e] & 0xffff0000u);
              }
            }
            {   // the LayerNorm backward sees the gradient as the stand-alone pair of kernels hands it over: rounded to bf16
              const uint32_t u[4] = {gt_pack_bf16(v[0], v[1]), gt_pack_bf16(v[2], v[3]), gt_pack_bf16(v[4], v[5]), gt_pack_bf16(v[6], v[7])};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v[2 * e] = __uint_as_float(u[e] << 16);
                v[2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u);
              }
            }
            float x[8], rr[8];
            {
              const uint32_t ux[4] = {sx.x, sx.y, sx.z, sx.w}, ur[4] = {sr.x, sr.y, sr.z, sr.w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                x[2 * e] = __uint_as_float(ux[e] << 16); x[2 * e + 1] = __uint_as_float(ux[e] & 0xffff0000u);
                rr[2 * e] = __uint_as_float(ur[e] << 16); rr[2 * e + 1] = __uint_as_float(ur[e] & 0xffff0000u);
              }
            }
            uint32_t keep = 0xffu;
            if (a.ln_thr) {
              keep = 0;
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const bool k = lin_hash(a.ln_s0, a.ln_s1, (uint32_t)m, (uint32_t)(col + e)) >= a.ln_thr;
                keep |= k ? (1u << e) : 0u;
                x[e] = k ? x[e] * a.ln_inv_keep : 0.f;
              }
            }
            const float4 w0 = *reinterpret_cast<const float4*>(a.ln_w + col), w1 = *reinterpret_cast<const float4*>(a.ln_w + col + 4);
            const float gw[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
            const float cnt = live ? 1.f : 0.f;
            float gg[8], xh[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              xh[e] = ((x[e] + rr[e]) - mu) * rs;
              gg[e] = v[e] * gw[e];
              lacc[(8 + e) * W1_THREADS] = fmaf(v[e], cnt, lacc[(8 + e) * W1_THREADS]);
              lacc[e * W1_THREADS] = fmaf(v[e] * cnt, xh[e], lacc[e * W1_THREADS]);
              gk[q][e] = gg[e];
              xk[q][e] = xh[e];
            }
            kp[q] = keep;
            rk[q] = rs;
            float s1 = ((gg[0] + gg[1]) + (gg[2] + gg[3])) + ((gg[4] + gg[5]) + (gg[6] + gg[7]));
            float s2 = ((gg[0] * xh[0] + gg[1] * xh[1]) + (gg[2] * xh[2] + gg[3] * xh[3])) + ((gg[4] * xh[4] + gg[5] * xh[5]) + (gg[6] * xh[6] + gg[7] * xh[7]));
#pragma unroll
            for (int sh = 1; sh < CPR; sh <<= 1) {   // the CPR adjacent lanes of a row
              s1 += __shfl_xor(s1, sh, 64);
              s2 += __shfl_xor(s2, sh, 64);
            }
            if ((ch % CPR) == 0) rowstat[(wm * 32 + i * 16 + r) * 4 + wn] = make_float2(s1, s2);
          }
          __syncthreads();
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            const int ch = lane + q * 64;
            const int r = ch / CPR, c8 = (ch % CPR) * 8;
            const int64_t m = (int64_t)tile * W1_TM + wm * 32 + i * 16 + r;
            const float2* rsp = rowstat + (wm * 32 + i * 16 + r) * 4;
            const float2 p0 = rsp[0], p1 = rsp[1], p2 = rsp[2], p3 = rsp[3];
            const float m1 = ((p0.x + p1.x) + (p2.x + p3.x)) * INV_D, m2 = ((p0.y + p1.y) + (p2.y + p3.y)) * INV_D;
            if (m < a.M) {
              const int64_t o = m * a.ldo + col0 + c8;
              float dz[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) dz[e] = rk[q] * (gk[q][e] - m1 - xk[q][e] * m2);
              if (a.lnb_dresid)
                *reinterpret_cast<uint4*>(a.lnb_dresid + o) =
                    make_uint4(gt_pack_bf16(dz[0], dz[1]), gt_pack_bf16(dz[2], dz[3]), gt_pack_bf16(dz[4], dz[5]), gt_pack_bf16(dz[6], dz[7]));
              if (a.lnb_dsub) {
                if (a.ln_thr) {
#pragma unroll
                  for (int e = 0; e < 8; ++e) dz[e] = ((kp[q] >> e) & 1u) ? dz[e] * a.ln_inv_keep : 0.f;
                }
                *reinterpret_cast<uint4*>(a.lnb_dsub + o) =
                    make_uint4(gt_pack_bf16(dz[0], dz[1]), gt_pack_bf16(dz[2], dz[3]), gt_pack_bf16(dz[4], dz[5]), gt_pack_bf16(dz[6], dz[7]));
              }
            }
          }
        }
      } else if constexpr (LN == 1) {
        // ---- LayerNorm epilogue: phase A per wave (a = bf16(acc + bias) stored; z = resid + dropout(a) kept in registers; the wave's
        // (mean, M2) over its NTW x 16 columns of every row -> LDS), block barrier, phase B (merge the 4 column waves, normalise, store)
        constexpr int CPR = NTW * 2, NCH = 16 * CPR, NQ = NCH / 64;
        static_assert(NCH % 64 == 0, "LN epilogue: whole waves of chunks");
        float z[2][NQ][8];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int j = 0; j < NTW; ++j) {
            *reinterpret_cast<float4*>(patch + n * PLD + j * 16 + g * 4) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          __builtin_amdgcn_wave_barrier();
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            const int ch = lane + q * 64;
            const int r = ch / CPR, c8 = (ch % CPR) * 8;
            int64_t m = (int64_t)tile * W1_TM + wm * 32 + i * 16 + r;
            const bool live = m < a.M;
            if (!live) m = a.M - 1;   // (tail rows compute on a valid row and store nothing)
            const int col = col0 + c8;
            float v[8];
            *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(patch + r * PLD + c8);
            *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(patch + r * PLD + c8 + 4);
            if (a.bias) {
              const float4 b0 = *reinterpret_cast<const float4*>(a.bias + col), b1 = *reinterpret_cast<const float4*>(a.bias + col + 4);
              v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
              v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
            }
            const int64_t o = m * a.ldo + col;
            const uint4 packed = make_uint4(gt_pack_bf16(v[0], v[1]), gt_pack_bf16(v[2], v[3]), gt_pack_bf16(v[4], v[5]), gt_pack_bf16(v[6], v[7]));
            if (live) *reinterpret_cast<uint4*>(a.out + o) = packed;
            // the LayerNorm sees the SAVED (bf16) sub-layer output, as the stand-alone kernel and the backward do
            const uint32_t u[4] = {packed.x, packed.y, packed.z, packed.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[2 * e] = __uint_as_float(u[e] << 16);
              v[2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u);
            }
            if (a.ln_thr) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = lin_hash(a.ln_s0, a.ln_s1, (uint32_t)m, (uint32_t)(col + e)) >= a.ln_thr ? v[e] * a.ln_inv_keep : 0.f;
            }
            if (a.ln_resid) {
              const uint4 ad = *reinterpret_cast<const uint4*>(a.ln_resid + o);
              const uint32_t w[4] = {ad.x, ad.y, ad.z, ad.w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v[2 * e] += __uint_as_float(w[e] << 16);
                v[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
              }
            }
            float mean = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            mean *= 0.125f;
            float m2 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              z[i][q][e] = v[e];
              m2 = fmaf(v[e] - mean, v[e] - mean, m2);
            }
            float cnt = 8.f;
#pragma unroll
            for (int sh = 1; sh < CPR; sh <<= 1, cnt *= 2.f)   // the CPR adjacent lanes of a row
              w1_merge(mean, m2, __shfl_xor(mean, sh, 64), __shfl_xor(m2, sh, 64), cnt);
            if ((ch % CPR) == 0) rowstat[(wm * 32 + i * 16 + r) * 4 + wn] = make_float2(mean, m2);
          }
          __builtin_amdgcn_wave_barrier();
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            const int ch = lane + q * 64;
            const int r = ch / CPR, c8 = (ch % CPR) * 8;
            const int64_t m = (int64_t)tile * W1_TM + wm * 32 + i * 16 + r;
            const float2* rsp = rowstat + (wm * 32 + i * 16 + r) * 4;
            const float2 p0 = rsp[0], p1 = rsp[1], p2 = rsp[2], p3 = rsp[3];
            float mean = p0.x, m2 = p0.y, mean_b = p2.x, m2_b = p2.y;
            w1_merge(mean, m2, p1.x, p1.y, (float)(NTW * 16));
            w1_merge(mean_b, m2_b, p3.x, p3.y, (float)(NTW * 16));
            w1_merge(mean, m2, mean_b, m2_b, (float)(NTW * 32));
            const float rs = 1.0f / sqrtf(m2 * (1.0f / (float)(NTW * 64)) + a.ln_eps);
            if (m < a.M) {
              const int col = col0 + c8;
              const float4 w0 = *reinterpret_cast<const float4*>(a.ln_w + col), w1 = *reinterpret_cast<const float4*>(a.ln_w + col + 4);
              const float4 b0 = *reinterpret_cast<const float4*>(a.ln_b + col), b1 = *reinterpret_cast<const float4*>(a.ln_b + col + 4);
              const float gw[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w}, gb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
              float y[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) y[e] = (z[i][q][e] - mean) * rs * gw[e] + gb[e];
              *reinterpret_cast<uint4*>(a.ln_out + m * a.ldo + col) =
                  make_uint4(gt_pack_bf16(y[0], y[1]), gt_pack_bf16(y[2], y[3]), gt_pack_bf16(y[4], y[5]), gt_pack_bf16(y[6], y[7]));
              if (wn == 0 && (ch % CPR) == 0) {
                a.ln_mean[m] = mean;
                a.ln_rstd[m] = rs;
              }
            }
          }
        }
      } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          *reinterpret_cast<float4*>(patch + n * PLD + j * 16 + g * 4) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
          acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        constexpr int CPR = NTW * 2;            // 8-column chunks per row
        constexpr int NCH = 16 * CPR;           // chunks of the patch
#pragma unroll
        for (int q = 0; q < (NCH + 63) / 64; ++q) {
          const int ch = lane + q * 64;
          const int r = ch / CPR, c8 = (ch % CPR) * 8;
          const int64_t m = (int64_t)tile * W1_TM + wm * 32 + i * 16 + r;
          if (ch < NCH && m < a.M) {
            const int col = col0 + c8;
            float v[8];
            *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(patch + r * PLD + c8);
            *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(patch + r * PLD + c8 + 4);
            if (a.bias) {
              const float4 b0 = *reinterpret_cast<const float4*>(a.bias + col), b1 = *reinterpret_cast<const float4*>(a.bias + col + 4);
              v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
              v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
            }
            if (a.act == 1) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (a.thr) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = lin_hash(a.s0, a.s1, (uint32_t)m, (uint32_t)(col + e)) >= a.thr ? v[e] * a.inv_keep : 0.f;
            }
            const int64_t o = m * a.ldo + col;
            if (a.gate) {
              const uint4 gm = *reinterpret_cast<const uint4*>(a.gate + o);
              const uint32_t u[4] = {gm.x, gm.y, gm.z, gm.w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v[2 * e] = gt_gate(v[2 * e], __uint_as_float(u[e] << 16), a.gate_inv_keep);
                v[2 * e + 1] = gt_gate(v[2 * e + 1], __uint_as_float(u[e] & 0xffff0000u), a.gate_inv_keep);
              }
            }
            if (a.add1) {
              const uint4 ad = *reinterpret_cast<const uint4*>(a.add1 + o);
              const uint32_t u[4] = {ad.x, ad.y, ad.z, ad.w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v[2 * e] += __uint_as_float(u[e] << 16);
                v[2 * e + 1] += __uint_as_float(u[e] & 0xffff0000u);
              }
            }
            if (a.add2) {
              const uint4 ad = *reinterpret_cast<const uint4*>(a.add2 + o);
              const uint32_t u[4] = {ad.x, ad.y, ad.z, ad.w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v[2 * e] += __uint_as_float(u[e] << 16);
                v[2 * e + 1] += __uint_as_float(u[e] & 0xffff0000u);
              }
            }
            *reinterpret_cast<uint4*>(a.out + o) =
                make_uint4(gt_pack_bf16(v[0], v[1]), gt_pack_bf16(v[2], v[3]), gt_pack_bf16(v[4], v[5]), gt_pack_bf16(v[6], v[7]));
          }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      }
    }
    __syncthreads();
    }
  }
#undef W1_CHUNK_PTR
  if constexpr (LN == 2) {
    // ---- the block's partial of the LayerNorm weight / bias gradient: lanes with the same lane % CPR hold the same 8 columns (16 row
    // lanes) -> lanes 0 .. CPR-1 of every wave -> the two row halves of a column quarter through LDS -> lnb_part[block][2][N]
    constexpr int CPR = NTW * 2, NC = NTW * 64;
    float lnb_w[8], lnb_b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { lnb_w[e] = lacc[e * W1_THREADS]; lnb_b[e] = lacc[(8 + e) * W1_THREADS]; }
    __syncthreads();   // (sred below aliases the stages, which the last tile's slower waves may still read)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int sh = CPR; sh < 64; sh <<= 1) {
        lnb_w[e] += __shfl_xor(lnb_w[e], sh, 64);
        lnb_b[e] += __shfl_xor(lnb_b[e], sh, 64);
      }
    }
    float* sred = reinterpret_cast<float*>(smem1);   // [2 row halves][2][NC] (the stages are idle: the tile loop ended behind a barrier)
    if (lane < CPR) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sred[(wm * 2 + 0) * NC + col0 + lane * 8 + e] = lnb_w[e];
        sred[(wm * 2 + 1) * NC + col0 + lane * 8 + e] = lnb_b[e];
      }
    }
    __syncthreads();
    if (tid < 2 * NC) a.lnb_part[(int64_t)blockIdx.x * 2 * NC + tid] = sred[tid] + sred[2 * NC + tid];
  }
}

// ---- shapes ---------------------------------------------------------------------------------------------------------
// n-tiles per wave for an (output columns R, contraction C) GEMM, 0 = not covered (the tiled kernels take it)
static inline int w1_pick_ntw(int64_t R, int64_t C) {
  if (R <= 0 || C <= 0 || R % 64 || C % 128 || C > 512) return 0;
  const int ks = (int)(C / 32), q = (int)(R / 64);   // q = n-tiles per wave if ONE column block covered R
  static const int cand[6][5] = {{4, 6, 4, 2, 0}, {8, 4, 2, 0, 0}, {12, 2, 0, 0, 0}, {16, 2, 0, 0, 0}, {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}};
  // (K = 768 / 1024 with ONE n-tile per wave -- the ER shapes 256 x 1024, 256 x 768 -- measured slower than the tiled kernels: not covered)
  // (NTW = 8 for 512 columns x K = 128 measured no faster than two column blocks of NTW = 4 and needs all 256 registers)
  for (int i = 0; i < 6; ++i)
    if (cand[i][0] == ks)
      for (int j = 1; j < 5 && cand[i][j]; ++j)
        if (q % cand[i][j] == 0) return cand[i][j];
  return 0;
}

template <int KS, int NTW, int LN = 0>
static inline bool w1_launch_one(dim3 grid, hipStream_t stream, const L1Args& a) {
  static std::mutex mu;
  static bool set[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  constexpr size_t lds = w1_lds_bytes<NTW>() + (LN == 2 ? 16 * W1_THREADS * sizeof(float) : 0);
  if (dev >= 0 && dev < 64) {
    std::lock_guard<std::mutex> lk(mu);
    if (!set[dev]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lin1<KS, NTW, LN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
      set[dev] = true;
    }
  }
  hipLaunchKernelGGL((k_lin1<KS, NTW, LN>), grid, dim3(W1_THREADS), lds, stream, a);
  return true;
}

// out[M][N] = epilogue(A[M][K] image^T); false = shape not covered / launch set-up failed
static inline bool w1_ln_covered(int64_t N, int64_t K) {
  const int ntw = w1_pick_ntw(N, K), ks = (int)(K / 32);
  return ntw && N == 64 * ntw && ((ntw == 2 && (ks == 4 || ks == 8 || ks == 16)) || (ntw == 4 && ks == 8));
}
// LayerNorm-backward epilogue: 128-column rows out of a 128 / 384 / 512-deep contraction (d_model = 128: out_proj^T-free shapes of the
// encoder layer -- linear1's dX (512 -> 128) and in_proj's dX (384 -> 128))
static inline bool w1_lnb_covered(int64_t N, int64_t K) {   // N = output columns (the LayerNorm dim), K = contraction
  return N == 128 && (K == 128 || K == 384 || K == 512) && w1_pick_ntw(N, K) == 2;
}
// blocks of the launch for M rows and N output columns (the LNB partial rows: one per block)
static inline int w1_grid_blocks(int64_t M, int64_t N, int64_t K) {
  const int ntw = w1_pick_ntw(N, K);
  if (!ntw || M <= 0) return 0;
  const int ncb = (int)(N / (64 * ntw));
  int sgroups = 32 / ncb;
  if (sgroups < 1) sgroups = 1;
  const int need = (int)gt_cdiv(gt_cdiv(M, W1_TM), 8);
  if (sgroups > need) sgroups = need;
  return 8 * ncb * sgroups;
}
static inline bool w1_launch(hipStream_t stream, L1Args& a) {
  const int ntw = w1_pick_ntw(a.N, a.K);
  if (!ntw || a.M <= 0) return false;
  a.ncb = a.N / (64 * ntw);
  a.row_tiles = (int)gt_cdiv(a.M, W1_TM);
  int sgroups = 32 / a.ncb;                           // ~one block per CU
  if (sgroups < 1) sgroups = 1;
  const int need = (int)gt_cdiv(a.row_tiles, 8);      // no more groups than row tiles
  if (sgroups > need) sgroups = need;
  a.sgroups = sgroups;
  const dim3 grid((unsigned)(8 * a.ncb * sgroups));
  const int ks = a.K / 32;
  if (a.ln_out) {   // LayerNorm epilogue: the row must lie in ONE column block
    if (a.ncb != 1) return false;
#define GT_W1_LN_CASE(KS_, NTW_) if (ks == KS_ && ntw == NTW_) return w1_launch_one<KS_, NTW_, 1>(grid, stream, a)
    GT_W1_LN_CASE(4, 2); GT_W1_LN_CASE(8, 2); GT_W1_LN_CASE(16, 2); GT_W1_LN_CASE(8, 4);
#undef GT_W1_LN_CASE
    return false;
  }
  if (a.lnb_part) {   // LayerNorm-backward epilogue (dX form): the row in ONE column block of 128 columns
    if (a.ncb != 1) return false;
    if (ks == 12 && ntw == 2) return w1_launch_one<12, 2, 2>(grid, stream, a);
    if (ks == 16 && ntw == 2) return w1_launch_one<16, 2, 2>(grid, stream, a);
    if (ks == 4 && ntw == 2) return w1_launch_one<4, 2, 2>(grid, stream, a);
    return false;
  }
#define GT_W1_CASE(KS_, NTW_) if (ks == KS_ && ntw == NTW_) return w1_launch_one<KS_, NTW_>(grid, stream, a)
  GT_W1_CASE(4, 6); GT_W1_CASE(4, 4); GT_W1_CASE(4, 2);
  GT_W1_CASE(8, 4); GT_W1_CASE(8, 2);
  GT_W1_CASE(12, 2); GT_W1_CASE(16, 2);
#undef GT_W1_CASE
  return false;
}

// ---- bound images (per host thread, as the bf16x3 table of linear3x.h) ----------------------------------------------------
struct W1Bound {
  const float* w;
  int64_t N, K;
  const void* img_fwd;   // image of W   [N][K]: forward
  const void* img_t;     // image of W^T [K][N]: dX
};
constexpr int W1_MAX_BOUND = 64;
struct W1Table {
  W1Bound e[W1_MAX_BOUND];
  int n = 0;
};
thread_local W1Table g_w1;

static inline const void* w1_lookup(const float* w, int64_t N, int64_t K, bool transposed) {
  for (int i = 0; i < g_w1.n; ++i)
    if (g_w1.e[i].w == w && g_w1.e[i].N == N && g_w1.e[i].K == K) return transposed ? g_w1.e[i].img_t : g_w1.e[i].img_fwd;
  return nullptr;
}
