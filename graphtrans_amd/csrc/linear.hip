// linear.hip — nn.Linear forward / backward on the CDNA4 matrix cores for the skinny GEMMs of the
// GraphTrans hot path (M = nodes or tokens ~ 3e4, K and N in 128..600), with the surrounding
// elementwise work fused in.
//
// Reference call sites (paths under /root/reference): GCNConv.linear modules/conv.py:44,51; GIN mlp
// conv.py:18-20; virtual-node MLP modules/gnn_module.py:161-170; gnn2transformer
// models/gnn_transformer.py:69-70,92; nn.TransformerEncoderLayer in_proj / out_proj / linear1 (+act
// +dropout) / linear2 modules/transformer_encoder.py:28-32 — cuBLAS GEMM + bias + relu + dropout
// as separate launches in the reference; the first profile here (profiles/r01a) showed the library
// GEMMs at 2-4 % of the MFMA rate on these shapes and ~450 elementwise/cast/reduce launches per step.
//
//   fwd : Y[M][N]  = act(X[M][K] W[N][K]^T + b) (* dropout)           k_linear_fwd   ("NT")
//   dX  : dX[M][K] = dZ[M][N] W[N][K],  dZ = dY * 1[Y>0] * inv_keep   k_linear_dx    ("NN")
//   dW  : dW[N][K] = dZ^T X,  db[N] = colsum(dZ)                      k_linear_dw    ("TN", split over M)
// W is always the fp32 master weight (converted while staging: no cast pass); X / Y storage fp32 or
// bf16; compute type bf16 (v_mfma_f32_16x16x32_bf16) or fp32 (v_mfma_f32_16x16x4_f32, exact).
// Orientation: every lane owns an output ROW index as its MFMA column (n = lane & 15) and 4
// consecutive output columns as accumulator registers -> 8/16-byte stores, float4 bias loads.
#include <mutex>

#include "gt_common.h"
#include "mfma_frag.h"

namespace {
using namespace gtf;

constexpr int LT = 256;  // threads
constexpr int BN = 128;

struct LinArgs {
  const void* a;      // fwd: X[M][K]; dx: dY[M][N]; dw: dY[M][N]
  const float* w;     // W[N][K] fp32
  const float* bias;  // [N] or null
  const void* x;      // dw: X[M][K]
  const void* ymask;  // dx/dw: forward output Y[M][N] when the forward fused relu(/dropout): dZ = dY*(Y>0)*inv_keep
  const void* add1;   // dx: optional addends [M][K] (storage type of dX): dX = dZ W + add1 + add2
  const void* add2;
  void* out;          // fwd: Y; dx: dX; dw: partial [splits][N][K] fp32
  float* dbpart;      // dw: [splits][N]
  int64_t M, N, K;
  int64_t ldy;        // row stride (elements) of Y / dY / ymask; >= N
  int64_t ldx;        // row stride (elements) of X / dX / the dX addends; >= K
  int act;            // 0 none, 1 relu, 2 gelu (erf)
  void* gout;         // fwd, act == 2: [M][ldy] (storage type of Y) receives d act/dz * dropout scale, the backward's multiplier; or null
  float inv_keep;     // 1/(1-p)
  uint32_t thr, s0, s1;
  int splits;
  int ntiles;           // fwd / dx: column tiles per row tile; dw: output tiles per M-split (see tile_of_block)
  int ntx;              // dw: tiles along N
  int64_t m_per_split;
  int64_t n_per_split;  // dx: contraction range per blockIdx.z when splits > 1 (partials [splits][M][K] fp32 in `out`)
  // grouped launch (blockIdx.y = group g, e.g. the towers of PNAConv): element offsets added per group
  int64_t g_x, g_y, g_w, g_b;   // X / dX / addends ; Y / dY / ymask ; W [N][K] ; bias [N]
  int64_t g_part;               // dw: floats between the groups' partial buffers (dW partials + db partials)
};

template <typename TA>
__device__ __forceinline__ const void* goff(const void* p, int64_t elems) {
  return p ? static_cast<const void*>(static_cast<const TA*>(p) + elems) : nullptr;
}

__device__ __forceinline__ uint32_t lin_hash(uint32_t s0, uint32_t s1, uint32_t row, uint32_t col) {
  uint32_t x = (row * 0x9E3779B1u + s0) ^ (col * 0x85EBCA77u + s1);
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

template <typename TC>
struct Tile {
  static constexpr int BK = 32;
  static constexpr int PAD = sizeof(TC) == 2 ? 8 : 4;   // 16 B row pad
  static constexpr int LD = BK + PAD;
};

// ---- register-staged tile loader (global -> registers -> LDS), 16-byte chunks -------------------
// load() issues every global load of the tile (and of the optional dZ mask tile) before anything
// consumes them; store() converts to the compute type and writes LDS.  Splitting the two lets the
// next stage's loads fly while the current stage's MFMAs run (T14 "issue early / write late").
template <typename TS, typename TC, int ROWS, int COLS, int LDS_LD, bool MASK>
struct Loader {
  static constexpr int EPC = 16 / sizeof(TS);  // elements per 16-byte chunk
  static constexpr int CH = COLS / EPC;
  static constexpr int NIT = ROWS * CH / LT;
  static_assert(ROWS * CH % LT == 0, "tile must be a multiple of the block's chunk count");
  uint4 v[NIT];
  uint4 m[MASK ? NIT : 1];
  int off[NIT];  // element offset of this thread's chunks relative to the tile origin (r * ld + cc)

  __device__ __forceinline__ void init(int64_t ld) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int c = threadIdx.x + i * LT;
      off[i] = (c / CH) * (int)ld + (c % CH) * EPC;
    }
  }

  // origin = address of tile element (0,0); rows_rem / cols_rem = valid extent from the origin
  __device__ __forceinline__ void load(const TS* origin, int64_t rows_rem, int64_t cols_rem, const TS* mask_origin) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int c = threadIdx.x + i * LT;
      const int r = c / CH, cc = (c % CH) * EPC;
      const bool ok = r < rows_rem && cc < cols_rem;
      v[i] = ok ? *reinterpret_cast<const uint4*>(origin + off[i]) : make_uint4(0, 0, 0, 0);
      if constexpr (MASK)
        m[i] = (ok && mask_origin) ? *reinterpret_cast<const uint4*>(mask_origin + off[i]) : make_uint4(0, 0, 0, 0);
    }
  }

  __device__ __forceinline__ void store(TC* lds, bool has_mask, float inv_keep) const {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int c = threadIdx.x + i * LT;
      const int r = c / CH, cc = (c % CH) * EPC;
      float f[EPC];
      if constexpr (sizeof(TS) == 4) {
        f[0] = __uint_as_float(v[i].x); f[1] = __uint_as_float(v[i].y);
        f[2] = __uint_as_float(v[i].z); f[3] = __uint_as_float(v[i].w);
      } else {
        const uint32_t u[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f[2 * q] = __uint_as_float(u[q] << 16);
          f[2 * q + 1] = __uint_as_float(u[q] & 0xffff0000u);
        }
      }
      if constexpr (MASK) {
        if (has_mask) {
          float y[EPC];
          if constexpr (sizeof(TS) == 4) {
            y[0] = __uint_as_float(m[i].x); y[1] = __uint_as_float(m[i].y);
            y[2] = __uint_as_float(m[i].z); y[3] = __uint_as_float(m[i].w);
          } else {
            const uint32_t u[4] = {m[i].x, m[i].y, m[i].z, m[i].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              y[2 * q] = __uint_as_float(u[q] << 16);
              y[2 * q + 1] = __uint_as_float(u[q] & 0xffff0000u);
            }
          }
#pragma unroll
          for (int e = 0; e < EPC; ++e) f[e] = gt_gate(f[e], y[e], inv_keep);
        }
      }
      TC* dst = lds + r * LDS_LD + cc;
      if constexpr (sizeof(TC) == 4) {
#pragma unroll
        for (int e = 0; e < EPC; e += 4) *reinterpret_cast<float4*>(dst + e) = make_float4(f[e], f[e + 1], f[e + 2], f[e + 3]);
      } else if constexpr (sizeof(TS) == 2) {
        // bf16 -> bf16 with no mask is a plain copy; with a mask the values were re-rounded above
        if (!MASK || !has_mask) {
          *reinterpret_cast<uint4*>(dst) = v[i];  // bf16 -> bf16: plain copy
        } else {
          uint4 o;
          o.x = gt_pack_bf16(f[0], f[1]);
          o.y = gt_pack_bf16(f[2], f[3]);
          o.z = gt_pack_bf16(f[4], f[5]);
          o.w = gt_pack_bf16(f[6], f[7]);
          *reinterpret_cast<uint4*>(dst) = o;
        }
      } else {
        uint2 o;
        o.x = gt_pack_bf16(f[0], f[1]);
        o.y = gt_pack_bf16(f[2], f[3]);
        *reinterpret_cast<uint2*>(dst) = o;
      }
    }
  }
};

// XCD-aware tile order for the row-parallel kernels: workgroup b runs on XCD b % 8 (each XCD has its
// own L2).  The NT column blocks of one row tile read the SAME activation rows, so they get ids that
// differ by 8 (same XCD, dispatched within 8*NT ids of each other): the rows cross the fabric once
// instead of NT times (PMC on the 300x300 GCN linears: 115 MB fetched for 38 MB of activations before).
// 1-D grid of 8*ceil(MT/8)*NT blocks; a different hardware mapping only costs the locality.
__device__ __forceinline__ void tile_of_block(int nt, int64_t& mt, int& ntile) {
  const int64_t b = blockIdx.x;
  const int64_t group = b / (8 * nt);
  const int r = (int)(b % (8 * nt));
  ntile = r / 8;
  mt = group * 8 + r % 8;
}

// Per-wave epilogue patch: 16 output rows x 64 output columns of fp32 staged through LDS so that the
// global stores are whole 256-byte row segments (the MFMA accumulator layout alone gives 32 B).
constexpr int PATCH_LD = 64 + 4;
constexpr int PATCH_FLOATS = 16 * PATCH_LD;

template <typename TO>
__device__ __forceinline__ void store_chunk(TO* p, float4 v) { gt_store4<TO>(p, v); }

// bias[col..col+3]; the last chunk of an N that is not a multiple of 4 must not read past the array
__device__ __forceinline__ float4 bias_chunk(const float* bias, int64_t col, int64_t N) {
  if (col + 4 <= N) return *reinterpret_cast<const float4*>(bias + col);
  float4 b = gt_zero4();
  if (col < N) b.x = bias[col];
  if (col + 1 < N) b.y = bias[col + 1];
  if (col + 2 < N) b.z = bias[col + 2];
  return b;
}

#include "linear32.h"
#include "linear3x.h"
#include "linear3r.h"
#include "linear1.h"
#include "linear2.h"
#include "linear_small.h"
#include "linear_heads.h"
#include "linear_dw16.h"

// ------------------------------------------------------------------------------------------------
// forward: Y = act(X W^T + b) [dropout]
// MFMA rows = output columns n (W rows), MFMA cols = output rows m (X rows).
// ------------------------------------------------------------------------------------------------
template <typename TX, typename TY, typename TC, int BMT>
__global__ void __launch_bounds__(LT) k_linear_fwd(LinArgs a) {
  constexpr int BK = Tile<TC>::BK, LD = Tile<TC>::LD;
  constexpr int BM = BMT, MI = BMT / 32;  // rows per block, m16-tiles per wave
  constexpr int TILE_ELEMS = (BM + BN) * LD;
  constexpr int PATCH_ELEMS = (int)(4 * PATCH_FLOATS * sizeof(float) / sizeof(TC));
  constexpr int LDS_ELEMS = 2 * TILE_ELEMS > PATCH_ELEMS ? 2 * TILE_ELEMS : PATCH_ELEMS;  // the epilogue patches reuse the tiles
  __shared__ __attribute__((aligned(16))) TC smem[LDS_ELEMS];
  TC* sX = smem;
  TC* sW = smem + BM * LD;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int wm = wid & 1, wn = wid >> 1;  // wave tile: rows wm*64.., cols wn*64..
  int64_t mt_;
  int nt_;
  tile_of_block(a.ntiles, mt_, nt_);
  const int64_t m0 = mt_ * BM, n0 = (int64_t)nt_ * BN;
  if (m0 >= a.M) return;
  if (blockIdx.y) {
    const int64_t grp = blockIdx.y;
    a.a = goff<TX>(a.a, grp * a.g_x);
    a.w += grp * a.g_w;
    if (a.bias) a.bias += grp * a.g_b;
    a.out = const_cast<void*>(goff<TY>(a.out, grp * a.g_y));
  }
  const TX* X = reinterpret_cast<const TX*>(a.a);
  f32x4 acc[4][MI];  // [n tile j][m tile i]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  Loader<TX, TC, BM, BK, LD, false> lx;
  Loader<float, TC, BN, BK, LD, false> lw;
  const int64_t Mx = a.M - m0, Nw = a.N - n0;
  const TX* xo = X + m0 * a.ldx;
  const float* wo = a.w + n0 * a.K;
  lx.init(a.ldx);
  lw.init(a.K);
  lx.load(xo, Mx, a.K, nullptr);
  lw.load(wo, Nw, a.K, nullptr);
  auto compute = [&](const TC* cX, const TC* cW) {
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      Frag<TC> fx[MI], fw[4];
#pragma unroll
      for (int i = 0; i < MI; ++i) fx[i] = frag_load(cX + (wm * (BM / 2) + i * 16 + n) * LD + kk * 32 + g * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) fw[j] = frag_load(cW + (wn * 64 + j * 16 + n) * LD + kk * 32 + g * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = mma(fw[j], fx[i], acc[j][i]);
    }
  };
  // two LDS stages, ONE barrier per k-step: stage s+1 is written while stage s is being multiplied (the
  // barrier that ended step s-1 freed that buffer); the global loads of stage s+2 are issued before the MFMAs
  // of stage s and land in registers during them
  lx.store(sX, false, 1.f);
  lw.store(sW, false, 1.f);
  if (BK < a.K) {
    lx.load(xo + BK, Mx, a.K - BK, nullptr);
    lw.load(wo + BK, Nw, a.K - BK, nullptr);
  }
  __syncthreads();
  int cur = 0;
  for (int64_t k0 = 0; k0 < a.K; k0 += BK, cur ^= 1) {
    if (k0 + BK < a.K) {
      lx.store(sX + (cur ^ 1) * TILE_ELEMS, false, 1.f);
      lw.store(sW + (cur ^ 1) * TILE_ELEMS, false, 1.f);
      if (k0 + 2 * BK < a.K) {
        lx.load(xo + k0 + 2 * BK, Mx, a.K - k0 - 2 * BK, nullptr);
        lw.load(wo + k0 + 2 * BK, Nw, a.K - k0 - 2 * BK, nullptr);
      }
    }
    compute(sX + cur * TILE_ELEMS, sW + cur * TILE_ELEMS);
    __syncthreads();
  }
  // epilogue: acc[j][i][r] = C[col n0+wn*64+j*16+g*4+r][row m0+wm*64+i*16+n] -> patch[row n][col ...]
  float* patch = reinterpret_cast<float*>(smem) + wid * PATCH_FLOATS;
  TY* Y = reinterpret_cast<TY*>(a.out);
#pragma unroll
  for (int i = 0; i < MI; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<float4*>(patch + n * PATCH_LD + j * 16 + g * 4) =
          make_float4(acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]);
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = lane + q * 64;       // 256 float4 chunks: row = c / 16, col4 = c % 16
      const int r = c >> 4, c4 = (c & 15) * 4;
      const int64_t m = m0 + wm * (BM / 2) + i * 16 + r;
      const int64_t col = n0 + wn * 64 + c4;
      if (m < a.M && col < a.N) {
        float4 v = *reinterpret_cast<const float4*>(patch + r * PATCH_LD + c4);
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
            const bool keep = lin_hash(a.s0, a.s1, (uint32_t)m, (uint32_t)(col + e)) >= a.thr;
            vv[e] = keep ? vv[e] * a.inv_keep : 0.f;
            gg[e] = keep ? gg[e] * a.inv_keep : 0.f;
          }
        }
        store_chunk<TY>(Y + m * a.ldy + col, v);
        if (a.act == 2 && a.gout) store_chunk<TY>(reinterpret_cast<TY*>(a.gout) + m * a.ldy + col, gm);
      }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// ------------------------------------------------------------------------------------------------
// dX[M][K] = dZ[M][N] W[N][K]      (contraction over n)
// MFMA rows = output columns k (A operand = W^T via transposed LDS read), MFMA cols = rows m.
// ------------------------------------------------------------------------------------------------
template <typename TY, typename TX, typename TC, int BMT>
__global__ void __launch_bounds__(LT) k_linear_dx(LinArgs a) {
  constexpr int BM = BMT, MI = BMT / 32;
  constexpr int BKc = Tile<TC>::BK;            // n-slots per stage
  constexpr int LDZ = Tile<TC>::LD;            // dZ tile [BM][BKc]
  constexpr int LDW = BN + Tile<TC>::PAD;      // W tile [BKc n][BN k]
  constexpr int TILE_ELEMS = BM * LDZ + BKc * LDW;
  constexpr int PATCH_ELEMS = (int)(4 * PATCH_FLOATS * sizeof(float) / sizeof(TC));
  constexpr int LDS_ELEMS = 2 * TILE_ELEMS > PATCH_ELEMS ? 2 * TILE_ELEMS : PATCH_ELEMS;  // two stages (see k_linear_fwd)
  __shared__ __attribute__((aligned(16))) TC smem[LDS_ELEMS];
  TC* sZ = smem;
  TC* sW = smem + BM * LDZ;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int wm = wid & 1, wk = wid >> 1;
  int64_t mt_;
  int nt_;
  tile_of_block(a.ntiles, mt_, nt_);
  const int64_t m0 = mt_ * BM, kk0 = (int64_t)nt_ * BN;  // output column tile (k)
  if (m0 >= a.M) return;
  if (blockIdx.y) {
    const int64_t grp = blockIdx.y;
    a.a = goff<TY>(a.a, grp * a.g_y);
    a.ymask = goff<TY>(a.ymask, grp * a.g_y);
    a.w += grp * a.g_w;
    a.add1 = goff<TX>(a.add1, grp * a.g_x);
    a.add2 = goff<TX>(a.add2, grp * a.g_x);
    a.out = const_cast<void*>(goff<TX>(a.out, grp * a.g_x));
  }
  const TY* dY = reinterpret_cast<const TY*>(a.a);
  const TY* Ym = reinterpret_cast<const TY*>(a.ymask);
  const bool has_mask = Ym != nullptr;
  f32x4 acc[4][MI];  // [k tile j][m tile i]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  Loader<TY, TC, BM, BKc, LDZ, true> lz;
  Loader<float, TC, BKc, BN, LDW, false> lw;
  const TY* zo = dY + m0 * a.ldy;
  const TY* mo = has_mask ? Ym + m0 * a.ldy : nullptr;
  const float* wo = a.w + kk0;
  // contraction range of this block (whole N unless the launch splits it over blockIdx.z)
  const int64_t cb = a.splits > 1 ? (int64_t)blockIdx.z * a.n_per_split : 0;
  const int64_t ce = a.splits > 1 ? (cb + a.n_per_split < a.N ? cb + a.n_per_split : a.N) : a.N;
  lz.init(a.ldy);
  lw.init(a.K);
  if (cb < ce) {
    lz.load(zo + cb, a.M - m0, ce - cb, has_mask ? mo + cb : nullptr);
    lw.load(wo + cb * a.K, ce - cb, a.K - kk0, nullptr);
  }
  auto load_stage = [&](int64_t c1) {
    lz.load(zo + c1, a.M - m0, ce - c1, has_mask ? mo + c1 : nullptr);
    lw.load(wo + c1 * a.K, ce - c1, a.K - kk0, nullptr);
  };
  if (cb < ce) {
    lz.store(sZ, has_mask, a.inv_keep);
    lw.store(sW, false, 1.f);
    if (cb + BKc < ce) load_stage(cb + BKc);
  }
  __syncthreads();
  int cur = 0;
  for (int64_t c0 = cb; c0 < ce; c0 += BKc, cur ^= 1) {
    if (c0 + BKc < ce) {
      lz.store(sZ + (cur ^ 1) * TILE_ELEMS, has_mask, a.inv_keep);
      lw.store(sW + (cur ^ 1) * TILE_ELEMS, false, 1.f);
      if (c0 + 2 * BKc < ce) load_stage(c0 + 2 * BKc);
    }
    const TC* cZ = sZ + cur * TILE_ELEMS;
    const TC* cW = sW + cur * TILE_ELEMS;
#pragma unroll
    for (int s = 0; s < BKc / 32; ++s) {
      Frag<TC> fz[MI], fw[4];
#pragma unroll
      for (int i = 0; i < MI; ++i) fz[i] = frag_load(cZ + (wm * (BM / 2) + i * 16 + n) * LDZ + s * 32 + g * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) fw[j] = frag_load_tr(cW, LDW, s * 32, wk * 64 + j * 16, n, g);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = mma(fw[j], fz[i], acc[j][i]);
    }
    __syncthreads();
  }
  float* patch = reinterpret_cast<float*>(smem) + wid * PATCH_FLOATS;
  TX* dX = reinterpret_cast<TX*>(a.out);
#pragma unroll
  for (int i = 0; i < MI; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<float4*>(patch + n * PATCH_LD + j * 16 + g * 4) =
          make_float4(acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]);
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = lane + q * 64;
      const int r = c >> 4, c4 = (c & 15) * 4;
      const int64_t m = m0 + wm * (BM / 2) + i * 16 + r;
      const int64_t col = kk0 + wk * 64 + c4;
      if (m < a.M && col < a.K && a.splits > 1) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + ((int64_t)blockIdx.z * a.M + m) * a.K + col) =
            *reinterpret_cast<const float4*>(patch + r * PATCH_LD + c4);
      } else if (m < a.M && col < a.K) {
        float4 v = *reinterpret_cast<const float4*>(patch + r * PATCH_LD + c4);
        if (a.add1) v = gt_add4(v, gt_load4<TX>(reinterpret_cast<const TX*>(a.add1) + m * a.ldx + col));
        if (a.add2) v = gt_add4(v, gt_load4<TX>(reinterpret_cast<const TX*>(a.add2) + m * a.ldx + col));
        store_chunk<TX>(dX + m * a.ldx + col, v);
      }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// ------------------------------------------------------------------------------------------------
// dW[N][K] = dZ^T X, db = colsum(dZ)      (contraction over m, split across blockIdx.z)
// MFMA rows = n (A = dZ^T via transposed LDS read), MFMA cols = k (B = X^T via transposed read).
// ------------------------------------------------------------------------------------------------
template <typename TY, typename TX, typename TC>
__global__ void __launch_bounds__(LT) k_linear_dw(LinArgs a) {
  constexpr int BMc = Tile<TC>::BK;            // m-slots per stage
  constexpr int LDZ = BN + Tile<TC>::PAD;      // dZ tile [BMc m][BN n]
  constexpr int LDX = BN + Tile<TC>::PAD;      // X tile  [BMc m][BN k]
  constexpr int TILE_ELEMS = BMc * (LDZ + LDX);
  constexpr int PATCH_ELEMS = (int)(4 * PATCH_FLOATS * sizeof(float) / sizeof(TC));
  constexpr int LDS_ELEMS = TILE_ELEMS > PATCH_ELEMS ? TILE_ELEMS : PATCH_ELEMS;
  __shared__ __attribute__((aligned(16))) TC smem[LDS_ELEMS];
  TC* sZ = smem;
  TC* sX = smem + BMc * LDZ;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int wn = wid & 1, wk = wid >> 1;
  // XCD-aware: the tiles of one M-split read the same dZ / X rows -> same XCD (see tile_of_block)
  int64_t split_;
  int tile_;
  tile_of_block(a.ntiles, split_, tile_);
  if (split_ >= a.splits) return;
  if (blockIdx.y) {
    const int64_t grp = blockIdx.y;
    a.a = goff<TY>(a.a, grp * a.g_y);
    a.ymask = goff<TY>(a.ymask, grp * a.g_y);
    a.x = goff<TX>(a.x, grp * a.g_x);
    a.out = static_cast<float*>(a.out) + grp * a.g_part;
    if (a.dbpart) a.dbpart += grp * a.g_part;
  }
  const int split = (int)split_;
  const int tx = tile_ % a.ntx, ty = tile_ / a.ntx;
  const int64_t n0 = (int64_t)tx * BN, k0 = (int64_t)ty * BN;
  const int64_t mb = (int64_t)split * a.m_per_split;
  const int64_t me = mb + a.m_per_split < a.M ? mb + a.m_per_split : a.M;
  const TY* dY = reinterpret_cast<const TY*>(a.a);
  const TY* Ym = reinterpret_cast<const TY*>(a.ymask);
  const bool has_mask = Ym != nullptr;
  const TX* X = reinterpret_cast<const TX*>(a.x);
  f32x4 acc[4][4];  // [n tile j][k tile i]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbacc = 0.f;  // thread t < BN: column n0 + t
  Loader<TY, TC, BMc, BN, LDZ, true> lz;
  Loader<TX, TC, BMc, BN, LDX, false> lx;
  lz.init(a.ldy);
  lx.init(a.ldx);
  if (mb < me) {
    lz.load(dY + mb * a.ldy + n0, me - mb, a.N - n0, has_mask ? Ym + mb * a.ldy + n0 : nullptr);
    lx.load(X + mb * a.ldx + k0, me - mb, a.K - k0, nullptr);
  }
  // (single LDS stage here: a second one costs this kernel, with its 64 accumulator registers and 17 KB tiles,
  // more occupancy than the saved barrier returns -- measured 4-8 % slower)
  for (int64_t m0 = mb; m0 < me; m0 += BMc) {
    __syncthreads();
    lz.store(sZ, has_mask, a.inv_keep);
    lx.store(sX, false, 1.f);
    __syncthreads();
    if (m0 + BMc < me) {
      const int64_t m1 = m0 + BMc;
      lz.load(dY + m1 * a.ldy + n0, me - m1, a.N - n0, has_mask ? Ym + m1 * a.ldy + n0 : nullptr);
      lx.load(X + m1 * a.ldx + k0, me - m1, a.K - k0, nullptr);
    }
    if (ty == 0 && threadIdx.x < BN) {
#pragma unroll 8
      for (int r = 0; r < BMc; ++r) {
        const TC v = sZ[r * LDZ + threadIdx.x];
        if constexpr (sizeof(TC) == 2) dbacc += gt_bf16_to_f32((gt_bf16)v);
        else dbacc += (float)v;
      }
    }
#pragma unroll
    for (int s = 0; s < BMc / 32; ++s) {
      Frag<TC> fz[4], fx[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) fz[j] = frag_load_tr(sZ, LDZ, s * 32, wn * 64 + j * 16, n, g);
#pragma unroll
      for (int i = 0; i < 4; ++i) fx[i] = frag_load_tr(sX, LDX, s * 32, wk * 64 + i * 16, n, g);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = mma(fz[j], fx[i], acc[j][i]);
    }
  }
  __syncthreads();
  // acc[j][i][r] = C[row n-index wn*64+j*16+g*4+r][col k-index wk*64+i*16+n] -> patch[16 n rows][64 k cols]
  float* patch = reinterpret_cast<float*>(smem) + wid * PATCH_FLOATS;
  float* part = reinterpret_cast<float*>(a.out) + (int64_t)split * a.N * a.K;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) patch[(g * 4 + r) * PATCH_LD + i * 16 + n] = acc[j][i][r];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = lane + q * 64;
      const int r = c >> 4, c4 = (c & 15) * 4;
      const int64_t row = n0 + wn * 64 + j * 16 + r;
      const int64_t col = k0 + wk * 64 + c4;
      if (row < a.N && col < a.K)
        *reinterpret_cast<float4*>(part + row * a.K + col) = *reinterpret_cast<const float4*>(patch + r * PATCH_LD + c4);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (ty == 0 && threadIdx.x < BN && n0 + threadIdx.x < a.N && a.dbpart)
    a.dbpart[(int64_t)split * a.N + n0 + threadIdx.x] = dbacc;
}

// out[i] = sum_s part[s][i]  (fixed order, 8 loads in flight).  One launch reduces two partial arrays:
// dW (len N*K -> out) and, when len2 > 0, db (len2 = N -> out2).
__global__ void __launch_bounds__(256) k_split_reduce(const float* __restrict__ part, int splits, int64_t len,
                                                      float* __restrict__ out, const float* __restrict__ part2, int64_t len2,
                                                      float* __restrict__ out2, int64_t g_part) {
  if (blockIdx.y) {   // grouped launch: the groups' partial buffers are g_part floats apart, outputs back to back
    part += blockIdx.y * g_part;
    out += blockIdx.y * len;
    if (part2) { part2 += blockIdx.y * g_part; out2 += blockIdx.y * len2; }
  }
  for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < len + len2; i0 += (int64_t)gridDim.x * 256) {
    const bool second = i0 >= len;
    const float* p = second ? part2 : part;
    const int64_t n = second ? len2 : len, i = second ? i0 - len : i0;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 8 <= splits; s += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += p[(int64_t)(s + u) * n + i];
    }
    for (; s < splits; ++s) acc[0] += p[(int64_t)s * n + i];
    (second ? out2 : out)[i] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  }
}

int check_lin(const char* fn, int x_dtype, int y_dtype, int compute, int64_t M, int64_t N, int64_t K, int64_t ldy) {
  if ((x_dtype != GT_F32 && x_dtype != GT_BF16) || (y_dtype != GT_F32 && y_dtype != GT_BF16) ||
      (compute != GT_F32 && compute != GT_BF16)) { gt_set_error("%s: bad dtype", fn); return GT_ERR_INVALID_ARG; }
  // exact-fp32 MFMA over mixed storage: fp32 x with bf16 y (gnn2transformer writing bf16 token rows; its backward reads
  // the bf16 token gradient) is supported; bf16 x with fp32 compute is not a combination of this path
  if (compute == GT_F32 && x_dtype != GT_F32) { gt_set_error("%s: fp32 compute needs fp32 x storage", fn); return GT_ERR_UNSUPPORTED; }
  if (M < 0 || N <= 0 || K <= 0) { gt_set_error("%s: bad sizes", fn); return GT_ERR_INVALID_ARG; }
  if (ldy < N) { gt_set_error("%s: ldy (%lld) < N (%lld)", fn, (long long)ldy, (long long)N); return GT_ERR_INVALID_ARG; }
  if (ldy % 4 != 0 || K % 4 != 0) { gt_set_error("%s: ldy (= N unless given) and K must be multiples of 4 (got %lld, %lld)", fn, (long long)ldy, (long long)K); return GT_ERR_UNSUPPORTED; }
  if ((x_dtype == GT_BF16 && K % 8 != 0) || (y_dtype == GT_BF16 && ldy % 8 != 0)) { gt_set_error("%s: bf16 storage needs K (x) / ldy (y) to be multiples of 8", fn); return GT_ERR_UNSUPPORTED; }
  return GT_OK;
}

void fill_drop(LinArgs& a, float dropout_p, uint64_t seed) {
  a.inv_keep = 1.0f / (1.0f - dropout_p);
  double thr = (double)dropout_p * 4294967296.0;
  a.thr = dropout_p > 0.f ? (uint32_t)(thr > 4294967295.0 ? 4294967295.0 : (thr < 1.0 ? 1.0 : thr)) : 0u;
  a.s0 = (uint32_t)seed;
  a.s1 = (uint32_t)(seed >> 32);
}

int dw_splits(int64_t M, int64_t N, int64_t K, int compute) {
  const int64_t bmc = compute == GT_BF16 ? 64 : 32;
  constexpr int target = 512;  // blocks per launch: partial traffic grows with it, parallelism too (flat 256..768 end to end)
  int64_t tiles = gt_cdiv(N, BN) * gt_cdiv(K, BN);
  int64_t s = target / tiles;
  int64_t maxs = gt_cdiv(M, bmc * 4);  // at least 4 stages per split
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  if (s > 256) s = 256;
  return (int)s;
}

// dX with a long contraction (N >> M, K: the 5 x 5002-way prediction heads) has too few output tiles
// to fill the chip: split N over blockIdx.z into fp32 partials, reduced in fixed order.
int dx_splits(int64_t M, int64_t N, int64_t K, int bm) {
  const int64_t tiles = gt_cdiv(M, bm) * gt_cdiv(K, BN);
  if (tiles >= 128 || N < 2048) return 1;
  int64_t s = 512 / tiles;
  const int64_t maxs = N / 256;  // at least 4 stages of 64 per split
  if (s > maxs) s = maxs;
  if (s > 256) s = 256;
  return s < 2 ? 1 : (int)s;
}

// dispatch over (storage of the M-sized operands, compute type)
#define GT_LIN_DISPATCH(KERNEL, grid, args)                                                                        \
  do {                                                                                                             \
    if (compute == GT_F32 && t0 == GT_F32 && t1 == GT_F32) hipLaunchKernelGGL((KERNEL<float, float, float>), grid, dim3(LT), 0, stream, args); \
    else if (compute == GT_F32 && t0 == GT_F32) hipLaunchKernelGGL((KERNEL<float, gt_bf16, float>), grid, dim3(LT), 0, stream, args); \
    else if (compute == GT_F32) hipLaunchKernelGGL((KERNEL<gt_bf16, float, float>), grid, dim3(LT), 0, stream, args); \
    else if (t0 == GT_F32 && t1 == GT_F32) hipLaunchKernelGGL((KERNEL<float, float, gt_bf16>), grid, dim3(LT), 0, stream, args);   \
    else if (t0 == GT_F32 && t1 == GT_BF16) hipLaunchKernelGGL((KERNEL<float, gt_bf16, gt_bf16>), grid, dim3(LT), 0, stream, args); \
    else if (t0 == GT_BF16 && t1 == GT_F32) hipLaunchKernelGGL((KERNEL<gt_bf16, float, gt_bf16>), grid, dim3(LT), 0, stream, args); \
    else hipLaunchKernelGGL((KERNEL<gt_bf16, gt_bf16, gt_bf16>), grid, dim3(LT), 0, stream, args);                 \
  } while (0)

#define GT_LIN_DISPATCH_BM(KERNEL, BMV, grid, args)                                                                \
  do {                                                                                                             \
    if (compute == GT_F32 && t0 == GT_F32 && t1 == GT_F32) hipLaunchKernelGGL((KERNEL<float, float, float, BMV>), grid, dim3(LT), 0, stream, args); \
    else if (compute == GT_F32 && t0 == GT_F32) hipLaunchKernelGGL((KERNEL<float, gt_bf16, float, BMV>), grid, dim3(LT), 0, stream, args); \
    else if (compute == GT_F32) hipLaunchKernelGGL((KERNEL<gt_bf16, float, float, BMV>), grid, dim3(LT), 0, stream, args); \
    else if (t0 == GT_F32 && t1 == GT_F32) hipLaunchKernelGGL((KERNEL<float, float, gt_bf16, BMV>), grid, dim3(LT), 0, stream, args);   \
    else if (t0 == GT_F32 && t1 == GT_BF16) hipLaunchKernelGGL((KERNEL<float, gt_bf16, gt_bf16, BMV>), grid, dim3(LT), 0, stream, args); \
    else if (t0 == GT_BF16 && t1 == GT_F32) hipLaunchKernelGGL((KERNEL<gt_bf16, float, gt_bf16, BMV>), grid, dim3(LT), 0, stream, args); \
    else hipLaunchKernelGGL((KERNEL<gt_bf16, gt_bf16, gt_bf16, BMV>), grid, dim3(LT), 0, stream, args);            \
  } while (0)

// ---- optional overlap of the weight-gradient GEMMs (gt_overlap_dw_*): per host thread ------------
struct DwPending {
  uintptr_t lo, hi;   // workspace range a forked dW GEMM (partials + its reduce) still uses
  hipEvent_t ev;      // recorded on the side stream behind it
};
constexpr int DW_RING = 32;   // an encoder stage books 6 launches (4 dW GEMMs, 2 LayerNorm finishes), two stages are in flight
struct DwOverlap {
  bool active = false;
  hipStream_t main = nullptr, side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool urgent = false;       // the forks from here on are the last of the backward: the optimizer waits for them
  DwPending ring[DW_RING];   // forks since the last full join, oldest first (the side stream runs them in this order)
  int n = 0;
};
thread_local DwOverlap g_dw;

void dw_forked(const void* workspace, size_t bytes);
// main waits for the side stream; nothing is pending afterwards
void dw_join_all() {
  (void)hipEventRecord(g_dw.ev_join, g_dw.side);
  (void)hipStreamWaitEvent(g_dw.main, g_dw.ev_join, 0);
  g_dw.n = 0;
}
// main waits for the forked dW GEMMs whose workspace overlaps [p, p + bytes) -- and, the side stream being in order, for
// everything forked before them; later forks on other workspaces keep running
void dw_release(const void* p, size_t bytes) {
  if (!g_dw.active || !g_dw.n) return;
  const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
  int last = -1;
  for (int i = 0; i < g_dw.n; ++i)
    if (g_dw.ring[i].lo < hi && lo < g_dw.ring[i].hi) last = i;
  if (last < 0) return;
  (void)hipStreamWaitEvent(g_dw.main, g_dw.ring[last].ev, 0);
  // drop entries 0..last; their events rotate to the free end of the ring
  DwPending keep[DW_RING];
  int k = 0;
  for (int i = last + 1; i < g_dw.n; ++i) keep[k++] = g_dw.ring[i];
  const int kept = k;
  for (int i = 0; i <= last; ++i) keep[k++] = g_dw.ring[i];
  for (int i = g_dw.n; i < DW_RING; ++i) keep[k++] = g_dw.ring[i];
  for (int i = 0; i < DW_RING; ++i) g_dw.ring[i] = keep[i];
  g_dw.n = kept;
}
// called right after a dW GEMM (and its reduce) went to the side stream
void dw_forked(const void* workspace, size_t bytes) {
  if (g_dw.n == DW_RING) dw_join_all();
  DwPending& e = g_dw.ring[g_dw.n++];
  e.lo = (uintptr_t)workspace;
  e.hi = e.lo + bytes;
  (void)hipEventRecord(e.ev, g_dw.side);
}

// ---- deferred partial-sum reduces ----------------------------------------------------------------------------------------------
// Every weight-gradient GEMM ends in a fixed-order sum over its M-split partials (k_split_reduce), every LayerNorm backward in a column
// finish over its block partials: ~30 launches of 5-15 us per step that produce PARAMETER gradients only -- nothing before the
// optimizer / the gradient all-reduce reads them.  Inside a section (gt_defer_begin: the whole-model backward, csrc/model.hip) the
// producers put their partials into the section's arena (gt_defer_take) and queue the sum (gt_defer_push); gt_defer_flush runs all
// queued sums as ONE launch (k_defer_reduce: job list in the kernel arguments, same summation order as k_split_reduce -> same bits).
// The arena is never reused inside a section, so a flush may run on any stream that is ordered behind the producers.
struct DeferJob {
  const float *part, *part2;   // part[s * stride + i], s < nparts
  float *out, *out2;
  int64_t len, len2, stride, stride2;
  int64_t ostride;             // out[i * ostride] (1 = contiguous; the second array is always contiguous)
  int32_t nparts, block0;
};
constexpr int DEFER_MAX_JOBS = 44;   // 44 x 80 bytes of kernel arguments
struct DeferJobs {
  DeferJob j[DEFER_MAX_JOBS];
  int n;
};
struct DeferState {
  bool active = false;
  char* arena = nullptr;
  size_t cap = 0, used = 0;
  size_t max_take = 0;   // gt_defer_limit: requests above this many bytes are refused (0 = no limit)
  DeferJobs jobs{};
  int blocks = 0;
};
thread_local DeferState g_defer;

__global__ void __launch_bounds__(256) k_defer_reduce(DeferJobs J) {
  int ji = 0;
  for (int i = 1; i < J.n; ++i)
    if ((int)blockIdx.x >= J.j[i].block0) ji = i;
  const DeferJob& q = J.j[ji];
  const int nb = (ji + 1 < J.n ? J.j[ji + 1].block0 : (int)gridDim.x) - q.block0;
  for (int64_t i0 = (int64_t)((int)blockIdx.x - q.block0) * 256 + threadIdx.x; i0 < q.len + q.len2; i0 += (int64_t)nb * 256) {
    const bool second = i0 >= q.len;
    const float* p = second ? q.part2 : q.part;
    const int64_t st = second ? q.stride2 : q.stride, i = second ? i0 - q.len : i0;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 8 <= q.nparts; s += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += p[(int64_t)(s + u) * st + i];
    }
    for (; s < q.nparts; ++s) acc[0] += p[(int64_t)s * st + i];
    const float t = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    if (second) q.out2[i] = t;
    else q.out[i * q.ostride] = t;
  }
}

// the end of a weight-gradient GEMM: queue the sum over its partials (they live in the section's arena) or run it now
void dw_reduce(hipStream_t stream, bool deferred, const float* part, int splits, int64_t len, float* out, const float* part2, int64_t len2,
               float* out2) {
  if (deferred) {
    (void)gt_defer_push(part, splits, len, len, out, part2, len2, len2, out2);
    return;
  }
  const int rg = (int)(gt_cdiv(len + len2, 256) < 2048 ? gt_cdiv(len + len2, 256) : 2048);
  hipLaunchKernelGGL(k_split_reduce, dim3(rg, 1), dim3(256), 0, stream, part, splits, len, out, part2, len2, out2, (int64_t)0);
}
// its partial buffer: the open section's arena when it has room, else the caller's workspace
float* dw_part(float* ws_part, int splits, int64_t N, int64_t K, bool* deferred) {
  void* p = gt_defer_take((size_t)splits * (size_t)(N * K + N) * sizeof(float));
  *deferred = p != nullptr;
  return p ? (float*)p : ws_part;
}

int pick_bm(int64_t M, int64_t K = 0, bool fwd = false) {
  // 128-row tiles (half the blocks) measured 0.3-0.7 % slower end to end even for the 256-row GEMMs -- except the forward of a long
  // contraction over very many rows (the Erdos-Renyi stress' linear2, 131 k x 256 x 1024: every 64-row block re-stages the fp32
  // master weight, 2 GB of L2 reads for 335 MB of operands): 180 -> 144 us with 128 rows; its dX form gets slower (236 us)
  if (fwd && M >= 65536 && K >= 768) return 128;
  return 64;
}


// Short-M launches: a wave per tile pair when the tiles alone fill the chip, else the four waves of a block split the contraction
// of one tile pair (see small_combine in linear_small.h).
enum SmallKind { SMALL_FWD, SMALL_DX, SMALL_DW };
static inline void small_launch(SmallKind kind, int compute, int64_t tiles, int64_t contraction, hipStream_t stream, const SmallArgs& sa) {
  const bool sk = tiles <= 2048 && contraction >= 128;
  const unsigned blocks = (unsigned)(sk ? tiles : gt_cdiv(tiles, 4));
#define GT_SMALL_GO(KERNEL)                                                                                              \
  do {                                                                                                                    \
    if (compute == GT_F32) { if (sk) hipLaunchKernelGGL((KERNEL<float, 4>), dim3(blocks), dim3(256), 0, stream, sa);      \
                             else hipLaunchKernelGGL((KERNEL<float, 1>), dim3(blocks), dim3(256), 0, stream, sa); }       \
    else { if (sk) hipLaunchKernelGGL((KERNEL<gt_bf16, 4>), dim3(blocks), dim3(256), 0, stream, sa);                      \
           else hipLaunchKernelGGL((KERNEL<gt_bf16, 1>), dim3(blocks), dim3(256), 0, stream, sa); }                       \
  } while (0)
  if (kind == SMALL_FWD) GT_SMALL_GO(k_small_fwd);
  else if (kind == SMALL_DX) GT_SMALL_GO(k_small_dx);
  else GT_SMALL_GO(k_small_dw);
#undef GT_SMALL_GO
}


// ---- short-M path (linear_small.h): one wave per output tile pair, no LDS, no partials ---------------------------------------
// forward and dW of a short-M GEMM with a WIDE output (the 5 x 5002-way prediction heads, models/gnn_transformer.py:120-126: 256 rows
// x 25 010 columns): the one-wave-per-tile kernels have no limit on N (their work list just grows); only the dX form would walk N
// as its contraction and stays on the tiled split-N kernel
bool small_wide_ok(int x_dtype, int y_dtype, int64_t M, int64_t K, int64_t ldx, int64_t ldy, int groups) {
  return groups == 1 && x_dtype == GT_F32 && y_dtype == GT_F32 && M > 0 && M <= 512 && K <= 2048 && ldx % 4 == 0 && ldy % 4 == 0;
}
bool small_eligible(int x_dtype, int y_dtype, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldy, int groups) {
  return groups == 1 && x_dtype == GT_F32 && y_dtype == GT_F32 && M > 0 && M <= 512 && N <= 2048 && K <= 2048 && ldx % 4 == 0 && ldy % 4 == 0;
}

// ---- exact-fp32 wide-tile path (linear32.h): the big-M GEMMs of the message-passing side --------------------------------
constexpr int64_t W32_MIN_M = 1024;   // below this the grid of 64-row blocks cannot fill the chip: 128-wide tiles + splits

bool w32_eligible(int compute, int x_dtype, int64_t M, int groups) {
  return compute == GT_F32 && x_dtype == GT_F32 && groups == 1 && M >= W32_MIN_M;
}
// grouped launches of the bf16x6 kernel (k_lin3, blockIdx.y = group): fp32 rows in and out, 16-byte chunks of every row
bool g3_eligible(int compute, int x_dtype, int y_dtype, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldy) {
  return compute == GT_F32 && x_dtype == GT_F32 && y_dtype == GT_F32 && M >= W32_MIN_M && N % 4 == 0 && K % 4 == 0 && ldx % 4 == 0 &&
         ldy % 4 == 0;
}

template <typename TA, typename TO, bool MASK, bool GELU = false, bool BNS = false>
void w32_launch_nt(int nt, dim3 grid, hipStream_t stream, const L32Args& a) {
  switch (nt) {
    case 19: hipLaunchKernelGGL((k_lin32<TA, TO, 19, MASK, GELU, BNS>), grid, dim3(256), 0, stream, a); break;
    case 16: hipLaunchKernelGGL((k_lin32<TA, TO, 16, MASK, GELU, BNS>), grid, dim3(256), 0, stream, a); break;
    case 12: hipLaunchKernelGGL((k_lin32<TA, TO, 12, MASK, GELU, BNS>), grid, dim3(256), 0, stream, a); break;
    default: hipLaunchKernelGGL((k_lin32<TA, TO, 8, MASK, GELU, BNS>), grid, dim3(256), 0, stream, a); break;
  }
}

// out[M][Nout] = epilogue(A[M][Kc] W[Nout][Kc]^T): ta / to = storage of A / out
template <bool MASK>
void w32_launch(int ta, int to, hipStream_t stream, L32Args& a) {
  const int nt = w32_pick_nt(a.Nout);
  a.ncb = (int)gt_cdiv(gt_cdiv(a.Nout, 16), nt);
  dim3 grid((unsigned)(gt_cdiv(gt_cdiv(a.M, W32_BM), 8) * 8 * a.ncb));
  if constexpr (!MASK) {
    if (a.act == 2) {   // gelu epilogue: fp32 rows in and out only (w32_eligible_fwd)
      w32_launch_nt<float, float, false, true>(nt, grid, stream, a);
      return;
    }
  } else {
    if (a.bn_part) {    // BatchNorm-backward statistics in the dX epilogue: fp32 rows only (gt_linear_bwd_bnstats_ok)
      w32_launch_nt<float, float, true, false, true>(nt, grid, stream, a);
      return;
    }
  }
  if (ta == GT_F32 && to == GT_F32) w32_launch_nt<float, float, MASK>(nt, grid, stream, a);
  else if (ta == GT_F32) w32_launch_nt<float, gt_bf16, MASK>(nt, grid, stream, a);
  else if (to == GT_F32) w32_launch_nt<gt_bf16, float, MASK>(nt, grid, stream, a);
  else w32_launch_nt<gt_bf16, gt_bf16, MASK>(nt, grid, stream, a);
}

// M-splits of the wide-tile dW GEMM.  Alone on the chip it wants ~768 blocks (3 per CU is what its 48 KB of LDS admits; the
// dispatcher fills CUs greedily, 2 per CU leaves a third idle).  Forked onto the overlap stream (gt_overlap_dw_*) it shares
// the chip with the critical path: one block per CU leaves every CU room for the other streams' kernels (the 5-25 us kernels
// of the virtual-node chain ran 2-5 x longer beside an 800-block GEMM) and writes a third of the partials -- the GEMM
// itself takes longer, but nothing waits for it (Code2 +1.7 %, Molpcba +2 % end to end).
int w32_dw_splits(int64_t M, int nkb, int nnb, bool overlapped) {
  int64_t s = (overlapped ? 256 : 768) / ((int64_t)nkb * nnb);   // every split costs N*K*4 bytes of partials twice
  const int64_t maxs = gt_cdiv(M, 16 * 8);          // at least 8 stages per split
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  if (s > 256) s = 256;
  return (int)s;
}

template <typename TY, typename TX>
void w32_launch_dw_nt(int nt, dim3 grid, hipStream_t stream, const L32DwArgs& a) {
  switch (nt) {
    case 19: hipLaunchKernelGGL((k_lin32_dw<TY, TX, 19, true>), grid, dim3(256), 0, stream, a); break;
    case 16: hipLaunchKernelGGL((k_lin32_dw<TY, TX, 16, true>), grid, dim3(256), 0, stream, a); break;
    case 12: hipLaunchKernelGGL((k_lin32_dw<TY, TX, 12, true>), grid, dim3(256), 0, stream, a); break;
    default: hipLaunchKernelGGL((k_lin32_dw<TY, TX, 8, true>), grid, dim3(256), 0, stream, a); break;
  }
}

}  // namespace

extern "C" int gt_linear_fwd(int x_dtype, int y_dtype, int compute, const void* x, const float* weight,
                             const float* bias, void* y, int64_t M, int64_t N, int64_t K, int act, float dropout_p,
                             uint64_t seed, gt_stream_t stream_) {
  return gt_linear_fwd_ld(x_dtype, y_dtype, compute, x, weight, bias, y, M, N, K, N, act, dropout_p, seed, stream_);
}

extern "C" int gt_linear_fwd_ld(int x_dtype, int y_dtype, int compute, const void* x, const float* weight,
                                const float* bias, void* y, int64_t M, int64_t N, int64_t K, int64_t ldy, int act,
                                float dropout_p, uint64_t seed, gt_stream_t stream_) {
  return gt_linear_fwd_ld2(x_dtype, y_dtype, compute, x, weight, bias, y, M, N, K, K, ldy, act, dropout_p, seed, stream_);
}

extern "C" int gt_linear_fwd_ld2(int x_dtype, int y_dtype, int compute, const void* x, const float* weight,
                                 const float* bias, void* y, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldy,
                                 int act, float dropout_p, uint64_t seed, gt_stream_t stream_) {
  return gt_linear_fwd_grouped(x_dtype, y_dtype, compute, x, weight, bias, y, M, N, K, ldx, ldy, 1, 0, 0, act, dropout_p, seed,
                               stream_);
}

static int linear_fwd_impl(int x_dtype, int y_dtype, int compute, const void* x, const float* weight, const float* bias, void* y,
                           void* gout, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldy, int groups, int64_t x_group_stride,
                           int64_t y_group_stride, int act, float dropout_p, uint64_t seed, gt_stream_t stream_);

// The row operand / the dX of ONE call as two matrices side by side (the JK = "cat" concatenation without its copy): set by the
// *_cat2 entry points for the duration of their call (per host thread), honoured by the bf16x6 path only.
struct Cat2Req {
  const void* x2 = nullptr;   // forward / dW: contraction columns [split, K) of X
  void* dx2 = nullptr;        // dX: output columns [split, K)
  int64_t split = 0, ld2 = 0;
};
thread_local Cat2Req g_cat2;
struct Cat2Scope {
  ~Cat2Scope() { g_cat2 = Cat2Req{}; }
};

// A row map for ONE call (gt_linear_set_rows; per host thread, consumed by the next gt_linear_fwd* / gt_linear_bwd* call whatever its
// outcome): forward stores output row m at row rows[m] of y, backward reads row m of dY from row rows[m] of dy; -1 = no such row
// (nothing stored / zeros read).  Honoured by the bf16x6 path only: gt_linear_rows_ok.
thread_local const int32_t* g_rows = nullptr;
struct RowsLn {   // + a LayerNorm of the stored output row in the forward's epilogue (gt_linear_set_rows_layernorm)
  const float *w = nullptr, *b = nullptr;
  void* out = nullptr;
  float *mean = nullptr, *rstd = nullptr;
  float eps = 0.f;
};
thread_local RowsLn g_rows_ln;
struct RowsTake {   // takes the request out of the thread state on entry: it can never leak into a later call
  const int32_t* rows;
  RowsLn ln;
  RowsTake() : rows(g_rows), ln(g_rows_ln) { g_rows = nullptr; g_rows_ln = RowsLn{}; }
};
struct RowsClear {   // for entry points that can fail in front of the call that takes the request
  ~RowsClear() { g_rows = nullptr; g_rows_ln = RowsLn{}; }
};
static inline bool rows_eligible(int compute, int x_dtype, int y_dtype, const float* weight, int64_t M, int64_t N, int64_t K) {
  return w32_eligible(compute, x_dtype, M, 1) && (y_dtype == GT_F32 || N % 8 == 0) && w3_lookup(weight, N, K, false) && w3_lookup(weight, N, K, true);
}

extern "C" int gt_linear_fwd_grouped(int x_dtype, int y_dtype, int compute, const void* x, const float* weight,
                                     const float* bias, void* y, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldy,
                                     int groups, int64_t x_group_stride, int64_t y_group_stride, int act, float dropout_p,
                                     uint64_t seed, gt_stream_t stream_) {
  GT_CHECK_ARG(act == 0 || act == 1, "act must be 0 (none) or 1 (relu); gelu: gt_linear_fwd_gelu");
  GT_CHECK_ARG(dropout_p == 0.f || act == 1, "fused dropout requires a fused activation");
  return linear_fwd_impl(x_dtype, y_dtype, compute, x, weight, bias, y, nullptr, M, N, K, ldx, ldy, groups, x_group_stride,
                         y_group_stride, act, dropout_p, seed, stream_);
}

// Y = dropout(gelu(X W^T + b)); gmul (nullable, storage type / pitch of Y) = gelu'(z) * dropout scale: the multiplier
// gt_linear_bwd_mul takes in place of the ReLU path's forward output
extern "C" int gt_linear_fwd_gelu(int x_dtype, int y_dtype, int compute, const void* x, const float* weight, const float* bias,
                                  void* y, void* gmul, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldy, float dropout_p,
                                  uint64_t seed, gt_stream_t stream_) {
  return linear_fwd_impl(x_dtype, y_dtype, compute, x, weight, bias, y, gmul, M, N, K, ldx, ldy, 1, 0, 0, 2, dropout_p, seed, stream_);
}

static int linear_fwd_impl(int x_dtype, int y_dtype, int compute, const void* x, const float* weight, const float* bias, void* y,
                           void* gout, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldy, int groups, int64_t x_group_stride,
                           int64_t y_group_stride, int act, float dropout_p, uint64_t seed, gt_stream_t stream_) {
  const RowsTake rows__;
  GT_CHECK_ARG(!rows__.rows || (groups == 1 && rows_eligible(compute, x_dtype, y_dtype, weight, M, N, K)),
               "gt_linear_set_rows: this GEMM does not take a row map (ask gt_linear_rows_ok)");
  GT_CHECK_ARG(groups >= 1 && groups <= 65535, "1..65535 groups");
  GT_CHECK_ARG(groups == 1 || (x_group_stride % (x_dtype == GT_BF16 ? 8 : 4) == 0 && y_group_stride % (y_dtype == GT_BF16 ? 8 : 4) == 0 &&
                               (N * K) % 4 == 0),
               "group strides must keep 16-byte alignment");
  int rc = check_lin("gt_linear_fwd", x_dtype, y_dtype, compute, M, N, K, ldy);
  if (rc == GT_OK && (ldx < (g_cat2.x2 ? g_cat2.split : K) || ldx % (x_dtype == GT_BF16 ? 8 : 4))) {   // (cat2: the first matrix holds columns [0, split))
    gt_set_error("gt_linear_fwd: ldx (%lld) must be >= K and a multiple of 16 bytes", (long long)ldx);
    rc = GT_ERR_UNSUPPORTED;
  }
  if (rc) return rc;
  GT_CHECK_ARG(x && weight && y, "null buffer");
  GT_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "dropout_p must be in [0,1)");
  if (M == 0) return GT_OK;
  GtProfScope prof__(GT_PROF_LINEAR, "gt_linear_fwd", stream_, {M, N, K, x_dtype, y_dtype, compute});
  hipStream_t stream = (hipStream_t)stream_;
  LinArgs a{};
  a.a = x; a.w = weight; a.bias = bias; a.out = y; a.M = M; a.N = N; a.K = K; a.ldy = ldy; a.ldx = ldx; a.act = act; a.gout = gout;
  fill_drop(a, dropout_p, seed);
  a.g_x = x_group_stride; a.g_y = y_group_stride; a.g_w = N * K; a.g_b = N;
  if (heads_shape_ok(x_dtype, y_dtype, M, N, K, ldx, ldy, groups) && act == 0 && !a.thr && !gout) {   // the prediction heads (linear_heads.h)
    HeadsArgs h{};
    h.x = (const float*)x; h.w = weight; h.bias = bias; h.out = (float*)y; h.M = M; h.N = N; h.ldx = ldx; h.ldy = ldy;
    {
      GtProfScope pk__(GT_PROF_GEMM_KERNEL, "k_heads_fwd", stream, {M, N, K, x_dtype, y_dtype, compute});
      heads_launch_fwd(compute, stream, h);
    }
    GT_CHECK_LAUNCH();
    return GT_OK;
  }
  if (small_eligible(x_dtype, y_dtype, M, N, K, ldx, ldy, groups) || small_wide_ok(x_dtype, y_dtype, M, K, ldx, ldy, groups)) {
    SmallArgs sa{};
    sa.x = (const float*)x; sa.w = weight; sa.bias = bias; sa.out = (float*)y; sa.M = M; sa.N = N; sa.K = K; sa.ldx = ldx; sa.ldy = ldy;
    sa.act = act; sa.gout = (float*)gout; sa.inv_keep = a.inv_keep; sa.thr = a.thr; sa.s0 = a.s0; sa.s1 = a.s1;
    {
      GtProfScope pk__(GT_PROF_GEMM_KERNEL, "k_small_fwd", stream, {M, N, K, x_dtype, y_dtype, compute});
      small_launch(SMALL_FWD, compute, gt_cdiv(M, 16) * gt_cdiv(N, 32), K, stream, sa);
    }
    GT_CHECK_LAUNCH();
    return GT_OK;
  }
  if (groups > 1 && g3_eligible(compute, x_dtype, y_dtype, M, N, K, ldx, ldy) && act != 2 && !g_cat2.x2) {
    // grouped big-M fp32 GEMMs whose group weights all have bound images (the towers of PNAConv): k_lin3 with blockIdx.y = group
    int64_t spacing = 0;
    if (const void* img = w3_lookup_grouped(weight, N, K, groups, false, &spacing)) {
      L32Args w{};
      w.a = x; w.bias = bias; w.out = y; w.M = M; w.Nout = N; w.Kc = K; w.lda = ldx; w.ldw = K; w.ldo = ldy;
      w.act = act; w.inv_keep = a.inv_keep; w.thr = a.thr; w.s0 = a.s0; w.s1 = a.s1;
      w.w3 = img; w.groups = groups; w.g_a = x_group_stride; w.g_o = y_group_stride; w.g_img = spacing; w.g_b = (int)N;
      {
        GtProfScope pk__(GT_PROF_GEMM_KERNEL, "k_lin3[fwd]", stream, {M, N, K, x_dtype, y_dtype, compute});
        w3_launch<false>(x_dtype, y_dtype, stream, w);
      }
      GT_CHECK_LAUNCH();
      return GT_OK;
    }
  }
  if (w32_eligible(compute, x_dtype, M, groups) && (act != 2 || (x_dtype == GT_F32 && y_dtype == GT_F32))) {
    L32Args w{};
    w.a = x; w.w = weight; w.bias = bias; w.out = y; w.M = M; w.Nout = N; w.Kc = K; w.lda = ldx; w.ldw = K; w.ldo = ldy;
    w.act = act; w.gout = gout; w.inv_keep = a.inv_keep; w.thr = a.thr; w.s0 = a.s0; w.s1 = a.s1;
    // a prepared bf16x3 image of this weight (gt_w3_bind): the fp32-accurate GEMM on the bf16 matrix pipe (linear3x.h);
    // bf16 rows need K % 8 (their 16-byte chunks are whole k-groups)
    w.w3 = (x_dtype == GT_F32 || K % 8 == 0) ? w3_lookup(weight, N, K, false) : nullptr;
    if (g_cat2.x2) {
      if (!w.w3 || x_dtype != GT_F32) { gt_set_error("gt_linear_fwd_cat2: needs a bound weight image and fp32 rows"); return GT_ERR_UNSUPPORTED; }
      w.a2 = g_cat2.x2; w.a_split = g_cat2.split; w.lda2 = g_cat2.ld2;
    }
    if (rows__.rows) {
      if (!w.w3) { gt_set_error("gt_linear_set_rows: needs a bound weight image"); return GT_ERR_UNSUPPORTED; }
      w.out_rows = rows__.rows;
      if (rows__.ln.out) {
        if (w3_pick_nt(N) * 16 != N || act != 0 || a.thr || ldy != N) { gt_set_error("gt_linear_set_rows_layernorm: the row must fill one column block (ask gt_linear_rows_layernorm_ok)"); return GT_ERR_UNSUPPORTED; }
        w.ln_w = rows__.ln.w; w.ln_b = rows__.ln.b; w.ln_out = rows__.ln.out; w.ln_mean = rows__.ln.mean; w.ln_rstd = rows__.ln.rstd;
        w.ln_eps = rows__.ln.eps;
      }
    }
    {
      const bool on_rows_kernel = w.w3 && w3r_ok(x_dtype, y_dtype, w);
      GtProfScope pk__(GT_PROF_GEMM_KERNEL, on_rows_kernel ? "k_lin3r[fwd]" : (w.w3 ? "k_lin3[fwd]" : "k_lin32[fwd]"), stream, {M, N, K, x_dtype, y_dtype, compute});
      if (on_rows_kernel) w3r_launch(stream, w);   // rows straight into fragments (linear3r.h)
      else if (w.w3) w3_launch<false>(x_dtype, y_dtype, stream, w);
      else w32_launch<false>(x_dtype, y_dtype, stream, w);
    }
    GT_CHECK_LAUNCH();
    return GT_OK;
  }
  // bf16 rows in and out with a bound fragment-order image of this weight (gt_w1_bind): the weight-stationary kernel (linear1.h)
  if (x_dtype == GT_BF16 && y_dtype == GT_BF16 && compute == GT_BF16 && groups == 1 && act != 2 && M >= W1_MIN_M) {
    if (const void* img = w1_lookup(weight, N, K, false)) {
      L1Args l{};
      l.a = (const gt_bf16*)x; l.img = (const unsigned char*)img; l.bias = bias; l.out = (gt_bf16*)y;
      l.M = M; l.lda = ldx; l.ldo = ldy; l.N = (int)N; l.K = (int)K; l.act = act;
      l.inv_keep = a.inv_keep; l.thr = a.thr; l.s0 = a.s0; l.s1 = a.s1;
      // many rows, or a contraction the registers cannot hold: both operands through the LDS ring (linear2.h)
      const bool ring = w2_take(l, w1_pick_ntw(N, K) != 0);
      bool ok;
      {
        GtProfScope pk__(GT_PROF_GEMM_KERNEL, ring ? "k_lin2[fwd]" : "k_lin1[fwd]", stream, {M, N, K, x_dtype, y_dtype, compute});
        ok = ring ? w2_launch(stream, l) : w1_launch(stream, l);
      }
      if (ok) { GT_CHECK_LAUNCH(); return GT_OK; }
    }
  }
  const int bm = pick_bm(M, K, true);
  a.ntiles = (int)gt_cdiv(N, BN);
  dim3 grid((unsigned)(gt_cdiv(gt_cdiv(M, bm), 8) * 8 * a.ntiles), (unsigned)groups);
  const int t0 = x_dtype, t1 = y_dtype;
  {
    GtProfScope pk__(GT_PROF_GEMM_KERNEL, "k_linear_fwd", stream, {M, N, K, x_dtype, y_dtype, compute});
    if (bm == 64) GT_LIN_DISPATCH_BM(k_linear_fwd, 64, grid, a);
    else GT_LIN_DISPATCH_BM(k_linear_fwd, 128, grid, a);
  }
  GT_CHECK_LAUNCH();
  return GT_OK;
}

extern "C" size_t gt_linear_bwd_workspace_bytes(int compute, int64_t M, int64_t N, int64_t K) {
  if (compute == GT_F32 && M >= W32_MIN_M) {   // wide-tile fp32 path: [dW / db partials | W^T for the dX GEMM]
    const int nt = w32_pick_nt(N);
    const int splits = w32_dw_splits(M, (int)gt_cdiv(K, 64), (int)gt_cdiv(gt_cdiv(N, 16), nt), false);   // the larger of the two configurations
    return (size_t)splits * (size_t)(N * K + N) * sizeof(float) + (size_t)N * K * sizeof(float) + 512;
  }
  const size_t dw = (size_t)dw_splits(M, N, K, compute) * (size_t)(N * K + N) * sizeof(float);
  const int dxs = dx_splits(M, N, K, pick_bm(M));
  size_t dx = dxs > 1 ? (size_t)dxs * M * K * sizeof(float) : 0;
  if (heads_shape_ok(GT_F32, GT_F32, M, N, K, 4, 4, 1) && heads_dx_workspace_bytes(M, N) > dx) dx = heads_dx_workspace_bytes(M, N);
  return (dw > dx ? dw : dx) + 256;  // the dx partials are consumed before dw reuses the space
}

extern "C" int gt_linear_bwd(int x_dtype, int y_dtype, int compute, const void* x, const float* weight, const void* dy,
                             const void* y_for_mask, const void* dx_add1, const void* dx_add2, void* dx, float* dweight,
                             float* dbias, int64_t M, int64_t N, int64_t K, float dropout_p, void* workspace,
                             size_t workspace_bytes, gt_stream_t stream_) {
  return gt_linear_bwd_ld(x_dtype, y_dtype, compute, x, weight, dy, y_for_mask, dx_add1, dx_add2, dx, dweight, dbias, M, N, K,
                          N, dropout_p, workspace, workspace_bytes, stream_);
}

extern "C" int gt_linear_bwd_ld(int x_dtype, int y_dtype, int compute, const void* x, const float* weight, const void* dy,
                                const void* y_for_mask, const void* dx_add1, const void* dx_add2, void* dx, float* dweight,
                                float* dbias, int64_t M, int64_t N, int64_t K, int64_t ldy, float dropout_p,
                                void* workspace, size_t workspace_bytes, gt_stream_t stream_) {
  return gt_linear_bwd_ld2(x_dtype, y_dtype, compute, x, weight, dy, y_for_mask, dx_add1, dx_add2, dx, dweight, dbias, M, N, K,
                           K, ldy, dropout_p, workspace, workspace_bytes, stream_);
}

// Options of ONE gt_linear_bwd* call that the plain entry points do not carry in their signatures: the variant entry points
// below set them for the duration of their call to gt_linear_bwd_grouped (per host thread; cleared when that call returns).
struct BnStatsReq {   // BatchNorm-backward statistics to accumulate in the dX epilogue (gt_linear_bwd_bnstats)
  const float *x = nullptr, *mean = nullptr, *rstd = nullptr, *w = nullptr, *b = nullptr;
  float* part = nullptr;
  int64_t ldx = 0;
  int relu = 0;
};
struct BwdCallOpts {
  bool mul_mask = false;             // y_for_mask is a MULTIPLIER (gt_linear_bwd_mul), not a forward output
  bool fork_dw_only = false;         // dW-only call that may still go to the overlap stream (gt_linear_bwd_dw_forked)
  const float* weight_t = nullptr;   // W^T [K][N] prepared by the caller (gt_linear_bwd_wt): no transpose launch
  BnStatsReq bns;                    // set by gt_linear_bwd_bnstats BEFORE the call it applies to
  bool gate_out = false;             // y_for_mask [M][ldx] gates the dX OUTPUT (gt_linear_bwd_gate_out), dY is used as it is
  const float* bcast = nullptr;      // dX += bcast[bcast_idx[row]] (gt_linear_bwd_bcast), set BEFORE the call it applies to
  const int32_t* bcast_idx = nullptr;
};
thread_local BwdCallOpts g_opt;
struct BwdOptScope {   // whatever was set is dropped when the call it was meant for returns
  ~BwdOptScope() { g_opt = BwdCallOpts{}; }
};

extern "C" int gt_linear_bwd_ld2(int x_dtype, int y_dtype, int compute, const void* x, const float* weight, const void* dy,
                                 const void* y_for_mask, const void* dx_add1, const void* dx_add2, void* dx, float* dweight,
                                 float* dbias, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldy, float dropout_p,
                                 void* workspace, size_t workspace_bytes, gt_stream_t stream_) {
  return gt_linear_bwd_grouped(x_dtype, y_dtype, compute, x, weight, dy, y_for_mask, dx_add1, dx_add2, dx, dweight, dbias, M, N, K,
                               ldx, ldy, 1, 0, 0, dropout_p, workspace, workspace_bytes, stream_);
}

// The dX of the NEXT gt_linear_bwd* call on this thread is the dy of a BatchNorm (input bn_x [M][ldx], saved mean / rstd, affine
// w / b, ReLU behind it or not): its epilogue also writes the row-tile partial sums part[ceil(M/64)][2][K] of that BatchNorm's
// backward (sum dy', sum dy' * xhat) -- what k_bn_bwd_partial would compute from a second pass over dy and bn_x.  Only the
// exact-fp32 wide-tile path does this: ask gt_linear_bwd_bnstats_ok first; the request is dropped after the next call.
extern "C" int gt_linear_bwd_bnstats_ok(int compute, int x_dtype, int y_dtype, int64_t M) {
  return (w32_eligible(compute, x_dtype, M, 1) && x_dtype == GT_F32 && y_dtype == GT_F32) ? 1 : 0;
}
extern "C" int64_t gt_linear_bwd_bnstats_rows(int64_t M) { return gt_cdiv(M, W32_BM); }
// the register-row bf16x6 kernel (linear3r.h) takes a dX call with BatchNorm statistics in its epilogue: one partial row per 128-row block
// (only with gt_option_set("bnstats_rows_kernel", 1): measured slower than the separate partial pass -- Code2 74.0 k against 74.85 k
// graphs/s, round 5 -- and gt_linear_bwd_bnstats' documented partial layout is the exact kernel's 64-row tiles; kept as a tested
// alternative, tests/test_hip_options.py)
static bool bns_on_rows_kernel(int x_dtype, int y_dtype, const float* weight, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldy) {
  if (!gt_opt(GT_OPT_BNSTATS_ROWS_KERNEL)) return false;
  const void* img = w3_lookup(weight, N, K, true);
  if (!img || x_dtype != GT_F32 || y_dtype != GT_F32) return false;
  L32Args w{};
  w.w3 = img;
  w.M = M; w.Nout = K; w.Kc = N; w.lda = ldy; w.ldo = ldx;
  return w3r_ok(GT_F32, GT_F32, w);
}
// Partial rows the dX call (weight [N][K], M rows) would write for a gt_linear_bwd_bnstats request under the CURRENT bindings of this
// thread: ceil(M / 128) when the register-row bf16x6 kernel takes it (a bound image of W^T, M >= 12288), else gt_linear_bwd_bnstats_rows(M)
// (the exact-fp32 kernel); 0 = no kernel does it for this call.
extern "C" int64_t gt_linear_bwd_bnstats_rows_for(int compute, int x_dtype, int y_dtype, const float* weight, int64_t M, int64_t N, int64_t K) {
  if (!gt_linear_bwd_bnstats_ok(compute, x_dtype, y_dtype, M)) return 0;
  return bns_on_rows_kernel(x_dtype, y_dtype, weight, M, N, K, K, N) ? gt_cdiv(M, 128) : gt_cdiv(M, W32_BM);
}
extern "C" int gt_linear_bwd_bnstats(const float* bn_x, int64_t ldx, const float* mean, const float* rstd, const float* w,
                                     const float* b, int relu, float* part) {
  GT_CHECK_ARG(bn_x && mean && rstd && w && b && part && ldx > 0, "null buffer");
  BnStatsReq& q = g_opt.bns;
  q.x = bn_x; q.ldx = ldx; q.mean = mean; q.rstd = rstd; q.w = w; q.b = b; q.relu = relu; q.part = part;
  return GT_OK;
}

// The dX of the NEXT gt_linear_bwd* call on this thread gets a broadcast addend: dx[m] += rows[idx[m]] (rows [.][K] fp32, pitch = the
// dX pitch; idx int32 [M]) -- the gradient that reaches every node of a graph from the virtual-node update, d_t0[batch[m]]
// (modules/gnn_module.py:219), added in the dX GEMM's epilogue instead of being written out per node first.  Only the kernel with
// the rows in registers does this (csrc/linear3r.h): ask gt_linear_bwd_bcast_ok first; the request is dropped after the next call.
extern "C" int gt_linear_bwd_bcast_ok(int compute, int x_dtype, int y_dtype, const float* weight, int64_t M, int64_t N, int64_t K) {
  if (!w32_eligible(compute, x_dtype, M, 1) || x_dtype != GT_F32 || y_dtype != GT_F32 || !w3_lookup(weight, N, K, true)) return 0;
  L32Args w{};
  w.w3 = w3_lookup(weight, N, K, true);
  w.M = M; w.Nout = K; w.Kc = N; w.lda = N; w.ldo = K;
  return w3r_ok(GT_F32, GT_F32, w) ? 1 : 0;
}
extern "C" int gt_linear_bwd_bcast(const float* rows, const int32_t* idx) {
  GT_CHECK_ARG((rows == nullptr) == (idx == nullptr), "rows and idx go together");
  g_opt.bcast = rows;
  g_opt.bcast_idx = idx;
  return GT_OK;
}

// gt_linear_bwd with the transposed weight W^T [K][N] supplied by the caller (weights do not change during a backward pass: one
// gt_transpose per weight, off the critical path, replaces a transpose launch in front of every wide fp32 dX GEMM); NULL = as
// gt_linear_bwd.  Paths that do not need W^T ignore it.
extern "C" int gt_linear_bwd_wt(int x_dtype, int y_dtype, int compute, const void* x, const float* weight, const float* weight_t,
                                const void* dy, const void* y_for_mask, const void* dx_add1, const void* dx_add2, void* dx,
                                float* dweight, float* dbias, int64_t M, int64_t N, int64_t K, float dropout_p, void* workspace,
                                size_t workspace_bytes, gt_stream_t stream_) {
  g_opt.weight_t = weight_t;
  return gt_linear_bwd_grouped(x_dtype, y_dtype, compute, x, weight, dy, y_for_mask, dx_add1, dx_add2, dx, dweight, dbias, M, N, K, K, N,
                               1, 0, 0, dropout_p, workspace, workspace_bytes, stream_);
}
// out [K][N] = in [N][K]^T (fp32)
extern "C" int gt_transpose(const float* in, float* out, int64_t N, int64_t K, gt_stream_t stream_) {
  GT_CHECK_ARG(in && out && N > 0 && K > 0, "bad arguments");
  hipLaunchKernelGGL(k_transpose32, dim3((unsigned)gt_cdiv(K, 32), (unsigned)gt_cdiv(N, 32)), dim3(256), 0, (hipStream_t)stream_, in, out, N, K);
  GT_CHECK_LAUNCH();
  return GT_OK;
}

// dW / db only (dx == NULL), and inside a gt_overlap_dw section still on the overlap stream, ordered behind everything queued on
// `stream` so far: lets a caller start the weight gradient of a GEMM BEFORE its dX GEMM (the encoder layer's in_proj: a dW GEMM
// that starts together with the next layer's first kernel is what the fused backward avoids)
extern "C" int gt_linear_bwd_dw_forked(int x_dtype, int y_dtype, int compute, const void* x, const float* weight, const void* dy,
                                       const void* y_for_mask, float* dweight, float* dbias, int64_t M, int64_t N, int64_t K,
                                       int64_t ldx, int64_t ldy, float dropout_p, void* workspace, size_t workspace_bytes,
                                       gt_stream_t stream_) {
  g_opt.fork_dw_only = true;
  return gt_linear_bwd_grouped(x_dtype, y_dtype, compute, x, weight, dy, y_for_mask, nullptr, nullptr, nullptr, dweight, dbias, M, N, K,
                               ldx, ldy, 1, 0, 0, dropout_p, workspace, workspace_bytes, stream_);
}

// gt_linear_bwd_dw_forked for a layer whose forward saved a multiplier (gt_linear_fwd_gelu): dZ = dY * gmul
extern "C" int gt_linear_bwd_mul_dw_forked(int x_dtype, int y_dtype, int compute, const void* x, const float* weight, const void* dy,
                                           const void* gmul, float* dweight, float* dbias, int64_t M, int64_t N, int64_t K,
                                           int64_t ldx, int64_t ldy, void* workspace, size_t workspace_bytes, gt_stream_t stream_) {
  GT_CHECK_ARG(gmul, "gt_linear_bwd_mul_dw_forked needs the multiplier");
  g_opt.fork_dw_only = true;
  g_opt.mul_mask = true;
  return gt_linear_bwd_grouped(x_dtype, y_dtype, compute, x, weight, dy, gmul, nullptr, nullptr, nullptr, dweight, dbias, M, N, K, ldx,
                               ldy, 1, 0, 0, 0.f, workspace, workspace_bytes, stream_);
}

// backward of gt_linear_fwd_gelu: `gmul` is the multiplier that forward saved (dZ = dY * gmul); everything else as gt_linear_bwd_ld2
extern "C" int gt_linear_bwd_mul(int x_dtype, int y_dtype, int compute, const void* x, const float* weight, const void* dy,
                                 const void* gmul, const void* dx_add1, const void* dx_add2, void* dx, float* dweight,
                                 float* dbias, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldy, void* workspace,
                                 size_t workspace_bytes, gt_stream_t stream_) {
  GT_CHECK_ARG(gmul, "gt_linear_bwd_mul needs the multiplier");
  g_opt.mul_mask = true;
  return gt_linear_bwd_grouped(x_dtype, y_dtype, compute, x, weight, dy, gmul, dx_add1, dx_add2, dx, dweight, dbias, M, N, K, ldx, ldy,
                               1, 0, 0, 0.f, workspace, workspace_bytes, stream_);
}

extern "C" size_t gt_linear_bwd_grouped_workspace_bytes(int compute, int64_t M, int64_t N, int64_t K, int groups) {
  return (size_t)(groups < 1 ? 1 : groups) * gt_linear_bwd_workspace_bytes(compute, M, N, K);
}

extern "C" int gt_linear_bwd_grouped(int x_dtype, int y_dtype, int compute, const void* x, const float* weight, const void* dy,
                                     const void* y_for_mask, const void* dx_add1, const void* dx_add2, void* dx, float* dweight,
                                     float* dbias, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldy, int groups,
                                     int64_t x_group_stride, int64_t y_group_stride, float dropout_p, void* workspace,
                                     size_t workspace_bytes, gt_stream_t stream_) {
  BwdOptScope opt_scope__;   // the per-call options live for exactly this call
  if (g_opt.bcast && dx && !gt_linear_bwd_bcast_ok(compute, x_dtype, y_dtype, weight, M, N, K)) {
    gt_set_error("gt_linear_bwd_bcast: this call does not run on the register-row kernel (ask gt_linear_bwd_bcast_ok)");
    return GT_ERR_UNSUPPORTED;
  }
  const RowsTake rows__;
  GT_CHECK_ARG(!rows__.rows || (groups == 1 && !y_for_mask && !g_opt.bns.part && !rows__.ln.out && rows_eligible(compute, x_dtype, y_dtype, weight, M, N, K)),
               "gt_linear_set_rows: this GEMM does not take a row map (ask gt_linear_rows_ok)");
  GT_CHECK_ARG(groups >= 1 && groups <= 65535, "1..65535 groups");
  GT_CHECK_ARG(groups == 1 || (x_group_stride % (x_dtype == GT_BF16 ? 8 : 4) == 0 && y_group_stride % (y_dtype == GT_BF16 ? 8 : 4) == 0 &&
                               (N * K) % 4 == 0),
               "group strides must keep 16-byte alignment");
  int rc = check_lin("gt_linear_bwd", x_dtype, y_dtype, compute, M, N, K, ldy);
  if (rc == GT_OK && (ldx < (g_cat2.x2 ? g_cat2.split : K) || ldx % (x_dtype == GT_BF16 ? 8 : 4))) {   // (cat2: the first matrix holds columns [0, split))
    gt_set_error("gt_linear_bwd: ldx (%lld) must be >= K and a multiple of 16 bytes", (long long)ldx);
    rc = GT_ERR_UNSUPPORTED;
  }
  if (rc) return rc;
  GT_CHECK_ARG(weight && dy, "null buffer");
  GT_CHECK_ARG(dx || dweight, "nothing to compute");
  GT_CHECK_ARG(!dweight || x, "dweight needs x");
  GT_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "dropout_p must be in [0,1)");
  hipStream_t stream = (hipStream_t)stream_;
  GtProfScope prof__(GT_PROF_LINEAR, dx ? (dweight ? "gt_linear_bwd" : "gt_linear_bwd_dx") : "gt_linear_bwd_dw", stream_,
                     {M, N, K, x_dtype, y_dtype, compute});
  LinArgs a{};
  a.w = weight; a.a = dy; a.ymask = y_for_mask; a.x = x; a.M = M; a.N = N; a.K = K; a.ldy = ldy; a.ldx = ldx;
  a.add1 = dx_add1; a.add2 = dx_add2;
  a.inv_keep = g_opt.mul_mask ? 0.f : 1.0f / (1.0f - dropout_p);   // 0 = multiplier mode (gt_gate)
  a.g_x = x_group_stride; a.g_y = y_group_stride; a.g_w = N * K; a.g_b = N;
  if (M == 0) {
    if (dweight) (void)hipMemsetAsync(dweight, 0, (size_t)groups * N * K * sizeof(float), stream);
    if (dbias) (void)hipMemsetAsync(dbias, 0, (size_t)groups * N * sizeof(float), stream);
    return GT_OK;
  }
  const size_t need1 = gt_linear_bwd_workspace_bytes(compute, M, N, K);   // per group
  const size_t need = (size_t)groups * need1;
  a.g_part = (int64_t)(need1 / sizeof(float));
  if (small_eligible(x_dtype, y_dtype, M, N, K, ldx, ldy, groups)) {
    SmallArgs sa{};
    sa.x = (const float*)x; sa.w = weight; sa.dy = (const float*)dy; sa.ymask = (const float*)y_for_mask;
    sa.add1 = (const float*)dx_add1; sa.add2 = (const float*)dx_add2; sa.M = M; sa.N = N; sa.K = K; sa.ldx = ldx; sa.ldy = ldy;
    sa.inv_keep = a.inv_keep;
    if (dx) {
      sa.out = (float*)dx;
      GtProfScope pk__(GT_PROF_GEMM_KERNEL, "k_small_dx", stream, {M, N, K, x_dtype, y_dtype, compute});
      small_launch(SMALL_DX, compute, gt_cdiv(M, 16) * gt_cdiv(K, 32), N, stream, sa);
    }
    if (dweight) {
      // forked only with a workspace to book it under: the kernel itself needs none, but gt_overlap_dw_release(range) is how the
      // caller learns when the dy / mask this GEMM reads (they live in the caller's workspace) may be overwritten
      const bool forked = g_dw.active && stream == g_dw.main && (dx || g_opt.fork_dw_only) && workspace && workspace_bytes &&
                          !(gt_prof_mask() & GT_PROF_LINEAR);
      if (forked) {
        (void)hipEventRecord(g_dw.ev_fork, stream);
        (void)hipStreamWaitEvent(g_dw.side, g_dw.ev_fork, 0);
        stream = g_dw.side;
      }
      sa.out = dweight; sa.db = dbias;
      {
        GtProfScope pk__(GT_PROF_GEMM_KERNEL, "k_small_dw", stream, {M, N, K, x_dtype, y_dtype, compute});
        small_launch(SMALL_DW, compute, gt_cdiv(N, 32) * gt_cdiv(K, 16), M, stream, sa);
      }
      if (forked) dw_forked(workspace, workspace_bytes);
    }
    GT_CHECK_LAUNCH();
    return GT_OK;
  }
  bool dx_done = false;
  if (dx && groups > 1 && g3_eligible(compute, x_dtype, y_dtype, M, N, K, ldx, ldy) && !g_opt.bns.part && !g_opt.bcast && !g_opt.gate_out &&
      !g_cat2.dx2 && !rows__.rows) {
    // dX of a grouped big-M fp32 GEMM on the bound images of the groups' W^T (k_lin3, blockIdx.y = group)
    int64_t spacing = 0;
    if (const void* img = w3_lookup_grouped(weight, N, K, groups, true, &spacing)) {
      L32Args w{};
      w.a = dy; w.amask = y_for_mask; w.out = dx; w.add1 = dx_add1; w.add2 = dx_add2;
      w.M = M; w.Nout = K; w.Kc = N; w.lda = ldy; w.ldw = N; w.ldo = ldx; w.inv_keep = a.inv_keep;
      w.w3 = img; w.groups = groups; w.g_a = y_group_stride; w.g_o = x_group_stride; w.g_img = spacing;
      GtProfScope pk__(GT_PROF_GEMM_KERNEL, "k_lin3[dx]", stream, {M, N, K, x_dtype, y_dtype, compute});
      w3_launch<true>(y_dtype, x_dtype, stream, w);
      dx_done = true;
    }
  }
  if (dweight && groups > 1 && g3_eligible(compute, x_dtype, y_dtype, M, N, K, ldx, ldy) && !g_cat2.x2 && !rows__.rows && workspace &&
      workspace_bytes >= need) {
    // the groups' weight gradients on the pipelined bf16x6 kernel (k_lin3r_dw, blockIdx.y = group) when their weights are bound
    int64_t spacing = 0;
    L32DwArgs d{};
    d.dy = dy; d.ymask = y_for_mask; d.x = x; d.inv_keep = a.inv_keep;
    d.M = M; d.N = N; d.K = K; d.ldy = ldy; d.ldx = ldx;
    d.groups = groups; d.g_y = y_group_stride; d.g_x = x_group_stride;
    if (w3_lookup_grouped(weight, N, K, groups, false, &spacing) && w3r_dw_ok(y_dtype, x_dtype, d)) {
      const bool will_fork = (g_dw.active && stream == g_dw.main && (dx || g_opt.fork_dw_only) && !(gt_prof_mask() & GT_PROF_LINEAR));
      const bool forked = will_fork;
      if (forked) {
        (void)hipEventRecord(g_dw.ev_fork, stream);
        (void)hipStreamWaitEvent(g_dw.side, g_dw.ev_fork, 0);
      }
      // (a local: a dX that falls through to the generic kernels below must stay on the caller's stream -- ADVICE r5)
      hipStream_t dw_stream = forked ? g_dw.side : stream;
      const int shape = w3r_dw_pick_shape(N, K);
      const int nkb3 = (int)gt_cdiv(K, w3r_dw_xt(shape)), nnb3 = (int)gt_cdiv(N, w3r_dw_zt(shape));
      int s3 = w3_dw_splits(M, nkb3 * nnb3 * groups);
      const int cap = w32_dw_splits(M, (int)gt_cdiv(K, 64), (int)gt_cdiv(gt_cdiv(N, 16), w32_pick_nt(N)), false);   // a group's share of the workspace holds this many partial copies
      if (s3 > cap) s3 = cap;
      const int64_t per = (int64_t)s3 * (N * K + N);
      // (reduced right behind the GEMM, not deferred to the end of the backward: the PNA driver gathers the image gradients before that)
      const bool deferred = false;
      float* base = reinterpret_cast<float*>(workspace);
      d.g_part = deferred ? per : a.g_part;
      d.part = base; d.dbpart = dbias ? base + (int64_t)s3 * N * K : nullptr;
      d.splits = s3; d.nkb = nkb3; d.nnb = nnb3;
      d.m_per_split = gt_cdiv(gt_cdiv(M, s3), 32) * 32;
      dim3 grid3((unsigned)(gt_cdiv(s3, 8) * 8 * nkb3 * nnb3), (unsigned)groups);
      {
        {   // (the bracket holds the GEMM kernel alone: one rocprofv3 row; its fixed-order reduce is k_split_reduce's row)
          GtProfScope pk__(GT_PROF_GEMM_KERNEL, "k_lin3r_dw", dw_stream, {M, N, K, x_dtype, y_dtype, compute});
          w3r_launch_dw(grid3, dw_stream, d, shape);
        }
        if (deferred) {
          for (int g = 0; g < groups; ++g)
            (void)gt_defer_push(base + g * per, s3, N * K, N * K, dweight + (int64_t)g * N * K, dbias ? base + g * per + (int64_t)s3 * N * K : nullptr,
                                dbias ? N : 0, dbias ? N : 0, dbias ? dbias + (int64_t)g * N : nullptr);
        } else {
          const int64_t len = N * K, len2 = dbias ? N : 0;
          const int rg = (int)(gt_cdiv(len + len2, 256) < 2048 ? gt_cdiv(len + len2, 256) : 2048);
          hipLaunchKernelGGL(k_split_reduce, dim3(rg, groups), dim3(256), 0, dw_stream, (const float*)base, s3, len, dweight, (const float*)d.dbpart,
                             len2, dbias, d.g_part);
        }
      }
      if (forked) dw_forked(workspace, workspace_bytes);
      dweight = nullptr; dbias = nullptr;   // done
      if (!dx || dx_done) { GT_CHECK_LAUNCH(); return GT_OK; }
    }
  }
  if (w32_eligible(compute, x_dtype, M, groups)) {
    if (!workspace || workspace_bytes < need) {
      gt_set_error("gt_linear_bwd: workspace too small (%zu < %zu)", workspace_bytes, need);
      return GT_ERR_WORKSPACE;
    }
    const int nt = w32_pick_nt(N);
    const int nkb = (int)gt_cdiv(K, 64), nnb = (int)gt_cdiv(gt_cdiv(N, 16), nt);
    const bool will_fork = (g_dw.active && stream == g_dw.main && (dx || g_opt.fork_dw_only) && dweight && !(gt_prof_mask() & GT_PROF_LINEAR));
    const int splits = w32_dw_splits(M, nkb, nnb, will_fork && !g_dw.urgent);
    float* part = reinterpret_cast<float*>(workspace);
    float* wt = part + (size_t)w32_dw_splits(M, nkb, nnb, false) * (size_t)(N * K + N) + 64;   // behind the larger partial area
    wt = reinterpret_cast<float*>(((uintptr_t)wt + 255) & ~(uintptr_t)255);
    if (dx) {   // dX = dZ (W^T)^T: the forward-form kernel on the transposed weight
      // W^T lives in the caller's workspace: a previous call's dW GEMM may still be running on the overlap stream with
      // its partials in the same workspace (callers hand ONE workspace to consecutive GEMMs, e.g. the four of an encoder
      // layer) -> wait for the forks that used this range (those on other workspaces keep running).
      const void* w3t = (y_dtype != GT_F32 && N % 8) ? nullptr : w3_lookup(weight, N, K, true);
      if (w3t && g_opt.bns.part && !bns_on_rows_kernel(x_dtype, y_dtype, weight, M, N, K, ldx, ldy)) w3t = nullptr;   // the exact kernel's epilogue then
      if (w3t) {
        wt = nullptr;   // the bound image of W^T: k_lin3, no transpose
      } else if (g_opt.weight_t) {
        wt = const_cast<float*>(g_opt.weight_t);   // read only
      } else {
        if (g_dw.active && stream == g_dw.main) dw_release(workspace, workspace_bytes);
        hipLaunchKernelGGL(k_transpose32, dim3((unsigned)gt_cdiv(K, 32), (unsigned)gt_cdiv(N, 32)), dim3(256), 0, stream, weight, wt, N, K);
      }
      L32Args w{};
      w.a = dy; w.amask = y_for_mask; w.w = wt; w.out = dx; w.add1 = dx_add1; w.add2 = dx_add2;
      w.M = M; w.Nout = K; w.Kc = N; w.lda = ldy; w.ldw = N; w.ldo = ldx; w.inv_keep = a.inv_keep;
      if (g_opt.bns.part && x_dtype == GT_F32 && y_dtype == GT_F32) {
        const BnStatsReq& q = g_opt.bns;
        w.bn_x = q.x; w.bn_ldx = q.ldx; w.bn_mean = q.mean; w.bn_rstd = q.rstd; w.bn_w = q.w; w.bn_b = q.b;
        w.bn_relu = q.relu; w.bn_part = q.part;
      }
      w.w3 = w3t;
      if (g_opt.bcast) {
        w.add_bc = g_opt.bcast; w.add_bidx = g_opt.bcast_idx;
        if (!w3t || !w3r_ok(y_dtype, x_dtype, w)) { gt_set_error("gt_linear_bwd_bcast: this call does not run on the register-row kernel (ask gt_linear_bwd_bcast_ok)"); return GT_ERR_UNSUPPORTED; }
      }
      if (rows__.rows) {
        if (!w3t) { gt_set_error("gt_linear_set_rows: needs the bound image of W^T"); return GT_ERR_UNSUPPORTED; }
        w.a_rows = rows__.rows;
      }
      if (g_cat2.dx2) {
        if (!w3t || dx_add1 || dx_add2) { gt_set_error("gt_linear_bwd_cat2: needs a bound weight image and no addends"); return GT_ERR_UNSUPPORTED; }
        w.out2 = g_cat2.dx2; w.out_split = g_cat2.split; w.ldo2 = g_cat2.ld2;
      }
      const bool on_rows_kernel = w3t && w3r_ok(y_dtype, x_dtype, w);
      GtProfScope pk__(GT_PROF_GEMM_KERNEL, on_rows_kernel ? "k_lin3r[dx]" : (w3t ? "k_lin3[dx]" : "k_lin32[dx]"), stream, {M, N, K, x_dtype, y_dtype, compute});
      if (on_rows_kernel) w3r_launch(stream, w);
      else if (w3t) w3_launch<true>(y_dtype, x_dtype, stream, w);
      else w32_launch<true>(y_dtype, x_dtype, stream, w);
    }
    if (dweight) {
      const bool forked = will_fork;
      if (forked) {
        (void)hipEventRecord(g_dw.ev_fork, stream);
        (void)hipStreamWaitEvent(g_dw.side, g_dw.ev_fork, 0);
        stream = g_dw.side;
      }
      L32DwArgs d{};
      d.dy = dy; d.ymask = y_for_mask; d.x = x; d.inv_keep = a.inv_keep;
      d.M = M; d.N = N; d.K = K; d.ldy = ldy; d.ldx = ldx;
      if (g_cat2.x2) { d.x2 = g_cat2.x2; d.x_split = g_cat2.split; d.ldx2 = g_cat2.ld2; }
      // weights with bound images run the bf16x6 dW kernel too (160 x 160 output tiles, split over M; linear3x.h)
      const bool split3 = x_dtype == GT_F32 && w3_lookup(weight, N, K, false) != nullptr && (y_dtype == GT_F32 || N % 8 == 0);
      if (rows__.rows && !split3) { gt_set_error("gt_linear_set_rows: needs the bound weight image"); return GT_ERR_UNSUPPORTED; }
      d.dy_rows = rows__.rows;
      if (split3) {
        const int nkb3 = (int)gt_cdiv(K, W3D_T), nnb3 = (int)gt_cdiv(N, W3D_T);
        int s3 = w3_dw_splits(M, nkb3 * nnb3);
        const int cap = w32_dw_splits(M, nkb, nnb, false);   // the workspace is sized for this many partial copies
        if (s3 > cap) s3 = cap;
        bool deferred;
        float* part3 = dw_part(part, s3, N, K, &deferred);
        d.part = part3; d.dbpart = dbias ? part3 + (size_t)s3 * N * K : nullptr;
        d.splits = s3; d.nkb = nkb3; d.nnb = nnb3;
        d.m_per_split = gt_cdiv(gt_cdiv(M, s3), 32) * 32;
        dim3 grid3((unsigned)(gt_cdiv(s3, 8) * 8 * nkb3 * nnb3));
        {
          const bool pipelined = w3r_dw_ok(y_dtype, x_dtype, d);
          {
            GtProfScope pk__(GT_PROF_GEMM_KERNEL, pipelined ? "k_lin3r_dw" : "k_lin3_dw", stream, {M, N, K, x_dtype, y_dtype, compute});
            if (pipelined) w3r_launch_dw(grid3, stream, d);   // stages pipelined (linear3r.h)
            else if (y_dtype == GT_F32) w3_launch_dw<float, float>(grid3, stream, d);
            else w3_launch_dw<gt_bf16, float>(grid3, stream, d);
          }
          dw_reduce(stream, deferred, part3, s3, N * K, dweight, d.dbpart, dbias ? N : 0, dbias);
        }
        if (forked) dw_forked(workspace, workspace_bytes);
        GT_CHECK_LAUNCH();
        return GT_OK;
      }
      bool deferred;
      float* part2w = dw_part(part, splits, N, K, &deferred);
      d.part = part2w; d.dbpart = dbias ? part2w + (size_t)splits * N * K : nullptr;
      d.splits = splits; d.nkb = nkb; d.nnb = nnb;
      d.m_per_split = gt_cdiv(gt_cdiv(M, splits), 16) * 16;
      dim3 grid((unsigned)(gt_cdiv(splits, 8) * 8 * nkb * nnb));
      {
        {
          GtProfScope pk__(GT_PROF_GEMM_KERNEL, "k_lin32_dw", stream, {M, N, K, x_dtype, y_dtype, compute});
          if (y_dtype == GT_F32) w32_launch_dw_nt<float, float>(nt, grid, stream, d);
          else w32_launch_dw_nt<gt_bf16, float>(nt, grid, stream, d);
        }
        dw_reduce(stream, deferred, part2w, splits, N * K, dweight, d.dbpart, dbias ? N : 0, dbias);
      }
      if (forked) dw_forked(workspace, workspace_bytes);
    }
    GT_CHECK_LAUNCH();
    return GT_OK;
  }
  if (dx && x_dtype == GT_BF16 && y_dtype == GT_BF16 && compute == GT_BF16 && groups == 1 && M >= W1_MIN_M && (!y_for_mask || g_opt.gate_out)) {
    // dX = dY W on the bound image of W^T (linear1.h); a gate here applies to the OUTPUT columns (gt_linear_bwd_gate_out)
    if (const void* img = w1_lookup(weight, N, K, true)) {
      L1Args l{};
      l.a = (const gt_bf16*)dy; l.img = (const unsigned char*)img; l.out = (gt_bf16*)dx;
      l.gate = g_opt.gate_out ? (const gt_bf16*)y_for_mask : nullptr; l.gate_inv_keep = a.inv_keep;
      l.add1 = (const gt_bf16*)dx_add1; l.add2 = (const gt_bf16*)dx_add2;
      l.M = M; l.lda = ldy; l.ldo = ldx; l.N = (int)K; l.K = (int)N;
      const bool ring = w2_take(l, w1_pick_ntw(K, N) != 0);
      {
        GtProfScope pk__(GT_PROF_GEMM_KERNEL, ring ? "k_lin2[dx]" : "k_lin1[dx]", stream, {M, N, K, x_dtype, y_dtype, compute});
        dx_done = ring ? w2_launch(stream, l) : w1_launch(stream, l);
      }
    }
  }
  if (g_opt.gate_out) {
    if (dx && !dx_done) { gt_set_error("gt_linear_bwd_gate_out: not covered (ask gt_linear_bwd_gate_out_ok)"); return GT_ERR_UNSUPPORTED; }
    a.ymask = nullptr;   // the gate belongs to the dX output, not to dY: the weight gradient below reads dY as it is
    y_for_mask = nullptr;
  }
  // (shape matching alone must not route a general GEMM with per-call options here: k_heads_dx knows none of them)
  if (dx && !dx_done && !y_for_mask && !g_opt.mul_mask && !g_opt.bns.part && !g_opt.gate_out && dropout_p == 0.f &&
      heads_shape_ok(x_dtype, y_dtype, M, N, K, ldx, ldy, groups) && workspace &&
      workspace_bytes >= heads_dx_workspace_bytes(M, N)) {   // the prediction heads: the contraction is their 25 010 columns (linear_heads.h)
    HeadsArgs h{};
    h.w = weight; h.dy = (const float*)dy; h.M = M; h.N = N; h.ldx = ldx; h.ldy = ldy;
    GtProfScope pk__(GT_PROF_GEMM_KERNEL, "k_heads_dx+reduce", stream, {M, N, K, x_dtype, y_dtype, compute});
    heads_launch_dx(compute, stream, h, workspace, (float*)dx, (const float*)dx_add1, (const float*)dx_add2);
    dx_done = true;
  }
  if (dx && !dx_done) {
    const int bm = pick_bm(M);
    // split-N partials are plain fp32 sums: only for fp32 dX without fused addends (and not for grouped launches)
    int splits = (x_dtype == GT_F32 && !dx_add1 && !dx_add2 && ldx == K && groups == 1) ? dx_splits(M, N, K, bm) : 1;
    if (splits > 1 && (!workspace || workspace_bytes < need)) splits = 1;
    a.splits = splits;
    a.n_per_split = gt_cdiv(gt_cdiv(N, splits), 64) * 64;
    a.out = splits > 1 ? workspace : dx;
    a.ntiles = (int)gt_cdiv(K, BN);
    dim3 grid((unsigned)(gt_cdiv(gt_cdiv(M, bm), 8) * 8 * a.ntiles), (unsigned)groups, (unsigned)splits);
    const int t0 = y_dtype, t1 = x_dtype;
    GtProfScope pk__(GT_PROF_GEMM_KERNEL, "k_linear_dx", stream, {M, N, K, x_dtype, y_dtype, compute});
    if (bm == 64) GT_LIN_DISPATCH_BM(k_linear_dx, 64, grid, a);
    else GT_LIN_DISPATCH_BM(k_linear_dx, 128, grid, a);
    if (splits > 1) {
      const int64_t len = M * K;
      const int rg = (int)(gt_cdiv(len, 256) < 2048 ? gt_cdiv(len, 256) : 2048);
      hipLaunchKernelGGL(k_split_reduce, dim3(rg), dim3(256), 0, stream, (const float*)workspace, splits, len,
                         reinterpret_cast<float*>(dx), (const float*)nullptr, (int64_t)0, (float*)nullptr, (int64_t)0);
    }
  }
  if (dweight && small_wide_ok(x_dtype, y_dtype, M, K, ldx, ldy, groups)) {   // wide short-M GEMM: dW on the one-wave-per-tile kernel
    const bool forked = g_dw.active && stream == g_dw.main && (dx || g_opt.fork_dw_only) && a.splits <= 1 && workspace && workspace_bytes &&
                        !(gt_prof_mask() & GT_PROF_LINEAR);
    if (forked) {
      (void)hipEventRecord(g_dw.ev_fork, stream);
      (void)hipStreamWaitEvent(g_dw.side, g_dw.ev_fork, 0);
      stream = g_dw.side;
    }
    SmallArgs sa{};
    sa.x = (const float*)x; sa.w = weight; sa.dy = (const float*)dy; sa.ymask = (const float*)y_for_mask;
    sa.M = M; sa.N = N; sa.K = K; sa.ldx = ldx; sa.ldy = ldy; sa.inv_keep = a.inv_keep;
    sa.out = dweight; sa.db = dbias;
    {
      GtProfScope pk__(GT_PROF_GEMM_KERNEL, "k_small_dw", stream, {M, N, K, x_dtype, y_dtype, compute});
      small_launch(SMALL_DW, compute, gt_cdiv(N, 32) * gt_cdiv(K, 16), M, stream, sa);
    }
    if (forked) dw_forked(workspace, workspace_bytes);
    GT_CHECK_LAUNCH();
    return GT_OK;
  }
  if (dweight) {
    const int splits = dw_splits(M, N, K, compute);
    if (!workspace || workspace_bytes < need) {
      gt_set_error("gt_linear_bwd: workspace too small (%zu < %zu)", workspace_bytes, need);
      return GT_ERR_WORKSPACE;
    }
    // dW is off the critical path of the backward (only the optimizer reads it): inside a
    // gt_overlap_dw_begin/_end section it runs on the side stream beside dX and whatever follows.
    // (not while the launch profiler brackets this call: its events sit on the caller's stream only)
    bool forked = false;
    if (g_dw.active && stream == g_dw.main && (dx || g_opt.fork_dw_only) && a.splits <= 1 && !(gt_prof_mask() & GT_PROF_LINEAR)) {
      (void)hipEventRecord(g_dw.ev_fork, stream);
      (void)hipStreamWaitEvent(g_dw.side, g_dw.ev_fork, 0);
      stream = g_dw.side;
      forked = true;
    }
    if (dw16_ok(x_dtype, y_dtype, compute, x, dy, a.ymask, M, N, K, ldx, ldy, groups)) {   // bf16 token rows: the LDS-DMA ring kernel (linear_dw16.h)
      Dw16Args d{};
      d.dy = (const gt_bf16*)dy; d.x = (const gt_bf16*)x; d.M = M; d.N = N; d.K = K; d.ldy = ldy; d.ldx = ldx;
      d.splits = dw16_splits(M, N, K, splits);
      d.m_per_split = gt_cdiv(gt_cdiv(M, d.splits), D16_ROWS) * D16_ROWS;
      bool deferred;
      d.part = dw_part(reinterpret_cast<float*>(workspace), d.splits, N, K, &deferred);
      d.dbpart = dbias ? d.part + (size_t)d.splits * N * K : nullptr;
      d.ntx = (int)(N / D16_T);
      d.ntiles = d.ntx * (int)(K / D16_T);
      {
        {
          GtProfScope pk__(GT_PROF_GEMM_KERNEL, "k_dw16", stream, {M, N, K, x_dtype, y_dtype, compute});
          hipLaunchKernelGGL(k_dw16, dim3((unsigned)(gt_cdiv(d.splits, 8) * 8 * d.ntiles)), dim3(256), 0, stream, d);
        }
        dw_reduce(stream, deferred, d.part, d.splits, N * K, dweight, d.dbpart, dbias ? N : 0, dbias);
      }
      if (forked) dw_forked(workspace, workspace_bytes);
      GT_CHECK_LAUNCH();
      return GT_OK;
    }
    const int64_t bmc = compute == GT_BF16 ? 64 : 32;
    a.splits = splits;
    a.m_per_split = gt_cdiv(gt_cdiv(M, splits), bmc) * bmc;
    a.out = workspace;
    a.dbpart = dbias ? reinterpret_cast<float*>(workspace) + (size_t)splits * N * K : nullptr;
    a.ntx = (int)gt_cdiv(N, BN);
    a.ntiles = a.ntx * (int)gt_cdiv(K, BN);
    dim3 grid((unsigned)(gt_cdiv(splits, 8) * 8 * a.ntiles), (unsigned)groups);
    const int t0 = y_dtype, t1 = x_dtype;
    {
      {
        GtProfScope pk__(GT_PROF_GEMM_KERNEL, "k_linear_dw", stream, {M, N, K, x_dtype, y_dtype, compute});
        GT_LIN_DISPATCH(k_linear_dw, grid, a);
      }
      const int64_t len = N * K, len2 = dbias ? N : 0;
      int rg = (int)(gt_cdiv(len + len2, 256) < 2048 ? gt_cdiv(len + len2, 256) : 2048);
      hipLaunchKernelGGL(k_split_reduce, dim3(rg, groups), dim3(256), 0, stream, (const float*)workspace, splits, len, dweight,
                         (const float*)a.dbpart, len2, dbias, a.g_part);
    }
    if (forked) dw_forked(workspace, workspace_bytes);
  }
  GT_CHECK_LAUNCH();
  return GT_OK;
}

// ---- JK = "cat" without the copy: X = [X1 | X2] (modules/gnn_module.py:104-105 feeding models/gnn_transformer.py:92) -------------
// 1 when gt_linear_fwd_cat2 / gt_linear_bwd_cat2 can run this GEMM (big-M exact-fp32 compute on a weight whose images are bound)
extern "C" int gt_linear_cat2_ok(int compute, const float* weight, int64_t M, int64_t N, int64_t K1, int64_t K2) {
  return (w32_eligible(compute, GT_F32, M, 1) && K1 > 0 && K2 > 0 && K1 % 4 == 0 && K2 % 4 == 0 && w3_lookup(weight, N, K1 + K2, false) &&
          w3_lookup(weight, N, K1 + K2, true)) ? 1 : 0;
}
// ---- a row map on the output (forward) / on dY (backward): gnn2transformer writing and reading the Transformer's token rows in place
// (models/gnn_transformer.py:92-96, modules/utils.py:5-29: no pad / unpad pass over the node rows) -----------------------------------
extern "C" int gt_linear_rows_ok(int compute, int x_dtype, int y_dtype, const float* weight, int64_t M, int64_t N, int64_t K) {
  return (rows_eligible(compute, x_dtype, y_dtype, weight, M, N, K)) ? 1 : 0;
}
extern "C" int gt_linear_set_rows(const int32_t* rows) {
  g_rows = rows;
  return GT_OK;
}
// ... and LayerNorm(ln_w, ln_b, eps) of every stored row in the same epilogue (forward only): ln_out (storage type and pitch of y) gets
// the normalised row, ln_mean / ln_rstd its statistics, all at the row the output goes to.  N must fill one column block of the kernel.
extern "C" int gt_linear_rows_layernorm_ok(int64_t N) { return (N > 0 && N % 16 == 0 && (int64_t)w3_pick_nt(N) * 16 == N) ? 1 : 0; }
extern "C" int gt_linear_set_rows_layernorm(const int32_t* rows, const float* ln_w, const float* ln_b, float eps, void* ln_out,
                                            float* ln_mean, float* ln_rstd) {
  GT_CHECK_ARG(rows && ln_w && ln_b && ln_out && ln_mean && ln_rstd, "null buffer");
  g_rows = rows;
  g_rows_ln.w = ln_w; g_rows_ln.b = ln_b; g_rows_ln.out = ln_out; g_rows_ln.mean = ln_mean; g_rows_ln.rstd = ln_rstd; g_rows_ln.eps = eps;
  return GT_OK;
}
// Y[M][N] = [X1 | X2] W^T + b with X1 [M][K1] (pitch ldx1), X2 [M][K2] (pitch ldx2), W [N][K1 + K2]; fp32 rows, y_dtype fp32 / bf16
extern "C" int gt_linear_fwd_cat2(int y_dtype, int compute, const void* x1, int64_t K1, int64_t ldx1, const void* x2, int64_t K2,
                                  int64_t ldx2, const float* weight, const float* bias, void* y, int64_t M, int64_t N, int64_t ldy,
                                  gt_stream_t stream_) {
  RowsClear rows_clear__;
  GT_CHECK_ARG(x1 && x2 && K1 > 0 && K2 > 0 && K1 % 4 == 0 && K2 % 4 == 0 && ldx2 >= K2 && ldx2 % 4 == 0, "bad second operand");
  Cat2Scope scope__;
  g_cat2.x2 = x2; g_cat2.split = K1; g_cat2.ld2 = ldx2;
  return linear_fwd_impl(GT_F32, y_dtype, compute, x1, weight, bias, y, nullptr, M, N, K1 + K2, ldx1, ldy, 1, 0, 0, 0, 0.f, 0, stream_);
}
// backward of the above: dX1 [M][K1] (pitch lddx1) and dX2 [M][K2] (pitch lddx2) from dY [M][N] (y_dtype, pitch ldy); dW [N][K1+K2], db
extern "C" int gt_linear_bwd_cat2(int y_dtype, int compute, const void* x1, int64_t K1, int64_t ldx1, const void* x2, int64_t K2,
                                  int64_t ldx2, const float* weight, const void* dy, void* dx1, int64_t lddx1, void* dx2, int64_t lddx2,
                                  float* dweight, float* dbias, int64_t M, int64_t N, int64_t ldy, void* workspace,
                                  size_t workspace_bytes, gt_stream_t stream_) {
  RowsClear rows_clear__;
  GT_CHECK_ARG(x1 && x2 && dx1 && dx2 && K1 > 0 && K2 > 0 && K1 % 4 == 0 && K2 % 4 == 0, "bad operands");
  GT_CHECK_ARG(ldx1 == lddx1 && ldx2 == lddx2, "dX pitches must equal the X pitches");   // (one pitch per matrix in the kernel arguments)
  Cat2Scope scope__;
  g_cat2.x2 = x2; g_cat2.dx2 = dx2; g_cat2.split = K1; g_cat2.ld2 = ldx2;
  return gt_linear_bwd_grouped(GT_F32, y_dtype, compute, x1, weight, dy, nullptr, nullptr, nullptr, dx1, dweight, dbias, M, N, K1 + K2,
                               ldx1, ldy, 1, 0, 0, 0.f, workspace, workspace_bytes, stream_);
}

// ---- bf16x3 weight images (linear3x.h) --------------------------------------------------------------------------------
// bytes of the image of a weight used as [rows][contraction] (forward: rows = N, contraction = K; dX form: rows = K, contraction = N)
extern "C" size_t gt_w3_image_bytes(int64_t rows, int64_t contraction) {
  return rows > 0 && contraction > 0 ? w3_image_bytes(rows, contraction) : 0;
}
// Builds `n` images in as few launches as the kernel-argument table allows (64 jobs each).  Job i: weight[i] is the fp32 matrix
// [N[i]][K[i]] (row pitch K[i]); transposed[i] == 0 -> image of W (rows N, contraction K) for the forward, != 0 -> image of W^T (rows K,
// contraction N) for the dX GEMM; image[i] has gt_w3_image_bytes(rows, contraction) bytes, 1024-byte aligned.
extern "C" int gt_w3_images(int n, const float* const* weight, const int64_t* N, const int64_t* K, const int* transposed,
                            void* const* image, gt_stream_t stream_) {
  GT_CHECK_ARG(n >= 0 && (n == 0 || (weight && N && K && transposed && image)), "bad arguments");
  for (int i0 = 0; i0 < n; i0 += W3_MAX_JOBS) {
    W3Jobs jobs{};
    int blocks = 0;
    const int cnt = n - i0 < W3_MAX_JOBS ? n - i0 : W3_MAX_JOBS;
    for (int i = 0; i < cnt; ++i) {
      const int s = i0 + i;
      GT_CHECK_ARG(weight[s] && image[s] && N[s] > 0 && K[s] > 0 && ((uintptr_t)image[s] & 1023) == 0, "null / unaligned buffer");
      W3Job& J = jobs.j[i];
      const int64_t rows = transposed[s] ? K[s] : N[s], contr = transposed[s] ? N[s] : K[s];
      J.w = weight[s]; J.img = (unsigned char*)image[s]; J.R = (int)rows; J.C = (int)contr; J.ldw = (int)K[s];
      J.transposed = transposed[s] ? 1 : 0; J.ntp = (int)w3_ntp(rows); J.ksteps = (int)gt_cdiv(contr, 32); J.block0 = blocks;
      blocks += J.ksteps * ((J.ntp + 3) / 4);
    }
    jobs.n = cnt;
    hipLaunchKernelGGL(k_w3_image, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, jobs);
  }
  GT_CHECK_LAUNCH();
  return GT_OK;
}
// Binds prepared images to weight pointers for the calling HOST THREAD: until gt_w3_unbind(), every big-M exact-fp32 GEMM
// (compute == GT_F32, M >= 1024, one group) of gt_linear_fwd* / gt_linear_bwd* whose weight (pointer, N, K) is in the table runs
// the bf16x6 kernel on the image (forward: image_fwd[i], dX: image_t[i]; a NULL entry keeps the exact-fp32 MFMA kernel for that
// direction).  The images must stay valid and current (rebuilt after every optimizer step) while bound.  At most 64 entries.
extern "C" int gt_w3_bind(int n, const float* const* weight, const int64_t* N, const int64_t* K, const void* const* image_fwd,
                          const void* const* image_t) {
  GT_CHECK_ARG(n >= 0 && n <= W3_MAX_BOUND && (n == 0 || (weight && N && K)), "at most 64 bound weights");
  for (int i = 0; i < n; ++i)
    g_w3.e[i] = W3Bound{weight[i], N[i], K[i], image_fwd ? image_fwd[i] : nullptr, image_t ? image_t[i] : nullptr};
  g_w3.n = n;
  return GT_OK;
}
extern "C" int gt_w3_unbind(void) {
  g_w3.n = 0;
  return GT_OK;
}

// ---- fragment-order bf16 images for the encoder layers' GEMMs (linear1.h) ------------------------------------------------
// 0 when the (rows, contraction) GEMM is not covered by the weight-stationary kernel (rows % 64, contraction % 128, <= 1024)
extern "C" size_t gt_w1_image_bytes(int64_t rows, int64_t contraction) {
  return (w1_pick_ntw(rows, contraction) || w2_covered(rows, contraction)) ? w1_image_bytes(rows, contraction) : 0;
}
// Job i as gt_w3_images: weight[i] = fp32 [N[i]][K[i]]; transposed[i] == 0 -> image of W (rows N, contraction K), != 0 -> image of W^T;
// image[i] has gt_w1_image_bytes(rows, contraction) bytes (non-zero), 16-byte aligned.
extern "C" int gt_w1_images(int n, const float* const* weight, const int64_t* N, const int64_t* K, const int* transposed,
                            void* const* image, gt_stream_t stream_) {
  GT_CHECK_ARG(n >= 0 && (n == 0 || (weight && N && K && transposed && image)), "bad arguments");
  for (int i0 = 0; i0 < n; i0 += W1_MAX_JOBS) {
    W1Jobs jobs{};
    int blocks = 0;
    const int cnt = n - i0 < W1_MAX_JOBS ? n - i0 : W1_MAX_JOBS;
    for (int i = 0; i < cnt; ++i) {
      const int s = i0 + i;
      GT_CHECK_ARG(weight[s] && image[s] && N[s] > 0 && K[s] > 0 && ((uintptr_t)image[s] & 15) == 0, "null / unaligned buffer");
      const int64_t rows = transposed[s] ? K[s] : N[s], contr = transposed[s] ? N[s] : K[s];
      GT_CHECK_ARG(w1_pick_ntw(rows, contr) != 0 || w2_covered(rows, contr), "shape not covered (gt_w1_image_bytes == 0)");
      W1Job& J = jobs.j[i];
      J.w = weight[s]; J.img = (unsigned char*)image[s]; J.R = (int)rows; J.C = (int)contr; J.ldw = (int)K[s];
      J.transposed = transposed[s] ? 1 : 0; J.block0 = blocks;
      blocks += (int)gt_cdiv((rows / 16) * (contr / 32), 4);
    }
    jobs.n = cnt;
    hipLaunchKernelGGL(k_w1_image, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, jobs);
  }
  GT_CHECK_LAUNCH();
  return GT_OK;
}
// Binds images for the calling HOST THREAD: until gt_w1_unbind(), gt_linear_fwd* / gt_linear_bwd* calls with bf16 rows in and out,
// bf16 compute, one group, no GELU and a weight (pointer, N, K) in the table run the weight-stationary kernel (forward: image_fwd[i];
// dX: image_t[i], when the call has no gate on dY; NULL keeps the tiled kernel for that direction).  Images must be current.
extern "C" int gt_w1_bind(int n, const float* const* weight, const int64_t* N, const int64_t* K, const void* const* image_fwd,
                          const void* const* image_t) {
  GT_CHECK_ARG(n >= 0 && n <= W1_MAX_BOUND && (n == 0 || (weight && N && K)), "at most 64 bound weights");
  for (int i = 0; i < n; ++i)
    g_w1.e[i] = W1Bound{weight[i], N[i], K[i], image_fwd ? image_fwd[i] : nullptr, image_t ? image_t[i] : nullptr};
  g_w1.n = n;
  return GT_OK;
}
extern "C" int gt_w1_unbind(void) {
  g_w1.n = 0;
  return GT_OK;
}
// gt_linear_bwd_ld2 whose gate (`y_or_mul` [M][ldx]: the forward output of the layer BELOW when dropout_p >= 0 -- dZ = dX * 1[y > 0]
// / (1 - p) -- or, with dropout_p < 0, a saved multiplier) applies to the dX OUTPUT of this call; dY is used as it is (also by the
// weight gradient).  Only on the weight-stationary path: ask gt_linear_bwd_gate_out_ok first.
extern "C" int gt_linear_bwd_gate_out_ok(int x_dtype, int y_dtype, int compute, const float* weight, int64_t M, int64_t N, int64_t K) {
  return (x_dtype == GT_BF16 && y_dtype == GT_BF16 && compute == GT_BF16 && M >= W1_MIN_M && w1_lookup(weight, N, K, true) &&
          w1_pick_ntw(K, N)) ? 1 : 0;
}
extern "C" int gt_linear_bwd_gate_out(int x_dtype, int y_dtype, int compute, const void* x, const float* weight, const void* dy,
                                      const void* y_or_mul, const void* dx_add1, const void* dx_add2, void* dx, float* dweight,
                                      float* dbias, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldy, float dropout_p,
                                      void* workspace, size_t workspace_bytes, gt_stream_t stream_) {
  GT_CHECK_ARG(y_or_mul && dx, "gt_linear_bwd_gate_out needs the gate tensor and dx");
  g_opt.gate_out = true;
  g_opt.mul_mask = dropout_p < 0.f;
  return gt_linear_bwd_grouped(x_dtype, y_dtype, compute, x, weight, dy, y_or_mul, dx_add1, dx_add2, dx, dweight, dbias, M, N, K, ldx, ldy,
                               1, 0, 0, dropout_p < 0.f ? 0.f : dropout_p, workspace, workspace_bytes, stream_);
}

// y = LayerNorm(resid + dropout(a)) * ln_w + ln_b with a = x W^T + b, as ONE launch: the GEMM's epilogue holds whole rows (N = the
// LayerNorm dim in one column block), saves a (the LayerNorm backward re-reads it), mean and rstd exactly as gt_linear_fwd followed by
// gt_layernorm_fwd(a, resid, ...) would (post-norm nn.TransformerEncoderLayer: x = norm1(x + dropout1(sa)), x = norm2(x + dropout2(ff));
// modules/transformer_encoder.py:28-32).  Only on the weight-stationary path: gt_linear_layernorm_fwd_ok says so (callers fall back to
// the two calls).
extern "C" int gt_linear_layernorm_fwd_ok(int dtype, int compute, const float* weight, int64_t M, int64_t N, int64_t K) {
  return (dtype == GT_BF16 && compute == GT_BF16 && M >= W1_MIN_M && w1_ln_covered(N, K) && w1_lookup(weight, N, K, false)) ? 1 : 0;
}
extern "C" int gt_linear_layernorm_fwd(int dtype, int compute, const void* x, const float* weight, const float* bias, void* a_out,
                                       int64_t M, int64_t N, int64_t K, const void* resid, const float* ln_weight, const float* ln_bias,
                                       float eps, float dropout_p, uint64_t seed, void* y, float* save_mean, float* save_rstd,
                                       gt_stream_t stream_) {
  GT_CHECK_ARG(x && weight && a_out && ln_weight && ln_bias && y && save_mean && save_rstd, "null buffer");
  GT_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "dropout_p must be in [0,1)");
  if (!gt_linear_layernorm_fwd_ok(dtype, compute, weight, M, N, K)) {
    gt_set_error("gt_linear_layernorm_fwd: not covered (ask gt_linear_layernorm_fwd_ok)");
    return GT_ERR_UNSUPPORTED;
  }
  if (M == 0) return GT_OK;
  hipStream_t stream = (hipStream_t)stream_;
  LinArgs d{};
  fill_drop(d, dropout_p, seed);
  L1Args l{};
  l.a = (const gt_bf16*)x; l.img = (const unsigned char*)w1_lookup(weight, N, K, false); l.bias = bias; l.out = (gt_bf16*)a_out;
  l.M = M; l.lda = K; l.ldo = N; l.N = (int)N; l.K = (int)K;
  l.ln_resid = (const gt_bf16*)resid; l.ln_w = ln_weight; l.ln_b = ln_bias; l.ln_out = (gt_bf16*)y; l.ln_mean = save_mean; l.ln_rstd = save_rstd;
  l.ln_eps = eps; l.ln_inv_keep = d.inv_keep; l.ln_thr = d.thr; l.ln_s0 = d.s0; l.ln_s1 = d.s1;
  bool ok;
  {
    GtProfScope pk__(GT_PROF_GEMM_KERNEL, "k_lin1[fwd+ln]", stream, {M, N, K, dtype, dtype, compute});
    ok = w1_launch(stream, l);
  }
  if (!ok) { gt_set_error("gt_linear_layernorm_fwd: launch set-up failed"); return GT_ERR_LAUNCH; }
  GT_CHECK_LAUNCH();
  return GT_OK;
}

// dX of a Linear whose OUTPUT gradient is at once the gradient of the LayerNorm below it (post-norm nn.TransformerEncoderLayer backward,
// modules/transformer_encoder.py:28-32: linear1's dX + the residual gradient -> norm1's backward; in_proj's dX + the residual gradient
// -> the previous layer's norm2): g = dY W (+ add1 + add2) never reaches memory, the GEMM's epilogue holds whole rows and runs
// gt_layernorm_bwd's arithmetic on them -- d_sub = d(sub-layer output), d_resid = d(residual branch), and the LayerNorm's weight / bias
// gradient through block partials in `workspace` (or the open deferred-reduce section).  weight [N][K], K = the LayerNorm dim.
// Only on the weight-stationary path: gt_linear_bwd_dx_layernorm_ok says so (callers fall back to gt_linear_bwd + gt_layernorm_bwd).
extern "C" int gt_linear_bwd_dx_layernorm_ok(int dtype, int compute, const float* weight, int64_t M, int64_t N, int64_t K) {
  return (dtype == GT_BF16 && compute == GT_BF16 && M >= W1_MIN_M && w1_lnb_covered(K, N) && w1_lookup(weight, N, K, true)) ? 1 : 0;
}
extern "C" size_t gt_linear_bwd_dx_layernorm_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  return (size_t)w1_grid_blocks(M, K, N) * 2 * (size_t)K * sizeof(float) + 256;
}
extern "C" int gt_linear_bwd_dx_layernorm(int dtype, int compute, const float* weight, const void* dy, const void* dx_add1, const void* dx_add2,
                                          int64_t M, int64_t N, int64_t K, const void* ln_x, const void* ln_resid, const float* ln_weight,
                                          const float* save_mean, const float* save_rstd, float dropout_p, uint64_t seed, void* d_sub,
                                          void* d_resid, float* ln_dweight, float* ln_dbias, void* workspace, size_t workspace_bytes,
                                          gt_stream_t stream_) {
  GT_CHECK_ARG(weight && dy && ln_x && ln_weight && save_mean && save_rstd && ln_dweight && ln_dbias, "null buffer");
  GT_CHECK_ARG(d_sub || d_resid, "nothing to compute");
  GT_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "dropout_p must be in [0,1)");
  if (!gt_linear_bwd_dx_layernorm_ok(dtype, compute, weight, M, N, K)) {
    gt_set_error("gt_linear_bwd_dx_layernorm: not covered (ask gt_linear_bwd_dx_layernorm_ok)");
    return GT_ERR_UNSUPPORTED;
  }
  if (M == 0) {
    (void)hipMemsetAsync(ln_dweight, 0, (size_t)K * sizeof(float), (hipStream_t)stream_);
    (void)hipMemsetAsync(ln_dbias, 0, (size_t)K * sizeof(float), (hipStream_t)stream_);
    return GT_OK;
  }
  hipStream_t stream = (hipStream_t)stream_;
  const int blocks = w1_grid_blocks(M, K, N);
  const size_t pbytes = (size_t)blocks * 2 * (size_t)K * sizeof(float);
  float* part = (float*)gt_defer_take(pbytes);   // inside a deferred-reduce section: the column sums join its one launch
  const bool deferred = part != nullptr;
  if (!deferred) {
    if (!workspace || workspace_bytes < pbytes) { gt_set_error("gt_linear_bwd_dx_layernorm: workspace too small"); return GT_ERR_WORKSPACE; }
    if (g_dw.active && stream == g_dw.main) dw_release(workspace, workspace_bytes);   // a forked GEMM may still read partials there
    part = (float*)workspace;
  }
  LinArgs d{};
  fill_drop(d, dropout_p, seed);
  L1Args l{};
  l.a = (const gt_bf16*)dy; l.img = (const unsigned char*)w1_lookup(weight, N, K, true);
  l.add1 = (const gt_bf16*)dx_add1; l.add2 = (const gt_bf16*)dx_add2;
  l.M = M; l.lda = N; l.ldo = K; l.N = (int)K; l.K = (int)N;
  l.lnb_x = (const gt_bf16*)ln_x; l.ln_resid = (const gt_bf16*)ln_resid; l.ln_w = ln_weight;
  l.ln_mean = const_cast<float*>(save_mean); l.ln_rstd = const_cast<float*>(save_rstd);
  l.ln_inv_keep = d.inv_keep; l.ln_thr = d.thr; l.ln_s0 = d.s0; l.ln_s1 = d.s1;
  l.lnb_dsub = (gt_bf16*)d_sub; l.lnb_dresid = (gt_bf16*)d_resid; l.lnb_part = part;
  bool ok;
  {
    GtProfScope pk__(GT_PROF_GEMM_KERNEL, "k_lin1[dx+lnb]", stream, {M, N, K, dtype, dtype, compute});
    ok = w1_launch(stream, l);
  }
  if (!ok) { gt_set_error("gt_linear_bwd_dx_layernorm: launch set-up failed"); return GT_ERR_LAUNCH; }
  GT_CHECK_LAUNCH();
  if (deferred) return gt_defer_push(part, blocks, K, 2 * K, ln_dweight, part + K, K, 2 * K, ln_dbias);
  return gt_layernorm_bwd_finish(part, blocks, K, ln_dweight, ln_dbias, stream_);
}

// ---- overlap section -------------------------------------------------------------------------------
extern "C" int gt_defer_begin(void* arena, size_t bytes) {
  g_defer.active = arena && bytes > 0;
  g_defer.arena = (char*)arena;
  g_defer.cap = bytes;
  g_defer.used = 0;
  g_defer.max_take = 0;
  g_defer.jobs.n = 0;
  g_defer.blocks = 0;
  return GT_OK;
}
// Only partial buffers of at most `max_take_bytes` join the open section (0 = every size): the big batches keep the weight-gradient
// GEMMs' immediate reduces (their partials stay in the reused, cache-resident workspaces) and defer the LayerNorms' column sums
extern "C" int gt_defer_limit(size_t max_take_bytes) {
  g_defer.max_take = max_take_bytes;
  return GT_OK;
}
extern "C" void* gt_defer_take(size_t bytes) {
  if (!g_defer.active || g_defer.jobs.n >= DEFER_MAX_JOBS || (g_defer.max_take && bytes > g_defer.max_take)) return nullptr;
  const size_t need = (bytes + 255) & ~(size_t)255;
  if (g_defer.used + need > g_defer.cap) return nullptr;
  void* p = g_defer.arena + g_defer.used;
  g_defer.used += need;
  return p;
}
extern "C" int gt_defer_room(int jobs) { return (g_defer.active && g_defer.jobs.n + jobs <= DEFER_MAX_JOBS) ? 1 : 0; }
extern "C" int gt_defer_push_strided(const float* part, int nparts, int64_t len, int64_t stride, float* out, int64_t ostride) {
  GT_CHECK_ARG(ostride >= 1, "bad output stride");
  const int rc = gt_defer_push(part, nparts, len, stride, out, nullptr, 0, 0, nullptr);
  if (rc) return rc;
  g_defer.jobs.j[g_defer.jobs.n - 1].ostride = ostride;
  return GT_OK;
}
extern "C" int gt_defer_push(const float* part, int nparts, int64_t len, int64_t stride, float* out, const float* part2, int64_t len2,
                             int64_t stride2, float* out2) {
  GT_CHECK_ARG(g_defer.active && g_defer.jobs.n < DEFER_MAX_JOBS, "no open section / job list full (a gt_defer_take came first?)");
  GT_CHECK_ARG(part && out && nparts >= 1 && len > 0 && (len2 == 0 || (part2 && out2)), "bad job");
  DeferJob& q = g_defer.jobs.j[g_defer.jobs.n++];
  q.part = part; q.part2 = len2 ? part2 : nullptr; q.out = out; q.out2 = len2 ? out2 : nullptr;
  q.len = len; q.len2 = len2; q.stride = stride; q.stride2 = stride2; q.nparts = nparts; q.ostride = 1;
  q.block0 = g_defer.blocks;
  const int64_t nb = gt_cdiv(len + len2, 256);
  g_defer.blocks += (int)(nb < 512 ? nb : 512);
  return GT_OK;
}
extern "C" int gt_defer_flush(gt_stream_t stream_) {
  if (!g_defer.active || g_defer.jobs.n == 0) return GT_OK;
  hipLaunchKernelGGL(k_defer_reduce, dim3((unsigned)g_defer.blocks), dim3(256), 0, (hipStream_t)stream_, g_defer.jobs);
  g_defer.jobs.n = 0;
  g_defer.blocks = 0;
  GT_CHECK_LAUNCH();
  return GT_OK;
}
extern "C" int gt_defer_end(void) {
  GT_CHECK_ARG(!g_defer.active || g_defer.jobs.n == 0, "queued sums were never flushed");
  g_defer.active = false;
  return GT_OK;
}

extern "C" int gt_overlap_dw_begin(gt_stream_t main_, gt_stream_t side_) {
  GT_CHECK_ARG(side_ && main_ != side_, "need a distinct side stream");
  if (!g_dw.ev_fork) {
    if (hipEventCreateWithFlags(&g_dw.ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g_dw.ev_join, hipEventDisableTiming) != hipSuccess) {
      gt_set_error("gt_overlap_dw_begin: event creation failed");
      return GT_ERR_LAUNCH;
    }
    for (int i = 0; i < DW_RING; ++i)
      if (hipEventCreateWithFlags(&g_dw.ring[i].ev, hipEventDisableTiming) != hipSuccess) {
        gt_set_error("gt_overlap_dw_begin: event creation failed");
        return GT_ERR_LAUNCH;
      }
  }
  g_dw.main = (hipStream_t)main_;
  g_dw.side = (hipStream_t)side_;
  g_dw.active = true;
  g_dw.urgent = false;
  g_dw.n = 0;
  return GT_OK;
}
extern "C" int gt_overlap_dw_sync(void) {
  if (!g_dw.active) return GT_OK;
  dw_join_all();
  return GT_OK;
}
// For the other entry points whose LAST launch only produces parameter gradients (the column finish of a LayerNorm backward,
// the partial reduce of an aggregate backward): inside an overlap section on the section's main stream, returns the overlap
// stream after ordering it behind everything queued on `stream` so far; otherwise `stream` itself.  A launch that went to the
// overlap stream is then booked with gt_overlap_dw_booked(workspace it reads, bytes) so that gt_overlap_dw_release covers it.
extern "C" gt_stream_t gt_overlap_dw_fork(gt_stream_t stream, unsigned prof_category) {
  // (not while the launch profiler brackets the caller's category: its events sit on the caller's stream only)
  if (!g_dw.active || (hipStream_t)stream != g_dw.main || (gt_prof_mask() & prof_category)) return stream;
  (void)hipEventRecord(g_dw.ev_fork, g_dw.main);
  (void)hipStreamWaitEvent(g_dw.side, g_dw.ev_fork, 0);
  return (gt_stream_t)g_dw.side;
}
extern "C" void gt_overlap_dw_booked(const void* workspace, size_t bytes) {
  if (g_dw.active) dw_forked(workspace, bytes);
}
extern "C" int gt_overlap_dw_urgent(int on) {
  g_dw.urgent = on != 0;
  return GT_OK;
}
extern "C" int gt_overlap_dw_release(const void* workspace, size_t bytes) {
  dw_release(workspace, bytes);
  return GT_OK;
}
extern "C" int gt_overlap_dw_end(void) {
  const int rc = gt_overlap_dw_sync();
  g_dw.active = false;
  return rc;
}
