// linear3x.h — fp32-accurate GEMMs on the bf16 matrix pipe ("bf16x6": three-way bf16 split of both operands, six products)
// for the big-M linears of the message-passing side.  Included by linear.hip inside its anonymous namespace, after linear32.h.
//
// Why: v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 MFMA rate; the exact-fp32 kernels of linear32.h are MFMA-bound at
// 0.47-0.50 of THAT peak (75 us for 31.6 k x 300 x 300) and were 25 % of the Code2 step's kernel time, 39 % of Molpcba's.
// An fp32 value splits EXACTLY into three bf16 values (8 + 8 + 8 significand bits, a = a1 + a2 + a3 with
// a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2): both differences are exact in fp32), and a product of two bf16 values is
// exact in fp32.  Keeping the six products of order <= 2^-16 (a1 b1; a1 b2, a2 b1; a1 b3, a2 b2, a3 b1) drops terms of relative size
// 2^-24 and below -- the rounding of ONE fp32 multiply -- so the result differs from an exact-fp32 GEMM by accumulation order
// only (measured: relative L2 to float64 1.4e-7 against 3.1e-7 for torch's fp32 GEMM on 4096 x 300 x 300; tests/test_hip_linear3x.py
// holds it to the same bar as the exact kernels).  Six v_mfma_f32_16x16x32_bf16 (6 x 16 cycles per 16x16x32 product block)
// replace eight v_mfma_f32_16x16x4_f32 (8 x 32 cycles): 2.7 x the matrix-pipe ceiling at fp32 accuracy.
// Non-finite inputs: inf - inf = NaN in the split, i.e. an inf activation yields NaN where the exact kernel yields inf.
//
// Weights are split ONCE per step into an "image" (k_w3_image, all weights of a model in one launch): bf16 planes laid out in
// the order the LDS wants them -- [k-step of 32][plane][16-row tile][1 KB: row r, 16-byte chunk (c ^ (r >> 1 & 3))] -- so a
// k-step's tiles reach the LDS by global_load_lds_dwordx4 (LDS-DMA, lane-linear, no staging registers, no ds_write) and the
// fragment reads (ds_read_b128 at row n, chunk g ^ (n >> 1 & 3)) are bank-conflict free in all four 16-lane service groups.
// The dX form runs the same kernel on the image of W^T (built by the same launch, transposed indexing): no transpose pass.
// The activation rows are split while they are staged (3 v_cvt_pk_bf16_f32 + 4 v_sub_f32 per pair).
//
// Block = 64 rows x NT n-tiles (NT = 10 or 8: N = 300 -> two column blocks of 160), 4 waves as 2 x 2, wave = 32 rows x NT/2
// n-tiles; W stage double-buffered (2 x 3 x NT KB), A stage single (12 KB): 72 KB -> two blocks per CU, one block's barriers and
// DMA waits hide under the other's MFMAs.
#pragma once

#ifndef W3_ABL
#define W3_ABL 0   // compile-time ablation mask of tools/gemm3_probe.hip (1 no DMA, 2 no row loads, 4 no MFMA, 8 no stores); 0 in the library
#endif

static inline int w3_pick_nt(int64_t R) {   // n-tiles per column block: fewest padded tiles, ties -> 10
  const int64_t tiles = gt_cdiv(R, 16);
  const int64_t w10 = gt_cdiv(tiles, 10) * 10 - tiles, w8 = gt_cdiv(tiles, 8) * 8 - tiles;
  return w8 < w10 ? 8 : 10;
}
static inline int64_t w3_ntp(int64_t R) {   // tiles per plane and k-step, padded to whole column blocks
  const int nt = w3_pick_nt(R);
  return gt_cdiv(gt_cdiv(R, 16), nt) * nt;
}
static inline size_t w3_image_bytes(int64_t R, int64_t C) { return (size_t)gt_cdiv(C, 32) * 3 * (size_t)w3_ntp(R) * 1024; }

// ---- image builder -------------------------------------------------------------------------------------------------
struct W3Job {
  const float* w;   // fp32 weight [rows][ldw]
  unsigned char* img;
  int R, C;         // image rows (output columns of the GEMM) and contraction length
  int ldw;
  int transposed;   // 0: element (r, c) = w[r * ldw + c]; 1: = w[c * ldw + r]  (the dX form: image of W^T)
  int ntp, ksteps;
  int block0;       // first block of this job
};
constexpr int W3_MAX_JOBS = 64;   // (3 KB of kernel arguments)
struct W3Jobs {
  W3Job j[W3_MAX_JOBS];
  int n;
};

__device__ __forceinline__ void w3_split_pair(float lo, float hi, uint32_t& u1, uint32_t& u2, uint32_t& u3) {
  u1 = gt_pack_bf16(lo, hi);
  const float rl = lo - __uint_as_float(u1 << 16), rh = hi - __uint_as_float(u1 & 0xffff0000u);   // exact
  u2 = gt_pack_bf16(rl, rh);
  u3 = gt_pack_bf16(rl - __uint_as_float(u2 << 16), rh - __uint_as_float(u2 & 0xffff0000u));
}

__global__ void __launch_bounds__(256) k_w3_image(W3Jobs jobs) {
  int ji = 0;
  for (int i = 1; i < jobs.n; ++i)
    if ((int)blockIdx.x >= jobs.j[i].block0) ji = i;
  const W3Job& J = jobs.j[ji];
  const int b = (int)blockIdx.x - J.block0;
  const int tpb = (J.ntp + 3) / 4;                  // blocks per k-step (4 tiles each)
  const int ks = b / tpb, jt = (b % tpb) * 4 + (threadIdx.x >> 6);
  if (ks >= J.ksteps || jt >= J.ntp) return;
  const int l = threadIdx.x & 63, r = l >> 2, c = l & 3;
  const int row = jt * 16 + r;
  float f[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = ks * 32 + c * 8 + e;
    f[e] = (row < J.R && k < J.C) ? (J.transposed ? J.w[(int64_t)k * J.ldw + row] : J.w[(int64_t)row * J.ldw + k]) : 0.f;
  }
  uint32_t p1[4], p2[4], p3[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) w3_split_pair(f[2 * q], f[2 * q + 1], p1[q], p2[q], p3[q]);
  const int64_t tile = ((int64_t)ks * 3) * J.ntp + jt;
  unsigned char* dst = J.img + tile * 1024 + r * 64 + ((c ^ ((r >> 1) & 3)) << 4);
  *reinterpret_cast<uint4*>(dst) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
  *reinterpret_cast<uint4*>(dst + (int64_t)J.ntp * 1024) = make_uint4(p2[0], p2[1], p2[2], p2[3]);
  *reinterpret_cast<uint4*>(dst + (int64_t)J.ntp * 2048) = make_uint4(p3[0], p3[1], p3[2], p3[3]);
}

// ---- the GEMM ------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void w3_lds_void;
typedef const __attribute__((address_space(1))) void w3_glb_void;

// out[M][Nout] = epilogue(A[M][Kc] Wimg^T); L32Args: a / amask / bias / add1 / add2 / out / gout / M / Nout / Kc / lda / ldo /
// act / inv_keep / thr / s0 / s1 / ncb as for k_lin32; w3 = the image, w3_ntp its tiles per plane
template <typename TA, typename TO, int NT, int MT, int WBUF, bool MASK, bool GELU = false>
__global__ void __launch_bounds__(256, 2) k_lin3(L32Args a) {
  constexpr int BM = 32 * MT;                               // 2 x 2 waves, MT m-tiles per wave: 64 (MT = 2) or 128 rows (MT = 4)
  constexpr int AR = BM / 64;                               // rows per staging thread
  constexpr int NPA = (sizeof(TA) == 2 && !MASK) ? 1 : 3;   // planes of the row operand (bf16 rows ARE their first plane)
  constexpr int HT = NT / 2;                                // n-tiles per wave
  constexpr int WSTAGE = 3 * NT * 1024, APLANE = BM * 64;
  constexpr int PLD = HT * 16 + 4;                          // epilogue patch pitch (floats)
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem3[];
  unsigned char* sA = smem3 + WBUF * WSTAGE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform in an SGPR: the DMA piece loop branches on it
  const int n = lane & 15, g = lane >> 4, wm = wid >> 1, wn = wid & 1;
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
  if (a.groups > 1) {   // grouped launch: this block's group has its own operands (uniform pointer arithmetic)
    const int64_t gi = blockIdx.y;
    a.a = reinterpret_cast<const TA*>(a.a) + gi * a.g_a;
    if (a.amask) a.amask = reinterpret_cast<const TA*>(a.amask) + gi * a.g_a;
    a.w3 = reinterpret_cast<const unsigned char*>(a.w3) + gi * a.g_img;
    if (a.bias) a.bias += gi * a.g_b;
    if (a.add1) a.add1 = reinterpret_cast<const TO*>(a.add1) + gi * a.g_o;
    if (a.add2) a.add2 = reinterpret_cast<const TO*>(a.add2) + gi * a.g_o;
    a.out = reinterpret_cast<TO*>(a.out) + gi * a.g_o;
  }
  const int64_t n0 = (int64_t)cb * NT * 16;
  const TA* A = reinterpret_cast<const TA*>(a.a);
  const TA* Am = reinterpret_cast<const TA*>(a.amask);
  const bool has_mask = MASK && Am != nullptr;
  const unsigned char* img = reinterpret_cast<const unsigned char*>(a.w3) + (int64_t)cb * NT * 1024 + lane * 16;
  const int64_t plane_stride = (int64_t)a.w3_ntp * 1024;    // bytes between the planes of one k-step

  // ---- row-operand staging: thread = (rows ar + 64 r, 8-element group q); out-of-range rows clamp to a valid row (never stored) ----
  const int ar = tid >> 2, q = tid & 3;
  const TA* a_src[AR];
  const TA* a2_src[AR];   // second row operand (virtual concatenation), columns [a_split, Kc)
  const TA* m_src[AR];
  bool a_none[AR];
  const bool cat2 = a.a2 != nullptr;
#pragma unroll
  for (int r = 0; r < AR; ++r) {
    const int64_t a_row = m0 + ar + 64 * r < a.M ? m0 + ar + 64 * r : a.M - 1;
    int64_t src_row = a_row;
    a_none[r] = false;
    if (a.a_rows) {   // the row operand's rows through the row map: a node without a token row contributes zeros
      const int32_t t = a.a_rows[a_row];
      a_none[r] = t < 0;
      src_row = t < 0 ? 0 : t;
    }
    a_src[r] = A + src_row * a.lda;
    a2_src[r] = cat2 ? reinterpret_cast<const TA*>(a.a2) + a_row * a.lda2 - a.a_split : nullptr;
    m_src[r] = has_mask ? Am + a_row * a.lda : nullptr;
  }
  struct ARegs {
    uint4 v0[AR], v1[AR], m0[AR], m1[AR];
    bool z0, z1;
  };
  ARegs R0;
#pragma unroll
  for (int r = 0; r < AR; ++r) R0.v0[r] = R0.v1[r] = R0.m0[r] = R0.m1[r] = make_uint4(0, 0, 0, 0);
  R0.z0 = R0.z1 = false;
  auto load = [&](ARegs& R, int ks) {
    const int64_t k = (int64_t)ks * 32 + q * 8;
    if constexpr (sizeof(TA) == 4) {
      R.z0 = k + 4 > a.Kc;
      R.z1 = k + 8 > a.Kc;
      const int64_t k0c = R.z0 ? a.Kc - 4 : k, k1c = R.z1 ? a.Kc - 4 : k + 4;
#pragma unroll
      for (int r = 0; r < AR; ++r) {
        R.v0[r] = *reinterpret_cast<const uint4*>((cat2 && k0c >= a.a_split ? a2_src[r] : a_src[r]) + k0c);
        R.v1[r] = *reinterpret_cast<const uint4*>((cat2 && k1c >= a.a_split ? a2_src[r] : a_src[r]) + k1c);
        if constexpr (MASK) {
          if (has_mask) {
            R.m0[r] = *reinterpret_cast<const uint4*>(m_src[r] + k0c);
            R.m1[r] = *reinterpret_cast<const uint4*>(m_src[r] + k1c);
          }
        }
      }
    } else {
      R.z0 = k + 8 > a.Kc;
      const int64_t kc = R.z0 ? a.Kc - 8 : k;
#pragma unroll
      for (int r = 0; r < AR; ++r) {
        R.v0[r] = *reinterpret_cast<const uint4*>(a_src[r] + kc);
        if constexpr (MASK) {
          if (has_mask) R.m0[r] = *reinterpret_cast<const uint4*>(m_src[r] + kc);
        }
      }
    }
  };
  auto store = [&](const ARegs& R) {
#pragma unroll
    for (int r = 0; r < AR; ++r) {
      const int row = ar + 64 * r;
      unsigned char* dst = sA + row * 64 + ((q ^ ((row >> 1) & 3)) << 4);
      if constexpr (NPA == 1) {
        *reinterpret_cast<uint4*>(dst) = (R.z0 || a_none[r]) ? make_uint4(0, 0, 0, 0) : R.v0[r];
      } else {
        float f[8];
        if constexpr (sizeof(TA) == 4) {
          const uint4 u0 = (R.z0 || a_none[r]) ? make_uint4(0, 0, 0, 0) : R.v0[r], u1 = (R.z1 || a_none[r]) ? make_uint4(0, 0, 0, 0) : R.v1[r];
          f[0] = __uint_as_float(u0.x); f[1] = __uint_as_float(u0.y); f[2] = __uint_as_float(u0.z); f[3] = __uint_as_float(u0.w);
          f[4] = __uint_as_float(u1.x); f[5] = __uint_as_float(u1.y); f[6] = __uint_as_float(u1.z); f[7] = __uint_as_float(u1.w);
        } else {
          chunk_to_f32<TA>((R.z0 || a_none[r]) ? make_uint4(0, 0, 0, 0) : R.v0[r], f);
        }
        if constexpr (MASK) {
          if (has_mask) {
            float y[8];
            if constexpr (sizeof(TA) == 4) {
              y[0] = __uint_as_float(R.m0[r].x); y[1] = __uint_as_float(R.m0[r].y); y[2] = __uint_as_float(R.m0[r].z); y[3] = __uint_as_float(R.m0[r].w);
              y[4] = __uint_as_float(R.m1[r].x); y[5] = __uint_as_float(R.m1[r].y); y[6] = __uint_as_float(R.m1[r].z); y[7] = __uint_as_float(R.m1[r].w);
            } else {
              chunk_to_f32<TA>(R.m0[r], y);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = gt_gate(f[e], y[e], a.inv_keep);
          }
        }
        uint32_t p1[4], p2[4], p3[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) w3_split_pair(f[2 * e], f[2 * e + 1], p1[e], p2[e], p3[e]);
        *reinterpret_cast<uint4*>(dst) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
        *reinterpret_cast<uint4*>(dst + APLANE) = make_uint4(p2[0], p2[1], p2[2], p2[3]);
        *reinterpret_cast<uint4*>(dst + 2 * APLANE) = make_uint4(p3[0], p3[1], p3[2], p3[3]);
      }
    }
  };
  // ---- weight tiles: LDS-DMA, piece i = plane i / NT, tile i % NT -> LDS offset i KB; the four waves take them round robin ----
  auto dma = [&](int ks, int buf) {
    const unsigned char* src = img + (int64_t)ks * 3 * plane_stride;
    unsigned char* dstb = smem3 + buf * WSTAGE;
#pragma unroll
    for (int i0 = 0; i0 < 3 * NT; i0 += 4) {
      const int i = i0 + wid;
      if (i < 3 * NT) {
        const int p = i / NT, j = i % NT;
        __builtin_amdgcn_global_load_lds((w3_glb_void*)(src + p * plane_stride + (int64_t)j * 1024), (w3_lds_void*)(dstb + i * 1024), 16, 0, 0);
      }
    }
  };

  f32x4 acc[MT][HT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < HT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nks = (int)((a.Kc + 31) / 32);
  const int swz = ((g ^ (n >> 1)) & 3) << 4;
  // One k-step ahead, on purpose no deeper: two register sets with counted s_waitcnt (row chunks two steps ahead, loads and DMA from
  // inline asm so that hipcc's vmcnt(0) in front of every LDS access behind an LDS-DMA does not drain them) measured 52 us against
  // 48 us for this form on 31.6 k x 300 x 300 -- the k-step is bound by issuing its 30 + 8 KB of vector-memory traffic beside 240
  // MFMAs (no loads at all: 34 us; no MFMAs: 39 us; tools/gemm3_probe results in DESIGN.md), not by their latency.
  if constexpr (!(W3_ABL & 1)) dma(0, 0);
  if constexpr (!(W3_ABL & 2)) load(R0, 0);
  for (int ks = 0; ks < nks; ++ks) {
    ARegs& R = R0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of step ks have landed, its row chunks are in registers
    if constexpr (WBUF == 2 && !(W3_ABL & 16)) __syncthreads();   // every wave is done with step ks - 1: the A stage and W buffer (ks+1)&1 are free
    if constexpr (!(W3_ABL & 32)) store(R);
    if constexpr (!(W3_ABL & 16)) __syncthreads();     // A(ks) and all of W(ks) are visible
    if (ks + 1 < nks) {
      if constexpr (WBUF == 2 && !(W3_ABL & 1)) dma(ks + 1, (ks + 1) & 1);
      if constexpr (!(W3_ABL & 2)) load(R, ks + 1);
    }
    const unsigned char* sW = smem3 + (WBUF == 2 ? (ks & 1) : 0) * WSTAGE + n * 64 + swz;
    const unsigned char* sAf = sA + (wm * 16 * MT + n) * 64 + swz;
    bf16x8_t fa[MT][NPA];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int p = 0; p < NPA; ++p) {
        if constexpr (W3_ABL & 64) fa[i][p] = __builtin_bit_cast(bf16x8_t, make_uint4(ks + i, p, lane, 7));
        else fa[i][p] = *reinterpret_cast<const bf16x8_t*>(sAf + p * APLANE + i * 16 * 64);
      }
#pragma unroll
    for (int jp = 0; jp < HT; jp += 2) {
      bf16x8_t fw[2][3];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
        if (jp + jj < HT) {
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            if constexpr (W3_ABL & 64) fw[jj][p] = __builtin_bit_cast(bf16x8_t, make_uint4(ks + jp, p + jj, lane, 9));
            else fw[jj][p] = *reinterpret_cast<const bf16x8_t*>(sW + (p * NT + wn * HT + jp + jj) * 1024);
          }
        }
      // the six products, small terms first; each round touches 2 x MT independent accumulators
      constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PA[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        if (PA[t] < NPA) {
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
            if (jp + jj < HT) {
#pragma unroll
              for (int i = 0; i < MT; ++i)
                if constexpr (W3_ABL & 4) acc[i][jp + jj][0] += __builtin_bit_cast(f32x4, fw[jj][PW[t]])[0] * __builtin_bit_cast(f32x4, fa[i][PA[t] < NPA ? PA[t] : 0])[0];
                else
                acc[i][jp + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[jj][PW[t]], fa[i][PA[t] < NPA ? PA[t] : 0], acc[i][jp + jj], 0, 0, 0);
            }
        }
      }
    }
    if constexpr (WBUF == 1) {   // one W buffer: the next k-step's tiles can only start once every wave has read this one's
      if constexpr (!(W3_ABL & 16)) __syncthreads();
      if (ks + 1 < nks) {
        if constexpr (!(W3_ABL & 1)) dma(ks + 1, 0);
      }
    }
  }
  __syncthreads();   // the stage buffers become the epilogue's patches

  // ---- epilogue: acc[i][j][r] = C[row m0 + wm*32 + i*16 + n][column n0 + (wn*HT + j)*16 + g*4 + r] ------------------------
  float* fs = reinterpret_cast<float*>(smem3);
  float* sB = fs + 4 * 16 * PLD;
  if (a.bias) {
    for (int c = tid; c < NT * 16; c += 256) sB[c] = n0 + c < a.Nout ? a.bias[n0 + c] : 0.f;
    __syncthreads();
  }
  float* patch = fs + wid * 16 * PLD;
  constexpr int CPR = HT * 4;   // 16-byte chunks per patch row; 16 rows -> HT chunks per lane
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int j = 0; j < HT; ++j)
      *reinterpret_cast<float4*>(patch + n * PLD + j * 16 + g * 4) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int64_t mrow0 = m0 + wm * 16 * MT + i * 16;
    const int64_t ncol0 = n0 + (int64_t)wn * HT * 16;
    float4 v[HT], e1[HT], e2[HT];
    bool ok[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      const int c = lane + t * 64;
      const int r = c / CPR, c4 = (c % CPR) * 4;
      const int64_t m = mrow0 + r, col = ncol0 + c4;
      ok[t] = m < a.M && col < a.Nout;
      if (ok[t]) {
        e1[t] = a.add1 ? gt_load4<TO>(reinterpret_cast<const TO*>(a.add1) + m * a.ldo + col) : gt_zero4();
        e2[t] = a.add2 ? gt_load4<TO>(reinterpret_cast<const TO*>(a.add2) + m * a.ldo + col) : gt_zero4();
      }
    }
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      const int c = lane + t * 64;
      const int r = c / CPR, c4 = (c % CPR) * 4;
      const int64_t m = mrow0 + r, col = ncol0 + c4;
      if (ok[t]) {
        v[t] = *reinterpret_cast<const float4*>(patch + r * PLD + c4);
        if (a.bias) v[t] = gt_add4(v[t], *reinterpret_cast<const float4*>(sB + wn * HT * 16 + c4));
        float* vv = reinterpret_cast<float*>(&v[t]);
        if (a.act == 1) v[t] = gt_relu4(v[t]);
        if constexpr (GELU) {
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
        if (a.thr) {
#pragma unroll
          for (int e = 0; e < 4; ++e) vv[e] = lin_hash(a.s0, a.s1, (uint32_t)m, (uint32_t)(col + e)) >= a.thr ? vv[e] * a.inv_keep : 0.f;
        }
        v[t] = gt_add4(gt_add4(v[t], e1[t]), e2[t]);
      }
    }
    if constexpr (!GELU) {   // (the gelu instantiations have no registers left for it, and no caller)
    if (a.ln_out) {
      // ---- LayerNorm of the row in the same epilogue (Nout = NT x 16: the block holds whole rows, half in each column wave):
      // two-pass statistics like k_ln_fwd on the STORED values (TO rounding first), the halves meet through LDS.  Every lane's HT
      // chunks belong to HT different rows (r = lane / 16 + 4 t at CPR = 16); the 16 lanes of a row are neighbours.
      static_assert(HT * 4 == CPR, "chunks per patch row");
      float2* stat = reinterpret_cast<float2*>(sB + NT * 16) + i * 64;   // [wm][wn][16 rows] x {sum | centred squares}, one slot per m-tile
      float mu[HT], rs[HT];
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        if constexpr (sizeof(TO) == 2) {   // the LayerNorm sees what is stored
          const uint32_t p0 = gt_pack_bf16(v[t].x, v[t].y), p1 = gt_pack_bf16(v[t].z, v[t].w);
          v[t] = make_float4(__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xffff0000u), __uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u));
        }
        float s = ok[t] ? (v[t].x + v[t].y) + (v[t].z + v[t].w) : 0.f;
#pragma unroll
        for (int sh = 1; sh < CPR; sh <<= 1) s += __shfl_xor(s, sh, 64);
        const int r = (lane + t * 64) / CPR;
        if ((lane & (CPR - 1)) == 0) stat[(wm * 2 + wn) * 16 + r].x = s;
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        const int r = (lane + t * 64) / CPR;
        mu[t] = (stat[(wm * 2 + 0) * 16 + r].x + stat[(wm * 2 + 1) * 16 + r].x) * (1.0f / (float)(NT * 16));
        const float dx = v[t].x - mu[t], dy = v[t].y - mu[t], dz = v[t].z - mu[t], dw = v[t].w - mu[t];
        float q = ok[t] ? (dx * dx + dy * dy) + (dz * dz + dw * dw) : 0.f;
#pragma unroll
        for (int sh = 1; sh < CPR; sh <<= 1) q += __shfl_xor(q, sh, 64);
        if ((lane & (CPR - 1)) == 0) stat[(wm * 2 + wn) * 16 + r].y = q;
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        const int c = lane + t * 64;
        const int r = c / CPR, c4 = (c % CPR) * 4;
        rs[t] = 1.0f / sqrtf((stat[(wm * 2 + 0) * 16 + r].y + stat[(wm * 2 + 1) * 16 + r].y) * (1.0f / (float)(NT * 16)) + a.ln_eps);
        if (!ok[t]) continue;
        const int64_t col = ncol0 + c4;
        const int64_t orow = a.out_rows ? (int64_t)a.out_rows[mrow0 + r] : mrow0 + r;
        if (orow < 0) continue;
        const float4 w4 = *reinterpret_cast<const float4*>(a.ln_w + col), b4 = *reinterpret_cast<const float4*>(a.ln_b + col);
        gt_store4<TO>(reinterpret_cast<TO*>(a.ln_out) + orow * a.ldo + col,
                      make_float4((v[t].x - mu[t]) * rs[t] * w4.x + b4.x, (v[t].y - mu[t]) * rs[t] * w4.y + b4.y,
                                  (v[t].z - mu[t]) * rs[t] * w4.z + b4.z, (v[t].w - mu[t]) * rs[t] * w4.w + b4.w));
        if (wn == 0 && c4 == 0) {
          a.ln_mean[orow] = mu[t];
          a.ln_rstd[orow] = rs[t];
        }
      }
    }
    }
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      const int c = lane + t * 64;
      const int r = c / CPR, c4 = (c % CPR) * 4;
      if (ok[t] && (!(W3_ABL & 8) || v[t].x == 12345.678f)) {
        const int64_t col = ncol0 + c4;
        if (a.out2 && col >= a.out_split) gt_store4<TO>(reinterpret_cast<TO*>(a.out2) + (mrow0 + r) * a.ldo2 + col - a.out_split, v[t]);
        else if (a.out_rows) {   // the output row through the row map (a token row), nothing for a node without one
          const int32_t orow = a.out_rows[mrow0 + r];
          if (orow >= 0) gt_store4<TO>(reinterpret_cast<TO*>(a.out) + (int64_t)orow * a.ldo + col, v[t]);
        } else gt_store4<TO>(reinterpret_cast<TO*>(a.out) + (mrow0 + r) * a.ldo + col, v[t]);
      }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

template <typename TA, typename TO, int NT, int MT, int WBUF, bool MASK, bool GELU>
void w3_launch_one(dim3 grid, hipStream_t stream, const L32Args& a) {
  constexpr int LDS = WBUF * 3 * NT * 1024 + 3 * 32 * MT * 64;
  static std::mutex mu;   // per instantiation: the > 64 KB dynamic-LDS opt-in is set once per device (C-ABI: one-time queries guarded)
  static bool done[16] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  {
    std::lock_guard<std::mutex> lk(mu);
    if (dev < 0 || dev >= 16 || !done[dev]) {
      (void)hipFuncSetAttribute((const void*)(k_lin3<TA, TO, NT, MT, WBUF, MASK, GELU>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      if (dev >= 0 && dev < 16) done[dev] = true;
    }
  }
  hipLaunchKernelGGL((k_lin3<TA, TO, NT, MT, WBUF, MASK, GELU>), grid, dim3(256), LDS, stream, a);
}

#ifndef W3_FORCE_MT
#define W3_FORCE_MT 0   // tools/gemm3_probe.hip only
#endif
#ifndef W3_WB_MT4
#define W3_WB_MT4 1     // W buffers of the 128-row / 64-row configurations (the probe builds the other combinations)
#endif
#ifndef W3_WB_MT2
#define W3_WB_MT2 2
#endif
// rows per block: 128 (one W buffer, two blocks per CU: every W byte feeds twice the MFMAs -- at 64 rows the kernel asks the L2 for
// 40 B / cycle / CU at the MFMA rate) when that still gives the chip >= 384 blocks, else 64 (two W buffers)
static inline int w3_pick_mt(int64_t M, int ncb) {
  if (W3_FORCE_MT) return W3_FORCE_MT;
  return gt_cdiv(M, 128) * ncb >= 384 ? 4 : 2;
}

template <bool MASK>
void w3_launch(int ta, int to, hipStream_t stream, L32Args& a) {
  if (a.groups > 1 && a.Nout <= 96 && ta == GT_F32 && to == GT_F32 && a.act != 2) {
    // a grouped launch of narrow GEMMs (the PNA pre stack's dX: 136 -> 68 columns per tower): 6 n-tiles, 64 rows, one W buffer =
    // 30 KB of LDS and 107 registers instead of 48 KB and 185 -- it runs beside the post stack's weight-gradient kernel (138 KB of LDS on
    // 192 CUs), where five of these blocks fit on a free CU against three: 126 -> 99 us in the step (24 us alone), Code2-PNA 33.0 ->
    // 33.9 k graphs/s.  The image was built for >= 8 tiles per plane: its first 6 are what this kernel reads.
    a.ncb = 1;
    a.w3_ntp = (int)w3_ntp(a.Nout);
    dim3 grid((unsigned)(gt_cdiv(gt_cdiv(a.M, 64), 8) * 8), (unsigned)a.groups);
    w3_launch_one<float, float, 6, 2, 1, MASK, false>(grid, stream, a);
    return;
  }
  const int nt = w3_pick_nt(a.Nout);
  a.ncb = (int)gt_cdiv(gt_cdiv(a.Nout, 16), nt);
  a.w3_ntp = (int)w3_ntp(a.Nout);
  const int ng = a.groups > 1 ? a.groups : 1;
  const int mt = w3_pick_mt(a.M, a.ncb * ng);
  dim3 grid((unsigned)(gt_cdiv(gt_cdiv(a.M, 32 * mt), 8) * 8 * a.ncb), (unsigned)ng);
#define GT_W3_MT(TA_, TO_, NT_, GELU_)                                                             \
  do {                                                                                             \
    if (mt == 4) w3_launch_one<TA_, TO_, NT_, 4, W3_WB_MT4, MASK, GELU_>(grid, stream, a);                 \
    else w3_launch_one<TA_, TO_, NT_, 2, W3_WB_MT2, MASK, GELU_>(grid, stream, a);                         \
  } while (0)
  if constexpr (!MASK) {
    if (a.act == 2) {   // gelu epilogue: fp32 rows in and out only
      if (nt == 10) GT_W3_MT(float, float, 10, true);
      else GT_W3_MT(float, float, 8, true);
      return;
    }
  }
#define GT_W3_GO(TA_, TO_)                                       \
  do {                                                           \
    if (nt == 10) GT_W3_MT(TA_, TO_, 10, false);                 \
    else GT_W3_MT(TA_, TO_, 8, false);                           \
  } while (0)
  if (ta == GT_F32 && to == GT_F32) GT_W3_GO(float, float);
  else if (ta == GT_F32) GT_W3_GO(float, gt_bf16);
  else if (to == GT_F32) GT_W3_GO(gt_bf16, float);
  else GT_W3_GO(gt_bf16, gt_bf16);
#undef GT_W3_GO
#undef GT_W3_MT
}

// ---- dW[N][K] = dZ^T X, db = colsum(dZ): contraction over M on the bf16 pipe -----------------------------------------------------
// Both operands are activations: they are split into bf16 planes while they are staged ([m][column] row-major per plane, 32 rows per
// stage) and read TRANSPOSED by ds_read_b64_tr_b16 (the contraction index m becomes the fragments' k-slots).  Block = 160 x 160
// outputs (N = K = 300: 2 x 2 tiles, 13 % padding instead of the 64 % of 128-wide tiles) x an M range; 4 waves as 2 x 2, wave =
// 5 x 5 accumulator tiles; the wave keeps the 15 X fragments of a stage in registers and streams the dZ fragments past them.
// Partials [split][N][K] + the fixed-order k_split_reduce as for the other dW kernels (bitwise reproducible).
constexpr int W3D_T = 160, W3D_LD = W3D_T + 8, W3D_PLANE = 32 * W3D_LD;   // bf16 elements

template <typename TS>
__device__ __forceinline__ void w3d_load_chunks(const TS* src, int64_t ld, int64_t row0, int64_t rows_end, int64_t col0, int64_t cols_end,
                                                uint4* v, const TS* msrc, uint4* vm, const TS* src2 = nullptr, int64_t split = 0,
                                                int64_t ld2 = 0, const int32_t* rows = nullptr, uint32_t* none = nullptr) {
  // a [32][160] tile as 16-byte chunks, chunk c = tid + 256 i: row c / CH, column chunk c % CH; out-of-range chunks are clamped to a
  // valid address (branch-free loads) and zeroed at store time
  constexpr int E = 16 / sizeof(TS), CH = W3D_T / E, NIT = (32 * CH + 255) / 256;
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int c = threadIdx.x + i * 256;
    const int r = c / CH, cc = (c % CH) * E;
    int64_t row = row0 + r, col = col0 + cc;
    row = row < rows_end ? row : rows_end - 1;
    col = col < cols_end ? col : 0;
    if (rows && r < 32) {   // rows through a row map (L32DwArgs::dy_rows): chunks of rows without a source are zeroed at store time
      const int32_t t = rows[row];
      if (t < 0) *none |= 1u << i;
      else *none &= ~(1u << i);
      row = t < 0 ? 0 : t;
    }
    if (r < 32) {
      if (src2 && col >= split) v[i] = *reinterpret_cast<const uint4*>(src2 + row * ld2 + (col - split));   // columns [split, ..) of a virtual concatenation
      else v[i] = *reinterpret_cast<const uint4*>(src + row * ld + col);
      if (msrc) vm[i] = *reinterpret_cast<const uint4*>(msrc + row * ld + col);
    }
  }
}

template <typename TS, bool MASK>
__device__ __forceinline__ void w3d_store_chunks(gt_bf16* planes, int64_t row0, int64_t rows_end, int64_t col0, int64_t cols_end,
                                                 const uint4* v, const uint4* vm, bool has_mask, float inv_keep, uint32_t none = 0) {
  constexpr int E = 16 / sizeof(TS), CH = W3D_T / E, NIT = (32 * CH + 255) / 256;
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int c = threadIdx.x + i * 256;
    const int r = c / CH, cc = (c % CH) * E;
    if (r >= 32) continue;
    const bool ok = row0 + r < rows_end && col0 + cc < cols_end && !((none >> i) & 1u);
    float f[E];
    chunk_to_f32<TS>(ok ? v[i] : make_uint4(0, 0, 0, 0), f);
    if constexpr (MASK) {
      if (has_mask) {
        float y[E];
        chunk_to_f32<TS>(vm[i], y);
#pragma unroll
        for (int e = 0; e < E; ++e) f[e] = ok ? gt_gate(f[e], y[e], inv_keep) : 0.f;
      }
    }
    uint32_t p1[E / 2], p2[E / 2], p3[E / 2];
#pragma unroll
    for (int e = 0; e < E / 2; ++e) w3_split_pair(f[2 * e], f[2 * e + 1], p1[e], p2[e], p3[e]);
    gt_bf16* dst = planes + r * W3D_LD + cc;
    if constexpr (E == 4) {
      *reinterpret_cast<uint2*>(dst) = make_uint2(p1[0], p1[1]);
      *reinterpret_cast<uint2*>(dst + W3D_PLANE) = make_uint2(p2[0], p2[1]);
      *reinterpret_cast<uint2*>(dst + 2 * W3D_PLANE) = make_uint2(p3[0], p3[1]);
    } else {
      *reinterpret_cast<uint4*>(dst) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
      *reinterpret_cast<uint4*>(dst + W3D_PLANE) = make_uint4(p2[0], p2[1], p2[2], p2[3]);
      *reinterpret_cast<uint4*>(dst + 2 * W3D_PLANE) = make_uint4(p3[0], p3[1], p3[2], p3[3]);
    }
  }
}

template <typename TY, typename TX, bool MASK>
__global__ void __launch_bounds__(256, 1) k_lin3_dw(L32DwArgs a) {
  constexpr int EY = 16 / sizeof(TY), EX = 16 / sizeof(TX);
  constexpr int ZIT = (32 * (W3D_T / EY) + 255) / 256, XIT = (32 * (W3D_T / EX) + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3d[];
  gt_bf16* sZ = reinterpret_cast<gt_bf16*>(smem3d);   // [3][32][W3D_LD]
  gt_bf16* sX = sZ + 3 * W3D_PLANE;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n = lane & 15, g = lane >> 4, wn = wid & 1, wk = wid >> 1;
  int64_t split_;
  int tile_;
  {  // XCD-aware: the tiles of one M-split read the same dZ / X rows -> ids 8 apart (same XCD, same L2)
    const int nt = a.nkb * a.nnb;
    const int64_t b = blockIdx.x;
    const int64_t group = b / (8 * nt);
    const int r = (int)(b % (8 * nt));
    tile_ = r / 8;
    split_ = group * 8 + r % 8;
  }
  if (split_ >= a.splits) return;
  const int kb = tile_ % a.nkb, nb = tile_ / a.nkb;
  const int64_t n0 = (int64_t)nb * W3D_T, k0 = (int64_t)kb * W3D_T;
  const int64_t mb = split_ * a.m_per_split;
  const int64_t me = mb + a.m_per_split < a.M ? mb + a.m_per_split : a.M;
  const TY* dY = reinterpret_cast<const TY*>(a.dy);
  const TY* Ym = reinterpret_cast<const TY*>(a.ymask);
  const bool has_mask = MASK && Ym != nullptr;
  const TX* X = reinterpret_cast<const TX*>(a.x);

  f32x4 acc[5][5];   // [n tile j][k tile i]
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int i = 0; i < 5; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbacc = 0.f;   // kb == 0, tid < 160: column n0 + tid

  uint4 vz[ZIT], vmk[MASK ? ZIT : 1], vx[XIT];
  uint32_t znone = 0;   // dy_rows: chunks of vz whose row has no source
  if (mb < me) {
    w3d_load_chunks<TY>(dY, a.ldy, mb, me, n0, a.N, vz, has_mask ? Ym : nullptr, vmk, nullptr, 0, 0, a.dy_rows, &znone);
    w3d_load_chunks<TX>(X, a.ldx, mb, me, k0, a.K, vx, nullptr, nullptr, reinterpret_cast<const TX*>(a.x2), a.x_split, a.ldx2);
  }
  for (int64_t m0 = mb; m0 < me; m0 += 32) {
    __syncthreads();   // every wave is done with the previous stage's planes
    w3d_store_chunks<TY, MASK>(sZ, m0, me, n0, a.N, vz, vmk, has_mask, a.inv_keep, znone);
    w3d_store_chunks<TX, false>(sX, m0, me, k0, a.K, vx, nullptr, false, 1.f);
    __syncthreads();
    if (m0 + 32 < me) {
      w3d_load_chunks<TY>(dY, a.ldy, m0 + 32, me, n0, a.N, vz, has_mask ? Ym : nullptr, vmk, nullptr, 0, 0, a.dy_rows, &znone);
      w3d_load_chunks<TX>(X, a.ldx, m0 + 32, me, k0, a.K, vx, nullptr, nullptr, reinterpret_cast<const TX*>(a.x2), a.x_split, a.ldx2);
    }
    if (kb == 0 && tid < W3D_T) {   // db: the three planes of a value add back to the value exactly
#pragma unroll 8
      for (int r = 0; r < 32; ++r) {
        const gt_bf16* q = sZ + r * W3D_LD + tid;
        dbacc += (gt_bf16_to_f32(q[0]) + gt_bf16_to_f32(q[W3D_PLANE])) + gt_bf16_to_f32(q[2 * W3D_PLANE]);
      }
    }
    Frag<gt_bf16> fx[5][3];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) fx[i][p] = frag_load_tr(sX + p * W3D_PLANE, W3D_LD, 0, wk * 80 + i * 16, n, g);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      Frag<gt_bf16> fz[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) fz[p] = frag_load_tr(sZ + p * W3D_PLANE, W3D_LD, 0, wn * 80 + j * 16, n, g);
      constexpr int PZ[6] = {2, 0, 1, 1, 0, 0}, PX[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < 5; ++i) acc[j][i] = mma(fz[PZ[t]], fx[i][PX[t]], acc[j][i]);
    }
  }
  __syncthreads();
  // acc[j][i][r] = C[row n0 + wn*80 + j*16 + g*4 + r][column k0 + wk*80 + i*16 + n] -> per-wave patch [16 n rows][80 k columns]
  constexpr int PLD = 80 + 4;
  float* patch = reinterpret_cast<float*>(smem3d) + wid * 16 * PLD;
  float* part = a.part + (int64_t)split_ * a.N * a.K;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) patch[(g * 4 + r) * PLD + i * 16 + n] = acc[j][i][r];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < 5; ++q) {   // 16 rows x 20 chunks = 320 chunks
      const int c = lane + q * 64;
      const int r = c / 20, c4 = (c % 20) * 4;
      const int64_t row = n0 + wn * 80 + j * 16 + r, col = k0 + wk * 80 + c4;
      if (row < a.N && col < a.K) *reinterpret_cast<float4*>(part + row * a.K + col) = *reinterpret_cast<const float4*>(patch + r * PLD + c4);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (kb == 0 && tid < W3D_T && n0 + tid < a.N && a.dbpart) a.dbpart[(int64_t)split_ * a.N + n0 + tid] = dbacc;
}

static inline int w3_dw_splits(int64_t M, int tiles) {
  // one block per CU (the kernel runs on the overlap stream beside the dX chain) -- and per XCD: block b runs on XCD b % 8 and the
  // kernels give the tiles of a split ids 8 apart, so XCD x gets the splits = x (mod 8); with 21 splits x 12 tiles (the PNA towers)
  // five XCDs got 36 blocks for their 32 CUs and the launch took two rounds (6.4 us per stage against 3.1: tools/gemm3r_dw_pna_probe).
  // Whole groups of 8 splits, at most 32 blocks per XCD.
  if (tiles < 1) tiles = 1;
  constexpr int per_xcd = 32;
  int64_t s = tiles <= per_xcd ? 8 * (per_xcd / tiles) : (8 * per_xcd) / tiles;
  const int64_t maxs = gt_cdiv(M, 32 * 8);           // at least 8 stages per split
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  return (int)s;
}

template <typename TY, typename TX>
void w3_launch_dw(dim3 grid, hipStream_t stream, const L32DwArgs& a) {
  constexpr int LDS = 6 * W3D_PLANE * 2;
  static std::mutex mu;
  static bool done[16] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  {
    std::lock_guard<std::mutex> lk(mu);
    if (dev < 0 || dev >= 16 || !done[dev]) {
      (void)hipFuncSetAttribute((const void*)(k_lin3_dw<TY, TX, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      if (dev >= 0 && dev < 16) done[dev] = true;
    }
  }
  hipLaunchKernelGGL((k_lin3_dw<TY, TX, true>), grid, dim3(256), LDS, stream, a);
}

// ---- per-thread table of prepared images (gt_w3_bind / gt_w3_unbind) -----------------------------------------------------
struct W3Bound {
  const float* w;
  int64_t N, K;          // the weight is [N][K]
  const void* img_fwd;   // image of W   (rows N, contraction K) or null
  const void* img_t;     // image of W^T (rows K, contraction N) or null
};
constexpr int W3_MAX_BOUND = 64;
struct W3Table {
  W3Bound e[W3_MAX_BOUND];
  int n = 0;
};
thread_local W3Table g_w3;

static inline const void* w3_lookup(const float* w, int64_t N, int64_t K, bool transposed) {
  for (int i = 0; i < g_w3.n; ++i)
    if (g_w3.e[i].w == w && g_w3.e[i].N == N && g_w3.e[i].K == K) return transposed ? g_w3.e[i].img_t : g_w3.e[i].img_fwd;
  return nullptr;
}
// `groups` weights [N][K] back to back from w, every one bound, their images evenly spaced: the first image and the spacing in bytes
static inline const void* w3_lookup_grouped(const float* w, int64_t N, int64_t K, int groups, bool transposed, int64_t* spacing) {
  const unsigned char* i0 = (const unsigned char*)w3_lookup(w, N, K, transposed);
  if (!i0 || groups < 2) { *spacing = 0; return i0; }
  const unsigned char* i1 = (const unsigned char*)w3_lookup(w + N * K, N, K, transposed);
  if (!i1) return nullptr;
  *spacing = (int64_t)(i1 - i0);
  for (int g = 2; g < groups; ++g)
    if ((const unsigned char*)w3_lookup(w + (int64_t)g * N * K, N, K, transposed) != i0 + (int64_t)g * *spacing) return nullptr;
  return i0;
}
