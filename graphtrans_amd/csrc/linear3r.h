// linear3r.h — k_lin3r: the bf16x6 GEMM (linear3x.h) with the ROW operand straight from global memory into MFMA fragments.
// Included by linear.hip inside its anonymous namespace, after linear3x.h (same weight images, same L32Args).
//
// The reference op: GCNConv / GINConv's nn.Linear on the node rows (modules/conv.py:44,51) and its input gradient --
// 31.6 k x 300 x 300 fp32 on Code2, five times per direction and step.
//
// Why a second kernel.  k_lin3 stages BOTH operands through the LDS: per 32-deep k-step a 64 x 160 block writes 12 KB of row
// planes, DMAs 30 KB of weight planes and reads 84 KB of fragments for 240 MFMAs -- the LDS port is as busy as the matrix pipe
// (15.8 against 15.4 us per CU on the Code2 shape), the row planes cost two barriers per k-step, and with N = 300 cut into two
// column blocks every activation row is loaded and split twice.  Here
//   * the activation rows never touch the LDS: lane (n, g) of a wave loads the 8 consecutive fp32 of row n, k-chunk g -- exactly its
//     MFMA fragment slot -- as two 16-byte global loads (the 4 lane groups of a row cover 128 contiguous bytes), one k-step ahead,
//     and splits them into the three bf16 planes in registers;
//   * a block is 128 rows x ALL columns (<= 20 n-tiles): 8 waves = 4 row groups x 2 column halves, wave = 32 rows x NTW n-tiles
//     (2 x NTW accumulator tiles: 80 VGPRs at NTW = 10, two waves per SIMD); the two column-half waves of a row group load the same
//     rows (the second from L1);
//   * only the weight planes live in the LDS: 3 x 2 NTW KB per k-step by LDS-DMA (global_load_lds_dwordx4, the image's tile order is
//     the LDS order), two stages, ONE barrier per k-step, the next stage's DMA and row loads in flight under the whole k-step's
//     120 MFMAs per wave;
//   * LDS traffic per MFMA falls from 0.35 KB (k_lin3) to 0.25 KB and nothing is written by ds_write: 1 824 LDS cycles against
//     3 648 matrix-pipe cycles per k-step and CU;
//   * the epilogue stores straight from the accumulators: lane (n, g) owns 4 consecutive columns of row n per tile (16-byte stores,
//     64 contiguous bytes per row and tile), bias / ReLU / two addends on the way.
// Covered: fp32 rows in and out, no gate / dropout / GELU / row maps / LayerNorm epilogue / virtual concatenation (those keep k_lin3),
// 8 < n-tiles per column block <= 20, M >= W3R_MIN_M (the grid is ceil(M / 128) x column blocks: below ~16 k rows k_lin3's 64-row
// blocks fill the chip better).
#pragma once

#ifndef W3R_ABL
#define W3R_ABL 0   // ablation mask of tools/gemm3r_probe (1 no DMA, 2 no row loads, 4 no MFMA, 8 no stores); 0 in the library
#endif

constexpr int64_t W3R_MIN_M = 12288;

template <int NTW, int CH>
__global__ void __launch_bounds__(256 * CH, CH) k_lin3r(L32Args a) {
  constexpr int NTB = CH * NTW;               // n-tiles per LDS stage and plane
  constexpr int NWV = 4 * CH;                 // waves: 4 row groups x CH column parts
  constexpr int WSTAGE = 3 * NTB * 1024;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem3r[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int rg = wid / CH, ch = wid % CH;
  const int cb = blockIdx.y;
  const int64_t m0 = (int64_t)blockIdx.x * 128 + rg * 32;
  const int tile0 = cb * a.ntb;                                   // first image tile of this column block
  const int ntl = a.w3_ntp - tile0 < NTB ? a.w3_ntp - tile0 : NTB;   // tiles the image holds for this stage (the rest of the stage is never stored from)
  const unsigned char* img = reinterpret_cast<const unsigned char*>(a.w3) + (int64_t)tile0 * 1024 + lane * 16;
  const int64_t plane_stride = (int64_t)a.w3_ntp * 1024;

  // ---- row operand: rows m0 + i*16 + n, floats [ks*32 + g*8, +8) ------------------------------------------------------------
  const float* arow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int64_t r = m0 + i * 16 + n;
    r = r < a.M ? r : a.M - 1;
    arow[i] = reinterpret_cast<const float*>(a.a) + r * a.lda;
  }
  float4 raw[2][2];
  bool z0 = false, z1 = false;
  auto load_rows = [&](int ks) {
    const int64_t k = (int64_t)ks * 32 + g * 8;
    z0 = k + 4 > a.Kc;
    z1 = k + 8 > a.Kc;
    const int64_t k0c = z0 ? a.Kc - 4 : k, k1c = z1 ? a.Kc - 4 : k + 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if constexpr (W3R_ABL & 32) {   // (probe only: the same bytes per wave, lane-linear)
        const float* base = reinterpret_cast<const float*>(a.a) + (m0 + i * 16 + (lane >> 3)) * a.lda + ks * 32 + (lane & 7) * 4;
        raw[i][0] = *reinterpret_cast<const float4*>(base);
        raw[i][1] = *reinterpret_cast<const float4*>(base + 8 * a.lda);
      } else {
        raw[i][0] = *reinterpret_cast<const float4*>(arow[i] + k0c);
        raw[i][1] = *reinterpret_cast<const float4*>(arow[i] + k1c);
      }
    }
  };
  // planes of the current k-step (fa) and of the next one (fn) as 32-bit words: word e of plane p of row tile i = k-pair e of the chunk
  uint32_t fa[2][3][4], fn[2][3][4];
  auto split_pair = [&](uint32_t (&dst)[2][3][4], int i, int e) {   // k-pair e (0..3) of row tile i
    const float4 u = e < 2 ? raw[i][0] : raw[i][1];
    const bool z = e < 2 ? z0 : z1;
    const float lo = z ? 0.f : ((e & 1) ? u.z : u.x), hi = z ? 0.f : ((e & 1) ? u.w : u.y);
    w3_split_pair(lo, hi, dst[i][0][e], dst[i][1][e], dst[i][2][e]);
  };
  auto frag = [&](const uint32_t (&src)[2][3][4], int i, int p) {
    return __builtin_bit_cast(bf16x8_t, make_uint4(src[i][p][0], src[i][p][1], src[i][p][2], src[i][p][3]));
  };
  // ---- weight planes: piece i = plane i / NTB, tile i % NTB -> LDS offset i KB of the stage; wave w takes pieces w, w + 8, ...
  // Branch-free (the pieces are issued BETWEEN the MFMAs of a k-step, one basic block): a piece index past the stage's last one
  // repeats the last piece (same bytes to the same place), a tile past the image's last one repeats the last tile (never stored from).
  constexpr int NPIECE = (3 * NTB + NWV - 1) / NWV;   // per wave and k-step
  auto dma_piece = [&](int ks, int buf, int q) {
    const unsigned char* src = img + (int64_t)ks * 3 * plane_stride;
    unsigned char* dstb = smem3r + buf * WSTAGE;
    int i = q * NWV + wid;
    i = i < 3 * NTB ? i : 3 * NTB - 1;
    const int p = i / NTB, j = i % NTB;
    const int js = j < ntl ? j : ntl - 1;
    __builtin_amdgcn_global_load_lds((w3_glb_void*)(src + p * plane_stride + (int64_t)js * 1024), (w3_lds_void*)(dstb + i * 1024), 16, 0, 0);
  };

  f32x4 acc[2][NTW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nks = (int)((a.Kc + 31) / 32);
  const int swz = ((g ^ (n >> 1)) & 3) << 4;
  constexpr int NP = NTW / 2;   // tile pairs per wave

  // One k-step = NP tile pairs x 24 MFMAs (6 products x 2 tiles x 2 row tiles), issued in groups of four.  The side work is placed BY
  // HAND between the groups and fenced (sched_barrier): hipcc's scheduler otherwise reads a pair's fragments right in front of its
  // MFMAs and waits for them, and issues the DMA pieces and the split in clumps during which the matrix pipe idles.
  //   * in front of group q (0..5) of pair p: fragment read q of pair p + 1 (one ds_read_b128);
  //   * MORE (a next stage exists): `raw` holds the NEXT k-step's rows, loaded during the previous k-step and complete since its
  //     closing barrier (vmcnt(0)).  They are split FIRST -- 8 k-pairs of ~13 VALU instructions in front of groups 0, 2, 3, 5 of
  //     pairs 0 and 1: hipcc guards the first use of a loaded register with s_waitcnt vmcnt(0), which costs nothing here and would
  //     wait for every LDS-DMA piece in flight anywhere later in the k-step (measured: +0.9 us per k-step) --, then the rows of
  //     k-step + 2 are loaded into the same registers (clamped to the last k-step: branch-free), and the DMA pieces of the next
  //     stage ride in front of groups 1 and 4 from pair 0 on (an LDS-DMA piece costs 60-180 issue cycles: never two in a row).
  auto kstep = [&](int ks, auto more_tag) {
    constexpr bool MORE = decltype(more_tag)::value;
    // (all four row loads are complete here; told to hipcc's wait-count pass in one piece -- it would otherwise place a vmcnt(0) in
    // front of the second row tile's split, behind the first DMA piece)
    if constexpr (MORE) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    const unsigned char* sW = smem3r + (ks & 1) * WSTAGE + (ch * NTW) * 1024 + n * 64 + swz;
    bf16x8_t fw[2][2][3];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int p = 0; p < 3; ++p) fw[0][jj][p] = *reinterpret_cast<const bf16x8_t*>(sW + (p * NTB + jj) * 1024);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pr = 0; pr < NP; ++pr) {
      const int cur = pr & 1, jp = 2 * pr;
      // the six products, small terms first; each group touches the pair's 4 accumulators once
      constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PA[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        if (pr + 1 < NP) fw[cur ^ 1][t / 3][t % 3] = *reinterpret_cast<const bf16x8_t*>(sW + ((t % 3) * NTB + jp + 2 + t / 3) * 1024);
        if constexpr (MORE) {
          if (pr < 2 && t != 1 && t != 4 && !(W3R_ABL & 16)) {
            const int e8 = pr * 4 + (t == 0 ? 0 : t == 2 ? 1 : t == 3 ? 2 : 3);   // 0..7
            split_pair(fn, e8 >> 2, e8 & 3);
          }
          if constexpr (!(W3R_ABL & 2)) {
            if (pr == 2 && t == 0) load_rows(ks + 2 < nks ? ks + 2 : nks - 1);
          }
        }
        if constexpr (MORE && !(W3R_ABL & 1)) {
          if ((t == 1 || t == 4) && 2 * pr + (t == 4 ? 1 : 0) < NPIECE) dma_piece(ks + 1, (ks + 1) & 1, 2 * pr + (t == 4 ? 1 : 0));
          if (pr == NP - 1 && t == 2) {   // (more pieces than 2 per pair: the rest here)
#pragma unroll
            for (int q = 2 * NP; q < NPIECE; ++q) dma_piece(ks + 1, (ks + 1) & 1, q);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            if constexpr (W3R_ABL & 4) acc[i][jp + jj][0] += __builtin_bit_cast(f32x4, fw[cur][jj][PW[t]])[0] * __builtin_bit_cast(f32x4, frag(fa, i, PA[t]))[0];
            else acc[i][jp + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[cur][jj][PW[t]], frag(fa, i, PA[t]), acc[i][jp + jj], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (MORE) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int e = 0; e < 4; ++e) fa[i][p][e] = (W3R_ABL & 16) ? (fa[i][p][e] ^ __float_as_uint(raw[i][e & 1].x)) : fn[i][p][e];
      __syncthreads();   // every wave has read stage ks & 1 and sees all of stage (ks + 1) & 1 (each wave waited for its own pieces)
    }
  };

  if constexpr (!(W3R_ABL & 1)) {
#pragma unroll
    for (int q = 0; q < NPIECE; ++q) dma_piece(0, 0, q);
  }
  if constexpr (!(W3R_ABL & 2)) load_rows(0);
  else { raw[0][0] = raw[0][1] = raw[1][0] = raw[1][1] = make_float4(1.f, 2.f, 3.f, 4.f); }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) split_pair(fa, i, e);
  if constexpr (!(W3R_ABL & 2)) load_rows(nks > 1 ? 1 : 0);
  __syncthreads();   // (vmcnt(0) + barrier) the first stage's tiles of every wave are visible
  for (int ks = 0; ks + 1 < nks; ++ks) kstep(ks, std::true_type{});
  kstep(nks - 1, std::false_type{});

  // ---- epilogue: acc[i][j][r] = C[row m0 + i*16 + n][column (tile0 + ch*NTW + j)*16 + g*4 + r] -------------------------------
  // (loads in batches, addresses clamped instead of branched around: a branch per tile serialises its load behind a wait)
  float* out = reinterpret_cast<float*>(a.out);
  const float* add1 = reinterpret_cast<const float*>(a.add1);
  const float* add2 = reinterpret_cast<const float*>(a.add2);
  int64_t colv[NTW];
  bool okc[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int64_t col = (int64_t)(tile0 + ch * NTW + j) * 16 + g * 4;
    okc[j] = ch * NTW + j < a.ntb && col < a.Nout;
    colv[j] = okc[j] ? col : 0;
  }
  float4 bv[NTW];
  if (a.bias) {
#pragma unroll
    for (int j = 0; j < NTW; ++j) bv[j] = *reinterpret_cast<const float4*>(a.bias + colv[j]);
  } else {
#pragma unroll
    for (int j = 0; j < NTW; ++j) bv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // BNS (dX form): this dX is the dy of a BatchNorm further down the backward pass (gt_linear_bwd_bnstats): the block also writes
  // bn_part[block][0][Nout] = sum_rows dy', [1][Nout] = sum_rows dy' xhat over its 128 rows -- k_bn_bwd_partial's second pass over dy
  // and the BatchNorm input happens here, on the values in the registers
  const bool bns = a.bn_part != nullptr;
  float4 s0[NTW], s1[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) s0[j] = s1[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int64_t m = m0 + i * 16 + n;
    const bool okm = m < a.M;
    const int64_t mc = okm ? m : a.M - 1;
    float4 v[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      v[j] = gt_add4(make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]), bv[j]);
      if (a.act == 1) v[j] = gt_relu4(v[j]);
    }
    if (add1) {
      float4 e[NTW];
#pragma unroll
      for (int j = 0; j < NTW; ++j) e[j] = *reinterpret_cast<const float4*>(add1 + mc * a.ldo + colv[j]);
#pragma unroll
      for (int j = 0; j < NTW; ++j) v[j] = gt_add4(v[j], e[j]);
    }
    if (add2) {
      float4 e[NTW];
#pragma unroll
      for (int j = 0; j < NTW; ++j) e[j] = *reinterpret_cast<const float4*>(add2 + mc * a.ldo + colv[j]);
#pragma unroll
      for (int j = 0; j < NTW; ++j) v[j] = gt_add4(v[j], e[j]);
    }
    if (a.add_bc) {   // + add_bc[add_bidx[row]]: a few hundred distinct rows, L2-resident
      const float* bc = a.add_bc + (int64_t)a.add_bidx[mc] * a.ldo;
      float4 e[NTW];
#pragma unroll
      for (int j = 0; j < NTW; ++j) e[j] = *reinterpret_cast<const float4*>(bc + colv[j]);
#pragma unroll
      for (int j = 0; j < NTW; ++j) v[j] = gt_add4(v[j], e[j]);
    }
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      if (okm && okc[j]) {
        if constexpr (!(W3R_ABL & 8)) *reinterpret_cast<float4*>(out + m * a.ldo + colv[j]) = v[j];
        else if (v[j].x == 12345.678f) *reinterpret_cast<float4*>(out + m * a.ldo + colv[j]) = v[j];
      }
    }
    if (bns) {
      float4 xr[NTW];
#pragma unroll
      for (int j = 0; j < NTW; ++j) xr[j] = *reinterpret_cast<const float4*>(a.bn_x + mc * a.bn_ldx + colv[j]);
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const float4 mu = *reinterpret_cast<const float4*>(a.bn_mean + colv[j]), rs = *reinterpret_cast<const float4*>(a.bn_rstd + colv[j]);
        const float4 xh = make_float4((xr[j].x - mu.x) * rs.x, (xr[j].y - mu.y) * rs.y, (xr[j].z - mu.z) * rs.z, (xr[j].w - mu.w) * rs.w);
        float4 gg = (okm && okc[j]) ? v[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bn_relu) {
          const float4 ww = *reinterpret_cast<const float4*>(a.bn_w + colv[j]), bb = *reinterpret_cast<const float4*>(a.bn_b + colv[j]);
          gg = make_float4(xh.x * ww.x + bb.x > 0.f ? gg.x : 0.f, xh.y * ww.y + bb.y > 0.f ? gg.y : 0.f, xh.z * ww.z + bb.z > 0.f ? gg.z : 0.f,
                           xh.w * ww.w + bb.w > 0.f ? gg.w : 0.f);
        }
        s0[j] = gt_add4(s0[j], gg);
        s1[j] = make_float4(fmaf(gg.x, xh.x, s1[j].x), fmaf(gg.y, xh.y, s1[j].y), fmaf(gg.z, xh.z, s1[j].z), fmaf(gg.w, xh.w, s1[j].w));
      }
    }
  }
  if (bns) {   // (uniform) sum over the 16 rows of a lane group (fixed butterfly), then over the four row groups through the LDS
    __syncthreads();   // every wave is past its last fragment read: the stage buffers become scratch
    float* sb = reinterpret_cast<float*>(smem3r);   // [2][4 row groups][NTB * 16 columns]
    constexpr int NC = NTB * 16;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      float t[8] = {s0[j].x, s0[j].y, s0[j].z, s0[j].w, s1[j].x, s1[j].y, s1[j].z, s1[j].w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) t[e] += __shfl_xor(t[e], o, 64);
      }
      if (n == 0) {
        const int cl = (ch * NTW + j) * 16 + g * 4;
        *reinterpret_cast<float4*>(sb + (0 * 4 + rg) * NC + cl) = make_float4(t[0], t[1], t[2], t[3]);
        *reinterpret_cast<float4*>(sb + (1 * 4 + rg) * NC + cl) = make_float4(t[4], t[5], t[6], t[7]);
      }
    }
    __syncthreads();
    for (int c = tid; c < 2 * NC; c += 256 * CH) {
      const int which = c / NC, cl = c % NC;
      const int64_t col = (int64_t)tile0 * 16 + cl;
      if (cl < a.ntb * 16 && col < a.Nout) {
        const float* q = sb + which * 4 * NC + cl;
        a.bn_part[((int64_t)blockIdx.x * 2 + which) * a.Nout + col] = (q[0] + q[NC]) + (q[2 * NC] + q[3 * NC]);
      }
    }
  }
}

template <int NTW, int CH>
void w3r_launch_one(dim3 grid, hipStream_t stream, const L32Args& a) {
  constexpr int LDS = 2 * 3 * CH * NTW * 1024;
  static std::mutex mu;   // per instantiation: the > 64 KB dynamic-LDS opt-in is set once per device
  static bool done[16] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  {
    std::lock_guard<std::mutex> lk(mu);
    if (dev < 0 || dev >= 16 || !done[dev]) {
      (void)hipFuncSetAttribute((const void*)(k_lin3r<NTW, CH>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      if (dev >= 0 && dev < 16) done[dev] = true;
    }
  }
  hipLaunchKernelGGL((k_lin3r<NTW, CH>), grid, dim3(256 * CH), LDS, stream, a);
}

// column blocks of the register-row kernel: as few as hold <= 20 tiles each; 0 = the shape is not covered
static inline int w3r_ncb(int64_t Nout) {
  const int64_t tiles = gt_cdiv(Nout, 16);
  const int ncb = (int)gt_cdiv(tiles, 20);
  const int64_t ntb = gt_cdiv(tiles, ncb);
  return ntb > 8 ? ncb : 0;
}

// the register-row kernel takes the call (MASK callers pass amask == null for an ungated dX)
static inline bool w3r_ok(int ta, int to, const L32Args& a) {
  if (ta != GT_F32 || to != GT_F32 || !a.w3) return false;
  if (a.amask || a.gout || a.thr || a.act > 1 || a.a2 || a.out2 || a.out_rows || a.a_rows || a.ln_out || a.groups > 1) return false;
  if (a.bn_part && (a.bn_ldx % 4 || (((uintptr_t)a.bn_x | (uintptr_t)a.bn_mean | (uintptr_t)a.bn_rstd | (uintptr_t)a.bn_w | (uintptr_t)a.bn_b) & 15))) return false;
  if ((((uintptr_t)a.a | (uintptr_t)a.out | (uintptr_t)a.add1 | (uintptr_t)a.add2 | (uintptr_t)a.bias | (uintptr_t)a.add_bc) & 15) != 0) return false;
  if (a.M < W3R_MIN_M || a.Nout % 4 || a.Kc % 4 || a.Kc < 4 || a.lda % 4 || a.ldo % 4) return false;
  return w3r_ncb(a.Nout) > 0;
}

static inline void w3r_launch(hipStream_t stream, L32Args& a) {
  const int64_t tiles = gt_cdiv(a.Nout, 16);
  a.ncb = w3r_ncb(a.Nout);
  a.ntb = (int)gt_cdiv(tiles, a.ncb);
  a.w3_ntp = (int)w3_ntp(a.Nout);
  dim3 grid((unsigned)gt_cdiv(a.M, 128), (unsigned)a.ncb);
#if defined(W3R_CH) && W3R_CH == 1   // (probe only: one wave per SIMD, every wave all columns -- measured slower, 43.8 against 39.8 us)
  if (a.ntb > 16) w3r_launch_one<20, 1>(grid, stream, a);
  else w3r_launch_one<16, 1>(grid, stream, a);
#else
  if (a.ntb > 16) w3r_launch_one<10, 2>(grid, stream, a);
  else w3r_launch_one<8, 2>(grid, stream, a);
#endif
}

#ifndef W3RD_ABL
#define W3RD_ABL 0   // ablation mask of tools/gemm3r_dw_probe (1 no MFMA, 2 no fragment reads, 4 no split / plane stores, 8 no row loads, 16 no partial stores); 0 in the library
#endif
// =====================================================================================================================================
// k_lin3r_dw — dW[N][K] = dZ^T X, db = colsum(dZ) on the bf16 pipe (bf16x6), the k_lin3_dw tiling with the stages PIPELINED.
// Reference: the weight gradient of GCNConv / GINConv's nn.Linear and of the encoder's linears in the fp32 mode (modules/conv.py:44,51,
// modules/transformer_encoder.py:59), M = 31.6 k rows against N x K = 300 x 300 on Code2.
//
// k_lin3_dw (linear3x.h) runs a stage as load-wait -> split + ds_write (260 VALU, ~1 000 cycles, matrix pipe idle) -> barrier ->
// transposed reads + 150 MFMAs (2 400 cycles) -> barrier with ONE wave per SIMD: 75 us in the Code2 step against 15 us of MFMA time,
// 0.147 of the bf16x6 ceiling with its reduce (BENCH_r04).  Here, same block (160 x 160 outputs x an M range, 4 waves as 2 x 2, wave =
// 5 x 5 accumulator tiles, 32 rows per stage, planes [m][column] read transposed by ds_read_b64_tr_b16):
//   * TWO LDS stage buffers (2 x 63 KB): while the MFMAs of stage s read buffer s & 1, the rows of stage s + 1 -- in registers since
//     the previous stage -- are split and written to the other buffer, one 16-byte chunk (26 VALU + 3 ds_write_b64) in front of every
//     15th MFMA, placed by hand and fenced (sched_barrier) like k_lin3r's k-step; ONE barrier per stage;
//   * TWO register sets of raw rows: the loads of stage s + 2 go out at the top of stage s and have the whole stage to land;
//   * the dZ fragments of n-tile j + 1 are read while the 30 MFMAs of n-tile j run;
//   * db: every thread's chunks sit in the same columns at every stage -- it keeps their running sums (20 VALU per stage) and the
//     block folds them once at the end (k-block 0 only), instead of re-reading the planes from the LDS at every stage.
// fp32 rows only (bf16 token-row gradients keep k_lin3_dw); same partials + fixed-order
// k_split_reduce, bitwise reproducible.
// PC: 8 waves, PRODUCER / CONSUMER -- waves 4..7 (one per SIMD) only load, split and store the next stage's planes, waves 0..3 only read
// fragments and issue MFMAs.  Measured (tools/gemm3r_dw_probe, profiles/r05_probes/gemm3r_dw_ablations.txt): alone on the chip 54.0 us
// against 55.5 us for the 4-wave form (69 us for k_lin3_dw) -- per 32-row stage the multiplying side alone needs 1.47 us, the staging
// side alone 1.73 us (the split is ~45 instructions per 16-byte chunk), together 2.75 us: they do not overlap much better from two
// waves per SIMD than from one.  IN THE TRAINING STEP the 8-wave form is the worse neighbour -- 512 threads x 247 registers + 126 KB
// of LDS leave a CU nothing for the main stream's kernels: Code2 71.0 k graphs/s against 73.3 k for the 4-wave form (= k_lin3_dw's
// 73.1-73.6 k: the weight gradients run beside the critical path, their duration does not enter the step).  The 4-wave form ships;
// GT_LIN3R_DW_PC=1 selects this one.
// TN x TK: accumulator tiles per wave -> block = 32 TN x 32 TK outputs (5 x 5: 160 x 160; 7 x 4: 224 x 128 for the PNA towers' 204 x 340)
template <bool MASK, bool ROWS, bool PC, int TN = 5, int TK = 5>
__global__ void __launch_bounds__(PC ? 512 : 256, PC ? 2 : 1) k_lin3r_dw(L32DwArgs a) {
  static_assert(TK <= TN && TN <= 7, "one X chunk per n-tile of the MFMA loop at most");
  constexpr int ZT = 32 * TN, XT = 32 * TK;                // the block's extents along N and K
  constexpr int LDZ = ZT + 8, LDX = XT + 8, PLZ = 32 * LDZ, PLX = 32 * LDX;   // plane pitches / sizes (bf16 elements)
  constexpr int NCZ = TN, NCX = TK;                        // 16-byte chunks of the [32][ZT] / [32][XT] fp32 tiles per thread: chunk c = tid + 256 i
  constexpr int STAGE_EL = 3 * PLZ + 3 * PLX;              // bf16 elements of one stage buffer: sZ[3] then sX[3]
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3rd[];
  gt_bf16* sbuf = reinterpret_cast<gt_bf16*>(smem3rd);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool producer = PC && wave >= 4;      // (wave-uniform)
  const bool stager = !PC || producer;        // this wave loads / splits / stores the planes
  const bool mather = !PC || !producer;       // this wave reads fragments and issues MFMAs
  const int tid = (int)threadIdx.x & 255;     // chunk ownership (stagers) and patch ownership (mathers)
  const int wid = wave & 3;
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
  const int64_t n0 = (int64_t)nb * ZT, k0 = (int64_t)kb * XT;
  const int64_t mb = split_ * a.m_per_split;
  const int64_t me = mb + a.m_per_split < a.M ? mb + a.m_per_split : a.M;
  if (a.groups > 1) {   // grouped launch: this block's group has its own operands and partials (uniform pointer arithmetic)
    const int64_t gi = blockIdx.y;
    a.dy = reinterpret_cast<const float*>(a.dy) + gi * a.g_y;
    if (a.ymask) a.ymask = reinterpret_cast<const float*>(a.ymask) + gi * a.g_y;
    a.x = reinterpret_cast<const float*>(a.x) + gi * a.g_x;
    a.part += gi * a.g_part;
    if (a.dbpart) a.dbpart += gi * a.g_part;
  }
  const float* dY = reinterpret_cast<const float*>(a.dy);
  const float* Ym = reinterpret_cast<const float*>(a.ymask);
  const bool has_mask = MASK && Ym != nullptr;
  const float* X = reinterpret_cast<const float*>(a.x);

  // chunk i of this thread: row cr[i] of the stage, columns cq[i] .. + 3 of the tile (the same at every stage)
  int crz[NCZ], cqz[NCZ], crx[NCX], cqx[NCX];
  bool zcol[NCZ], xcol[NCX];   // the chunk's columns exist
  int64_t zoff[NCZ], xld[NCX];
  const float* xsrc[NCX];   // the chunk's X column in row 0 of its matrix (columns [x_split, K) of a virtual concatenation live in x2)
#pragma unroll
  for (int i = 0; i < NCZ; ++i) {
    const int c = tid + 256 * i;
    crz[i] = c / (ZT / 4);
    cqz[i] = (c % (ZT / 4)) * 4;
    zcol[i] = n0 + cqz[i] < a.N;
    zoff[i] = (int64_t)crz[i] * a.ldy + (zcol[i] ? n0 + cqz[i] : 0);
  }
#pragma unroll
  for (int i = 0; i < NCX; ++i) {
    const int c = tid + 256 * i;
    crx[i] = c / (XT / 4);
    cqx[i] = (c % (XT / 4)) * 4;
    xcol[i] = k0 + cqx[i] < a.K;
    const int64_t col = xcol[i] ? k0 + cqx[i] : 0;
    const bool second = a.x2 && col >= a.x_split;
    xld[i] = second ? a.ldx2 : a.ldx;
    xsrc[i] = (second ? reinterpret_cast<const float*>(a.x2) + (col - a.x_split) : X + col) + (int64_t)crx[i] * xld[i];
  }
  struct Raw {
    float4 z[NCZ], x[NCX], m[MASK ? NCZ : 1];
    uint32_t none;   // dy_rows: bit i = chunk i's row of dY has no source (zeros)
  };
  auto load_stage = [&](Raw& R, int64_t m0) {   // rows m0 + cr[i]; rows past the range clamp to its last row (zeroed at split time)
    R.none = 0;
    if constexpr (W3RD_ABL & 8) {
#pragma unroll
      for (int i = 0; i < NCZ; ++i) R.z[i] = make_float4(1.f, 2.f, (float)m0, 4.f);
#pragma unroll
      for (int i = 0; i < NCX; ++i) R.x[i] = make_float4(4.f, 3.f, 2.f, (float)m0);
      return;
    }
#pragma unroll
    for (int i = 0; i < NCZ; ++i) {
      const int64_t rowc = m0 + crz[i] < me ? m0 : me - 1 - crz[i];
      if constexpr (ROWS) {   // dY's rows through the row map: L32DwArgs::dy_rows
        const int32_t t = a.dy_rows[rowc + crz[i]];
        if (t < 0) R.none |= 1u << i;
        R.z[i] = *reinterpret_cast<const float4*>(dY + (int64_t)(t < 0 ? 0 : t) * a.ldy + (zoff[i] - (int64_t)crz[i] * a.ldy));
      } else
      R.z[i] = *reinterpret_cast<const float4*>(dY + rowc * a.ldy + zoff[i]);
      if constexpr (MASK) {
        if (has_mask) R.m[i] = *reinterpret_cast<const float4*>(Ym + rowc * a.ldy + zoff[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < NCX; ++i) {
      const int64_t rowc = m0 + crx[i] < me ? m0 : me - 1 - crx[i];
      R.x[i] = *reinterpret_cast<const float4*>(xsrc[i] + rowc * xld[i]);
    }
  };
  float4 dbs[NCZ];
#pragma unroll
  for (int i = 0; i < NCZ; ++i) dbs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  // split one chunk of a raw stage into the planes of buffer `buf`: which = 0 dZ chunk i (gated, summed into db), 1 X chunk i
  auto split_chunk = [&](const Raw& R, int64_t m0, int buf, int which, int i) {
    gt_bf16* planes = sbuf + buf * STAGE_EL + which * 3 * PLZ;
    const int cr_ = which ? crx[i] : crz[i], cq_ = which ? cqx[i] : cqz[i];
    const int pl_ = which ? PLX : PLZ;
    const bool ok = m0 + cr_ < me && (which ? xcol[i] : (zcol[i] && !((R.none >> i) & 1u)));
    float4 v = which ? R.x[i] : R.z[i];
    if constexpr (MASK) {
      if (!which && has_mask) {
        const float4 y = R.m[i];
        v = make_float4(gt_gate(v.x, y.x, a.inv_keep), gt_gate(v.y, y.y, a.inv_keep), gt_gate(v.z, y.z, a.inv_keep), gt_gate(v.w, y.w, a.inv_keep));
      }
    }
    if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!which) dbs[i] = gt_add4(dbs[i], v);
    if constexpr (W3RD_ABL & 4) return;
    uint32_t p1[2], p2[2], p3[2];
    w3_split_pair(v.x, v.y, p1[0], p2[0], p3[0]);
    w3_split_pair(v.z, v.w, p1[1], p2[1], p3[1]);
    gt_bf16* dst = planes + cr_ * (which ? LDX : LDZ) + cq_;
    *reinterpret_cast<uint2*>(dst) = make_uint2(p1[0], p1[1]);
    *reinterpret_cast<uint2*>(dst + pl_) = make_uint2(p2[0], p2[1]);
    *reinterpret_cast<uint2*>(dst + 2 * pl_) = make_uint2(p3[0], p3[1]);
  };

  const int64_t nst = mb < me ? (me - mb + 31) / 32 : 0;
  const bool want_db = kb == 0 && a.dbpart != nullptr;   // (uniform)

  // ---- the staging side: rows of stage s + 1 split into buffer (s + 1) & 1 while stage s is multiplied --------------------------
  // PC: the whole life of waves 4..7 (their registers hold raw rows and the db sums, no accumulators); !PC: called chunk by chunk
  // from between the MFMAs below.
  auto db_fold = [&]() {   // (every wave of the block passes the two barriers)
    __syncthreads();
    float* sdb = reinterpret_cast<float*>(smem3rd);   // [32][ZT]
    if (stager) {
#pragma unroll
      for (int i = 0; i < NCZ; ++i) *reinterpret_cast<float4*>(sdb + crz[i] * ZT + cqz[i]) = dbs[i];
    }
    __syncthreads();
    if (stager && tid < ZT && n0 + tid < a.N) {
      float t = 0.f;
#pragma unroll 8
      for (int r = 0; r < 32; ++r) t += sdb[r * ZT + tid];
      a.dbpart[(int64_t)split_ * a.N + n0 + tid] = t;
    }
  };
  if constexpr (PC) {
    if (producer) {
      Raw RA;
      if (nst > 0) {
        load_stage(RA, mb);
#pragma unroll
        for (int i = 0; i < NCZ; ++i) { split_chunk(RA, mb, 0, 0, i); if (i < NCX) split_chunk(RA, mb, 0, 1, i); }
        load_stage(RA, nst > 1 ? mb + 32 : mb);
      }
      __syncthreads();
      for (int64_t s = 0; s < nst; ++s) {
        const int buf = (int)(s & 1);
        const int64_t m1 = mb + (s + 1) * 32, m2 = mb + (s + 2) * 32;
#pragma unroll
        for (int i = 0; i < NCZ; ++i) {   // (a stage past the range: zeros into the idle buffer)
          split_chunk(RA, m1 < me ? m1 : me, buf ^ 1, 0, i);
          if (i < NCX) split_chunk(RA, m1 < me ? m1 : me, buf ^ 1, 1, i);
        }
        load_stage(RA, m2 < me ? m2 : (me > 32 ? me - 32 : mb));   // in flight across the barrier: these waves have the time
        __syncthreads();
      }
      if (want_db) db_fold();
      return;
    }
  }

  // ---- the multiplying side ------------------------------------------------------------------------------------------------------
  f32x4 acc[TN][TK];   // [n tile j][k tile i]
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TK; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  Raw RA, RB;   // (!PC) RA: the stage that is split during the current stage; RB: the one after it, in flight
  if constexpr (!PC) {
    if (nst > 0) {
      load_stage(RA, mb);
#pragma unroll
      for (int i = 0; i < NCZ; ++i) { split_chunk(RA, mb, 0, 0, i); if (i < NCX) split_chunk(RA, mb, 0, 1, i); }
      load_stage(RA, nst > 1 ? mb + 32 : mb);
    }
  }
  __syncthreads();
  for (int64_t s = 0; s < nst; ++s) {
    const int buf = (int)(s & 1);
    const int64_t m1 = mb + (s + 1) * 32, m2 = mb + (s + 2) * 32;
    if constexpr (!PC) load_stage(RB, m2 < me ? m2 : (me > 32 ? me - 32 : mb));
    __builtin_amdgcn_sched_barrier(0);
    const gt_bf16* sZ = sbuf + buf * STAGE_EL;
    const gt_bf16* sX = sZ + 3 * PLZ;
    auto fload = [&](const gt_bf16* pl, int ld, int col0) {
      if constexpr (W3RD_ABL & 2) { Frag<gt_bf16> f; f.v = make_uint4((uint32_t)col0, (uint32_t)s, 3u, (uint32_t)lane); return f; }
      else return frag_load_tr(pl, ld, 0, col0, n, g);
    };
    Frag<gt_bf16> fx[TK][3];
#pragma unroll
    for (int i = 0; i < TK; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) fx[i][p] = fload(sX + p * PLX, LDX, wk * 16 * TK + i * 16);
    Frag<gt_bf16> fz[2][3];
#pragma unroll
    for (int p = 0; p < 3; ++p) fz[0][p] = fload(sZ + p * PLZ, LDZ, wn * 16 * TN);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int cur = j & 1;
      constexpr int PZ[6] = {2, 0, 1, 1, 0, 0}, PX[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        // side work in front of this group of 5 MFMAs: the next n-tile's fragments (t < 3); one wave per SIMD (!PC): one chunk of the
        // next stage's split (t = 1, 4)
        if (j + 1 < TN && t < 3) fz[cur ^ 1][t] = fload(sZ + t * PLZ, LDZ, wn * 16 * TN + (j + 1) * 16);
        if constexpr (!PC) {
          if (t == 1) split_chunk(RA, m1 < me ? m1 : me, buf ^ 1, 0, j);
          if (t == 4 && j < NCX) split_chunk(RA, m1 < me ? m1 : me, buf ^ 1, 1, j);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TK; ++i) {
          if constexpr (W3RD_ABL & 1) acc[j][i][0] += __uint_as_float(fz[cur][PZ[t]].v.x) * __uint_as_float(fx[i][PX[t]].v.y);
          else acc[j][i] = mma(fz[cur][PZ[t]], fx[i][PX[t]], acc[j][i]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (!PC) RA = RB;
    __syncthreads();   // every wave has read buffer `buf` and the next stage's planes are complete in the other one
  }

  // acc[j][i][r] = C[row n0 + wn*80 + j*16 + g*4 + r][column k0 + wk*80 + i*16 + n] -> per-wave patch [16 n rows][80 k columns]
  constexpr int PLD = 16 * TK + 4;
  float* patch = reinterpret_cast<float*>(smem3rd) + wid * 16 * PLD;
  float* part = a.part + (int64_t)split_ * a.N * a.K;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
#pragma unroll
    for (int i = 0; i < TK; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) patch[(g * 4 + r) * PLD + i * 16 + n] = acc[j][i][r];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < TK; ++q) {   // 16 rows x 4 TK chunks
      const int c = lane + q * 64;
      const int r = c / (4 * TK), c4 = (c % (4 * TK)) * 4;
      const int64_t row = n0 + wn * 16 * TN + j * 16 + r, col = k0 + wk * 16 * TK + c4;
      if (row < a.N && col < a.K && (!(W3RD_ABL & 16) || patch[r * PLD + c4] == 12345.678f)) *reinterpret_cast<float4*>(part + row * a.K + col) = *reinterpret_cast<const float4*>(patch + r * PLD + c4);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (want_db) db_fold();
}

// the pipelined kernel takes the call: fp32 operands, 16-byte aligned rows
static inline bool w3r_dw_ok(int ty, int tx, const L32DwArgs& a) {
  if (ty != GT_F32 || tx != GT_F32 || (a.dy_rows && a.ymask) || (a.groups > 1 && (a.dy_rows || a.x2))) return false;
  if (a.N % 4 || a.K % 4 || a.ldy % 4 || a.ldx % 4 || (a.x2 && (a.ldx2 % 4 || a.x_split % 4))) return false;
  return (((uintptr_t)a.dy | (uintptr_t)a.x | (uintptr_t)a.ymask | (uintptr_t)a.x2) & 15) == 0;
}
// block shapes of k_lin3r_dw: 0 = 160 x 160 (5 x 5 tiles per wave), 1 = 224 x 128 (7 x 4: the PNA towers' 204 x 340 post-Linear
// pads to 224 x 384 instead of 320 x 480)
// (a 224 x 96 shape -- 7 x 3 tiles, 126 KB of LDS and 372 registers, so that a 30-KB block of the main stream fits beside it -- measured the
// same as 224 x 128 on Code2-PNA: 33.7-33.9 k graphs/s both; not kept)
static inline int w3r_dw_zt(int shape) { return shape == 1 ? 224 : 160; }
static inline int w3r_dw_xt(int shape) { return shape == 1 ? 128 : 160; }
static inline int w3r_dw_pick_shape(int64_t N, int64_t K) {   // least padded area, ties -> 0
  int best = 0;
  int64_t area = -1;
  for (int sh = 0; sh < 2; ++sh) {
    const int64_t ar = gt_cdiv(N, w3r_dw_zt(sh)) * w3r_dw_zt(sh) * gt_cdiv(K, w3r_dw_xt(sh)) * w3r_dw_xt(sh);
    if (area < 0 || ar < area) { area = ar; best = sh; }
  }
  return best;
}
template <int TN, int TK>
static inline void w3r_launch_dw_shape(dim3 grid, hipStream_t stream, const L32DwArgs& a) {
  constexpr int LDS = 2 * 2 * (3 * 32 * (32 * TN + 8) + 3 * 32 * (32 * TK + 8));
  static std::mutex mu;
  static bool done[16] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  {
    std::lock_guard<std::mutex> lk(mu);
    if (dev < 0 || dev >= 16 || !done[dev]) {
      (void)hipFuncSetAttribute((const void*)(k_lin3r_dw<true, false, false, TN, TK>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      (void)hipFuncSetAttribute((const void*)(k_lin3r_dw<false, false, false, TN, TK>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      if (dev >= 0 && dev < 16) done[dev] = true;
    }
  }
  if (a.ymask) hipLaunchKernelGGL((k_lin3r_dw<true, false, false, TN, TK>), grid, dim3(256), LDS, stream, a);
  else hipLaunchKernelGGL((k_lin3r_dw<false, false, false, TN, TK>), grid, dim3(256), LDS, stream, a);
}
static inline void w3r_launch_dw(dim3 grid, hipStream_t stream, const L32DwArgs& a, int shape = 0) {
  if (shape == 1) { w3r_launch_dw_shape<7, 4>(grid, stream, a); return; }   // (no row map there: w3r_dw_ok with groups)
  constexpr int LDS = 2 * 6 * W3D_PLANE * 2;
  static std::mutex mu;
  static bool done[16] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  {
    std::lock_guard<std::mutex> lk(mu);
    if (dev < 0 || dev >= 16 || !done[dev]) {
      (void)hipFuncSetAttribute((const void*)(k_lin3r_dw<true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      (void)hipFuncSetAttribute((const void*)(k_lin3r_dw<false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      (void)hipFuncSetAttribute((const void*)(k_lin3r_dw<false, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      if (dev >= 0 && dev < 16) done[dev] = true;
    }
  }
  if (a.dy_rows) hipLaunchKernelGGL((k_lin3r_dw<false, true, true>), grid, dim3(512), LDS, stream, a);   // (w3r_dw_ok: no gate with a row map)
  else if (a.ymask) hipLaunchKernelGGL((k_lin3r_dw<true, false, true>), grid, dim3(512), LDS, stream, a);
  else hipLaunchKernelGGL((k_lin3r_dw<false, false, false>), grid, dim3(256), LDS, stream, a);   // (the 8-wave producer / consumer form of this case: 3 % slower in the step, see the kernel's header)
}
