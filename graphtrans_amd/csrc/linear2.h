// linear2.h — k_lin2: the bf16 GEMMs of the Transformer encoder layers on MANY token rows (in_proj / linear1 / linear2, forward and dX,
// modules/transformer_encoder.py:28-32) with BOTH operands streamed through an LDS ring by LDS-DMA.  Included by linear.hip inside its
// anonymous namespace, after linear1.h (same fragment-order weight images, same L1Args, same epilogue arithmetic).
//
// Why a second kernel.  k_lin1 keeps the weight in registers and moves only the activation tile -- through REGISTERS (two 16-byte loads
// per thread and 128-deep chunk, one chunk ahead): 32 KB in flight per CU.  At Code2's 32 k tokens (two row tiles per block) that is the
// right trade; at the Erdos-Renyi stress (131 k tokens, d_model 256, ffn 1024) the kernel runs at the rate of its loads in flight --
// 2.1-2.4 TB/s on algorithmic bytes (linear1 forward 163 us, linear2's dX 179 us: `tools/gemm1_bench.py 131328`) --, and a contraction of
// 768 / 1024 does not fit the registers at all (the tiled kernels: 145-190 us).  Here
//   * a block is 256 rows x 256 output columns (8 waves = 2 row halves x 4 column quarters, wave = 128 x 64 = 8 x 4 accumulator tiles);
//   * a stage = one 32-deep k-step of both operands: the 256 x 32 activation rows (16 KB) and the 16 weight fragments of the column
//     block (16 KB, the image's fragment order IS the LDS order), brought in by global_load_lds_dwordx4; a ring of four stages keeps
//     three in flight (96 KB per CU), waits are counted (s_waitcnt vmcnt(N)), the barrier is the bare s_barrier, and the fragment reads
//     are inline asm (hipcc's wait-count pass drains vmcnt in front of LDS reads it can see behind an LDS-DMA: linear_dw16.h);
//   * the ring runs across work items (row tile x column block): the next item's first stages are in flight under this item's epilogue;
//   * activation rows are 64 bytes in the stage; the 16-byte k-slice c of row r sits at slot c ^ F[(r >> 2) & 3], F = {0, 3, 2, 1}:
//     conflict-free ds_read_b128 in all four 16-lane service groups (MI355X_MICROARCH.md, LDS); the DMA writes lane-linear, so the
//     swizzle is applied to each lane's SOURCE address;
//   * the column blocks of a row tile run on one XCD (one L2: the activation tile leaves HBM once);
//   * the epilogue runs from the accumulators: lanes g and g ^ 1 exchange halves so that every lane owns 8 consecutive columns of its
//     row (16-byte loads of the gate / addends, 16-byte stores), bias / ReLU / dropout / gradient gate / two addends as k_lin1's.
// Covered: output columns 128 or a multiple of 256, contraction a multiple of 128 up to 1024 (k_lin1's image shapes), >= W2_MIN_M rows, 16-byte aligned rows; no LayerNorm
// epilogues (those stay with k_lin1 / the stand-alone kernels).
#pragma once

constexpr int W2_TM = 256;                 // rows per work item
constexpr int W2_STAGES = 4;
constexpr int W2_A_BYTES = W2_TM * 64;     // 256 rows x 32 k of bf16
constexpr int W2_THREADS = 512;
constexpr int W2_BLOCKS = 256;             // one block per CU
constexpr int64_t W2_MIN_M = 2048;

typedef uint32_t w2_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void w2_lds_void;
typedef const __attribute__((address_space(1))) void w2_glb_void;

template <int OFF>
__device__ __forceinline__ void w2_rd(w2_u32x4& d, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void w2_vmwait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Epilogue of one work item, from the accumulators: acc[i][j][r] = out[row i*16 + n][column j*16 + g*4 + r] of the wave's 128 x (NTW*16)
// patch.  Lanes g and g ^ 1 swap halves -- even g keeps tile 2jp (its 4 columns + the partner's next 4), odd g takes tile 2jp + 1 -- so
// every lane owns 8 consecutive columns of its row.  FWD: bias / ReLU / dropout (no loads besides the bias, fetched once per item);
// !FWD (dX): gradient gate / two addends -- their loads are UNCONDITIONAL (an absent tensor reads the activation base: one cached line
// for the whole wave) so that the six loads of a row tile are issued together instead of one round trip each.  FULL: every row of the
// tile exists (no row guards: one basic block per row tile).
template <int NTW, bool FWD, bool FULL>
__device__ __forceinline__ void w2_epilogue(const L1Args& a, f32x4 (&acc)[8][NTW], int cnt, int64_t mrow0, int col0, int g) {
  constexpr int NP = NTW / 2;
  const bool odd = (g & 1) != 0;
  int colp[NP];
  float bv[NP][8];
#pragma unroll
  for (int jp = 0; jp < NP; ++jp) {
    colp[jp] = col0 + (2 * jp + (odd ? 1 : 0)) * 16 + (g & 2) * 4;
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[jp][e] = 0.f;
    if constexpr (FWD) {
      if (a.bias) {
        const float4 b0 = *reinterpret_cast<const float4*>(a.bias + colp[jp]), b1 = *reinterpret_cast<const float4*>(a.bias + colp[jp] + 4);
        bv[jp][0] = b0.x; bv[jp][1] = b0.y; bv[jp][2] = b0.z; bv[jp][3] = b0.w;
        bv[jp][4] = b1.x; bv[jp][5] = b1.y; bv[jp][6] = b1.z; bv[jp][7] = b1.w;
      }
    }
  }
  const bool has_gate = a.gate != nullptr, has_a1 = a.add1 != nullptr, has_a2 = a.add2 != nullptr;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (i >= 4 && i >= cnt) break;   // (wave-uniform: the row half holds 4 .. 8 tiles)
    const int64_t m = mrow0 + i * 16;
    const int64_t mc = FULL ? m : (m < a.M ? m : a.M - 1);
    uint4 gm[NP], d1[NP], d2[NP];
    if constexpr (!FWD) {
#pragma unroll
      for (int jp = 0; jp < NP; ++jp) {
        const int64_t o = mc * a.ldo + colp[jp];
        gm[jp] = *reinterpret_cast<const uint4*>(has_gate ? a.gate + o : a.a);
        d1[jp] = *reinterpret_cast<const uint4*>(has_a1 ? a.add1 + o : a.a);
        d2[jp] = *reinterpret_cast<const uint4*>(has_a2 ? a.add2 + o : a.a);
      }
    }
#pragma unroll
    for (int jp = 0; jp < NP; ++jp) {
      const f32x4 o0 = acc[i][2 * jp], o1 = acc[i][2 * jp + 1];
      acc[i][2 * jp] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[i][2 * jp + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float rcv = __shfl_xor(odd ? o0[r] : o1[r], 16, 64);
        v[r] = odd ? rcv : o0[r];
        v[4 + r] = odd ? o1[r] : rcv;
      }
      const int col = colp[jp];
      if constexpr (FWD) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bv[jp][e];
        if (a.act == 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (a.thr) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = lin_hash(a.s0, a.s1, (uint32_t)mc, (uint32_t)(col + e)) >= a.thr ? v[e] * a.inv_keep : 0.f;
        }
      } else {
        const uint32_t ug[4] = {gm[jp].x, gm[jp].y, gm[jp].z, gm[jp].w};
        const uint32_t u1[4] = {d1[jp].x, d1[jp].y, d1[jp].z, d1[jp].w};
        const uint32_t u2[4] = {d2[jp].x, d2[jp].y, d2[jp].z, d2[jp].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float lo = v[2 * e], hi = v[2 * e + 1];
          lo = has_gate ? gt_gate(lo, __uint_as_float(ug[e] << 16), a.gate_inv_keep) : lo;
          hi = has_gate ? gt_gate(hi, __uint_as_float(ug[e] & 0xffff0000u), a.gate_inv_keep) : hi;
          lo += has_a1 ? __uint_as_float(u1[e] << 16) : 0.f;
          hi += has_a1 ? __uint_as_float(u1[e] & 0xffff0000u) : 0.f;
          lo += has_a2 ? __uint_as_float(u2[e] << 16) : 0.f;
          hi += has_a2 ? __uint_as_float(u2[e] & 0xffff0000u) : 0.f;
          v[2 * e] = lo;
          v[2 * e + 1] = hi;
        }
      }
      const uint4 packed = make_uint4(gt_pack_bf16(v[0], v[1]), gt_pack_bf16(v[2], v[3]), gt_pack_bf16(v[4], v[5]), gt_pack_bf16(v[6], v[7]));
      if (FULL || m < a.M) *reinterpret_cast<uint4*>(a.out + mc * a.ldo + col) = packed;
    }
  }
}

// NTW = n-tiles per wave: the column block is 4 * NTW * 16 columns (NTW = 4: 256, NTW = 2: 128)
template <int NTW, bool FWD>
__global__ void __launch_bounds__(W2_THREADS, 1) k_lin2(L1Args a) {
  static_assert(NTW == 2 || NTW == 4, "column blocks of 128 or 256");
  constexpr int NT = 4 * NTW;                       // n-tiles (weight fragments per k-step) of a column block
  constexpr int STAGE = W2_A_BYTES + NT * 1024;
  constexpr int NWP = (NT + 7) / 8;                 // weight pieces per wave and stage
  constexpr int PER = 2 + NWP;                      // DMA instructions per wave and stage
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem2[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4, wm = wid >> 2, wn = wid & 3;
  const int KS = a.K / 32;
  const int ncb = a.ncb;
  // rows per work item (a.tm, a multiple of 16 chosen by the launcher so that the items fill whole rounds of blocks): mt 16-row tiles,
  // the first h0 belong to the waves of row half 0, the rest to row half 1 (4 .. 8 each)
  const int mt = a.tm >> 4, h0 = (mt + 1) >> 1;
  const int cnt = wm ? mt - h0 : h0, tile0 = wm ? h0 : 0;
  // work items of this block: XCD x owns the row tiles x, x + 8, ...; its items (row tile, column block), column block fastest, are
  // dealt to the XCD's blocks round-robin -- the column blocks of a row tile run side by side on one XCD
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int nrt = a.row_tiles > xcd ? (a.row_tiles - xcd + 7) / 8 : 0;
  const int nitems = nrt * ncb;
  const int my_items = slot < nitems ? (nitems - slot + nslot - 1) / nslot : 0;
  const int nsteps = my_items * KS;
  if (nsteps == 0) return;

  // ---- DMA side: pieces of a stage = mt x (16 activation rows x 64 B) + NT weight fragments; wave w issues activation pieces w, w + 8
  // and weight pieces w (, w + 8); an index past the last piece repeats the last one (same bytes to the same place).  Lane L of an
  // activation piece lands at (row L >> 2, slot L & 3) and FETCHES k-slice (L & 3) ^ F[(row >> 2) & 3] of that row; (row >> 2) & 3 == L >> 4.
  const int a_ksl = (lane & 3) ^ ((4 - g) & 3);
  const int a_rl = lane >> 2;
  const int pa0 = wid < mt ? wid : mt - 1, pa1 = wid + 8 < mt ? wid + 8 : mt - 1;
  const int pw0 = wid < NT ? wid : NT - 1, pw1 = wid + 8 < NT ? wid + 8 : NT - 1;
  const unsigned char* img_l = a.img + lane * 16;
  const gt_bf16* a_src0;
  const gt_bf16* a_src1;
  const unsigned char* w_src0;
  const unsigned char* w_src1;
  int is_q = slot, is_kc = 0;   // the item / k-step the NEXT issue belongs to
#define W2_SET_ITEM()                                                                                              \
  do {                                                                                                             \
    const int rt_ = xcd + 8 * (is_q / ncb), cb_ = is_q % ncb;                                                      \
    int64_t r0_ = (int64_t)rt_ * a.tm + pa0 * 16 + a_rl, r1_ = (int64_t)rt_ * a.tm + pa1 * 16 + a_rl;             \
    r0_ = r0_ < a.M ? r0_ : a.M - 1;                                                                               \
    r1_ = r1_ < a.M ? r1_ : a.M - 1;                                                                               \
    a_src0 = a.a + r0_ * a.lda + a_ksl * 8;                                                                        \
    a_src1 = a.a + r1_ * a.lda + a_ksl * 8;                                                                        \
    w_src0 = img_l + ((int64_t)(cb_ * NT + pw0) * KS) * 1024;                                                      \
    w_src1 = img_l + ((int64_t)(cb_ * NT + pw1) * KS) * 1024;                                                      \
  } while (0)
  // piece P (0, 1: activation rows; 2, 3: weight fragments) of the stage behind `dst_`
#define W2_PIECE(P, dst_)                                                                                          \
  do {                                                                                                             \
    if constexpr ((P) == 0) __builtin_amdgcn_global_load_lds((w2_glb_void*)(a_src0 + is_kc * 32), (w2_lds_void*)((dst_) + pa0 * 1024), 16, 0, 0); \
    if constexpr ((P) == 1) __builtin_amdgcn_global_load_lds((w2_glb_void*)(a_src1 + is_kc * 32), (w2_lds_void*)((dst_) + pa1 * 1024), 16, 0, 0); \
    if constexpr ((P) == 2) __builtin_amdgcn_global_load_lds((w2_glb_void*)(w_src0 + (int64_t)is_kc * 1024), (w2_lds_void*)((dst_) + W2_A_BYTES + pw0 * 1024), 16, 0, 0); \
    if constexpr ((P) == 3 && NWP == 2) __builtin_amdgcn_global_load_lds((w2_glb_void*)(w_src1 + (int64_t)is_kc * 1024), (w2_lds_void*)((dst_) + W2_A_BYTES + pw1 * 1024), 16, 0, 0); \
  } while (0)
#define W2_ADVANCE()                                                                                               \
  do {                                                                                                             \
    if (++is_kc == KS) {                                                                                           \
      is_kc = 0;                                                                                                   \
      is_q += nslot;                                                                                               \
      if (is_q < nitems) W2_SET_ITEM();                                                                            \
    }                                                                                                              \
  } while (0)
  W2_SET_ITEM();

  // ---- fragment addresses relative to a stage
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem2;
  const uint32_t offA = (uint32_t)((tile0 * 16 + n) * 64 + ((g ^ ((4 - (n >> 2)) & 3)) << 4));
  const uint32_t offW = (uint32_t)(W2_A_BYTES + wn * NTW * 1024 + lane * 16);

  f32x4 acc[8][NTW];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int s = 0; s < W2_STAGES - 1 && s < nsteps; ++s) {
    unsigned char* dst = smem2 + s * STAGE;
    W2_PIECE(0, dst); W2_PIECE(1, dst); W2_PIECE(2, dst); W2_PIECE(3, dst);
    W2_ADVANCE();
  }
  int q = slot, kc = 0;
  for (int s = 0; s < nsteps; ++s) {
    // Counted wait: the pieces of stages s + 1, s + 2 may stay in flight.  The stores (and consumed loads) of an epilogue sit in the same
    // counter: loads complete in order among loads, so "at most 2 PER operations outstanding" still means stage s has landed -- it only
    // waits for the stores as well.
    const int left = nsteps - 1 - s;
    if (left == 0) w2_vmwait<0>();
    else if (left == 1) w2_vmwait<PER>();
    else w2_vmwait<2 * PER>();
    __builtin_amdgcn_s_barrier();   // stage s is complete (every wave's pieces) and every wave has left stage s - 1
    const bool more = s + W2_STAGES - 1 < nsteps;
    unsigned char* dst = smem2 + ((s + W2_STAGES - 1) & (W2_STAGES - 1)) * STAGE;
    const uint32_t sb = lds0 + (uint32_t)((s & (W2_STAGES - 1)) * STAGE);
    // all fragment reads up front, in the order the MFMA groups need them (lgkmcnt counts them down in order); the next stage's DMA
    // pieces ride one at a time in front of the four MFMA groups: an LDS-DMA instruction costs ~100 issue cycles, during which the
    // SIMD's other wave has the matrix pipe
    w2_u32x4 xa[8], wb[NTW];
    {
      const uint32_t pa = sb + offA, pw = sb + offW;
      w2_rd<0>(wb[0], pw);
      w2_rd<1024>(wb[1], pw);
      if constexpr (NTW == 4) {
        w2_rd<2048>(wb[2], pw);
        w2_rd<3072>(wb[3], pw);
      }
      w2_rd<0>(xa[0], pa);
      w2_rd<1024>(xa[1], pa);
      w2_rd<2048>(xa[2], pa);
      w2_rd<3072>(xa[3], pa);
      w2_rd<4096>(xa[4], pa);
      w2_rd<5120>(xa[5], pa);
      w2_rd<6144>(xa[6], pa);
      w2_rd<7168>(xa[7], pa);
    }
#define W2_MMA(i_)                                                                                                                      \
  _Pragma("unroll") for (int j = 0; j < NTW; ++j)                                                                                      \
    acc[i_][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wb[j]), __builtin_bit_cast(bf16x8_t, xa[i_]), acc[i_][j], 0, 0, 0)
    if (more) W2_PIECE(0, dst);
    if constexpr (NTW == 4) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(wb[0]), "+v"(wb[1]), "+v"(wb[2]), "+v"(wb[3]), "+v"(xa[0]), "+v"(xa[1]) :: "memory");
    else asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(wb[0]), "+v"(wb[1]), "+v"(xa[0]), "+v"(xa[1]) :: "memory");
    W2_MMA(0);
    W2_MMA(1);
    __builtin_amdgcn_sched_barrier(0);
    if (more) W2_PIECE(1, dst);
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(xa[2]), "+v"(xa[3]) :: "memory");
    W2_MMA(2);
    W2_MMA(3);
    __builtin_amdgcn_sched_barrier(0);
    if (more) W2_PIECE(2, dst);
    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(xa[4]), "+v"(xa[5]) :: "memory");
    if (cnt > 4) { W2_MMA(4); }
    if (cnt > 5) { W2_MMA(5); }
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
      W2_PIECE(3, dst);
      W2_ADVANCE();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xa[6]), "+v"(xa[7]) :: "memory");
    if (cnt > 6) { W2_MMA(6); }
    if (cnt > 7) { W2_MMA(7); }
#undef W2_MMA
    if (kc != KS - 1) {
      ++kc;
      continue;
    }
    // ---- epilogue of this item (w2_epilogue above); tail tiles take the row-guarded form
    const int rt = xcd + 8 * (q / ncb), cb = q % ncb;
    const int col0 = (cb * NT + wn * NTW) * 16;
    const int64_t mrow0 = (int64_t)rt * a.tm + tile0 * 16 + n;
    if ((int64_t)rt * a.tm + a.tm <= a.M) w2_epilogue<NTW, FWD, true>(a, acc, cnt, mrow0, col0, g);
    else w2_epilogue<NTW, FWD, false>(a, acc, cnt, mrow0, col0, g);
    kc = 0;
    q += nslot;
  }
#undef W2_SET_ITEM
#undef W2_PIECE
#undef W2_ADVANCE
}

// ---- shapes -------------------------------------------------------------------------------------------------------------------
// (output columns R, contraction C) this kernel covers; the images are k_w1_image's (any R % 16 == 0, C % 32 == 0)
static inline bool w2_covered(int64_t R, int64_t C) {
  return R > 0 && C >= 128 && C % 128 == 0 && C <= 1024 && (R == 128 || R % 256 == 0) && R <= 4096;
}
static inline bool w2_args_ok(const L1Args& a) {
  if ((a.gate || a.add1 || a.add2) && (a.bias || a.act || a.thr)) return false;   // forward OR dX epilogue
  return w2_covered(a.N, a.K) && a.M >= W2_MIN_M && !a.ln_out && !a.lnb_part && (a.act == 0 || a.act == 1) && a.lda % 8 == 0 && a.ldo % 8 == 0 &&
         ((uintptr_t)a.a & 15) == 0 && ((uintptr_t)a.out & 15) == 0 && (!a.gate || ((uintptr_t)a.gate & 15) == 0) &&
         (!a.add1 || ((uintptr_t)a.add1 & 15) == 0) && (!a.add2 || ((uintptr_t)a.add2 & 15) == 0);
}

template <int NTW, bool FWD>
static inline bool w2_launch_one(dim3 grid, hipStream_t stream, const L1Args& a) {
  static std::mutex mu;
  static bool set[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  constexpr size_t lds = (size_t)W2_STAGES * (W2_A_BYTES + 4 * NTW * 1024);
  if (dev >= 0 && dev < 64) {
    std::lock_guard<std::mutex> lk(mu);
    if (!set[dev]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lin2<NTW, FWD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
      set[dev] = true;
    }
  }
  hipLaunchKernelGGL((k_lin2<NTW, FWD>), grid, dim3(W2_THREADS), lds, stream, a);
  return true;
}

static inline bool w2_launch(hipStream_t stream, L1Args& a) {
  if (!w2_args_ok(a)) return false;
  const int ntw = a.N == 128 ? 2 : 4;
  a.ncb = a.N / (64 * ntw);
  // rows per work item: the height (a multiple of 16) whose items fill whole rounds of the 256 blocks best -- 131 328 rows (the ER
  // stress: 513 tiles of 256) would run 3 rounds for 2.004 rounds of work; 176-row items run 3 rounds of 0.69
  int best_tm = W2_TM;
  int64_t best_cost = INT64_MAX;
  for (int tm = W2_TM; tm >= 128; tm -= 16) {
    const int64_t items_ = gt_cdiv(a.M, tm) * a.ncb;
    const int64_t cost = gt_cdiv(items_, W2_BLOCKS) * (tm + 24);   // (+ the per-item epilogue / hand-over, in row units)
    if (cost < best_cost) { best_cost = cost; best_tm = tm; }
  }
  a.tm = best_tm;
  a.row_tiles = (int)gt_cdiv(a.M, a.tm);
  int64_t items = (int64_t)a.row_tiles * a.ncb;
  int blocks = items < W2_BLOCKS ? (int)gt_cdiv(items, 8) * 8 : W2_BLOCKS;
  const dim3 grid((unsigned)blocks);
  const bool fwd = !a.gate && !a.add1 && !a.add2;
  if (fwd) return ntw == 2 ? w2_launch_one<2, true>(grid, stream, a) : w2_launch_one<4, true>(grid, stream, a);
  return ntw == 2 ? w2_launch_one<2, false>(grid, stream, a) : w2_launch_one<4, false>(grid, stream, a);
}

// k_lin1 is built for <= a few row tiles per block; from this many rows on the ring kernel takes the plain-epilogue GEMMs it covers
// (tools/gemm1_bench.py: M = 131 k, d_model 256 / ffn 1024).  gt_option "lin_ring": 1 = never, 2 = whenever covered.
constexpr int64_t W2_PREFER_M = 65536;
static inline bool w2_take(const L1Args& a, bool stationary_covers) {
  const int opt = gt_opt(GT_OPT_LIN_RING);
  if (opt == 1 || !w2_args_ok(a)) return false;
  if (opt == 3) return !stationary_covers;   // (probe: only the shapes k_lin1 does not cover)
  // (256 x 256 -- out_proj and its dX at d_model 256 -- stays with k_lin1: 40 against 42 us at 131 k rows)
  return opt == 2 || !stationary_covers || (a.M >= W2_PREFER_M && (a.N > 256 || a.K > 256));
}
