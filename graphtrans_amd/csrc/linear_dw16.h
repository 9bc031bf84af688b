// linear_dw16.h — weight gradient of the encoder linears on bf16 token rows: dW[N][K] = dY^T X, db = colsum(dY), both operands
// [M][cols] bf16 in memory, contraction over the M = 32 k token rows (modules/transformer_encoder.py:28-32: in_proj / out_proj /
// linear1 / linear2 of nn.TransformerEncoderLayer, 16 weight gradients per Code2 step).  Included by linear.hip inside its anonymous
// namespace.
//
// Why a second dW kernel: k_linear_dw stages both operands through registers into ONE LDS stage (two barriers per 64 rows, the loads
// of the next stage one step ahead) and needs ~400 blocks x 127 M-splits to hide that latency -- 25 MB of fp32 partials for 33 MB of
// operands, 19 + 12 us per call with its reduce (0.10 of HBM, the largest kernel total of the step).  Here nothing passes through
// registers: a stage (32 token rows x 128 columns of dY and of X, 16 KB) reaches the LDS by global_load_lds_dwordx4 (LDS-DMA), a ring
// of four stages keeps three in flight (48 KB per CU, ~1.5 us of latency covered at the L2 rate), and the fragments are read
// TRANSPOSED by ds_read_b64_tr_b16 (the token index becomes the k-slot).  Waits are counted (s_waitcnt vmcnt(8 / 4 / 0)) and the
// barrier is the bare s_barrier: __syncthreads() carries a workgroup fence, i.e. vmcnt(0), which would drain the ring every step.
// One block per CU (64 KB) streams its M range at the rate the L2 delivers; ~256 blocks -> 64-85 splits, a third of the partials.
//
// LDS stage: [operand 0 = dY, 1 = X][32 rows][256 B]; the 16-byte chunk c of row r sits at slot c ^ ((r & 3) << 1): the four rows a
// 16-lane group reads in one transposed read (rows r..r+3, 32 B each) fall into four different 32-byte bank groups.  The DMA writes
// lane-linear (1 KB = 4 rows per instruction), so the swizzle is applied to the SOURCE address of each lane.
#pragma once

constexpr int D16_STAGES = 4;
constexpr int D16_ROWS = 32;               // token rows per stage = one 32-deep MFMA step
constexpr int D16_T = 128;                 // output tile: 128 n x 128 k, 4 waves as 2 x 2, wave = 64 x 64
constexpr int D16_OP_BYTES = D16_ROWS * D16_T * 2;   // 8 KB
constexpr int D16_STAGE_BYTES = 2 * D16_OP_BYTES;    // 16 KB

struct Dw16Args {
  const gt_bf16* dy;
  const gt_bf16* x;
  float* part;     // [splits][N][K]
  float* dbpart;   // [splits][N] or null
  int64_t M, N, K, ldy, ldx;
  int splits, ntx, ntiles;
  int64_t m_per_split;   // multiple of 32
};

typedef __attribute__((address_space(3))) void d16_lds_void;
typedef const __attribute__((address_space(1))) void d16_glb_void;

__global__ void __launch_bounds__(256) k_dw16(Dw16Args a) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[D16_STAGES * D16_STAGE_BYTES];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int wn = wid & 1, wk = wid >> 1;
  int64_t split_;
  int tile_;
  tile_of_block(a.ntiles, split_, tile_);   // the tiles of one M-split share an XCD (same dY / X rows, same L2)
  if (split_ >= a.splits) return;
  const int split = (int)split_;
  const int tx = tile_ % a.ntx, ty = tile_ / a.ntx;
  const int64_t n0 = (int64_t)tx * D16_T, k0 = (int64_t)ty * D16_T;
  const int64_t mb = (int64_t)split * a.m_per_split;
  const int64_t me = mb + a.m_per_split < a.M ? mb + a.m_per_split : a.M;
  const int nsteps = me > mb ? (int)((me - mb + D16_ROWS - 1) / D16_ROWS) : 0;

  // ---- DMA: instruction i of a stage = operand i >> 3, rows (i & 7) * 4 .. + 3; wave w issues i = w, w + 4, w + 8, w + 12.
  // Lane L lands at slot (row L >> 4, chunk position L & 15) and therefore FETCHES chunk (L & 15) ^ ((L >> 4) << 1) of that row.
  const int rl = lane >> 4;
  const int src_chunk = (lane & 15) ^ (rl << 1);
  const gt_bf16* src_base[4];
  int64_t src_ld[4];
  int row_in_stage[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = wid + 4 * q;
    const bool isx = i >= 8;
    row_in_stage[q] = (i & 7) * 4 + rl;
    src_ld[q] = isx ? a.ldx : a.ldy;
    src_base[q] = (isx ? a.x + k0 : a.dy + n0) + src_chunk * 8;
  }
  auto issue = [&](int st) {
    unsigned char* dst = smem + (st % D16_STAGES) * D16_STAGE_BYTES;
    const int64_t m0 = mb + (int64_t)st * D16_ROWS;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int64_t r = m0 + row_in_stage[q];
      r = r < a.M ? r : a.M - 1;   // rows past the end are zeroed in the LDS before they are read (below)
      __builtin_amdgcn_global_load_lds((d16_glb_void*)(src_base[q] + r * src_ld[q]), (d16_lds_void*)(dst + (wid + 4 * q) * 1024), 16, 0, 0);
    }
  };

  // ---- fragment addresses (LDS byte addresses relative to a stage): transposed read q of a fragment covers rows
  // g*8 + q*4 + (n >> 2), 4 elements at column c0 + (n & 3) * 4.
  // The reads are inline asm on purpose: hipcc's waitcnt pass puts s_waitcnt vmcnt(0) in front of every ds_read_b64_tr_b16
  // INTRINSIC that follows an LDS-DMA (the intrinsic carries no memory operand it could disambiguate; plain ds_read_b128 does
  // and is left alone) -- which drains the ring at every step.  The asm form is invisible to that pass; its lgkmcnt wait is
  // written out below, with the fragments as operands so that the MFMAs cannot move above it.
  uint32_t offz[4], offx[4];
  {
    const int rr = g * 8 + (n >> 2);
    const int sw = (n >> 2) << 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cz = (wn * 64 + j * 16 + (n & 3) * 4) >> 3, cx = (wk * 64 + j * 16 + (n & 3) * 4) >> 3;
      offz[j] = (uint32_t)(rr * 256 + ((cz ^ sw) << 4) + (n & 1) * 8);
      offx[j] = (uint32_t)(rr * 256 + ((cx ^ sw) << 4) + (n & 1) * 8);
    }
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
#define D16_TR(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #OFF : "=&v"(dst) : "v"(addr) : "memory")

  f32x4 acc[4][4];   // [n tile j][k tile i]
  f32x4 accb[4];     // db: dY^T . ones (k-tile 0 blocks, wk == 0 waves)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    accb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool want_db = a.dbpart != nullptr && ty == 0 && wk == 0;
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u));

  for (int s = 0; s < D16_STAGES - 1 && s < nsteps; ++s) issue(s);
  for (int s = 0; s < nsteps; ++s) {
    const int ahead = nsteps - 1 - s < D16_STAGES - 2 ? nsteps - 1 - s : D16_STAGES - 2;   // stages issued behind s: they may stay in flight
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // stage s is complete (every wave's pieces) and every wave has left stage s - 1
    if (s + D16_STAGES - 1 < nsteps) issue(s + D16_STAGES - 1);
    unsigned char* st = smem + (s % D16_STAGES) * D16_STAGE_BYTES;
    const int64_t m0 = mb + (int64_t)s * D16_ROWS;
    if (m0 + D16_ROWS > a.M) {   // the last rows of the matrix: zero the row slots past M (their DMA fetched row M - 1 again)
      const int first = (int)(a.M - m0);
      for (int c = threadIdx.x; c < 2 * (D16_ROWS - first) * 16; c += 256) {
        const int op = c / ((D16_ROWS - first) * 16), rem = c % ((D16_ROWS - first) * 16);
        *reinterpret_cast<uint4*>(st + op * D16_OP_BYTES + (first + rem / 16) * 256 + (rem % 16) * 16) = make_uint4(0, 0, 0, 0);
      }
      __syncthreads();
    }
    uint2 zl[4], zh[4], xl[4], xh[4];
    const uint32_t sb = lds0 + (uint32_t)((s % D16_STAGES) * D16_STAGE_BYTES);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t az = sb + offz[j], ax = sb + offx[j];
      D16_TR(zl[j], az, 0);
      D16_TR(zh[j], az, 1024);
      D16_TR(xl[j], ax, 8192);
      D16_TR(xh[j], ax, 9216);
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(zl[0]), "+v"(zl[1]), "+v"(zl[2]), "+v"(zl[3]), "+v"(zh[0]), "+v"(zh[1]), "+v"(zh[2]), "+v"(zh[3]),
                   "+v"(xl[0]), "+v"(xl[1]), "+v"(xl[2]), "+v"(xl[3]), "+v"(xh[0]), "+v"(xh[1]), "+v"(xh[2]), "+v"(xh[3])
                 :: "memory");
    bf16x8_t fz[4], fx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      fz[j] = __builtin_bit_cast(bf16x8_t, make_uint4(zl[j].x, zl[j].y, zh[j].x, zh[j].y));
      fx[j] = __builtin_bit_cast(bf16x8_t, make_uint4(xl[j].x, xl[j].y, xh[j].x, xh[j].y));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fz[j], fx[i], acc[j][i], 0, 0, 0);
    if (want_db) {
#pragma unroll
      for (int j = 0; j < 4; ++j) accb[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fz[j], ones, accb[j], 0, 0, 0);
    }
  }
  __syncthreads();   // the ring becomes the epilogue's patches

  // acc[j][i][r] = dW[n-index wn*64 + j*16 + g*4 + r][k-index wk*64 + i*16 + n] -> patch[16 n rows][64 k cols] -> 256-byte row segments
  float* patch = reinterpret_cast<float*>(smem) + wid * PATCH_FLOATS;
  float* part = a.part + (int64_t)split * a.N * a.K;
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
      *reinterpret_cast<float4*>(part + row * a.K + col) = *reinterpret_cast<const float4*>(patch + r * PATCH_LD + c4);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (want_db && n == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) a.dbpart[(int64_t)split * a.N + n0 + wn * 64 + j * 16 + g * 4 + r] = accb[j][r];
  }
}

#undef D16_TR

constexpr int D16_BLOCKS = 256;   // one block per CU
// bf16 rows on both sides, whole 128 x 128 tiles, 16-byte aligned rows, enough rows to fill the ring
static inline bool dw16_ok(int x_dtype, int y_dtype, int compute, const void* x, const void* dy, const void* ymask, int64_t M, int64_t N,
                           int64_t K, int64_t ldx, int64_t ldy, int groups) {
  return x_dtype == GT_BF16 && y_dtype == GT_BF16 && compute == GT_BF16 && groups == 1 && !ymask && M >= 1024 &&
         N % D16_T == 0 && K % D16_T == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0;
}
static inline int dw16_splits(int64_t M, int64_t N, int64_t K, int cap) {
  const int64_t tiles = (N / D16_T) * (K / D16_T);
  // two blocks per CU (2 x 64 KB of LDS) once the contraction is long enough that the doubled partials stay small beside the operands:
  // at M = 131 k (the Erdos-Renyi stress) 512 blocks run the N or K = 1024 shapes in 125-133 us against 174-185 us (tools/dw16_bench.py)
  const int64_t target = M >= 65536 ? 2 * D16_BLOCKS : D16_BLOCKS;
  int64_t s = target / tiles;
  const int64_t maxs = gt_cdiv(M, 4 * D16_ROWS);   // at least four stages per split
  if (s > maxs) s = maxs;
  if (s > cap) s = cap;
  return s < 1 ? 1 : (int)s;
}
