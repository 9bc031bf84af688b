// norm_coop.h — training-mode BatchNorm1d as ONE launch per direction: statistics, a grid-wide barrier, apply.
// Included by norm.hip inside its anonymous namespace (uses BnDrop / bn_drop4 / bn_gate of that file).
//
// Reference: h = self.batch_norms[layer](h) of every message-passing layer and its backward (modules/gnn_module.py:84,204; GIN's
// mlp BatchNorm1d(2D), modules/conv.py:18-20) -- N = 31.6 k node rows x 300 channels on Code2, five times per direction.
//
// The three-launch scheme (k_bn_stats_partial -> k_bn_stats_finish -> k_bn_apply; backward: partial -> finish -> apply) is a chain of
// three dependent launches that reads the rows twice (three times backward).  Per Code2 step it was 46 launches and 0.57 ms of kernel
// time, 0.58 ms on the critical path (profiles/r05_timeline_code2.json).  A BatchNorm needs ONE global exchange -- 2 D column sums --
// and the matrix is small against the chip: 9.5 M floats over 250 blocks x 256 threads are 37 float4 per thread.  So:
//   * grid = RB row ranges x CS column slabs of 32 columns (one 128-byte line per row and slab), at most one block per CU
//     (<= the device's CU count: every block can be resident at once -- the barrier below needs that);
//   * block = 256 threads = 8 column chunks x 32 row lanes; a thread walks rows r0 + lane, + 32, ... of its chunk and KEEPS the first
//     32 of them in registers (128 VGPRs: the kernel stays below 200 so that a block still fits a CU whose SIMDs each hold one
//     312-register wave of the weight-gradient GEMM that runs beside it on the overlap stream);
//   * phase 1: shifted column sums (pivot = row 0, as k_bn_stats_partial) / backward: sum dy', sum dy' xhat; fixed-order reduce over
//     the row lanes through LDS; the block's 2 x 32 partials go to the workspace;
//   * grid barrier: one agent-scope counter per call (release add, relaxed spin with s_sleep, acquire fence; bounded spin -- a hung
//     barrier flags the slot and falls through instead of hanging the queue).  The counter must be ZERO at launch: gt_bn_coop_slots
//     hands the library a pool of zeroed slots (the whole-model driver clears one pool per step), otherwise the call clears its own
//     counter with a 4-byte memset in front of the launch;
//   * every block sums the RB partials of ITS slab in rank order (the same value in every block: bitwise reproducible), row range 0
//     also writes save_mean / save_rstd / the running statistics (backward: dweight / dbias);
//   * phase 2 applies from the registers (forward: + ReLU, dropout, residual, the next layer's virtual-node add; backward: x re-read
//     from L2 / Infinity Cache) and writes.  Rows beyond the 32 x 32 a block keeps are re-read.
// Rows are read once (backward: x twice), written once; no finish launch, no second partial pass.
#pragma once

constexpr int CO_NT = 256, CO_CH = 8, CO_RL = 32, CO_RPT_MAX = 16, CO_COLS = 32;
constexpr int CO_BCG = 64;   // broadcast rows staged per block
constexpr uint32_t CO_SPIN_LIMIT = 1u << 22;   // x ~0.3 us of sleep: ~1 s

struct BnCoopArgs {
  const float* x;        // [N][D] the BatchNorm input
  const float* dy;       // backward: [N][D]
  float* out;            // forward: y ; backward: dx
  const float *w, *b;
  const float* resid;    // forward: added behind the dropout, or null
  const float* bcast;    // forward: [.][D] rows added through bidx, or null
  const int32_t* bidx;
  float *mean, *rstd;    // forward: written (saved for the backward); backward: read
  float *rmean, *rvar;   // forward: running statistics or null
  int64_t* nbt;
  float* dweight;        // backward outputs
  float* dbias;
  float* part;           // [RB][CS][2][32]
  uint32_t* ctr;         // the barrier's counter (zero at launch); ctr[1] = 1 when the spin gave up
  int64_t N, D, rows_per;
  int RB, CS, relu;
  float momentum, eps;
  BnDrop drop;
};

// The exchanged data (the blocks' partials) is written and read with agent-scope RELAXED atomics -- coherent across the XCDs' L2s by
// themselves -- and ordered against the counter by completion (s_waitcnt vmcnt(0) in front of the add; the reads sit behind the
// spin's exit branch).  An agent-scope release / acquire pair would write back and invalidate the whole L2 from every block:
// measured 46 us against 29 us for the three launches on 31.6 k x 300 (an L2 walk per block and a cold L2 for phase 2).
__device__ __forceinline__ void co_grid_barrier(uint32_t* ctr, uint32_t target) {
  __syncthreads();   // the partials were stored by wave 0 (row lane 0 = threads 0..7), which also holds thread 0
  if (threadIdx.x == 0) {
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's partial stores are complete
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > CO_SPIN_LIMIT) {
        __hip_atomic_store(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  __syncthreads();
}
__device__ __forceinline__ void co_st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float co_ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ float4 co_ld(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <bool BWD, int CO_RPT>
__global__ void __launch_bounds__(CO_NT, 3) k_bn_coop(BnCoopArgs a) {
  __shared__ float4 sred[2][CO_RL][CO_CH];
  __shared__ float stot[2][CO_COLS];
  __shared__ float sst[2][CO_COLS];   // forward: mean, rstd ; backward: sum dy' / N, sum dy' xhat / N
  // forward with a broadcast addend (the next layer's virtual-node rows, bcast[bidx[row]]): a block's rows belong to a handful of
  // graphs, so the rows g0 .. g0 + 63 of this slab wait in the LDS (fetched beside phase 1): phase 2 has no index -> row chain of
  // dependent global loads
  __shared__ float4 sbc[BWD ? 1 : CO_BCG][CO_CH];
  const int tid = threadIdx.x, ch = tid & (CO_CH - 1), rl = tid / CO_CH;
  const int cs = (int)(blockIdx.x % (unsigned)a.CS), rb = (int)(blockIdx.x / (unsigned)a.CS);
  const int64_t D = a.D;
  const int64_t c4 = (int64_t)cs * CO_COLS + ch * 4;
  const bool cact = c4 < D;
  const int64_t cc = cact ? c4 : 0;
  // (32-bit rows and element offsets against uniform base pointers: co_plan admits N * D < 2^30 only)
  const int r0 = rb * (int)a.rows_per;
  const int r1 = r0 + (int)a.rows_per < (int)a.N ? r0 + (int)a.rows_per : (int)a.N;
  const uint32_t Du = (uint32_t)D, ccu = (uint32_t)cc;
  const float* X = a.x;
  // Row j of this thread = r0 + rl + 32 j.  The rows are walked in ROUNDS of CO_RPT: every pass over a round has its CO_RPT row loads
  // in flight together; phase 1 takes the rounds last to first, so that round 0 is still in the registers when phase 2 starts
  // (the other rounds are read again there).  Address = (uniform base advanced by 32 j rows) + ONE per-thread element offset: the
  // scalar unit does the row arithmetic; rows past the range are not loaded (exec mask).
  const uint32_t voff = (uint32_t)(r0 + rl) * Du + ccu;
  const int64_t rstep = (int64_t)CO_RL * D;
  const int nrounds = (int)((a.rows_per + CO_RL * CO_RPT - 1) / (CO_RL * CO_RPT));
  auto rowof = [&](int j) { return r0 + rl + CO_RL * j; };
  auto ldrow = [&](const float* base, int j) {
    float4 v = gt_zero4();
    if (rowof(j) < r1) v = co_ld(base + (int64_t)j * rstep + voff);
    return v;
  };
  float4 keep[CO_RPT];
  float4 acc0 = gt_zero4(), acc1 = gt_zero4();
  float4 mu = gt_zero4(), rs = gt_zero4(), ww = co_ld(a.w + cc), bb = co_ld(a.b + cc);
  int g0 = 0, gspan = -1;   // forward: first graph of the block's rows, graphs staged in the LDS (-1: none)

  if constexpr (!BWD) {
    if (a.bcast && r1 > r0) {   // (uniform) two dependent round trips that overlap the first round's loads
      g0 = a.bidx[r0];
      const int glast = a.bidx[r1 - 1];
      gspan = glast - g0 + 1 < CO_BCG ? glast - g0 + 1 : CO_BCG;
      for (int j = rl; j < gspan; j += CO_RL) sbc[j][ch] = co_ld(a.bcast + ((uint32_t)(g0 + j) * Du + ccu));
    }
    const float4 piv = co_ld(X + ccu);   // row 0 of the matrix: shifted sums
    auto add_row = [&](float4 v) {
      v = make_float4(v.x - piv.x, v.y - piv.y, v.z - piv.z, v.w - piv.w);
      acc0 = gt_add4(acc0, v);
      acc1 = make_float4(fmaf(v.x, v.x, acc1.x), fmaf(v.y, v.y, acc1.y), fmaf(v.z, v.z, acc1.z), fmaf(v.w, v.w, acc1.w));
    };
    for (int rd = nrounds - 1; rd >= 0; --rd) {
#pragma unroll
      for (int i = 0; i < CO_RPT; ++i) keep[i] = ldrow(X, rd * CO_RPT + i);
#pragma unroll
      for (int i = 0; i < CO_RPT; ++i)
        if (rowof(rd * CO_RPT + i) < r1) add_row(keep[i]);
    }
  } else {
    mu = co_ld(a.mean + cc); rs = co_ld(a.rstd + cc);
  }
  // backward: dy' of a round into the registers (dy loads in flight together, x in batches beside them), optionally summed
  auto bwd_round = [&](int rd, bool sum) {
    constexpr int NB = 4;
#pragma unroll
    for (int i = 0; i < CO_RPT; ++i) keep[i] = ldrow(a.dy, rd * CO_RPT + i);
#pragma unroll
    for (int i0 = 0; i0 < CO_RPT; i0 += NB) {
      float4 v[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) v[u] = ldrow(X, rd * CO_RPT + i0 + u);
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int r = rowof(rd * CO_RPT + i0 + u);
        float4 g = keep[i0 + u];
        if (a.drop.thr) g = bn_drop4(g, a.drop, (uint32_t)r, (uint32_t)cc);
        if (a.relu) g = bn_gate(g, v[u], mu, rs, ww, bb);
        keep[i0 + u] = g;
        if (sum && r < r1) {
          acc0 = gt_add4(acc0, g);
          acc1 = make_float4(fmaf(g.x, (v[u].x - mu.x) * rs.x, acc1.x), fmaf(g.y, (v[u].y - mu.y) * rs.y, acc1.y),
                             fmaf(g.z, (v[u].z - mu.z) * rs.z, acc1.z), fmaf(g.w, (v[u].w - mu.w) * rs.w, acc1.w));
        }
      }
    }
  };
  if constexpr (BWD) {
    for (int rd = nrounds - 1; rd >= 0; --rd) bwd_round(rd, true);
  }

  // ---- the block's column partials: fixed-order sum over the 32 row lanes, stored by row lane 0
  sred[0][rl][ch] = acc0;
  sred[1][rl][ch] = acc1;
  __syncthreads();
  if (rl == 0) {
    float4 t0 = sred[0][0][ch], t1 = sred[1][0][ch];
#pragma unroll 8
    for (int q = 1; q < CO_RL; ++q) {
      t0 = gt_add4(t0, sred[0][q][ch]);
      t1 = gt_add4(t1, sred[1][q][ch]);
    }
    float* p = a.part + ((int64_t)(rb * a.CS + cs) * 2) * CO_COLS + ch * 4;
    co_st_agent(p + 0, t0.x); co_st_agent(p + 1, t0.y); co_st_agent(p + 2, t0.z); co_st_agent(p + 3, t0.w);
    co_st_agent(p + CO_COLS + 0, t1.x); co_st_agent(p + CO_COLS + 1, t1.y); co_st_agent(p + CO_COLS + 2, t1.z); co_st_agent(p + CO_COLS + 3, t1.w);
  }
  co_grid_barrier(a.ctr, (uint32_t)(a.RB * a.CS));

  // ---- totals of this slab, in row-range order (identical in every block of the slab)
  if (tid < 2 * CO_COLS) {
    const int k = tid / CO_COLS, col = tid % CO_COLS;
    // (all <= 32 partials in flight, then the fixed-order sum: one round trip, not RB of them)
    float pv[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) pv[q] = co_ld_agent(a.part + ((int64_t)((q < a.RB ? q : a.RB - 1) * a.CS + cs) * 2 + k) * CO_COLS + col);
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) t += q < a.RB ? pv[q] : 0.f;
    stot[k][col] = t;
  }
  __syncthreads();
  if (tid < CO_COLS) {
    const int64_t c = (int64_t)cs * CO_COLS + tid;
    const float s1 = stot[0][tid], s2 = stot[1][tid];
    const float inv_n = 1.0f / (float)a.N;
    if constexpr (!BWD) {
      const float piv = c < D ? a.x[c] : 0.f;
      const float m1 = s1 * inv_n;
      float var = s2 * inv_n - m1 * m1;
      var = var < 0.f ? 0.f : var;
      const float m = piv + m1, r = 1.0f / sqrtf(var + a.eps);
      sst[0][tid] = m;
      sst[1][tid] = r;
      if (rb == 0 && c < D) {
        a.mean[c] = m;
        a.rstd[c] = r;
        if (a.rmean) {
          a.rmean[c] = (1.f - a.momentum) * a.rmean[c] + a.momentum * m;
          const float unbiased = a.N > 1 ? var * ((float)a.N / (float)(a.N - 1)) : var;
          a.rvar[c] = (1.f - a.momentum) * a.rvar[c] + a.momentum * unbiased;
        }
      }
    } else {
      sst[0][tid] = s1 * inv_n;
      sst[1][tid] = s2 * inv_n;
      if (rb == 0 && c < D) {
        a.dbias[c] = s1;
        a.dweight[c] = s2;
      }
    }
  }
  if (!BWD && blockIdx.x == 0 && tid == 0 && a.nbt) a.nbt[0] += 1;
  __syncthreads();
  if (!cact) return;
  const float4 t0 = *reinterpret_cast<const float4*>(&sst[0][ch * 4]), t1 = *reinterpret_cast<const float4*>(&sst[1][ch * 4]);
  float* O = a.out;

  if constexpr (!BWD) {
    mu = t0; rs = t1;
    for (int rd = 0; rd < nrounds; ++rd) {
      if (rd > 0) {
#pragma unroll
        for (int i = 0; i < CO_RPT; ++i) keep[i] = ldrow(X, rd * CO_RPT + i);
      }
      // the rows' graphs (all in flight), then per batch of 4 rows: residual rows + broadcast rows (LDS, or memory past the staged graphs)
      int32_t bi[CO_RPT];
      if (a.bcast) {
#pragma unroll
        for (int i = 0; i < CO_RPT; ++i) bi[i] = rowof(rd * CO_RPT + i) < r1 ? a.bidx[rowof(rd * CO_RPT + i)] : g0;
      }
#pragma unroll
      for (int i0 = 0; i0 < CO_RPT; i0 += 4) {
        float4 res[4], bc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          res[u] = a.resid ? ldrow(a.resid, rd * CO_RPT + i0 + u) : gt_zero4();
          bc[u] = gt_zero4();
          if (a.bcast) {
            const int rel = bi[i0 + u] - g0;
            if (rel >= 0 && rel < gspan) bc[u] = sbc[rel][ch];
            else bc[u] = co_ld(a.bcast + ((uint32_t)bi[i0 + u] * Du + ccu));
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = rd * CO_RPT + i0 + u, r = rowof(j);
          if (r >= r1) continue;
          float4 v = keep[i0 + u];
          v = make_float4((v.x - mu.x) * rs.x * ww.x + bb.x, (v.y - mu.y) * rs.y * ww.y + bb.y, (v.z - mu.z) * rs.z * ww.z + bb.z,
                          (v.w - mu.w) * rs.w * ww.w + bb.w);
          if (a.relu) v = gt_relu4(v);
          if (a.drop.thr) v = bn_drop4(v, a.drop, (uint32_t)r, (uint32_t)cc);
          v = gt_add4(gt_add4(v, res[u]), bc[u]);
          *reinterpret_cast<float4*>(O + (int64_t)j * rstep + voff) = v;
        }
      }
    }
  } else {
    for (int rd = 0; rd < nrounds; ++rd) {
      if (rd > 0) bwd_round(rd, false);   // dy' of the round again (dy re-read, gate recomputed)
#pragma unroll
      for (int i0 = 0; i0 < CO_RPT; i0 += 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ldrow(X, rd * CO_RPT + i0 + u);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = rd * CO_RPT + i0 + u;
          if (rowof(j) >= r1) continue;
          const float4 g = keep[i0 + u];
          float4 o;
          o.x = ww.x * rs.x * (g.x - t0.x - (v[u].x - mu.x) * rs.x * t1.x);
          o.y = ww.y * rs.y * (g.y - t0.y - (v[u].y - mu.y) * rs.y * t1.y);
          o.z = ww.z * rs.z * (g.z - t0.z - (v[u].z - mu.z) * rs.z * t1.z);
          o.w = ww.w * rs.w * (g.w - t0.w - (v[u].w - mu.w) * rs.w * t1.w);
          *reinterpret_cast<float4*>(O + (int64_t)j * rstep + voff) = o;
        }
      }
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
struct BnCoopSlots {   // per host thread: zeroed barrier counters handed over by the caller (gt_bn_coop_slots), 64 bytes apart
  uint32_t* base = nullptr;
  int n = 0, next = 0;
};
thread_local BnCoopSlots g_co_slots;
constexpr int CO_SLOT_BYTES = 64;

static int co_cu_count() {   // blocks that can be resident at once: one per CU
  static int cus[16] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16) return 0;
  if (!cus[dev]) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = -1;
    cus[dev] = v;
  }
  return cus[dev] > 0 ? cus[dev] : 0;
}
static int g_co_on = -1;   // -1: the environment decides (GT_BN_COOP, A/B knob); gt_bn_coop_set overrides (tests compare the two schemes)
static bool co_enabled() {
  // OFF unless asked for: in the training step it gained nothing (Molpcba 95.5 k -> 95.4 k graphs/s, Code2 at 32 graphs 15.98 k -> 15.49 k,
  // Code2 b256 with four rounds 73.7 k -> 67.4 k; profiles/r05_probes/bn_one_launch.txt) -- its 4 waves per CU move ~3 TB/s where the
  // three launches' full-occupancy passes move 4-5, and its blocks hold their CUs while they spin.  Kept as a measured, tested variant.
  static const bool env_on = [] { const char* e = getenv("GT_BN_COOP"); return e && atoi(e) != 0; }();
  return g_co_on < 0 ? env_on : g_co_on != 0;
}
// grid shape of the one-launch BatchNorm; false = the three-launch scheme takes the call
static bool co_plan(int dtype, int64_t rows, int64_t dim, int* RB, int* CS, int64_t* rows_per) {
  if (!co_enabled() || dtype != GT_F32 || dim % 4 || rows <= 1024 || rows * dim >= ((int64_t)1 << 30)) return false;
  const int cus = co_cu_count();
  const int cs = (int)gt_cdiv(dim, CO_COLS);
  if (cus < 64 || cs > cus) return false;
  int rb = cus / cs;
  if (rb > 32) rb = 32;   // (the totals loop and the workspace's partial area are sized for <= 32 row ranges)
  const int64_t rp = gt_cdiv(rows, rb);
  // the rows a block cannot keep in registers are read twice: beyond 4 rounds of the register rows the three-launch scheme is the better one
  // One round only (every row in the registers): measured alone on the chip (tools/bn_bench.py, profiles/r05_probes/bn_one_launch.txt) the
  // one-launch kernel moves ~3 TB/s with its 4 waves per CU where the three-launch scheme's full-occupancy passes move 4-5 TB/s --
  // at Code2's 31.6 k x 300 it loses (45 against 29 us forward, 61 against 36 backward), below ~12 k rows it ties or wins and saves
  // two launches per direction (Molpcba, the 32-graph shard of a strong-scaled Code2 step: host-bound steps)
  static const int max_rounds = [] { const char* e = getenv("GT_BN_COOP_ROUNDS"); const int v = e ? atoi(e) : 1; return v >= 1 && v <= 4 ? v : 1; }();
  if (rp > (int64_t)max_rounds * CO_RL * CO_RPT_MAX) return false;
  *RB = rb; *CS = cs; *rows_per = rp;
  return true;
}

template <bool BWD>
static void co_launch(const BnCoopArgs& a, hipStream_t stream) {
  const dim3 grid((unsigned)(a.RB * a.CS)), block(CO_NT);
  if (a.rows_per <= 8 * CO_RL) hipLaunchKernelGGL((k_bn_coop<BWD, 8>), grid, block, 0, stream, a);
  else if (!BWD || a.rows_per <= 16 * CO_RL) hipLaunchKernelGGL((k_bn_coop<BWD, 16>), grid, block, 0, stream, a);   // (forward: its side loads leave room for 16 rows)
  else hipLaunchKernelGGL((k_bn_coop<true, 20>), grid, block, 0, stream, a);
}
// partial area + the call's own counter slot
static size_t co_workspace_need(int RB, int CS) { return (size_t)RB * CS * 2 * CO_COLS * sizeof(float) + 2 * CO_SLOT_BYTES; }
// the counter of this call: the next zeroed slot of the caller's pool, else the tail of the workspace behind a 4-byte... 8-byte clear
static uint32_t* co_counter(void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (g_co_slots.base && g_co_slots.next < g_co_slots.n) {
    uint32_t* c = g_co_slots.base + (size_t)g_co_slots.next * (CO_SLOT_BYTES / 4);
    ++g_co_slots.next;
    return c;
  }
  uintptr_t p = ((uintptr_t)workspace + workspace_bytes - CO_SLOT_BYTES) & ~(uintptr_t)(CO_SLOT_BYTES - 1);
  (void)hipMemsetAsync((void*)p, 0, 8, stream);
  return (uint32_t*)p;
}
