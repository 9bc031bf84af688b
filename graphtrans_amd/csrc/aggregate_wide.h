// aggregate_wide.h — the fp32 aggregate kernels with W consecutive floats per lane (included by aggregate.hip, inside its namespace).
//
// The float4-chunk mapping of aggregate.hip covers a row of D = 300 (the reference's default width, both headline
// workloads) with 64 + 11 chunks: two chunks per lane of which the second is live on 11 lanes, i.e. 8 registers per lane
// and row where 5 would do, ~150 VGPRs, 3 waves per SIMD.  These kernels are a chain of dependent gathers per node
// (ptr -> edge indices -> neighbour rows) and run at the speed of the number of such chains in flight, so registers are
// what they cost.  Here lane l owns columns [l W, l W + W), W = D / 64 rounded up (5 for D = 300: 60 live lanes), loaded as
// one 16-byte + (W - 4) 4-byte accesses (dword aligned; the wave still reads the row contiguously).  Same walk, same
// software pipeline, same summation order per column as k_agg_fwd / k_agg_bwd.
#pragma once

#ifndef AGGW_U
#define AGGW_U 2   // edges gathered per trip
#endif

template <int W>
struct Row {
  float v[W];
};

struct __attribute__((packed, aligned(4))) F4Unaligned {
  float a, b, c, d;
};

template <int W>
__device__ __forceinline__ Row<W> row_load(const float* p) {
  Row<W> r;
  const F4Unaligned q = *reinterpret_cast<const F4Unaligned*>(p);
  r.v[0] = q.a; r.v[1] = q.b; r.v[2] = q.c; r.v[3] = q.d;
#pragma unroll
  for (int e = 4; e < W; ++e) r.v[e] = p[e];
  return r;
}

template <int W>
__device__ __forceinline__ void row_store(float* p, const Row<W>& r) {
  F4Unaligned q;
  q.a = r.v[0]; q.b = r.v[1]; q.c = r.v[2]; q.d = r.v[3];
  *reinterpret_cast<F4Unaligned*>(p) = q;
#pragma unroll
  for (int e = 4; e < W; ++e) p[e] = r.v[e];
}

template <int W>
__device__ __forceinline__ Row<W> row_zero() {
  Row<W> r;
#pragma unroll
  for (int e = 0; e < W; ++e) r.v[e] = 0.f;
  return r;
}

// per-lane edge-encoder state: Linear keeps its K weight columns and the bias in registers
template <int W, int EDGE>
struct EdgeStateW {
  float w[is_linear<EDGE>() ? kmax<EDGE>() : 1][W];
  float b[W];
};

template <int W, int EDGE>
__device__ __forceinline__ void edge_state_init_w(EdgeStateW<W, EDGE>& s, const AggArgs& a, int col, bool act) {
  if constexpr (is_linear<EDGE>()) {
#pragma unroll
    for (int e = 0; e < W; ++e) {
      s.b[e] = act ? a.b[col + e] : 0.f;
#pragma unroll
      for (int k = 0; k < kmax<EDGE>(); ++k) s.w[k][e] = (k < a.K && act) ? a.w[(int64_t)(col + e) * a.K + k] : 0.f;
    }
  }
}

template <int W, int EDGE>
__device__ __forceinline__ Row<W> edge_embed_w(const EdgeStateW<W, EDGE>& s, const AggArgs& a, int col, int eid, const float* av,
                                               const int* ti) {
  Row<W> r = row_zero<W>();
  if constexpr (is_linear<EDGE>()) {
#pragma unroll
    for (int e = 0; e < W; ++e) {
      float x = s.b[e];
#pragma unroll
      for (int k = 0; k < kmax<EDGE>(); ++k) x = fmaf(s.w[k][e], av[k], x);   // weight columns past K are zero
      r.v[e] = x;
    }
  } else if constexpr (EDGE == GT_EDGE_TABLES) {
#pragma unroll
    for (int k = 0; k < MAX_K; ++k)
      if (k < a.K) {
        const float* t = a.w + (int64_t)ti[k] * a.D + col;
#pragma unroll
        for (int e = 0; e < W; ++e) r.v[e] += t[e];
      }
  } else if constexpr (EDGE == GT_EDGE_DENSE) {
    r = row_load<W>(reinterpret_cast<const float*>(a.dense) + (int64_t)eid * a.D + col);
  }
  return r;
}

// The gather phases below are written branch-free (clamped indices instead of predicated loads, the conv kind a template
// parameter, attribute columns past K loaded from column 0 and multiplied by a zero weight): every `cond ? load : 0` becomes
// a basic block with its own s_waitcnt vmcnt(0), which is how a node's five rows turned into three dependent round trips.
template <int EDGE>
__device__ __forceinline__ void edge_attr_load_w(const AggArgs& a, int eid, float* av, int* ti) {
  if constexpr (is_linear<EDGE>()) {
    const float* p = reinterpret_cast<const float*>(a.attr) + (int64_t)eid * a.K;
#pragma unroll
    for (int k = 0; k < kmax<EDGE>(); ++k) av[k] = p[k < a.K ? k : 0];
  } else if constexpr (EDGE == GT_EDGE_TABLES) {
    const int64_t* p = reinterpret_cast<const int64_t*>(a.attr) + (int64_t)eid * a.K;
#pragma unroll
    for (int k = 0; k < MAX_K; ++k) ti[k] = a.tab_off[k < a.K ? k : 0] + (int)p[k < a.K ? k : 0];
  }
}

// ---- forward -------------------------------------------------------------------------------------------------------------
template <int W, int EDGE, bool GCN>
__global__ void __launch_bounds__(AGG_THREADS) k_aggw_fwd(AggArgs a) {
  constexpr int U = W <= 4 ? 2 * AGGW_U : AGGW_U;   // in-edges gathered per trip (4-float rows: twice as many for the same registers)
  if constexpr (EDGE == GT_EDGE_TABLES) {
    extern __shared__ __attribute__((aligned(16))) float tab_lds[];
    if (a.table_rows > 0) {
      for (int64_t i = threadIdx.x * 4; i < (int64_t)a.table_rows * a.D; i += AGG_THREADS * 4)
        *reinterpret_cast<float4*>(tab_lds + i) = *reinterpret_cast<const float4*>(a.w + i);
      __syncthreads();
      a.w = tab_lds;
    }
  }
  const int lane = threadIdx.x & 63;
  const int64_t blk = (int64_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8;   // XCD-aware, as k_agg_fwd
  const int64_t wave = blk * AGG_WAVES + (threadIdx.x >> 6);
  const bool act = (int64_t)lane * W < a.D;
  const int col = act ? lane * W : 0;
  const int chunk = a.chunk;
  const int64_t v_lo = wave * chunk;
  if (v_lo >= a.N) return;
  EdgeStateW<W, EDGE> es;
  edge_state_init_w<W, EDGE>(es, a, col, act);
  const float* h = reinterpret_cast<const float*>(a.h);
  float* out = reinterpret_cast<float*>(a.out);
  const float one_eps = GCN ? 0.f : 1.0f + a.self_param[0];
  Row<W> root = row_zero<W>();
  if constexpr (GCN) root = row_load<W>(a.self_param + col);
  const int64_t last = a.N - 1;

  auto load_ptr = [&](int64_t v, int& b, int& e) {   // [b, e) empty past the last node
    const int64_t vv = v < last ? v : last;
    b = a.ptr[vv];
    const int e1 = a.ptr[vv + 1];
    e = v <= last ? e1 : b;
  };
  auto load_idx = [&](int b, int e, int (&src)[U], int (&eid)[U]) {   // a.nbr / a.eid hold >= 1 entry (host)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int q = b + u < e ? b + u : e - 1;
      q = q > 0 ? q : 0;
      src[u] = a.nbr[q];
      eid[u] = a.eid[q];
    }
  };
  int beg0, end0, beg1, end1;
  int src0[U], eid0[U];
  load_ptr(v_lo, beg0, end0);
  load_ptr(v_lo + 1, beg1, end1);
  load_idx(beg0, end0, src0, eid0);

#pragma unroll 1
  for (int it = 0; it < chunk; ++it) {
    const int64_t v = v_lo + it;
    if (v >= a.N) break;
    int src1[U], eid1[U], beg2, end2;
    load_idx(beg1, end1, src1, eid1);
    load_ptr(it + 2 < chunk ? v + 2 : a.N, beg2, end2);
    float dv = 1.0f, degv = 1.0f;
    if constexpr (GCN) { dv = a.dis[v]; degv = a.deg[v]; }
    const Row<W> hv = row_load<W>(h + v * a.D + col);
    Row<W> acc = row_zero<W>();
    for (int p = beg0; p < end0; p += U) {
      int src[U], eid[U];
      if (p == beg0) {
#pragma unroll
        for (int u = 0; u < U; ++u) { src[u] = src0[u]; eid[u] = eid0[u]; }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int q = p + u < end0 ? p + u : end0 - 1;
          src[u] = a.nbr[q];
          eid[u] = a.eid[q];
        }
      }
      float wgt[U];
      Row<W> row[U];
      float av[U][MAX_K];
      int ti[U][MAX_K];
#pragma unroll
      for (int u = 0; u < U; ++u) {   // edges past the node's last repeat it (same cache lines) and are skipped below
        row[u] = row_load<W>(h + (int64_t)src[u] * a.D + col);
        if constexpr (GCN) wgt[u] = a.dis[src[u]]; else wgt[u] = 1.0f;
        edge_attr_load_w<EDGE>(a, eid[u], av[u], ti[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        // no `break` past the last edge: a conditional region is where the compiler sinks that edge's loads to, one round
        // trip per edge; the repeated edge is computed and selected away instead
        const bool valid = p + u < end0;
        const Row<W> e = edge_embed_w<W, EDGE>(es, a, col, eid[u], av[u], ti[u]);
#pragma unroll
        for (int c = 0; c < W; ++c) acc.v[c] = fmaf(valid ? fmaxf(row[u].v[c] + e.v[c], 0.f) : 0.f, wgt[u], acc.v[c]);
      }
    }
    if (act) {
      Row<W> r;
      const float inv_deg = GCN ? 1.0f / degv : 0.f;
#pragma unroll
      for (int c = 0; c < W; ++c) {
        // GCN: relu(x + root) * 1.0 / deg (conv.py:63-65); GIN: (1 + eps) x + sum
        r.v[c] = GCN ? acc.v[c] * dv + fmaxf(hv.v[c] + root.v[c], 0.f) * inv_deg : fmaf(hv.v[c], one_eps, acc.v[c]);
      }
      row_store<W>(out + v * a.D + col, r);
    }
    beg0 = beg1; end0 = end1; beg1 = beg2; end1 = end2;
#pragma unroll
    for (int u = 0; u < U; ++u) { src0[u] = src1[u]; eid0[u] = eid1[u]; }
  }
}

// ---- backward ------------------------------------------------------------------------------------------------------------
template <int W, int EDGE, bool GCN>
__global__ void __launch_bounds__(AGG_THREADS) k_aggw_bwd(AggArgs a) {
  constexpr int NREG = reg_slots<EDGE>();
  constexpr int UB = AGGW_U;   // (four per trip measured the same on the stress batch: 271 vs 268 us)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int64_t D = a.D;
  const bool act = (int64_t)lane * W < D;
  const int col = act ? lane * W : 0;
  EdgeStateW<W, EDGE> es;
  edge_state_init_w<W, EDGE>(es, a, col, act);
  const float* h = reinterpret_cast<const float*>(a.h);
  const float* g = reinterpret_cast<const float*>(a.g);
  float* dh = reinterpret_cast<float*>(a.out);
  const float one_eps = GCN ? 0.f : 1.0f + a.self_param[0];
  const int nslots = (is_linear<EDGE>() ? a.K + 2 : 1) + (EDGE == GT_EDGE_TABLES ? a.table_rows : 0);

  // slot order in `partial` as k_agg_bwd: 0 = self ; Linear: 1..K = weight columns, K+1 = bias ; Tables: 1.. = rows
  float racc[NREG][W];
#pragma unroll
  for (int s = 0; s < NREG; ++s)
#pragma unroll
    for (int e = 0; e < W; ++e) racc[s][e] = 0.f;

  float* wl = nullptr;  // per-wave LDS table-gradient rows [table_rows][D]
  if constexpr (EDGE == GT_EDGE_TABLES) {
    wl = lds + (int64_t)wid * a.table_rows * D;
    for (int64_t i = lane; i < (int64_t)a.table_rows * D; i += 64) wl[i] = 0.f;
    float* tab = lds + (int64_t)AGG_WAVES * a.table_rows * D;   // the tables for the relu gate, as k_agg_bwd
    for (int64_t i = threadIdx.x * 4; i < (int64_t)a.table_rows * D; i += AGG_THREADS * 4)
      *reinterpret_cast<float4*>(tab + i) = *reinterpret_cast<const float4*>(a.w + i);
    __syncthreads();
    a.w = tab;
  }
  Row<W> root = row_zero<W>();
  if constexpr (GCN) root = row_load<W>(a.self_param + col);
  const int64_t last = a.N - 1;

  const int64_t total_waves = (int64_t)gridDim.x * AGG_WAVES;
  const int64_t blk = gridDim.x % 8 == 0 ? (int64_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : (int64_t)blockIdx.x;
  const int64_t wave0 = blk * AGG_WAVES + wid;
  const int64_t cpw = (a.N + total_waves - 1) / total_waves;   // source nodes per wave-tile
  const int64_t u_lo = wave0 * cpw;
  auto load_ptr = [&](int64_t v, int& b, int& e) {   // [b, e) empty past the last node
    const int64_t vv = v < last ? v : last;
    b = a.ptr[vv];
    const int e1 = a.ptr[vv + 1];
    e = v <= last ? e1 : b;
  };
  auto load_idx = [&](int b, int e, int (&dst)[UB], int (&eid)[UB]) {   // a.nbr / a.eid hold >= 1 entry (host)
#pragma unroll
    for (int ub = 0; ub < UB; ++ub) {
      int q = b + ub < e ? b + ub : e - 1;
      q = q > 0 ? q : 0;
      dst[ub] = a.nbr[q];
      eid[ub] = a.eid[q];
    }
  };
  int beg0, end0, beg1, end1;
  int dst0[UB], eid0[UB];
  load_ptr(u_lo, beg0, end0);
  load_ptr(u_lo + 1, beg1, end1);
  load_idx(beg0, end0, dst0, eid0);
#pragma unroll 1
  for (int64_t it = 0; it < cpw; ++it) {
    const int64_t u = u_lo + it;
    if (u >= a.N) break;
    int dst1[UB], eid1[UB], beg2, end2;
    load_idx(beg1, end1, dst1, eid1);
    load_ptr(it + 2 < cpw ? u + 2 : a.N, beg2, end2);
    const int beg = beg0, end = end0;
    const Row<W> hu = row_load<W>(h + u * D + col);
    const Row<W> gu = row_load<W>(g + u * D + col);
    Row<W> acc = row_zero<W>();
    float du = 1.0f, degu = 1.0f;
    if constexpr (GCN) { du = a.dis[u]; degu = a.deg[u]; }
    for (int p0 = beg; p0 < end; p0 += UB) {
      int dsts[UB], eids[UB];
      if (p0 == beg) {
#pragma unroll
        for (int ub = 0; ub < UB; ++ub) { dsts[ub] = dst0[ub]; eids[ub] = eid0[ub]; }
      } else {
#pragma unroll
        for (int ub = 0; ub < UB; ++ub) {
          const int qq = p0 + ub < end ? p0 + ub : end - 1;
          dsts[ub] = a.nbr[qq];
          eids[ub] = a.eid[qq];
        }
      }
      Row<W> gds[UB];
      float avs[UB][MAX_K], wks[UB];
      int tis[UB][MAX_K];
#pragma unroll
      for (int ub = 0; ub < UB; ++ub) {   // edges past the node's last repeat it (same cache lines) and are skipped below
        gds[ub] = row_load<W>(g + (int64_t)dsts[ub] * D + col);
        if constexpr (GCN) wks[ub] = a.dis[dsts[ub]]; else wks[ub] = 1.0f;
        edge_attr_load_w<EDGE>(a, eids[ub], avs[ub], tis[ub]);
      }
#pragma unroll
      for (int ub = 0; ub < UB; ++ub) {
        const bool valid = p0 + ub < end;   // no `break`: see the forward
        const float* av = avs[ub];
        const int* ti = tis[ub];
        const float wk = wks[ub] * du;
        const Row<W> e = edge_embed_w<W, EDGE>(es, a, col, eids[ub], av, ti);
        Row<W> t;
#pragma unroll
        for (int c = 0; c < W; ++c) {
          t.v[c] = (valid && hu.v[c] + e.v[c] > 0.f) ? gds[ub].v[c] * wk : 0.f;
          acc.v[c] += t.v[c];
        }
        if constexpr (is_linear<EDGE>()) {
#pragma unroll
          for (int c = 0; c < W; ++c) {
#pragma unroll
            for (int k = 0; k < kmax<EDGE>(); ++k) racc[1 + k][c] = fmaf(t.v[c], av[k], racc[1 + k][c]);   // slots past K are dropped
            racc[kmax<EDGE>() + 1][c] += t.v[c];
          }
        } else if constexpr (EDGE == GT_EDGE_TABLES) {
          if (act) {
#pragma unroll
            for (int k = 0; k < MAX_K; ++k)
              if (k < a.K) {
                float* r = wl + (int64_t)ti[k] * D + col;   // lanes own distinct columns, the wave owns the rows
#pragma unroll
                for (int c = 0; c < W; ++c) r[c] += t.v[c];
              }
          }
        } else if constexpr (EDGE == GT_EDGE_DENSE) {
          if (act && valid) row_store<W>(reinterpret_cast<float*>(a.d_dense) + (int64_t)eids[ub] * D + col, t);
        }
      }
    }
    Row<W> r;
#pragma unroll
    for (int c = 0; c < W; ++c) {
      if constexpr (GCN) {
        const float s = hu.v[c] + root.v[c] > 0.f ? gu.v[c] * (1.0f / degu) : 0.f;
        racc[0][c] += s;
        r.v[c] = acc.v[c] + s;
      } else {
        racc[0][c] = fmaf(gu.v[c], hu.v[c], racc[0][c]);
        r.v[c] = fmaf(gu.v[c], one_eps, acc.v[c]);
      }
    }
    if (act) row_store<W>(dh + u * D + col, r);
    beg0 = beg1; end0 = end1; beg1 = beg2; end1 = end2;
#pragma unroll
    for (int ub = 0; ub < UB; ++ub) { dst0[ub] = dst1[ub]; eid0[ub] = eid1[ub]; }
  }

  // ---- block reduction of the register accumulators: waves (LDS) -> partial
  if constexpr (EDGE == GT_EDGE_TABLES) __syncthreads();  // every wave is done with the staged tables (same LDS as `stage`)
  float* stage = lds + (EDGE == GT_EDGE_TABLES ? (int64_t)AGG_WAVES * a.table_rows * D : 0);  // [AGG_WAVES][D]
  float* part = a.partial + (int64_t)blockIdx.x * nslots * D;
#pragma unroll
  for (int s = 0; s < NREG; ++s) {
    int slot = s;
    if constexpr (is_linear<EDGE>()) {
      if (s >= 1 && s <= kmax<EDGE>()) {
        if (s - 1 >= a.K) continue;
      } else if (s == kmax<EDGE>() + 1) {
        slot = a.K + 1;
      }
    }
    if (act) {
#pragma unroll
      for (int c = 0; c < W; ++c) stage[(int64_t)wid * D + col + c] = racc[s][c];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += AGG_THREADS) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < AGG_WAVES; ++w) t += stage[(int64_t)w * D + c];
      part[(int64_t)slot * D + c] = t;
    }
    __syncthreads();
  }
  if constexpr (EDGE == GT_EDGE_TABLES) {
    __syncthreads();
    for (int64_t i = threadIdx.x; i < (int64_t)a.table_rows * D; i += AGG_THREADS) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < AGG_WAVES; ++w) t += lds[(int64_t)w * a.table_rows * D + i];
      part[D + i] = t;  // slots 1.. = table rows
    }
  }
}
