// aggregate.hip — fused GCN / GIN message + scatter-aggregate + self term, forward and backward.
//
// Reference math (paths under /root/reference):
//   GCNConv  modules/conv.py:50-71   out[v] = sum_{k: col_k=v} dis[row_k] dis[col_k] relu(h[row_k] + e_k)
//                                              + relu(h[v] + root_emb) / deg[v]
//   GINConv  modules/conv.py:26-36   out[v] = (1 + eps) h[v] + sum_{k: col_k=v} relu(h[row_k] + e_k)
// The reference materialises x_j, e, the sum, the relu and the scaled message as five E x D
// tensors and scatter-adds them with atomics (torch-scatter); here one wave-tile per destination
// node walks the destination-sorted CSR, gathers neighbour rows with 16-byte lane loads, applies
// edge embedding / relu / norm in registers and writes each output row once: no atomics, fixed
// summation order (deterministic), E x D never touches memory.
//
// Lane mapping: a node row of D (<= 1024, D % 4 == 0) values is covered by LPN lanes x NCH float4
// chunks.  D <= 64 -> 16 lanes/node (4 nodes per wave), D <= 128 -> 32 lanes/node, else the whole
// wave with NCH = ceil(D/256) chunks per lane (D = 300 -> 75 chunks: 64 + 11).
//
// Backward walks the source-sorted CSC (one wave-tile per SOURCE node u, persistent grid):
//   t_k = w_k 1[h[u]+e_k>0] g[col_k];  dh[u] = sum_k t_k + self';  edge-parameter gradients are
//   accumulated in registers (Linear) or per-wave LDS rows (embedding tables), reduced per block
//   and finished by a second kernel in a fixed order.
#include <mutex>
#include "gt_common.h"
#include <type_traits>

namespace {

constexpr int AGG_THREADS = 256;
constexpr int AGG_WAVES = AGG_THREADS / GT_WAVE;
constexpr int BWD_BLOCKS = 2048;  // upper bound (sizes the partial workspace); the launch takes one round of resident blocks (bwd_grid)
constexpr int MAX_K = 4;
// internal edge mode: Linear edge encoder with K <= 2 (the Code2 case, dataset/code.py:117): half the
// weight registers / accumulators of the generic K <= 4 variant -> higher occupancy
constexpr int EDGE_LINEAR2 = 4;
template <int EDGE>
__host__ __device__ constexpr bool is_linear() { return EDGE == GT_EDGE_LINEAR || EDGE == EDGE_LINEAR2; }
template <int EDGE>
__host__ __device__ constexpr int kmax() { return EDGE == EDGE_LINEAR2 ? 2 : MAX_K; }

struct AggArgs {
  int conv, K;
  int64_t N, E, D;
  const void* h;
  const void* g;  // bwd: grad_out
  const int32_t* ptr;
  const int32_t* nbr;  // fwd: in_src ; bwd: out_dst
  const int32_t* eid;
  const float* deg;
  const float* dis;
  const float* self_param;
  const void* attr;
  const float* w;  // Linear weight [D][K] or tables [rows][D]
  const float* b;  // Linear bias [D]
  int tab_off[MAX_K];
  int table_rows;
  int chunk;      // fwd: consecutive nodes per wave-tile (<= FWD_CHUNK)
  const void* dense;
  void* out;      // fwd: out ; bwd: grad_h
  void* d_dense;  // bwd
  float* partial; // bwd: [blocks][slots][D]
};

template <int LPN, int NCH>
struct LaneMap {
  static constexpr int NPW = 64 / LPN;
  int sub, sl;
  int col[NCH];
  bool act[NCH];
  __device__ __forceinline__ LaneMap(int lane, int64_t D) {
    sub = lane / LPN;
    sl = lane % LPN;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      col[j] = (sl + j * 64) * 4;
      act[j] = col[j] < D;
      if (!act[j]) col[j] = 0;  // safe address; results discarded
    }
  }
};

// per-lane edge-embedding state
template <int EDGE, int NCH>
struct EdgeState {
  float4 w[is_linear<EDGE>() ? kmax<EDGE>() : 1][NCH];
  float4 b[NCH];
};

template <int EDGE, int NCH, int LPN>
__device__ __forceinline__ void edge_state_init(EdgeState<EDGE, NCH>& s, const AggArgs& a, const LaneMap<LPN, NCH>& m) {
  if constexpr (is_linear<EDGE>()) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      s.b[j] = m.act[j] ? *reinterpret_cast<const float4*>(a.b + m.col[j]) : gt_zero4();
#pragma unroll
      for (int k = 0; k < kmax<EDGE>(); ++k) {
        if (k < a.K && m.act[j]) {
          const float* w = a.w + (int64_t)m.col[j] * a.K + k;  // W[d][k], d = col..col+3
          s.w[k][j] = make_float4(w[0], w[a.K], w[2 * a.K], w[3 * a.K]);
        } else {
          s.w[k][j] = gt_zero4();
        }
      }
    }
  }
}

// e_k chunk j for original edge id `eid`
template <typename T, int EDGE, int NCH, int LPN>
__device__ __forceinline__ float4 edge_embed(const EdgeState<EDGE, NCH>& s, const AggArgs& a,
                                             const LaneMap<LPN, NCH>& m, int j, int eid, const float* av,
                                             const int* ti) {
  if constexpr (EDGE == GT_EDGE_NONE) {
    return gt_zero4();
  } else if constexpr (is_linear<EDGE>()) {
    float4 e = s.b[j];
#pragma unroll
    for (int k = 0; k < kmax<EDGE>(); ++k)
      if (k < a.K) e = gt_fma4(s.w[k][j], av[k], e);
    return e;
  } else if constexpr (EDGE == GT_EDGE_TABLES) {
    float4 e = gt_zero4();
#pragma unroll
    for (int k = 0; k < MAX_K; ++k)
      if (k < a.K) e = gt_add4(e, *reinterpret_cast<const float4*>(a.w + (int64_t)ti[k] * a.D + m.col[j]));
    return e;
  } else {
    return gt_load4<T>(reinterpret_cast<const T*>(a.dense) + (int64_t)eid * a.D + m.col[j]);
  }
}

template <int EDGE>
__device__ __forceinline__ void edge_attr_load(const AggArgs& a, int eid, float* av, int* ti) {
  if constexpr (is_linear<EDGE>()) {
    const float* p = reinterpret_cast<const float*>(a.attr) + (int64_t)eid * a.K;
#pragma unroll
    for (int k = 0; k < kmax<EDGE>(); ++k) av[k] = k < a.K ? p[k] : 0.f;
  } else if constexpr (EDGE == GT_EDGE_TABLES) {
    const int64_t* p = reinterpret_cast<const int64_t*>(a.attr) + (int64_t)eid * a.K;
#pragma unroll
    for (int k = 0; k < MAX_K; ++k) ti[k] = k < a.K ? a.tab_off[k] + (int)p[k] : 0;
  }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// One wave-tile walks FWD_CHUNK consecutive destination nodes.  A node is three dependent memory round
// trips (ptr -> in-edge indices -> neighbour rows); with ~3 in-edges there is nothing inside one node
// to hide them behind, so the walk is software pipelined across nodes: while node i's rows are gathered,
// the indices of node i+1 and the ptr pair of node i+2 are already in flight.
constexpr int FWD_CHUNK = 8;   // at most; small batches use shorter walks so that the chip still fills (a.chunk)

template <typename T, int LPN, int NCH, int EDGE>
__global__ void __launch_bounds__(AGG_THREADS) k_agg_fwd(AggArgs a) {
  constexpr int NPW = 64 / LPN;
  constexpr int U = NCH >= 3 ? 2 : 4;
  if constexpr (EDGE == GT_EDGE_TABLES) {
    // Bond-style embedding tables are a dozen rows: every edge sums K of them, so the block reads them
    // from LDS instead of issuing K dependent global row loads per edge (a.table_rows > 0 = staged)
    extern __shared__ __attribute__((aligned(16))) float tab_lds[];
    if (a.table_rows > 0) {
      for (int64_t i = threadIdx.x * 4; i < (int64_t)a.table_rows * a.D; i += AGG_THREADS * 4)
        *reinterpret_cast<float4*>(tab_lds + i) = *reinterpret_cast<const float4*>(a.w + i);
      __syncthreads();
      a.w = tab_lds;
    }
  }
  const int lane = threadIdx.x & 63;
  // XCD-aware block order (grid is a multiple of 8): workgroup b runs on XCD b % 8; XCD x walks the x-th
  // contiguous eighth of the nodes, so the re-gathers of a row (self + ~3 neighbours of the same graph)
  // hit one L2 instead of crossing the fabric from several XCDs.
  const int64_t blk = (int64_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8;
  const int64_t wave = blk * AGG_WAVES + (threadIdx.x >> 6);
  LaneMap<LPN, NCH> m(lane, a.D);
  const int chunk = a.chunk;
  const int64_t v_lo = wave * (chunk * NPW) + m.sub;   // this sub-group's nodes: v_lo, v_lo + NPW, ...
  if (v_lo >= a.N) return;
  EdgeState<EDGE, NCH> es;
  edge_state_init<EDGE, NCH, LPN>(es, a, m);
  const T* h = reinterpret_cast<const T*>(a.h);
  T* out = reinterpret_cast<T*>(a.out);
  const bool gcn = a.conv == GT_CONV_GCN;
  const float one_eps = gcn ? 0.f : 1.0f + a.self_param[0];
  float4 root[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j)
    root[j] = (gcn && m.act[j]) ? *reinterpret_cast<const float4*>(a.self_param + m.col[j]) : gt_zero4();

  auto load_ptr = [&](int64_t v, int& b, int& e) {
    const bool ok = v < a.N;
    b = ok ? a.ptr[v] : 0;
    e = ok ? a.ptr[v + 1] : 0;
  };
  auto load_idx = [&](int b, int e, int (&src)[U], int (&eid)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = b + u < e ? b + u : (e > b ? e - 1 : -1);
      src[u] = q >= 0 ? a.nbr[q] : 0;
      eid[u] = q >= 0 ? a.eid[q] : 0;
    }
  };

  int beg0, end0, beg1, end1;
  int src0[U], eid0[U];
  load_ptr(v_lo, beg0, end0);
  load_ptr(v_lo + NPW, beg1, end1);
  load_idx(beg0, end0, src0, eid0);

#pragma unroll 1
  for (int it = 0; it < chunk; ++it) {
    const int64_t v = v_lo + (int64_t)it * NPW;
    if (v >= a.N) break;
    // ---- prefetch: indices of the next node, ptr pair of the one after
    int src1[U], eid1[U], beg2, end2;
    load_idx(beg1, end1, src1, eid1);
    load_ptr(it + 2 < chunk ? v + 2 * NPW : a.N, beg2, end2);
    // ---- current node
    const float dv = gcn ? a.dis[v] : 1.0f;
    const float degv = gcn ? a.deg[v] : 1.0f;
    float4 hv[NCH], acc[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      hv[j] = m.act[j] ? gt_load4<T>(h + v * a.D + m.col[j]) : gt_zero4();
      acc[j] = gt_zero4();
    }
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
      float4 row[U][NCH];
      float av[U][MAX_K];
      int ti[U][MAX_K];
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) row[u][j] = gt_load4<T>(h + (int64_t)src[u] * a.D + m.col[j]);
        wgt[u] = gcn ? a.dis[src[u]] : 1.0f;
        if (p + u >= end0) wgt[u] = 0.f;
        edge_attr_load<EDGE>(a, eid[u], av[u], ti[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
          float4 e = edge_embed<T, EDGE, NCH, LPN>(es, a, m, j, eid[u], av[u], ti[u]);
          acc[j] = gt_fma4(gt_relu4(gt_add4(row[u][j], e)), wgt[u], acc[j]);
        }
      }
    }
    const float inv_deg = gcn ? 1.0f / degv : 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      if (!m.act[j]) continue;
      float4 r;
      if (gcn) {
        // relu(x + root) * 1.0 / deg   (conv.py:63-65)
        float4 s = gt_relu4(gt_add4(hv[j], root[j]));
        r = make_float4(acc[j].x * dv + s.x * inv_deg, acc[j].y * dv + s.y * inv_deg, acc[j].z * dv + s.z * inv_deg,
                        acc[j].w * dv + s.w * inv_deg);
      } else {
        r = gt_fma4(hv[j], one_eps, acc[j]);
      }
      gt_store4<T>(out + v * a.D + m.col[j], r);
    }
    // ---- rotate the pipeline registers
    beg0 = beg1; end0 = end1; beg1 = beg2; end1 = end2;
#pragma unroll
    for (int u = 0; u < U; ++u) { src0[u] = src1[u]; eid0[u] = eid1[u]; }
  }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 gate4(float4 pre, float4 v) {
  return make_float4(pre.x > 0.f ? v.x : 0.f, pre.y > 0.f ? v.y : 0.f, pre.z > 0.f ? v.z : 0.f, pre.w > 0.f ? v.w : 0.f);
}

template <int EDGE>
__host__ __device__ constexpr int reg_slots() {
  // register-accumulated D-vectors per lane: [self] (+ K weight columns + bias for Linear)
  return is_linear<EDGE>() ? kmax<EDGE>() + 2 : 1;
}

template <typename T, int LPN, int NCH, int EDGE>
__global__ void __launch_bounds__(AGG_THREADS) k_agg_bwd(AggArgs a) {
  constexpr int NPW = 64 / LPN;
  constexpr int NREG = reg_slots<EDGE>();
  constexpr int UB = NCH >= 2 ? 2 : 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  LaneMap<LPN, NCH> m(lane, a.D);
  EdgeState<EDGE, NCH> es;
  edge_state_init<EDGE, NCH, LPN>(es, a, m);
  const T* h = reinterpret_cast<const T*>(a.h);
  const T* g = reinterpret_cast<const T*>(a.g);
  T* dh = reinterpret_cast<T*>(a.out);
  const bool gcn = a.conv == GT_CONV_GCN;
  const float one_eps = gcn ? 0.f : 1.0f + a.self_param[0];
  const int64_t D = a.D;
  const int nslots = (is_linear<EDGE>() ? a.K + 2 : 1) + (EDGE == GT_EDGE_TABLES ? a.table_rows : 0);

  // slot order in `partial`: 0 = self ; Linear: 1..K = weight columns, K+1 = bias ; Tables: 1.. = rows
  float4 racc[NREG][NCH];
#pragma unroll
  for (int s = 0; s < NREG; ++s)
#pragma unroll
    for (int j = 0; j < NCH; ++j) racc[s][j] = gt_zero4();

  float* wl = nullptr;  // per-wave LDS table-gradient rows [table_rows][D]
  if constexpr (EDGE == GT_EDGE_TABLES) {
    wl = lds + (int64_t)wid * a.table_rows * D;
    for (int64_t i = lane; i < (int64_t)a.table_rows * D; i += 64) wl[i] = 0.f;
    // the tables themselves (needed for the relu gate h[u] + e_k > 0) are parked in LDS as in the forward: read
    // from global they are a third dependent round trip per edge (eid -> attr -> table row), ~1.4 us per edge.
    // They share the space of the reduction stage, which is only used after the walk.
    float* tab = lds + (int64_t)AGG_WAVES * a.table_rows * D;
    for (int64_t i = threadIdx.x * 4; i < (int64_t)a.table_rows * D; i += AGG_THREADS * 4)
      *reinterpret_cast<float4*>(tab + i) = *reinterpret_cast<const float4*>(a.w + i);
    __syncthreads();
    a.w = tab;
  }

  // persistent grid: every wave-tile walks its own contiguous run of source nodes, software pipelined
  // like the forward (indices of node i+1 and the ptr pair of node i+2 in flight while node i's
  // gradient rows are gathered).  The runs depend on (N, grid) only: partial sums stay reproducible.
  const int64_t total_waves = (int64_t)gridDim.x * AGG_WAVES;
  const int64_t blk = gridDim.x % 8 == 0 ? (int64_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : (int64_t)blockIdx.x;
  const int64_t wave0 = blk * AGG_WAVES + wid;   // XCD-aware like the forward: contiguous node runs per XCD
  const int64_t cpw = (a.N + total_waves * NPW - 1) / (total_waves * NPW);   // nodes per sub-group
  const int64_t u_lo = wave0 * cpw * NPW + m.sub;
  auto load_ptr = [&](int64_t v, int& b, int& e) {
    const bool ok = v < a.N;
    b = ok ? a.ptr[v] : 0;
    e = ok ? a.ptr[v + 1] : 0;
  };
  auto load_idx = [&](int b, int e, int (&dst)[UB], int (&eid)[UB]) {
#pragma unroll
    for (int ub = 0; ub < UB; ++ub) {
      const int q = b + ub < e ? b + ub : (e > b ? e - 1 : -1);
      dst[ub] = q >= 0 ? a.nbr[q] : 0;
      eid[ub] = q >= 0 ? a.eid[q] : 0;
    }
  };
  int beg0, end0, beg1, end1;
  int dst0[UB], eid0[UB];
  load_ptr(u_lo, beg0, end0);
  load_ptr(u_lo + NPW, beg1, end1);
  load_idx(beg0, end0, dst0, eid0);
#pragma unroll 1
  for (int64_t it = 0; it < cpw; ++it) {
    const int64_t u = u_lo + it * NPW;
    if (u >= a.N) break;
    int dst1[UB], eid1[UB], beg2, end2;
    load_idx(beg1, end1, dst1, eid1);
    load_ptr(it + 2 < cpw ? u + 2 * NPW : a.N, beg2, end2);
    const int beg = beg0, end = end0;
    float4 hu[NCH], gu[NCH], acc[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      hu[j] = gt_load4<T>(h + u * D + m.col[j]);
      gu[j] = gt_load4<T>(g + u * D + m.col[j]);
      acc[j] = gt_zero4();
    }
    const float du = gcn ? a.dis[u] : 1.0f;
    const float degu = gcn ? a.deg[u] : 1.0f;
    // out-edges UB at a time: the indices of UB edges, then every gradient row / attribute of the group,
    // then the arithmetic in edge order (one edge per trip is a chain of 2 dependent round trips per edge)
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
      float4 gds[UB][NCH];
      float avs[UB][MAX_K], wks[UB];
      int tis[UB][MAX_K];
#pragma unroll
      for (int ub = 0; ub < UB; ++ub) {
        edge_attr_load<EDGE>(a, eids[ub], avs[ub], tis[ub]);
        wks[ub] = gcn ? du * a.dis[dsts[ub]] : 1.0f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) gds[ub][j] = gt_load4<T>(g + (int64_t)dsts[ub] * D + m.col[j]);
      }
#pragma unroll
      for (int ub = 0; ub < UB; ++ub) {
      if (p0 + ub >= end) break;
      const int eid = eids[ub];
      const float* av = avs[ub];
      const int* ti = tis[ub];
      const float wk = wks[ub];
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const float4 gd = gds[ub][j];
        float4 e = edge_embed<T, EDGE, NCH, LPN>(es, a, m, j, eid, av, ti);
        float4 t = gate4(gt_add4(hu[j], e), gt_scale4(gd, wk));
        acc[j] = gt_add4(acc[j], t);
        if constexpr (is_linear<EDGE>()) {
#pragma unroll
          for (int k = 0; k < kmax<EDGE>(); ++k)
            if (k < a.K) racc[1 + k][j] = gt_fma4(t, av[k], racc[1 + k][j]);
          racc[kmax<EDGE>() + 1][j] = gt_add4(racc[kmax<EDGE>() + 1][j], t);
        } else if constexpr (EDGE == GT_EDGE_TABLES) {
          if (m.act[j]) {
#pragma unroll
            for (int k = 0; k < MAX_K; ++k)
              if (k < a.K) {
                float4* r = reinterpret_cast<float4*>(wl + (int64_t)ti[k] * D + m.col[j]);
                // lanes of one sub-group own distinct columns; sub-groups of a wave may hit the same
                // row -> serialise the NPW sub-groups to keep the sum order fixed
                if constexpr (NPW == 1) {
                  *r = gt_add4(*r, t);
                } else {
                  // the compiler reasons per thread and would merge these predicated updates into
                  // one wave-wide read-modify-write (a cross-lane race on a shared row): fence each.
#pragma unroll
                  for (int sg = 0; sg < NPW; ++sg) {
                    if (m.sub == sg) *r = gt_add4(*r, t);
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                  }
                }
              }
          }
        } else if constexpr (EDGE == GT_EDGE_DENSE) {
          if (m.act[j]) gt_store4<T>(reinterpret_cast<T*>(a.d_dense) + (int64_t)eid * D + m.col[j], t);
        }
      }
      }
    }
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      float4 r;
      if (gcn) {
        float4 root = *reinterpret_cast<const float4*>(a.self_param + m.col[j]);
        float4 s = gate4(gt_add4(hu[j], root), gt_scale4(gu[j], 1.0f / degu));
        racc[0][j] = gt_add4(racc[0][j], s);
        r = gt_add4(acc[j], s);
      } else {
        racc[0][j] = make_float4(fmaf(gu[j].x, hu[j].x, racc[0][j].x), fmaf(gu[j].y, hu[j].y, racc[0][j].y),
                                 fmaf(gu[j].z, hu[j].z, racc[0][j].z), fmaf(gu[j].w, hu[j].w, racc[0][j].w));
        r = gt_fma4(gu[j], one_eps, acc[j]);
      }
      if (m.act[j]) gt_store4<T>(dh + u * D + m.col[j], r);
    }
    beg0 = beg1; end0 = end1; beg1 = beg2; end1 = end2;
#pragma unroll
    for (int ub = 0; ub < UB; ++ub) { dst0[ub] = dst1[ub]; eid0[ub] = eid1[ub]; }
  }

  // ---- block reduction of the register accumulators: sub-groups (shuffle) -> waves (LDS) -> partial
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
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      float4 x = racc[s][j];
#pragma unroll
      for (int o = LPN; o < 64; o <<= 1) {
        x.x += __shfl_xor(x.x, o, 64);
        x.y += __shfl_xor(x.y, o, 64);
        x.z += __shfl_xor(x.z, o, 64);
        x.w += __shfl_xor(x.w, o, 64);
      }
      if (m.sub == 0 && m.act[j]) *reinterpret_cast<float4*>(stage + (int64_t)wid * D + m.col[j]) = x;
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

// final fixed-order reduction over the per-block partials.
// grid (ceil(D/64), nslots); out pointers by slot; GIN self slot is additionally summed over columns.
struct ReduceArgs {
  const float* partial;
  int nblocks, nslots;
  int64_t D;
  int conv, edge, K, table_rows;
  float* d_self;
  float* d_w;  // Linear [D][K] or tables [rows][D]
  float* d_b;
};

constexpr int RED_WAVES = 8;    // the backward leaves ~1000 partial rows (one round of resident blocks): 8 waves x 8 loads in flight
                                // keep the chain of dependent round trips at 16 (4 waves: 31, 43 us for a 20-block kernel; 1024-thread blocks wait for a CU with 16 free slots)
__global__ void __launch_bounds__(RED_WAVES * 64) k_agg_reduce(ReduceArgs r) {
  __shared__ float sm[RED_WAVES][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int slot = blockIdx.y;
  const int64_t c = (int64_t)blockIdx.x * 64 + lane;
  float t = 0.f;
  if (c < r.D) {
    // fixed order, 8 independent loads in flight per thread (a serial chain of L2 round trips made
    // this kernel cost as much as the main pass: profiles/r01a)
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* p = r.partial + (int64_t)slot * r.D + c;
    const int64_t stride = (int64_t)r.nslots * r.D;
    int b = wid;
    // (32 loads in flight first: the ~1000 partial rows of a Code2 backward are 126 per thread -- 4 round trips instead of 16;
    // 40 -> ~12 us for the 20-block launch, which runs beside the dX GEMM on the overlap stream and took CU time from it)
    for (; b + 31 * RED_WAVES < r.nblocks; b += 32 * RED_WAVES) {
      float v[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) v[u] = p[(int64_t)(b + RED_WAVES * u) * stride];
#pragma unroll
      for (int u = 0; u < 32; ++u) acc[u & 7] += v[u];
    }
    for (; b + 7 * RED_WAVES < r.nblocks; b += 8 * RED_WAVES) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += p[(int64_t)(b + RED_WAVES * u) * stride];
    }
    for (; b < r.nblocks; b += RED_WAVES) acc[0] += p[(int64_t)b * stride];
    t = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  }
  sm[wid][lane] = t;
  __syncthreads();
  if (wid != 0) return;
  t = 0.f;
#pragma unroll
  for (int w = 0; w < RED_WAVES; ++w) t += sm[w][lane];
  if (slot == 0) {
    if (r.conv == GT_CONV_GCN) {
      if (c < r.D && r.d_self) r.d_self[c] = t;
    } else {
      // GIN: d_eps = sum over all columns; one block column-tile at a time -> write tile sums to
      // d_self[1 + blockIdx.x]; the host-side wrapper launches k_eps_finish.
      if (c >= r.D) t = 0.f;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
      if (lane == 0 && r.d_self) r.d_self[1 + blockIdx.x] = t;
    }
    return;
  }
  if (c >= r.D) return;
  if (r.edge == GT_EDGE_LINEAR) {
    if (slot <= r.K) {
      if (r.d_w) r.d_w[c * r.K + (slot - 1)] = t;
    } else if (r.d_b) {
      r.d_b[c] = t;
    }
  } else if (r.edge == GT_EDGE_TABLES) {
    if (r.d_w) r.d_w[(int64_t)(slot - 1) * r.D + c] = t;
  }
}

__global__ void k_eps_finish(float* d_self, int ntiles) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < ntiles; ++i) t += d_self[1 + i];
    d_self[0] = t;
  }
}

#include "aggregate_wide.h"

// ------------------------------------------------------------------------------------------------
// dispatch
// ------------------------------------------------------------------------------------------------
// fp32 rows of 193..448 columns that split evenly over the lanes take the W-floats-per-lane kernels (aggregate_wide.h): W = 5..7
// for the widths the float4-chunk mapping wastes registers on (300, 384, ...), W = 4 (D = 256, the stress configuration) for
// their branch-free gather phases
template <typename T>
__host__ inline int wide_w(int64_t D) {
  if constexpr (!std::is_same<T, float>::value) return 0;
#ifdef AGGW_DISABLE
  return 0;
#endif
  if (D <= 192 || D > 448) return 0;
  const int w = (int)gt_cdiv(D, 64);
  return D % w == 0 ? w : 0;
}

// Persistent backward grid.  Every wave-tile walks N / tiles source nodes one after the other, so the launch takes as long as one
// walk times the number of ROUNDS of resident blocks: the grid is the largest that is resident at once (registers / LDS of this
// instantiation, asked of the runtime once), i.e. the shortest walks that still finish in one round -- 1.08 rounds cost two
// (Molpcba: 555 blocks 51 us, 416 blocks 38 us).  Depends on (kernel, N) only: the partial sums stay reproducible.
inline int bwd_grid(void (*kernel)(AggArgs), size_t lds_bytes, int64_t N, int npw) {
  // one-time queries, cached per (device, kernel, LDS bytes) under a mutex: the entry points are called from the caller's
  // thread AND from autograd's worker thread (C-ABI contract: no unguarded mutable globals), and the occupancy of a
  // kernel -- hence the partial-sum order -- belongs to the device that runs it, not to whichever device asked first
  struct Seen { int dev; const void* k; size_t lds; int per_cu; };
  constexpr int MAX_DEV = 16, MAX_SEEN = 256;
  static std::mutex mu;
  static Seen seen[MAX_SEEN];
  static int nseen = 0;
  static int cus_of[MAX_DEV] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  int cus = 0, per_cu = 0;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (dev >= 0 && dev < MAX_DEV) cus = cus_of[dev];
    if (!cus) {
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
      if (dev >= 0 && dev < MAX_DEV) cus_of[dev] = cus;
    }
    for (int i = 0; i < nseen; ++i)
      if (seen[i].dev == dev && seen[i].k == (const void*)kernel && seen[i].lds == lds_bytes) per_cu = seen[i].per_cu;
    if (!per_cu) {
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kernel, AGG_THREADS, lds_bytes) != hipSuccess || per_cu < 1) per_cu = 1;
      if (nseen < MAX_SEEN) seen[nseen++] = Seen{dev, (const void*)kernel, lds_bytes, per_cu};
    }
  }
  int64_t cap = (int64_t)cus * per_cu;
  if (cap > BWD_BLOCKS) cap = BWD_BLOCKS;
  int64_t nodes = gt_cdiv(N > 0 ? N : 1, cap * AGG_WAVES * npw);   // per wave-tile
  if (nodes < 2) nodes = 2;
  return (int)gt_cdiv(gt_cdiv(N > 0 ? N : 1, npw * nodes), AGG_WAVES);
}

template <int W, int EDGE, bool GCN, bool BWD>
void launch_wide_conv(const AggArgs& a, size_t lds_bytes, int* grid_bwd, hipStream_t stream) {
  if constexpr (BWD) {
    if (lds_bytes > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)(k_aggw_bwd<W, EDGE, GCN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    *grid_bwd = bwd_grid(k_aggw_bwd<W, EDGE, GCN>, lds_bytes, a.N, 1);
    hipLaunchKernelGGL((k_aggw_bwd<W, EDGE, GCN>), dim3(*grid_bwd), dim3(AGG_THREADS), lds_bytes, stream, a);
  } else {
    const int64_t waves = gt_cdiv(a.N, a.chunk);
    hipLaunchKernelGGL((k_aggw_fwd<W, EDGE, GCN>), dim3((unsigned)(gt_cdiv(gt_cdiv(waves, AGG_WAVES), 8) * 8)), dim3(AGG_THREADS),
                       lds_bytes, stream, a);
  }
}

template <int W, int EDGE, bool BWD>
void launch_wide(const AggArgs& a0, size_t lds_bytes, int* grid_bwd, hipStream_t stream) {
  AggArgs a = a0;
  if (a.E == 0) a.nbr = a.eid = a.ptr;   // the index prefetch reads entry 0 unconditionally: any readable int32 will do
  if (a.conv == GT_CONV_GCN) launch_wide_conv<W, EDGE, true, BWD>(a, lds_bytes, grid_bwd, stream);
  else launch_wide_conv<W, EDGE, false, BWD>(a, lds_bytes, grid_bwd, stream);
}

template <typename T, int EDGE, bool BWD>
int launch_cfg(const AggArgs& a, size_t lds_bytes, int* grid_bwd, hipStream_t stream) {
  const int64_t D = a.D;
  switch (wide_w<T>(D)) {
    case 4: launch_wide<4, EDGE, BWD>(a, lds_bytes, grid_bwd, stream); return GT_OK;
    case 5: launch_wide<5, EDGE, BWD>(a, lds_bytes, grid_bwd, stream); return GT_OK;
    case 6: launch_wide<6, EDGE, BWD>(a, lds_bytes, grid_bwd, stream); return GT_OK;
    case 7: launch_wide<7, EDGE, BWD>(a, lds_bytes, grid_bwd, stream); return GT_OK;
    default: break;
  }
#define GT_AGG_LAUNCH(LPN, NCH)                                                                              \
  do {                                                                                                       \
    constexpr int NPW = 64 / (LPN);                                                                          \
    if constexpr (BWD) {                                                                                     \
      if (lds_bytes > 48 * 1024)                                                                             \
        (void)hipFuncSetAttribute((const void*)(k_agg_bwd<T, LPN, NCH, EDGE>),                              \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);               \
      *grid_bwd = bwd_grid(k_agg_bwd<T, LPN, NCH, EDGE>, lds_bytes, a.N, NPW);                              \
      hipLaunchKernelGGL((k_agg_bwd<T, LPN, NCH, EDGE>), dim3(*grid_bwd), dim3(AGG_THREADS), lds_bytes,     \
                         stream, a);                                                                         \
    } else {                                                                                                 \
      int64_t waves = gt_cdiv(a.N, NPW * a.chunk);                                                           \
      hipLaunchKernelGGL((k_agg_fwd<T, LPN, NCH, EDGE>), dim3((unsigned)(gt_cdiv(gt_cdiv(waves, AGG_WAVES), 8) * 8)), \
                         dim3(AGG_THREADS), lds_bytes, stream, a);                                           \
    }                                                                                                        \
  } while (0)
  if (D <= 64) GT_AGG_LAUNCH(16, 1);
  else if (D <= 128) GT_AGG_LAUNCH(32, 1);
  else if (D <= 256) GT_AGG_LAUNCH(64, 1);
  else if (D <= 512) GT_AGG_LAUNCH(64, 2);
  else if (D <= 768) GT_AGG_LAUNCH(64, 3);
  else GT_AGG_LAUNCH(64, 4);
#undef GT_AGG_LAUNCH
  return GT_OK;
}

template <typename T, bool BWD>
int launch_edge(int edge_mode, const AggArgs& a, size_t lds_bytes, int* grid_bwd, hipStream_t stream) {
  switch (edge_mode) {
    case GT_EDGE_NONE: return launch_cfg<T, GT_EDGE_NONE, BWD>(a, lds_bytes, grid_bwd, stream);
    case GT_EDGE_LINEAR:
      if (a.K <= 2) return launch_cfg<T, EDGE_LINEAR2, BWD>(a, lds_bytes, grid_bwd, stream);
      return launch_cfg<T, GT_EDGE_LINEAR, BWD>(a, lds_bytes, grid_bwd, stream);
    case GT_EDGE_TABLES: return launch_cfg<T, GT_EDGE_TABLES, BWD>(a, lds_bytes, grid_bwd, stream);
    case GT_EDGE_DENSE: return launch_cfg<T, GT_EDGE_DENSE, BWD>(a, lds_bytes, grid_bwd, stream);
  }
  return GT_ERR_INVALID_ARG;
}

int check_common(const char* fn, int conv, int edge_mode, int dtype, int64_t N, int64_t E, int64_t D, int64_t K,
                 const void* attr, const float* w, const float* b, const int32_t* tab_off, const void* dense) {
  if (conv != GT_CONV_GCN && conv != GT_CONV_GIN) { gt_set_error("%s: bad conv %d", fn, conv); return GT_ERR_INVALID_ARG; }
  if (dtype != GT_F32 && dtype != GT_BF16) { gt_set_error("%s: bad dtype %d", fn, dtype); return GT_ERR_INVALID_ARG; }
  if (N < 0 || D <= 0) { gt_set_error("%s: bad sizes", fn); return GT_ERR_INVALID_ARG; }
  if (D % 4 != 0 || D > 1024) { gt_set_error("%s: dim %lld unsupported (need dim %% 4 == 0 and dim <= 1024)", fn, (long long)D); return GT_ERR_UNSUPPORTED; }
  if (edge_mode == GT_EDGE_LINEAR) {
    if (K < 1 || K > MAX_K) { gt_set_error("%s: Linear edge encoder needs 1 <= K <= %d (got %lld)", fn, MAX_K, (long long)K); return GT_ERR_UNSUPPORTED; }
    if ((!attr && E > 0) || !w || !b) { gt_set_error("%s: null Linear edge-encoder buffer", fn); return GT_ERR_INVALID_ARG; }
  } else if (edge_mode == GT_EDGE_TABLES) {
    if (K < 1 || K > MAX_K) { gt_set_error("%s: table edge encoder needs 1 <= K <= %d", fn, MAX_K); return GT_ERR_UNSUPPORTED; }
    if ((!attr && E > 0) || !w || !tab_off) { gt_set_error("%s: null table edge-encoder buffer", fn); return GT_ERR_INVALID_ARG; }
  } else if (edge_mode == GT_EDGE_DENSE) {
    if (!dense && E > 0) { gt_set_error("%s: null dense edge embedding", fn); return GT_ERR_INVALID_ARG; }
  } else if (edge_mode != GT_EDGE_NONE) {
    gt_set_error("%s: bad edge_mode %d", fn, edge_mode);
    return GT_ERR_INVALID_ARG;
  }
  return GT_OK;
}

size_t bwd_lds_bytes(int edge_mode, int64_t D, int64_t table_rows) {
  size_t stage = (size_t)AGG_WAVES * D * sizeof(float);
  if (edge_mode == GT_EDGE_TABLES) {   // per-wave gradient rows + max(reduction stage, staged tables)
    const size_t tab = (size_t)table_rows * D * sizeof(float);
    return (size_t)AGG_WAVES * table_rows * D * sizeof(float) + (stage > tab ? stage : tab);
  }
  return stage;
}

int bwd_slots(int edge_mode, int64_t K, int64_t table_rows) {
  if (edge_mode == GT_EDGE_LINEAR) return (int)K + 2;
  if (edge_mode == GT_EDGE_TABLES) return 1 + (int)table_rows;
  return 1;
}

}  // namespace

extern "C" int gt_aggregate_fwd(int conv, int edge_mode, int dtype, const void* h, int64_t N, int64_t E, int64_t D,
                                const int32_t* in_ptr, const int32_t* in_src, const int32_t* in_eid, const float* deg,
                                const float* dis, const float* self_param, const void* edge_attr, int64_t K,
                                const float* edge_w, const float* edge_b, const int32_t* tab_off_host,
                                int64_t table_rows, const void* edge_dense, void* out, gt_stream_t stream_) {
  int rc = check_common("gt_aggregate_fwd", conv, edge_mode, dtype, N, E, D, K, edge_attr, edge_w, edge_b,
                        tab_off_host, edge_dense);
  if (rc != GT_OK) return rc;
  GT_CHECK_ARG(h && out && in_ptr && self_param, "null buffer");
  GT_CHECK_ARG(E == 0 || (in_src && in_eid), "null CSR");
  GT_CHECK_ARG(conv != GT_CONV_GCN || (deg && dis), "GCN needs deg/dis");
  if (N == 0) return GT_OK;
  GtProfScope prof__(GT_PROF_AGGREGATE, "gt_aggregate_fwd", stream_, {N, E, D, dtype == GT_F32 ? 4 : 2,
                     edge_mode == GT_EDGE_LINEAR ? K * 4 : (edge_mode == GT_EDGE_TABLES ? K * 8 : 0), edge_mode});
  AggArgs a{};
  a.conv = conv; a.K = (int)K; a.N = N; a.E = E; a.D = D; a.h = h; a.ptr = in_ptr; a.nbr = in_src; a.eid = in_eid;
  a.deg = deg; a.dis = dis; a.self_param = self_param; a.attr = edge_attr; a.w = edge_w; a.b = edge_b;
  a.dense = edge_dense; a.out = out;
  if (edge_mode == GT_EDGE_TABLES)
    for (int k = 0; k < K; ++k) a.tab_off[k] = tab_off_host[k];
  hipStream_t stream = (hipStream_t)stream_;
  {  // nodes per wave-tile: 8 when that still gives >= 4096 wave-tiles, fewer for small batches
    const int64_t npw = D <= 64 ? 4 : (D <= 128 ? 2 : 1);
    int64_t c = N / (4096 * npw);
    a.chunk = (int)(c < 1 ? 1 : (c > FWD_CHUNK ? FWD_CHUNK : c));
  }
  // embedding-table edge encoders: the (few) table rows are parked in LDS per block when they fit
  size_t fwd_lds = 0;
  if (edge_mode == GT_EDGE_TABLES && table_rows > 0 && (size_t)table_rows * D * 4 <= 48 * 1024) {
    a.table_rows = (int)table_rows;
    fwd_lds = (size_t)table_rows * D * 4;
  }
  rc = dtype == GT_F32 ? launch_edge<float, false>(edge_mode, a, fwd_lds, nullptr, stream)
                       : launch_edge<gt_bf16, false>(edge_mode, a, fwd_lds, nullptr, stream);
  if (rc != GT_OK) return rc;
  GT_CHECK_LAUNCH();
  return GT_OK;
}

extern "C" size_t gt_aggregate_bwd_workspace_bytes(int conv, int edge_mode, int64_t D, int64_t K, int64_t table_rows) {
  (void)conv;
  return (size_t)BWD_BLOCKS * bwd_slots(edge_mode, K, table_rows) * D * sizeof(float) + 256;
}

extern "C" int gt_aggregate_bwd(int conv, int edge_mode, int dtype, const void* h, const void* grad_out, int64_t N,
                                int64_t E, int64_t D, const int32_t* out_ptr, const int32_t* out_dst,
                                const int32_t* out_eid, const float* deg, const float* dis, const float* self_param,
                                const void* edge_attr, int64_t K, const float* edge_w, const float* edge_b,
                                const int32_t* tab_off_host, int64_t table_rows, const void* edge_dense, void* grad_h,
                                float* d_self, float* d_edge_w, float* d_edge_b, void* d_dense, void* workspace,
                                size_t workspace_bytes, gt_stream_t stream_) {
  int rc = check_common("gt_aggregate_bwd", conv, edge_mode, dtype, N, E, D, K, edge_attr, edge_w, edge_b,
                        tab_off_host, edge_dense);
  if (rc != GT_OK) return rc;
  GT_CHECK_ARG(h && grad_out && grad_h && out_ptr && self_param, "null buffer");
  GT_CHECK_ARG(E == 0 || (out_dst && out_eid), "null CSC");
  GT_CHECK_ARG(conv != GT_CONV_GCN || (deg && dis), "GCN needs deg/dis");
  GT_CHECK_ARG(edge_mode != GT_EDGE_DENSE || d_dense, "dense mode needs d_dense");
  GT_CHECK_ARG(conv != GT_CONV_GIN || !d_self || true, "");
  if (edge_mode == GT_EDGE_TABLES) {
    size_t lds = bwd_lds_bytes(edge_mode, D, table_rows);
    if (table_rows < 1 || lds > 160 * 1024) {
      gt_set_error("gt_aggregate_bwd: %lld table rows x dim %lld does not fit the per-wave LDS accumulators",
                   (long long)table_rows, (long long)D);
      return GT_ERR_UNSUPPORTED;
    }
  }
  size_t need = gt_aggregate_bwd_workspace_bytes(conv, edge_mode, D, K, table_rows);
  if (!workspace || workspace_bytes < need) {
    gt_set_error("gt_aggregate_bwd: workspace too small (%zu < %zu)", workspace_bytes, need);
    return GT_ERR_WORKSPACE;
  }
  hipStream_t stream = (hipStream_t)stream_;
  GtProfScope prof__(GT_PROF_AGGREGATE, "gt_aggregate_bwd", stream_, {N, E, D, dtype == GT_F32 ? 4 : 2,
                     edge_mode == GT_EDGE_LINEAR ? K * 4 : (edge_mode == GT_EDGE_TABLES ? K * 8 : 0), edge_mode});
  const int nslots = bwd_slots(edge_mode, K, table_rows);
  int grid = 0;   // chosen per kernel instantiation (bwd_grid)
  AggArgs a{};
  a.conv = conv; a.K = (int)K; a.N = N; a.E = E; a.D = D; a.h = h; a.g = grad_out; a.ptr = out_ptr; a.nbr = out_dst;
  a.eid = out_eid; a.deg = deg; a.dis = dis; a.self_param = self_param; a.attr = edge_attr; a.w = edge_w;
  a.b = edge_b; a.dense = edge_dense; a.out = grad_h; a.d_dense = d_dense; a.partial = (float*)workspace;
  // GCN inside a deferred-reduce section (gt_defer_begin: the whole-model backward): the block partials go to the section's arena
  // (the caller's workspace is reused by the next layer) and their sums are queued below instead of launched
  float* dpart = nullptr;
  {
    const int jobs = (d_self ? 1 : 0) + (edge_mode == GT_EDGE_LINEAR ? (d_edge_w ? (int)K : 0) + (d_edge_b ? 1 : 0)
                                                                      : (edge_mode == GT_EDGE_TABLES && d_edge_w ? 1 : 0));
    // (small batches only: a step made of launches gains the five launches -- NCI1 +3 % --, at Code2's 31 k rows the arena copy of
    // the partials is cold memory every layer where the workspace stays in the Infinity Cache: -1 %)
    if (conv == GT_CONV_GCN && N <= 2048 && jobs > 0 && gt_defer_room(jobs)) dpart = (float*)gt_defer_take(need);
    if (dpart) a.partial = dpart;
  }
  a.table_rows = (int)table_rows;
  if (edge_mode == GT_EDGE_TABLES)
    for (int k = 0; k < K; ++k) a.tab_off[k] = tab_off_host[k];
  size_t lds = bwd_lds_bytes(edge_mode, D, table_rows);
  rc = dtype == GT_F32 ? launch_edge<float, true>(edge_mode, a, lds, &grid, stream)
                       : launch_edge<gt_bf16, true>(edge_mode, a, lds, &grid, stream);
  if (rc != GT_OK) return rc;
  ReduceArgs r{};
  r.partial = a.partial; r.nblocks = grid; r.nslots = nslots; r.D = D; r.conv = conv; r.edge = edge_mode;
  r.K = (int)K; r.table_rows = (int)table_rows; r.d_self = d_self; r.d_w = d_edge_w; r.d_b = d_edge_b;
  int ctiles = (int)gt_cdiv(D, 64);
  if (dpart) {   // the partials sit in the deferred-reduce section's arena: their sums join the section's one reduce launch
    const int64_t pstride = (int64_t)nslots * D;
    if (d_self) (void)(gt_defer_push(dpart, grid, D, pstride, d_self, nullptr, 0, 0, nullptr));
    if (edge_mode == GT_EDGE_LINEAR) {
      if (d_edge_w)
        for (int k = 0; k < (int)K; ++k) (void)(gt_defer_push_strided(dpart + (int64_t)(1 + k) * D, grid, D, pstride, d_edge_w + k, K));
      if (d_edge_b) (void)(gt_defer_push(dpart + (int64_t)(1 + K) * D, grid, D, pstride, d_edge_b, nullptr, 0, 0, nullptr));
    } else if (edge_mode == GT_EDGE_TABLES && d_edge_w) {
      // slots 1 .. table_rows of a block row are contiguous: one job over table_rows x D outputs, d_w[(slot - 1) * D + c]
      (void)(gt_defer_push(dpart + D, grid, (int64_t)table_rows * D, pstride, d_edge_w, nullptr, 0, 0, nullptr));
    }
    GT_CHECK_LAUNCH();
    return GT_OK;
  }
  // the block partials only hold parameter gradients (root / eps, edge-encoder weights): their reduce goes to the overlap stream
  // when there is one -- the next kernel of the backward (the dX GEMM) does not wait for it
  hipStream_t rstream = (hipStream_t)gt_overlap_dw_fork(stream_, 0 /* forked while profiled too: the brackets then hold the gather kernel alone, under the schedule the step really runs */);
  hipLaunchKernelGGL(k_agg_reduce, dim3(ctiles, nslots), dim3(RED_WAVES * 64), 0, rstream, r);
  if (conv == GT_CONV_GIN && d_self) hipLaunchKernelGGL(k_eps_finish, dim3(1), dim3(64), 0, rstream, d_self, ctiles);
  if (rstream != stream) gt_overlap_dw_booked(workspace, workspace_bytes);
  GT_CHECK_LAUNCH();
  return GT_OK;
}
