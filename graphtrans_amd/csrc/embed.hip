// embed.hip — input node encoders: sum of embedding-table rows, forward and backward.
//
// Reference (paths under /root/reference): ASTNodeEncoder.forward dataset/utils.py:28-30
//   type_emb[x[:,0]] + attr_emb[x[:,1]] + depth_emb[min(depth, max_depth)]        (Code2)
// and ogb AtomEncoder (dataset/mol.py:83): sum_i atom_embedding_list[i][x[:,i]]   (Molpcba);
// torch runs one gather per table plus adds, and a sort-based embedding_dense_backward per table
// (~40 launches, 0.54 ms per Code2 step, profiles/r01g).
//
// Forward: one wave-tile per node gathers the T table rows with 16-byte lane loads and writes the
// sum once.  Backward: d_table[r] = sum_{n: idx[n] = r} g[n] is a scatter with heavy collisions
// (21 depth rows for 31 k nodes).  To stay DETERMINISTIC without sorting, gradients are accumulated
// in 64-bit fixed point with integer atomics (integer addition is associative, so the result does
// not depend on the arrival order): scale = 2^30 / max|g| (one abs-max pass), then one scatter pass
// of int64 atomics (LDS slabs for the small tables, see k_embed_scatter) and one conversion pass.  Precision: 2^-30 relative to the largest gradient
// element, i.e. the same order as an fp32 sum that contains that element.
#include "gt_common.h"

namespace {

constexpr int ET = 256;
constexpr int MAX_TABLES = 16;

struct EmbArgs {
  int T;
  int64_t N, D;
  const int64_t* idx[MAX_TABLES];
  int64_t stride[MAX_TABLES];
  int64_t clamp[MAX_TABLES];   // max index (inclusive) or -1
  const float* table[MAX_TABLES];
  int64_t row_off[MAX_TABLES]; // bwd: row offset of table t in the concatenated accumulator
  int64_t rows[MAX_TABLES];
  float* dtable[MAX_TABLES];
  const float* g;
  float* out;
  long long* acc;       // [total_rows][D] int64 fixed point
  unsigned* absmax;     // bit pattern of max |g|
};

__device__ __forceinline__ int64_t emb_index(const EmbArgs& a, int t, int64_t n) {
  int64_t i = a.idx[t][n * a.stride[t]];
  if (a.clamp[t] >= 0 && i > a.clamp[t]) i = a.clamp[t];
  return i;
}

__global__ void __launch_bounds__(ET) k_embed_fwd(EmbArgs a) {
  const int64_t C = a.D / 4, total = a.N * C;
  for (int64_t i = (int64_t)blockIdx.x * ET + threadIdx.x; i < total; i += (int64_t)gridDim.x * ET) {
    const int64_t n = i / C, c = (i % C) * 4;
    float4 s = gt_zero4();
    for (int t = 0; t < a.T; ++t)
      s = gt_add4(s, *reinterpret_cast<const float4*>(a.table[t] + emb_index(a, t, n) * a.D + c));
    *reinterpret_cast<float4*>(a.out + n * a.D + c) = s;
  }
}

__global__ void __launch_bounds__(ET) k_embed_absmax(const float* __restrict__ g, int64_t total4, unsigned* __restrict__ absmax) {
  __shared__ float red[ET / 64];
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * ET + threadIdx.x; i < total4; i += (int64_t)gridDim.x * ET) {
    const float4 v = *reinterpret_cast<const float4*>(g + i * 4);
    m = fmaxf(fmaxf(fmaxf(m, fabsf(v.x)), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < ET / 64; ++w) m = fmaxf(m, red[w]);
    // positive floats order like their bit patterns; the plain read only skips atomics that cannot win
    const unsigned bits = __float_as_uint(m);
    if (m > 0.f && bits > __hip_atomic_load(absmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(absmax, bits);
  }
}

__device__ __forceinline__ float emb_scale(const unsigned* absmax) {
  const float m = __uint_as_float(*absmax);
  if (!(m > 0.f) || !isfinite(m)) return 1.0f;
  int e;
  frexpf(m, &e);                 // m = f * 2^e, f in [0.5, 1)
  return ldexpf(1.0f, 30 - e);   // max|g| * scale in [2^29, 2^30): exact power of two
}

// Scatter: block = (node chunk, 32-column slab); thread = (column, one of 8 node lanes).  Tables with
// few rows (depth: 21 rows for ~31 k nodes) would serialise thousands of global atomics per address,
// so every table marked `small` is accumulated in an LDS copy of its slab first (ds_add_u64) and
// flushed with one global atomic per (row, column) per block; large tables (attribute: 10 k rows)
// go straight to global atomics where collisions are rare.
constexpr int SC_COLS = 32, SC_LANES = ET / SC_COLS;
// nodes per block: 512 for big batches (fewer slab flushes), 128 for small ones (more blocks in flight)
constexpr int SC_LDS_ROWS = 256;  // sum of rows over the small tables must fit: 256 * 32 * 8 B = 64 KB

__global__ void __launch_bounds__(ET) k_embed_scatter(EmbArgs a, unsigned small_mask, int lds_rows, int SC_NODES) {
  extern __shared__ unsigned long long slab[];  // [lds_rows][SC_COLS]
  const int cl = threadIdx.x % SC_COLS, nl = threadIdx.x / SC_COLS;
  const int64_t c = (int64_t)blockIdx.y * SC_COLS + cl;
  const int64_t n0 = (int64_t)blockIdx.x * SC_NODES;
  const int64_t n1 = n0 + SC_NODES < a.N ? n0 + SC_NODES : a.N;
  for (int i = threadIdx.x; i < lds_rows * SC_COLS; i += ET) slab[i] = 0ull;
  __syncthreads();
  const float scale = emb_scale(a.absmax);
  if (c < a.D) {
    // small tables: LDS slab
    if (small_mask) {
      for (int64_t n = n0 + nl; n < n1; n += SC_LANES) {
        const long long q = __float2ll_rn(a.g[n * a.D + c] * scale);
        if (q == 0) continue;
        int lrow = 0;
        for (int t = 0; t < a.T; ++t) {
          if (!a.dtable[t] || !(small_mask >> t & 1)) continue;
          atomicAdd(&slab[(lrow + emb_index(a, t, n)) * SC_COLS + cl], (unsigned long long)q);
          lrow += (int)a.rows[t];
        }
      }
    }
    // large tables: straight to global atomics, but consecutive nodes of a thread that hit the SAME row are
    // summed in a register first (attribute vocabularies are skewed: one hot row would otherwise serialise
    // thousands of atomics on 300 addresses; uniform data pays nothing for the run-length check)
    for (int t = 0; t < a.T; ++t) {
      if (!a.dtable[t] || (small_mask >> t & 1)) continue;
      int64_t cur_row = -1;
      long long cur_sum = 0;
      for (int64_t n = n0 + nl; n < n1; n += SC_LANES) {
        const long long q = __float2ll_rn(a.g[n * a.D + c] * scale);
        const int64_t r = emb_index(a, t, n);
        if (r != cur_row) {
          if (cur_sum) atomicAdd(reinterpret_cast<unsigned long long*>(a.acc + (a.row_off[t] + cur_row) * a.D + c),
                                 (unsigned long long)cur_sum);
          cur_row = r;
          cur_sum = 0;
        }
        cur_sum += q;
      }
      if (cur_sum) atomicAdd(reinterpret_cast<unsigned long long*>(a.acc + (a.row_off[t] + cur_row) * a.D + c),
                             (unsigned long long)cur_sum);
    }
  }
  __syncthreads();
  if (c < a.D) {
    int lrow = 0;
    for (int t = 0; t < a.T; ++t) {
      if (!a.dtable[t] || !(small_mask >> t & 1)) continue;
      for (int r = nl; r < (int)a.rows[t]; r += SC_LANES) {
        const unsigned long long v = slab[(lrow + r) * SC_COLS + cl];
        if (v) atomicAdd(reinterpret_cast<unsigned long long*>(a.acc + (a.row_off[t] + r) * a.D + c), v);
      }
      lrow += (int)a.rows[t];
    }
  }
}

__global__ void __launch_bounds__(ET) k_embed_convert(EmbArgs a) {
  const float inv = 1.0f / emb_scale(a.absmax);
  for (int t = 0; t < a.T; ++t) {
    if (!a.dtable[t]) continue;
    const int64_t total = a.rows[t] * a.D;
    const long long* src = a.acc + a.row_off[t] * a.D;
    for (int64_t i = (int64_t)blockIdx.x * ET + threadIdx.x; i < total; i += (int64_t)gridDim.x * ET)
      a.dtable[t][i] = (float)src[i] * inv;
  }
}

int emb_check(const char* fn, int T, int64_t N, int64_t D) {
  if (T < 1 || T > MAX_TABLES) { gt_set_error("%s: 1..%d tables supported", fn, MAX_TABLES); return GT_ERR_UNSUPPORTED; }
  if (N < 0 || D <= 0 || D % 4) { gt_set_error("%s: dim must be a positive multiple of 4", fn); return GT_ERR_UNSUPPORTED; }
  return GT_OK;
}

int grid_for(int64_t items) {
  int64_t g = gt_cdiv(items, ET * 2);
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

extern "C" int gt_embed_sum_fwd(int num_tables, const int64_t* const* idx_ptrs_host, const int64_t* idx_strides_host,
                                const int64_t* clamp_max_host, const float* const* tables_host, int64_t N, int64_t D,
                                float* out, gt_stream_t stream_) {
  int rc = emb_check("gt_embed_sum_fwd", num_tables, N, D);
  if (rc) return rc;
  if (N == 0) return GT_OK;
  GT_CHECK_ARG(idx_ptrs_host && idx_strides_host && clamp_max_host && tables_host && out, "null buffer");
  EmbArgs a{};
  a.T = num_tables; a.N = N; a.D = D; a.out = out;
  for (int t = 0; t < num_tables; ++t) {
    a.idx[t] = idx_ptrs_host[t]; a.stride[t] = idx_strides_host[t]; a.clamp[t] = clamp_max_host[t]; a.table[t] = tables_host[t];
    GT_CHECK_ARG(a.idx[t] && a.table[t], "null table / index pointer");
  }
  hipLaunchKernelGGL(k_embed_fwd, dim3(grid_for(N * (D / 4))), dim3(ET), 0, (hipStream_t)stream_, a);
  GT_CHECK_LAUNCH();
  return GT_OK;
}

extern "C" size_t gt_embed_sum_bwd_workspace_bytes(int num_tables, const int64_t* table_rows_host, int64_t D) {
  int64_t rows = 0;
  for (int t = 0; t < num_tables && t < MAX_TABLES; ++t) rows += table_rows_host[t];
  return (size_t)rows * D * sizeof(long long) + 256;
}

extern "C" int gt_embed_sum_bwd(int num_tables, const int64_t* const* idx_ptrs_host, const int64_t* idx_strides_host,
                                const int64_t* clamp_max_host, const int64_t* table_rows_host, const float* grad_out,
                                int64_t N, int64_t D, float* const* d_tables_host, void* workspace, size_t workspace_bytes,
                                gt_stream_t stream_) {
  int rc = emb_check("gt_embed_sum_bwd", num_tables, N, D);
  if (rc) return rc;
  GT_CHECK_ARG(idx_ptrs_host && idx_strides_host && clamp_max_host && table_rows_host && (grad_out || N == 0) && d_tables_host, "null buffer");
  const size_t need = gt_embed_sum_bwd_workspace_bytes(num_tables, table_rows_host, D);
  if (!workspace || workspace_bytes < need) { gt_set_error("gt_embed_sum_bwd: workspace too small"); return GT_ERR_WORKSPACE; }
  hipStream_t stream = (hipStream_t)stream_;
  EmbArgs a{};
  a.T = num_tables; a.N = N; a.D = D; a.g = grad_out;
  int64_t off = 0;
  for (int t = 0; t < num_tables; ++t) {
    a.idx[t] = idx_ptrs_host[t]; a.stride[t] = idx_strides_host[t]; a.clamp[t] = clamp_max_host[t];
    a.rows[t] = table_rows_host[t]; a.row_off[t] = off; a.dtable[t] = d_tables_host[t];
    off += table_rows_host[t];
  }
  // workspace: [absmax word (256 B)] [int64 accumulators]
  a.absmax = reinterpret_cast<unsigned*>(workspace);
  a.acc = reinterpret_cast<long long*>(reinterpret_cast<char*>(workspace) + 256);
  (void)hipMemsetAsync(workspace, 0, need, stream);
  if (N > 0) {
    int64_t ag = gt_cdiv(N * (D / 4), ET * 4);
    hipLaunchKernelGGL(k_embed_absmax, dim3((unsigned)(ag > 1024 ? 1024 : ag)), dim3(ET), 0, stream, grad_out, N * (D / 4),
                       a.absmax);
    // smallest tables first into the LDS slab until it is full
    unsigned small_mask = 0;
    int lds_rows = 0;
    for (;;) {
      int best = -1;
      for (int t = 0; t < num_tables; ++t)
        if (a.dtable[t] && !(small_mask >> t & 1) && (best < 0 || a.rows[t] < a.rows[best])) best = t;
      if (best < 0 || lds_rows + a.rows[best] > SC_LDS_ROWS) break;
      small_mask |= 1u << best;
      lds_rows += (int)a.rows[best];
    }
    const int sc_nodes = N >= 16384 ? 512 : 128;
    hipLaunchKernelGGL(k_embed_scatter, dim3((unsigned)gt_cdiv(N, sc_nodes), (unsigned)gt_cdiv(D, SC_COLS)), dim3(ET),
                       (size_t)lds_rows * SC_COLS * sizeof(unsigned long long), stream, a, small_mask, lds_rows, sc_nodes);
  }
  hipLaunchKernelGGL(k_embed_convert, dim3(grid_for(off * D)), dim3(ET), 0, stream, a);
  GT_CHECK_LAUNCH();
  return GT_OK;
}
