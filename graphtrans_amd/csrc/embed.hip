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
  // Every thread walks its nodes in batches of SC_U: all gradient values and all row indices of a batch are
  // loaded first (independent loads in flight), then accumulated -- one node per trip is a chain of dependent
  // L2 round trips (64 of them per thread) and made this kernel 4x slower than its atomics.
  constexpr int SC_U = 8;
  if (c < a.D) {
    // small tables: LDS slab
    if (small_mask) {
      for (int64_t nb = n0 + nl; nb < n1; nb += (int64_t)SC_U * SC_LANES) {
        long long q[SC_U];
#pragma unroll
        for (int u = 0; u < SC_U; ++u) {
          const int64_t n = nb + (int64_t)u * SC_LANES;
          q[u] = n < n1 ? __float2ll_rn(a.g[n * a.D + c] * scale) : 0;
        }
        int lrow = 0;
        for (int t = 0; t < a.T; ++t) {
          if (!a.dtable[t] || !(small_mask >> t & 1)) continue;
          int r[SC_U];
#pragma unroll
          for (int u = 0; u < SC_U; ++u) {
            const int64_t n = nb + (int64_t)u * SC_LANES;
            r[u] = n < n1 ? (int)emb_index(a, t, n) : 0;
          }
#pragma unroll
          for (int u = 0; u < SC_U; ++u)
            if (q[u]) atomicAdd(&slab[(lrow + r[u]) * SC_COLS + cl], (unsigned long long)q[u]);
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
      for (int64_t nb = n0 + nl; nb < n1; nb += (int64_t)SC_U * SC_LANES) {
        long long q[SC_U];
        int64_t r[SC_U];
#pragma unroll
        for (int u = 0; u < SC_U; ++u) {
          const int64_t n = nb + (int64_t)u * SC_LANES;
          const bool ok = n < n1;
          q[u] = ok ? __float2ll_rn(a.g[n * a.D + c] * scale) : 0;
          r[u] = ok ? emb_index(a, t, n) : -1;
        }
#pragma unroll
        for (int u = 0; u < SC_U; ++u) {
          if (r[u] < 0) continue;
          if (r[u] != cur_row) {
            if (cur_sum) atomicAdd(reinterpret_cast<unsigned long long*>(a.acc + (a.row_off[t] + cur_row) * a.D + c),
                                   (unsigned long long)cur_sum);
            cur_row = r[u];
            cur_sum = 0;
          }
          cur_sum += q[u];
        }
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

// ================================================================================================
// Sorted path: d_table[r] = sum over the nodes of row r IN NODE ORDER, plain fp32, no atomics.
//
// gt_embed_sort (once per batch, off the critical path) sorts the node ids of every table by row with a stable
// counting sort (per-block histograms -> scan -> ranked fill); gt_embed_sum_bwd_sorted then reduces the
// gradient rows as SEGMENTS of that order: a block takes 64 consecutive sorted positions and walks them
// with the running sum in a register, starting a new sum whenever the row changes.  Rows that lie entirely
// inside the chunk are stored directly; the (at most two) rows that cross the chunk's ends leave a partial
// in head[chunk] / tail[chunk], and a second kernel adds the partials of each crossing row in chunk order.
// Fixed summation order -> bitwise reproducible, exact fp32 (no fixed-point rounding), and one hot row
// (skewed vocabularies) costs a serial pass over its partials instead of serialised atomics.
// ================================================================================================
constexpr int SB = 1024;     // keys per sort block
constexpr int SEG_CH = 64;   // sorted positions per reduce block
constexpr int SEG_T = 320;   // threads per reduce block (one column each per pass)
constexpr int MAX_SORT_ROWS = 16384;

struct SortArgs {
  int T;
  int64_t N;
  const int64_t* idx[MAX_TABLES];
  int64_t stride[MAX_TABLES];
  int64_t clamp[MAX_TABLES];
  int64_t rows[MAX_TABLES];
  int64_t row_off[MAX_TABLES];  // offset of table t's bins in the concatenated histogram
  int64_t total_rows;
  int nblk;
  int32_t* hist;       // [nblk][total_rows] transient: counts, then exclusive prefixes over the blocks
  int32_t* tot;        // [total_rows] transient: keys per bin
  int32_t* bin_start;  // [total_rows + T]  table t at row_off[t] + t, rows[t] + 1 entries
  int32_t* order;      // [T][N] node id at sorted position
  int32_t* keys;       // [T][N] row at sorted position
};

__device__ __forceinline__ int sort_key(const SortArgs& a, int t, int64_t n) {
  int64_t i = a.idx[t][n * a.stride[t]];
  if (a.clamp[t] >= 0 && i > a.clamp[t]) i = a.clamp[t];
  return (int)i;
}

// S1: per-block histogram of the keys of block b (grid: nblk x T)
__global__ void __launch_bounds__(SB) k_esort_hist(SortArgs a) {
  extern __shared__ int lh[];
  const int t = blockIdx.y, b = blockIdx.x;
  const int rows = (int)a.rows[t];
  for (int k = threadIdx.x; k < rows; k += SB) lh[k] = 0;
  __syncthreads();
  const int64_t n = (int64_t)b * SB + threadIdx.x;
  if (n < a.N) {
    const int k = sort_key(a, t, n);
    if (k >= 0 && k < rows) atomicAdd(&lh[k], 1);
  }
  __syncthreads();
  int32_t* out = a.hist + (int64_t)b * a.total_rows + a.row_off[t];
  for (int k = threadIdx.x; k < rows; k += SB) out[k] = lh[k];
}

// S2a: per bin (all tables concatenated), exclusive prefix over the blocks and the bin total
__global__ void __launch_bounds__(256) k_esort_colscan(SortArgs a) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= a.total_rows) return;
  int32_t* col = a.hist + k;
  int run = 0, b = 0;
  for (; b + 8 <= a.nblk; b += 8) {
    int v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = col[(int64_t)(b + u) * a.total_rows];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      col[(int64_t)(b + u) * a.total_rows] = run;
      run += v[u];
    }
  }
  for (; b < a.nblk; ++b) {
    const int v = col[(int64_t)b * a.total_rows];
    col[(int64_t)b * a.total_rows] = run;
    run += v;
  }
  a.tot[k] = run;
}

// S2b: exclusive scan of the bin totals of table t (grid: T, 1024 threads); bin_start[rows] = number of keys
__global__ void __launch_bounds__(1024) k_esort_binscan(SortArgs a) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int rows = (int)a.rows[t];
  const int32_t* tot = a.tot + a.row_off[t];
  int32_t* bs = a.bin_start + a.row_off[t] + t;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int k0 = 0; k0 < rows; k0 += 1024) {
    const int k = k0 + tid;
    const int total = k < rows ? tot[k] : 0;
    int incl = total;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(incl, o, 64);
      if (lane >= o) incl += u;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int pre = carry_s;
    for (int q = 0; q < w; ++q) pre += wsum[q];
    if (k < rows) bs[k] = pre + incl - total;
    __syncthreads();
    if (tid == 1023) carry_s = pre + incl;
    __syncthreads();
  }
  if (tid == 0) bs[rows] = carry_s;
}

// S3: stable fill: position = bin start + keys of the same bin in earlier blocks + earlier keys in this block
__global__ void __launch_bounds__(SB) k_esort_fill(SortArgs a) {
  __shared__ int lk[SB];
  const int t = blockIdx.y, b = blockIdx.x;
  const int rows = (int)a.rows[t];
  const int64_t n = (int64_t)b * SB + threadIdx.x;
  int k = -1;
  if (n < a.N) {
    k = sort_key(a, t, n);
    if (k < 0 || k >= rows) k = -1;   // out-of-range indices are dropped (the forward would have faulted)
  }
  lk[threadIdx.x] = k;
  __syncthreads();
  if (k < 0) return;
  int rank = 0;
  for (int j = 0; j < (int)threadIdx.x; ++j) rank += lk[j] == k;
  const int pos = a.bin_start[a.row_off[t] + t + k] + a.hist[(int64_t)b * a.total_rows + a.row_off[t] + k] + rank;
  a.order[(int64_t)t * a.N + pos] = (int32_t)n;
  a.keys[(int64_t)t * a.N + pos] = k;
}

struct SegArgs {
  int T;
  int64_t N, D;
  int nchunks;
  int64_t row_off[MAX_TABLES];
  int64_t rows[MAX_TABLES];
  const int32_t* bin_start;
  const int32_t* order;
  const int32_t* keys;
  const float* g;
  float* dtable[MAX_TABLES];
  float* head;  // [T][nchunks][D]
  float* tail;  // [T][nchunks][D]
};

// R1 (grid: nchunks x T)
__global__ void __launch_bounds__(SEG_T) k_eseg_reduce(SegArgs a) {
  const int t = blockIdx.y, ch = blockIdx.x;
  float* dt = a.dtable[t];
  if (!dt) return;
  const int32_t* ord = a.order + (int64_t)t * a.N;
  const int32_t* ks = a.keys + (int64_t)t * a.N;
  const int32_t* bs = a.bin_start + a.row_off[t] + t;
  const int64_t p0 = (int64_t)ch * SEG_CH;
  const int64_t npos = bs[a.rows[t]];  // sorted positions of this table (out-of-range indices were dropped)
  const int64_t p1 = p0 + SEG_CH < npos ? p0 + SEG_CH : npos;
  if (p0 >= p1) return;
  const int first_row = ks[p0], last_row = ks[p1 - 1];
  const bool before = bs[first_row] < p0;                       // first row started in an earlier chunk
  const bool after = (int64_t)bs[last_row + 1] > p1;            // last row continues in a later chunk
  float* head = a.head + ((int64_t)t * a.nchunks + ch) * a.D;
  float* tail = a.tail + ((int64_t)t * a.nchunks + ch) * a.D;
  for (int64_t c = threadIdx.x; c < a.D; c += SEG_T) {
    int cur = first_row;
    float acc = 0.f;
    auto flush = [&](int row, float v) {
      const bool b = row == first_row && before, af = row == last_row && after;
      if (b) head[c] = v;            // continuation (also a chunk that lies wholly inside one row)
      else if (af) tail[c] = v;      // starts here, continues
      else dt[(int64_t)row * a.D + c] = v;
    };
    for (int64_t pb = p0; pb < p1; pb += 8) {
      int node[8], row[8];
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool ok = pb + u < p1;
        node[u] = ok ? ord[pb + u] : 0;
        row[u] = ok ? ks[pb + u] : -1;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = row[u] >= 0 ? a.g[(int64_t)node[u] * a.D + c] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (row[u] < 0) continue;
        if (row[u] != cur) {
          flush(cur, acc);
          cur = row[u];
          acc = 0.f;
        }
        acc += v[u];
      }
    }
    flush(cur, acc);
  }
}

// R2 (grid: nchunks x T): the block of the chunk in which a crossing row STARTS adds its partials in chunk order
__global__ void __launch_bounds__(SEG_T) k_eseg_fixup(SegArgs a) {
  const int t = blockIdx.y, ch = blockIdx.x;
  float* dt = a.dtable[t];
  if (!dt) return;
  const int32_t* ks = a.keys + (int64_t)t * a.N;
  const int32_t* bs = a.bin_start + a.row_off[t] + t;
  const int64_t p0 = (int64_t)ch * SEG_CH;
  const int64_t npos = bs[a.rows[t]];
  const int64_t p1 = p0 + SEG_CH < npos ? p0 + SEG_CH : npos;
  if (p0 >= p1) return;
  const int row = ks[p1 - 1];
  const int64_t seg_end = bs[row + 1];
  if (seg_end <= p1 || (int64_t)bs[row] < p0) return;   // does not cross, or started earlier (someone else's)
  const int last_chunk = (int)((seg_end - 1) / SEG_CH);  // chunks ch+1 .. last_chunk hold one head piece each
  const float* tail = a.tail + ((int64_t)t * a.nchunks + ch) * a.D;
  const float* head = a.head + (int64_t)t * a.nchunks * a.D;
  for (int64_t c = threadIdx.x; c < a.D; c += SEG_T) {
    float acc = tail[c];
    int j = ch + 1;
    for (; j + 7 <= last_chunk; j += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = head[(int64_t)(j + u) * a.D + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; j <= last_chunk; ++j) acc += head[(int64_t)j * a.D + c];
    dt[(int64_t)row * a.D + c] = acc;
  }
}

struct SortLayout {
  int64_t total_rows = 0, row_off[MAX_TABLES] = {};
  size_t bin_off = 0, order_off = 0, keys_off = 0, bytes = 0;
};
SortLayout sort_layout(int T, const int64_t* rows, int64_t N) {
  SortLayout L;
  for (int t = 0; t < T; ++t) { L.row_off[t] = L.total_rows; L.total_rows += rows[t]; }
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  L.bin_off = 0;
  L.order_off = up((size_t)(L.total_rows + T) * 4);
  L.keys_off = L.order_off + up((size_t)T * N * 4);
  L.bytes = L.keys_off + up((size_t)T * N * 4);
  return L;
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

extern "C" size_t gt_embed_sort_plan_bytes(int num_tables, const int64_t* table_rows_host, int64_t N) {
  if (num_tables < 1 || num_tables > MAX_TABLES) return 0;
  return sort_layout(num_tables, table_rows_host, N).bytes;
}

extern "C" size_t gt_embed_sort_workspace_bytes(int num_tables, const int64_t* table_rows_host, int64_t N) {
  if (num_tables < 1 || num_tables > MAX_TABLES) return 0;
  const SortLayout L = sort_layout(num_tables, table_rows_host, N);
  return ((size_t)gt_cdiv(N > 0 ? N : 1, SB) + 1) * L.total_rows * sizeof(int32_t) + 256;
}

extern "C" int gt_embed_sort(int num_tables, const int64_t* const* idx_ptrs_host, const int64_t* idx_strides_host,
                             const int64_t* clamp_max_host, const int64_t* table_rows_host, int64_t N, void* plan,
                             size_t plan_bytes, void* workspace, size_t workspace_bytes, gt_stream_t stream_) {
  int rc = emb_check("gt_embed_sort", num_tables, N, 4);
  if (rc) return rc;
  GT_CHECK_ARG(idx_ptrs_host && idx_strides_host && clamp_max_host && table_rows_host, "null buffer");
  GT_CHECK_ARG(N < (int64_t)1 << 31, "more than 2^31 nodes");
  for (int t = 0; t < num_tables; ++t)
    if (table_rows_host[t] < 1 || table_rows_host[t] > MAX_SORT_ROWS) {
      gt_set_error("gt_embed_sort: table %d has %lld rows (1..%d supported)", t, (long long)table_rows_host[t], MAX_SORT_ROWS);
      return GT_ERR_UNSUPPORTED;
    }
  const SortLayout L = sort_layout(num_tables, table_rows_host, N);
  if (!plan || plan_bytes < L.bytes) { gt_set_error("gt_embed_sort: plan buffer too small"); return GT_ERR_WORKSPACE; }
  if (!workspace || workspace_bytes < gt_embed_sort_workspace_bytes(num_tables, table_rows_host, N)) {
    gt_set_error("gt_embed_sort: workspace too small");
    return GT_ERR_WORKSPACE;
  }
  hipStream_t stream = (hipStream_t)stream_;
  SortArgs a{};
  a.T = num_tables; a.N = N; a.total_rows = L.total_rows; a.nblk = (int)gt_cdiv(N > 0 ? N : 1, SB);
  int64_t max_rows = 0;
  for (int t = 0; t < num_tables; ++t) {
    a.idx[t] = idx_ptrs_host[t]; a.stride[t] = idx_strides_host[t]; a.clamp[t] = clamp_max_host[t];
    a.rows[t] = table_rows_host[t]; a.row_off[t] = L.row_off[t];
    GT_CHECK_ARG(a.idx[t] || N == 0, "null index pointer");
    if (a.rows[t] > max_rows) max_rows = a.rows[t];
  }
  char* pb = static_cast<char*>(plan);
  a.hist = static_cast<int32_t*>(workspace);
  a.tot = a.hist + (size_t)a.nblk * L.total_rows;
  a.bin_start = reinterpret_cast<int32_t*>(pb + L.bin_off);
  a.order = reinterpret_cast<int32_t*>(pb + L.order_off);
  a.keys = reinterpret_cast<int32_t*>(pb + L.keys_off);
  hipLaunchKernelGGL(k_esort_hist, dim3(a.nblk, num_tables), dim3(SB), (size_t)max_rows * sizeof(int), stream, a);
  hipLaunchKernelGGL(k_esort_colscan, dim3((unsigned)gt_cdiv(L.total_rows, 256)), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(k_esort_binscan, dim3(num_tables), dim3(1024), 0, stream, a);
  if (N > 0) hipLaunchKernelGGL(k_esort_fill, dim3(a.nblk, num_tables), dim3(SB), 0, stream, a);
  GT_CHECK_LAUNCH();
  return GT_OK;
}

extern "C" size_t gt_embed_sum_bwd_sorted_workspace_bytes(int num_tables, int64_t N, int64_t D) {
  return (size_t)2 * num_tables * gt_cdiv(N > 0 ? N : 1, SEG_CH) * D * sizeof(float) + 256;
}

extern "C" int gt_embed_sum_bwd_sorted(int num_tables, const int64_t* table_rows_host, const float* grad_out, int64_t N,
                                       int64_t D, const void* plan, float* const* d_tables_host, void* workspace,
                                       size_t workspace_bytes, gt_stream_t stream_) {
  int rc = emb_check("gt_embed_sum_bwd_sorted", num_tables, N, D);
  if (rc) return rc;
  GT_CHECK_ARG(table_rows_host && d_tables_host && (plan || N == 0) && (grad_out || N == 0), "null buffer");
  if (!workspace || workspace_bytes < gt_embed_sum_bwd_sorted_workspace_bytes(num_tables, N, D)) {
    gt_set_error("gt_embed_sum_bwd_sorted: workspace too small");
    return GT_ERR_WORKSPACE;
  }
  hipStream_t stream = (hipStream_t)stream_;
  const SortLayout L = sort_layout(num_tables, table_rows_host, N);
  // rows nobody indexes get zero: one memset per run of tables that lie back to back (the fused path's flat gradient buffer
  // holds them contiguously: one launch instead of one per table)
  for (int t = 0; t < num_tables;) {
    if (!d_tables_host[t]) { ++t; continue; }
    char* lo = (char*)d_tables_host[t];
    char* hi = lo + (size_t)table_rows_host[t] * D * sizeof(float);
    int u = t + 1;
    while (u < num_tables && d_tables_host[u] && (char*)d_tables_host[u] >= hi && (char*)d_tables_host[u] - hi < 64) {
      hi = (char*)d_tables_host[u] + (size_t)table_rows_host[u] * D * sizeof(float);   // (a gap is the 16-byte segment padding)
      ++u;
    }
    (void)hipMemsetAsync(lo, 0, (size_t)(hi - lo), stream);
    t = u;
  }
  if (N == 0) return GT_OK;
  SegArgs a{};
  a.T = num_tables; a.N = N; a.D = D; a.nchunks = (int)gt_cdiv(N, SEG_CH); a.g = grad_out;
  const char* pb = static_cast<const char*>(plan);
  a.bin_start = reinterpret_cast<const int32_t*>(pb + L.bin_off);
  a.order = reinterpret_cast<const int32_t*>(pb + L.order_off);
  a.keys = reinterpret_cast<const int32_t*>(pb + L.keys_off);
  for (int t = 0; t < num_tables; ++t) { a.row_off[t] = L.row_off[t]; a.rows[t] = table_rows_host[t]; a.dtable[t] = d_tables_host[t]; }
  a.head = static_cast<float*>(workspace);
  a.tail = a.head + (size_t)num_tables * a.nchunks * D;
  hipLaunchKernelGGL(k_eseg_reduce, dim3(a.nchunks, num_tables), dim3(SEG_T), 0, stream, a);
  hipLaunchKernelGGL(k_eseg_fixup, dim3(a.nchunks, num_tables), dim3(SEG_T), 0, stream, a);
  GT_CHECK_LAUNCH();
  return GT_OK;
}
