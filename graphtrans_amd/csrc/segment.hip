// segment.hip — per-graph segment ops and node-row <-> token-row layout kernels.
//
// Reference semantics (paths under /root/reference):
//   h + vn[batch]                         modules/gnn_module.py:199   (index + add)
//   global_add_pool(h, batch) + vn        modules/gnn_module.py:219   (torch_scatter atomics)
//   pad_batch / unpad_batch               modules/utils.py:5-53       (python loop over graphs)
//   CLS concat + mask extend              modules/transformer_encoder.py:50-55
// `batch` is sorted (PyG collation), so every graph is a contiguous row range [ptr[b], ptr[b+1]):
// segment sums need no atomics and pad/unpad are pure index arithmetic on graph_ptr.
#include "gt_common.h"

namespace {

constexpr int SEG_THREADS = 256;

// out[n] = x[n] + seg[node_graph[n]] ; flat over node rows (no per-graph grid: graph sizes are ragged)
template <typename T>
__global__ void __launch_bounds__(SEG_THREADS) k_bcast_add(const T* __restrict__ x, const T* __restrict__ seg,
                                                           const int32_t* __restrict__ node_graph, int64_t N,
                                                           int64_t D, T* __restrict__ out) {
  const int64_t C = D / 4;  // float4 chunks per row
  const int64_t total = N * C;
  for (int64_t i = (int64_t)blockIdx.x * SEG_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * SEG_THREADS) {
    int64_t r = i / C, c = (i % C) * 4;
    float4 s = gt_load4<T>(seg + (int64_t)node_graph[r] * D + c);
    if (x) s = gt_add4(s, gt_load4<T>(x + r * D + c));
    gt_store4<T>(out + r * D + c, s);
  }
}

// out[b] = add[b] + sum_{n in graph b} x[n] ; grid (column tiles of 64 chunks, B); 4 waves stride rows.
template <typename T>
__global__ void __launch_bounds__(SEG_THREADS) k_segment_sum(const T* __restrict__ x, const T* __restrict__ add,
                                                             const int32_t* __restrict__ gptr, int64_t D,
                                                             T* __restrict__ out) {
  __shared__ float4 sm[4][64];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t c = ((int64_t)blockIdx.x * 64 + lane) * 4;
  const bool act = c < D;
  const int beg = gptr[b], end = gptr[b + 1];
  float4 acc = gt_zero4();
  if (act) {
    int r = beg + wid;
    // 8 independent row loads in flight per wave (a serial chain of L2 round trips otherwise)
    for (; r + 28 < end; r += 32) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = gt_load4<T>(x + (int64_t)(r + 4 * u) * D + c);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = gt_add4(acc, v[u]);
    }
    for (; r < end; r += 4) acc = gt_add4(acc, gt_load4<T>(x + (int64_t)r * D + c));
  }
  sm[wid][lane] = acc;
  __syncthreads();
  if (wid == 0 && act) {
    float4 t = gt_add4(gt_add4(sm[0][lane], sm[1][lane]), gt_add4(sm[2][lane], sm[3][lane]));
    if (add) t = gt_add4(t, gt_load4<T>(add + (int64_t)b * D + c));
    gt_store4<T>(out + (int64_t)b * D + c, t);
  }
}

// ---- the same sum in two load-balanced phases (needs a workspace): graph sizes are ragged (11 .. 2000 nodes in
// a Code2 batch) and with one block per graph the largest graph is the kernel's tail (2.4 MB through one CU).
// Phase 1: a block walks SS_CH consecutive ROWS with the running sum in a register (8 rows in flight), starting
// a new sum at every graph boundary; graphs inside the chunk are stored directly, the (at most two) graphs
// that cross its ends leave head / tail partials.  Phase 2: one block per graph adds its partials in chunk
// order (and handles empty graphs).  Fixed order -> reproducible.
constexpr int SS_CH = 64, SS_T = 320;

template <typename T>
__device__ __forceinline__ float ldf(const T* p) {
  if constexpr (sizeof(T) == 4) return (float)*reinterpret_cast<const float*>(p);
  else return gt_bf16_to_f32(*reinterpret_cast<const gt_bf16*>(p));
}
template <typename T>
__device__ __forceinline__ void stf(T* p, float v) {
  if constexpr (sizeof(T) == 4) *reinterpret_cast<float*>(p) = v;
  else *reinterpret_cast<gt_bf16*>(p) = gt_f32_to_bf16(v);
}

template <typename T>
__global__ void __launch_bounds__(SS_T) k_segsum_chunks(const T* __restrict__ x, const T* __restrict__ add,
                                                        const int32_t* __restrict__ gptr, int B, int64_t N, int64_t D,
                                                        T* __restrict__ out, float* __restrict__ head, float* __restrict__ tail) {
  const int64_t p0 = (int64_t)blockIdx.x * SS_CH;
  const int64_t p1 = p0 + SS_CH < N ? p0 + SS_CH : N;
  if (p0 >= p1) return;
  // graph of row p0: last b with gptr[b] <= p0 (empty graphs share a start: take the one that owns the row)
  int lo = 0, hi = B;   // invariant: gptr[lo] <= p0 < gptr[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (gptr[mid] <= p0) lo = mid; else hi = mid;
  }
  const int g0 = lo;
  float* hd = head + (int64_t)blockIdx.x * D;
  float* tl = tail + (int64_t)blockIdx.x * D;
  for (int64_t c = threadIdx.x; c < D; c += SS_T) {
    int g = g0;
    int64_t gend = gptr[g + 1];
    float acc = 0.f;
    auto flush = [&](int gg, float v) {
      const bool before = gptr[gg] < p0, after = (int64_t)gptr[gg + 1] > p1;
      if (before) hd[c] = v;
      else if (after) tl[c] = v;
      else stf<T>(out + (int64_t)gg * D + c, v + (add ? ldf<T>(add + (int64_t)gg * D + c) : 0.f));
    };
    for (int64_t rb = p0; rb < p1; rb += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = rb + u < p1 ? ldf<T>(x + (rb + u) * D + c) : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t r = rb + u;
        if (r >= p1) break;
        if (r == gend) {            // row r opens the next non-empty graph
          flush(g, acc);
          acc = 0.f;
          do { ++g; gend = gptr[g + 1]; } while (gend <= r);
        }
        acc += v[u];
      }
    }
    flush(g, acc);
  }
}

template <typename T>
__global__ void __launch_bounds__(SS_T) k_segsum_fixup(const T* __restrict__ add, const int32_t* __restrict__ gptr, int64_t D,
                                                       const float* __restrict__ head, const float* __restrict__ tail,
                                                       T* __restrict__ out) {
  const int b = blockIdx.x;
  const int64_t beg = gptr[b], end = gptr[b + 1];
  if (end > beg && beg / SS_CH == (end - 1) / SS_CH) return;   // inside one chunk: phase 1 stored it
  const int64_t c0 = beg / SS_CH, c1 = end > beg ? (end - 1) / SS_CH : 0;
  for (int64_t c = threadIdx.x; c < D; c += SS_T) {
    float acc = 0.f;
    if (end > beg) {
      acc = tail[c0 * D + c];
      int64_t j = c0 + 1;
      for (; j + 7 <= c1; j += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = head[(j + u) * D + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
      }
      for (; j <= c1; ++j) acc += head[j * D + c];
    }
    stf<T>(out + (int64_t)b * D + c, acc + (add ? ldf<T>(add + (int64_t)b * D + c) : 0.f));
  }
}

// tokens[row(b,p)] <- node row | cls | 0 ; grid (position tiles, B)
template <typename T>
__global__ void __launch_bounds__(SEG_THREADS) k_seq_gather(const T* __restrict__ h, const T* __restrict__ cls,
                                                            const float* __restrict__ cls32,
                                                            const int32_t* __restrict__ gptr,
                                                            const int32_t* __restrict__ desc, int64_t row_stride,
                                                            int64_t max_npos, int with_cls, int64_t D,
                                                            T* __restrict__ tokens, uint8_t* __restrict__ pad_mask,
                                                            int pos_per_block) {
  const int b = blockIdx.y;
  const int row0 = desc[b * 4 + 0], npos = desc[b * 4 + 1], kv_off = desc[b * 4 + 2], kv_len = desc[b * 4 + 3];
  const int kept = kv_len - with_cls;
  const int node0 = gptr[b + 1] - kept;  // first kept node (graphs keep their LAST `kept` nodes)
  const int p0 = blockIdx.x * pos_per_block;
  if (p0 >= npos) return;
  const int p1 = p0 + pos_per_block < npos ? p0 + pos_per_block : npos;
  const int64_t C = D / 4;
  const int64_t total = (int64_t)(p1 - p0) * C;
  for (int64_t i = threadIdx.x; i < total; i += SEG_THREADS) {
    int p = p0 + (int)(i / C);
    int64_t c = (i % C) * 4;
    int j = p - kv_off;
    float4 v = gt_zero4();
    if (j >= 0 && j < kept) v = gt_load4<T>(h + (int64_t)(node0 + j) * D + c);
    else if (with_cls && j == kept) v = cls32 ? gt_load4<float>(cls32 + c) : gt_load4<T>(cls + c);
    gt_store4<T>(tokens + ((int64_t)row0 + (int64_t)p * row_stride) * D + c, v);
  }
  if (pad_mask)
    for (int p = p0 + threadIdx.x; p < p1; p += SEG_THREADS)
      pad_mask[(int64_t)b * max_npos + p] = (p < kv_off || p >= kv_off + kv_len) ? 1 : 0;
}

// adjoint / unpad: node rows <- token rows ; flat over node rows, CLS rows by the first B blocks
template <typename T>
__global__ void __launch_bounds__(SEG_THREADS) k_seq_scatter(const T* __restrict__ tokens, const T* __restrict__ base,
                                                             const int32_t* __restrict__ gptr,
                                                             const int32_t* __restrict__ node_graph,
                                                             const int32_t* __restrict__ desc, int64_t num_seqs,
                                                             int64_t row_stride, int with_cls, int64_t N, int64_t D,
                                                             T* __restrict__ h_out, T* __restrict__ cls_out) {
  const int64_t C = D / 4;
  if (cls_out && with_cls) {
    for (int64_t i = (int64_t)blockIdx.x * SEG_THREADS + threadIdx.x; i < num_seqs * C;
         i += (int64_t)gridDim.x * SEG_THREADS) {
      int64_t b = i / C, c = (i % C) * 4;
      int row0 = desc[b * 4 + 0], kv_off = desc[b * 4 + 2], kv_len = desc[b * 4 + 3];
      gt_store4<T>(cls_out + b * D + c,
                   gt_load4<T>(tokens + ((int64_t)row0 + (int64_t)(kv_off + kv_len - 1) * row_stride) * D + c));
    }
  }
  const int64_t total = N * C;
  for (int64_t i = (int64_t)blockIdx.x * SEG_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * SEG_THREADS) {
    int64_t r = i / C, c = (i % C) * 4;
    int b = node_graph[r];
    int row0 = desc[b * 4 + 0], kv_off = desc[b * 4 + 2], kv_len = desc[b * 4 + 3];
    int node0 = gptr[b + 1] - (kv_len - with_cls);
    float4 v;
    if (r >= node0) v = gt_load4<T>(tokens + ((int64_t)row0 + (int64_t)(kv_off + (int)r - node0) * row_stride) * D + c);
    else v = base ? gt_load4<T>(base + r * D + c) : gt_zero4();
    gt_store4<T>(h_out + r * D + c, v);
  }
}

int check(const char* fn, int dtype, int64_t D) {
  if (dtype != GT_F32 && dtype != GT_BF16) { gt_set_error("%s: bad dtype", fn); return GT_ERR_INVALID_ARG; }
  if (D <= 0 || D % 4 != 0) { gt_set_error("%s: dim %lld must be a positive multiple of 4", fn, (long long)D); return GT_ERR_UNSUPPORTED; }
  return GT_OK;
}

// rows handled per block so that a block moves ~32 KB
int rows_per_block(int64_t D, int elt) {
  int64_t r = (32 * 1024) / (D * elt);
  return (int)(r < 1 ? 1 : (r > 64 ? 64 : r));
}

}  // namespace

static int flat_grid(int64_t items) {
  int64_t g = gt_cdiv(items, SEG_THREADS * 4);
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

extern "C" int gt_segment_bcast_add(int dtype, const void* x, const void* seg, const int32_t* node_graph, int64_t N,
                                    int64_t B, int64_t D, void* out, gt_stream_t stream_) {
  (void)B;
  int rc = check("gt_segment_bcast_add", dtype, D);
  if (rc) return rc;
  GT_CHECK_ARG(seg && node_graph && out, "null buffer");
  if (N == 0) return GT_OK;
  hipStream_t stream = (hipStream_t)stream_;
  dim3 grid(flat_grid(N * (D / 4)));
  if (dtype == GT_F32)
    hipLaunchKernelGGL(k_bcast_add<float>, grid, dim3(SEG_THREADS), 0, stream, (const float*)x, (const float*)seg,
                       node_graph, N, D, (float*)out);
  else
    hipLaunchKernelGGL(k_bcast_add<gt_bf16>, grid, dim3(SEG_THREADS), 0, stream, (const gt_bf16*)x,
                       (const gt_bf16*)seg, node_graph, N, D, (gt_bf16*)out);
  GT_CHECK_LAUNCH();
  return GT_OK;
}

extern "C" int gt_segment_sum(int dtype, const void* x, const void* add, const int32_t* graph_ptr, int64_t N,
                              int64_t B, int64_t D, void* out, gt_stream_t stream_) {
  (void)N;
  int rc = check("gt_segment_sum", dtype, D);
  if (rc) return rc;
  GT_CHECK_ARG(x && graph_ptr && out, "null buffer");
  if (B == 0) return GT_OK;
  GT_CHECK_ARG(B <= 65535, "more than 65535 graphs per batch");
  hipStream_t stream = (hipStream_t)stream_;
  dim3 grid((unsigned)gt_cdiv(D / 4, 64), (unsigned)B);
  if (dtype == GT_F32)
    hipLaunchKernelGGL(k_segment_sum<float>, grid, dim3(SEG_THREADS), 0, stream, (const float*)x, (const float*)add,
                       graph_ptr, D, (float*)out);
  else
    hipLaunchKernelGGL(k_segment_sum<gt_bf16>, grid, dim3(SEG_THREADS), 0, stream, (const gt_bf16*)x,
                       (const gt_bf16*)add, graph_ptr, D, (gt_bf16*)out);
  GT_CHECK_LAUNCH();
  return GT_OK;
}

static int seq_gather_impl(const char* fn, int dtype, const void* h, const void* cls, const float* cls32, const int32_t* graph_ptr,
                           const int32_t* seq_desc, int64_t num_seqs, int64_t row_stride, int64_t max_npos,
                           int with_cls, int64_t D, void* tokens, uint8_t* pad_mask, gt_stream_t stream_) {
  int rc = check(fn, dtype, D);
  if (rc) return rc;
  if (!(h && graph_ptr && seq_desc && tokens)) { gt_set_error("%s: null buffer", fn); return GT_ERR_INVALID_ARG; }
  if (with_cls && !cls && !cls32) { gt_set_error("%s: with_cls needs the cls row", fn); return GT_ERR_INVALID_ARG; }
  if (num_seqs == 0 || max_npos == 0) return GT_OK;
  if (num_seqs > 65535) { gt_set_error("%s: more than 65535 sequences", fn); return GT_ERR_INVALID_ARG; }
  hipStream_t stream = (hipStream_t)stream_;
  int ppb = rows_per_block(D, dtype == GT_F32 ? 4 : 2);
  dim3 grid((unsigned)gt_cdiv(max_npos, ppb), (unsigned)num_seqs);
  if (dtype == GT_F32)
    hipLaunchKernelGGL(k_seq_gather<float>, grid, dim3(SEG_THREADS), 0, stream, (const float*)h, (const float*)cls, cls32,
                       graph_ptr, seq_desc, row_stride, max_npos, with_cls, D, (float*)tokens, pad_mask, ppb);
  else
    hipLaunchKernelGGL(k_seq_gather<gt_bf16>, grid, dim3(SEG_THREADS), 0, stream, (const gt_bf16*)h,
                       (const gt_bf16*)cls, cls32, graph_ptr, seq_desc, row_stride, max_npos, with_cls, D, (gt_bf16*)tokens,
                       pad_mask, ppb);
  GT_CHECK_LAUNCH();
  return GT_OK;
}

extern "C" int gt_seq_gather(int dtype, const void* h, const void* cls, const int32_t* graph_ptr,
                             const int32_t* seq_desc, int64_t num_seqs, int64_t row_stride, int64_t max_npos,
                             int with_cls, int64_t D, void* tokens, uint8_t* pad_mask, gt_stream_t stream_) {
  return seq_gather_impl("gt_seq_gather", dtype, h, cls, nullptr, graph_ptr, seq_desc, num_seqs, row_stride, max_npos, with_cls, D,
                         tokens, pad_mask, stream_);
}

extern "C" int gt_seq_gather_cls32(int dtype, const void* h, const float* cls32, const int32_t* graph_ptr,
                                   const int32_t* seq_desc, int64_t num_seqs, int64_t row_stride, int64_t max_npos,
                                   int with_cls, int64_t D, void* tokens, gt_stream_t stream_) {
  return seq_gather_impl("gt_seq_gather_cls32", dtype, h, nullptr, cls32, graph_ptr, seq_desc, num_seqs, row_stride, max_npos,
                         with_cls, D, tokens, nullptr, stream_);
}

// out[c] = sum over rows of x[r][c], fp32, fixed order: (256 / CW) row groups x CW column chunks per block pass, CW = 16 / 32 / 64 by
// the row width; a thread takes every (256 / CW)-th row, 8 rows (index, then row) in flight.  (r5: with 64 chunk columns fixed, the
// CLS-row gradient of a d = 128 encoder -- 256 rows through row_idx -- ran on 128 threads as 64 chained index -> row round trips:
// 37 us of the main stream for 64 KB, profiles/r05 timeline; now ~5.)
template <typename T>
__global__ void __launch_bounds__(256) k_colsum_f32(const T* __restrict__ x, const int64_t* __restrict__ row_idx, int64_t rows, int64_t D,
                                                    float* __restrict__ out) {
  __shared__ float4 part[256];
  const int64_t C = D / 4;
  const int CW = C >= 64 ? 64 : (C >= 32 ? 32 : 16);
  const int groups = 256 / CW;
  const int g = threadIdx.x / CW, cl = threadIdx.x % CW;
  for (int64_t c0 = (int64_t)blockIdx.x * CW; c0 < C; c0 += (int64_t)gridDim.x * CW) {
    const int64_t c = c0 + cl;
    float4 acc = gt_zero4();
    if (c < C) {
      int64_t r = g;
      for (; r + 7 * groups < rows; r += 8 * groups) {
        int64_t idx[8];
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) idx[u] = row_idx ? row_idx[r + u * groups] : r + u * groups;
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = gt_load4<T>(x + idx[u] * D + c * 4);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = gt_add4(acc, v[u]);
      }
      for (; r < rows; r += groups) acc = gt_add4(acc, gt_load4<T>(x + (row_idx ? row_idx[r] : r) * D + c * 4));
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (g == 0 && c < C) {
      float4 s = part[cl];
      for (int k = 1; k < groups; ++k) s = gt_add4(s, part[k * CW + cl]);
      gt_store4<float>(out + c * 4, s);
    }
    __syncthreads();
  }
}

static int colsum_impl(const char* fn, int dtype, const void* x, const int64_t* row_idx, int64_t rows, int64_t D, float* out, gt_stream_t stream_) {
  int rc = check(fn, dtype, D);
  if (rc) return rc;
  if (!(x && out)) { gt_set_error("%s: null buffer", fn); return GT_ERR_INVALID_ARG; }
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t C = D / 4;
  dim3 grid((unsigned)gt_cdiv(C, C >= 64 ? 64 : (C >= 32 ? 32 : 16)));
  if (dtype == GT_F32) hipLaunchKernelGGL(k_colsum_f32<float>, grid, dim3(256), 0, stream, (const float*)x, row_idx, rows, D, out);
  else hipLaunchKernelGGL(k_colsum_f32<gt_bf16>, grid, dim3(256), 0, stream, (const gt_bf16*)x, row_idx, rows, D, out);
  GT_CHECK_LAUNCH();
  return GT_OK;
}
extern "C" int gt_colsum_f32(int dtype, const void* x, int64_t rows, int64_t D, float* out, gt_stream_t stream_) {
  return colsum_impl("gt_colsum_f32", dtype, x, nullptr, rows, D, out, stream_);
}
// ... over the rows row_idx[0 .. rows) of x (the CLS rows of the token matrix: the gradient of the cls embedding without a gather)
extern "C" int gt_colsum_rows_f32(int dtype, const void* x, const int64_t* row_idx, int64_t rows, int64_t D, float* out, gt_stream_t stream_) {
  GT_CHECK_ARG(row_idx, "null row index");
  return colsum_impl("gt_colsum_rows_f32", dtype, x, row_idx, rows, D, out, stream_);
}

// The token row of every node (the row map of gt_linear_set_rows) + the CLS rows of the token matrix, for layouts WITHOUT pad rows
// (the packed layout: kv_off = 0, npos = kv_len): rows[r] = row0(b) + (kv_off + r - first kept node of b) * row_stride, -1 for the
// leading nodes a truncated graph drops (modules/utils.py:17-21 keeps the LAST max_num_nodes); tokens[cls row of b] = cls32.
template <typename T>
__global__ void __launch_bounds__(SEG_THREADS) k_seq_token_rows(const float* __restrict__ cls32, const int32_t* __restrict__ gptr,
                                                                const int32_t* __restrict__ node_graph, const int32_t* __restrict__ desc,
                                                                int64_t num_seqs, int64_t row_stride, int with_cls, int64_t N, int64_t D,
                                                                T* __restrict__ tokens, int32_t* __restrict__ rows,
                                                                const float* __restrict__ ln_w, const float* __restrict__ ln_b, float ln_eps,
                                                                T* __restrict__ ln_out, float* __restrict__ ln_mean, float* __restrict__ ln_rstd) {
  const int64_t C = D / 4;
  const int64_t t0 = (int64_t)blockIdx.x * SEG_THREADS + threadIdx.x, step = (int64_t)gridDim.x * SEG_THREADS;
  for (int64_t r = t0; r < N; r += step) {
    const int b = node_graph[r];
    const int4 d = *reinterpret_cast<const int4*>(desc + b * 4);   // {row0, npos, kv_off, kv_len}
    const int node0 = gptr[b + 1] - (d.w - with_cls);
    rows[r] = r >= node0 ? (int32_t)(d.x + (int64_t)(d.z + (int)r - node0) * row_stride) : -1;
  }
  if (!with_cls) return;
  if (!ln_out) {
    for (int64_t i = t0; i < num_seqs * C; i += step) {
      const int64_t b = i / C, c = (i % C) * 4;
      const int4 d = *reinterpret_cast<const int4*>(desc + b * 4);
      gt_store4<T>(tokens + ((int64_t)d.x + (int64_t)(d.z + d.w - 1) * row_stride) * D + c, gt_load4<float>(cls32 + c));
    }
    return;
  }
  // with the LayerNorm of the token rows (norm_input): one wave per CLS row; the statistics of the STORED (T-rounded) row, two passes
  const int lane = threadIdx.x & 63;
  const int64_t w0 = t0 >> 6, nw = step >> 6;
  for (int64_t b = w0; b < num_seqs; b += nw) {
    const int4 d = *reinterpret_cast<const int4*>(desc + b * 4);
    const int64_t row = (int64_t)d.x + (int64_t)(d.z + d.w - 1) * row_stride;
    float s = 0.f;
    for (int64_t c = lane; c < C; c += 64) {
      gt_store4<T>(tokens + row * D + c * 4, gt_load4<float>(cls32 + c * 4));
      const float4 v = gt_load4<T>(tokens + row * D + c * 4);   // (this lane's own store: what the LayerNorm kernels would read)
      s += (v.x + v.y) + (v.z + v.w);
    }
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) s += __shfl_xor(s, sh, 64);
    const float mu = s / (float)D;
    float q = 0.f;
    for (int64_t c = lane; c < C; c += 64) {
      const float4 v = gt_load4<T>(tokens + row * D + c * 4);
      const float dx = v.x - mu, dy = v.y - mu, dz = v.z - mu, dw = v.w - mu;
      q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) q += __shfl_xor(q, sh, 64);
    const float rs = 1.0f / sqrtf(q / (float)D + ln_eps);
    for (int64_t c = lane; c < C; c += 64) {
      const float4 v = gt_load4<T>(tokens + row * D + c * 4);
      const float4 w4 = *reinterpret_cast<const float4*>(ln_w + c * 4), b4 = *reinterpret_cast<const float4*>(ln_b + c * 4);
      gt_store4<T>(ln_out + row * D + c * 4, make_float4((v.x - mu) * rs * w4.x + b4.x, (v.y - mu) * rs * w4.y + b4.y,
                                                          (v.z - mu) * rs * w4.z + b4.z, (v.w - mu) * rs * w4.w + b4.w));
    }
    if (lane == 0) {
      ln_mean[row] = mu;
      ln_rstd[row] = rs;
    }
  }
}
static int seq_token_rows_impl(const char* fn, int dtype, const float* cls32, const int32_t* graph_ptr, const int32_t* node_graph,
                               const int32_t* seq_desc, int64_t num_seqs, int64_t row_stride, int with_cls, int64_t N, int64_t D, void* tokens,
                               int32_t* rows, const float* ln_w, const float* ln_b, float ln_eps, void* ln_out, float* ln_mean, float* ln_rstd,
                               gt_stream_t stream_) {
  int rc = check(fn, dtype, D);
  if (rc) return rc;
  if (!(graph_ptr && node_graph && seq_desc && tokens && rows)) { gt_set_error("%s: null buffer", fn); return GT_ERR_INVALID_ARG; }
  if (with_cls && !cls32) { gt_set_error("%s: with_cls needs the cls row", fn); return GT_ERR_INVALID_ARG; }
  if (num_seqs == 0) return GT_OK;
  hipStream_t stream = (hipStream_t)stream_;
  int64_t items = N > num_seqs * (D / 4) ? N : num_seqs * (D / 4);
  if (ln_out && items < num_seqs * 64) items = num_seqs * 64;   // a wave per CLS row
  dim3 grid((unsigned)(gt_cdiv(items, SEG_THREADS) < 1024 ? gt_cdiv(items, SEG_THREADS) : 1024));
  if (dtype == GT_F32)
    hipLaunchKernelGGL(k_seq_token_rows<float>, grid, dim3(SEG_THREADS), 0, stream, cls32, graph_ptr, node_graph, seq_desc, num_seqs, row_stride,
                       with_cls, N, D, (float*)tokens, rows, ln_w, ln_b, ln_eps, (float*)ln_out, ln_mean, ln_rstd);
  else
    hipLaunchKernelGGL(k_seq_token_rows<gt_bf16>, grid, dim3(SEG_THREADS), 0, stream, cls32, graph_ptr, node_graph, seq_desc, num_seqs, row_stride,
                       with_cls, N, D, (gt_bf16*)tokens, rows, ln_w, ln_b, ln_eps, (gt_bf16*)ln_out, ln_mean, ln_rstd);
  GT_CHECK_LAUNCH();
  return GT_OK;
}
extern "C" int gt_seq_token_rows(int dtype, const float* cls32, const int32_t* graph_ptr, const int32_t* node_graph, const int32_t* seq_desc,
                                 int64_t num_seqs, int64_t row_stride, int with_cls, int64_t N, int64_t D, void* tokens, int32_t* rows,
                                 gt_stream_t stream_) {
  return seq_token_rows_impl("gt_seq_token_rows", dtype, cls32, graph_ptr, node_graph, seq_desc, num_seqs, row_stride, with_cls, N, D, tokens, rows,
                             nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, stream_);
}
// ... whose CLS rows also get the LayerNorm of gt_linear_set_rows_layernorm (the node rows get it in the GEMM's epilogue)
extern "C" int gt_seq_token_rows_layernorm(int dtype, const float* cls32, const int32_t* graph_ptr, const int32_t* node_graph,
                                           const int32_t* seq_desc, int64_t num_seqs, int64_t row_stride, int with_cls, int64_t N, int64_t D,
                                           void* tokens, int32_t* rows, const float* ln_w, const float* ln_b, float ln_eps, void* ln_out,
                                           float* ln_mean, float* ln_rstd, gt_stream_t stream_) {
  GT_CHECK_ARG(ln_w && ln_b && ln_out && ln_mean && ln_rstd, "null LayerNorm buffer");
  return seq_token_rows_impl("gt_seq_token_rows_layernorm", dtype, cls32, graph_ptr, node_graph, seq_desc, num_seqs, row_stride, with_cls, N, D,
                             tokens, rows, ln_w, ln_b, ln_eps, ln_out, ln_mean, ln_rstd, stream_);
}

extern "C" int gt_seq_scatter(int dtype, const void* tokens, const void* base, const int32_t* graph_ptr,
                              const int32_t* node_graph, const int32_t* seq_desc, int64_t num_seqs,
                              int64_t row_stride, int with_cls, int64_t N, int64_t D, void* h_out, void* cls_out,
                              gt_stream_t stream_) {
  int rc = check("gt_seq_scatter", dtype, D);
  if (rc) return rc;
  GT_CHECK_ARG(tokens && graph_ptr && node_graph && seq_desc && h_out, "null buffer");
  if (num_seqs == 0) return GT_OK;
  hipStream_t stream = (hipStream_t)stream_;
  int64_t items = (N > num_seqs ? N : num_seqs) * (D / 4);
  dim3 grid(flat_grid(items));
  if (dtype == GT_F32)
    hipLaunchKernelGGL(k_seq_scatter<float>, grid, dim3(SEG_THREADS), 0, stream, (const float*)tokens,
                       (const float*)base, graph_ptr, node_graph, seq_desc, num_seqs, row_stride, with_cls, N, D,
                       (float*)h_out, (float*)cls_out);
  else
    hipLaunchKernelGGL(k_seq_scatter<gt_bf16>, grid, dim3(SEG_THREADS), 0, stream, (const gt_bf16*)tokens,
                       (const gt_bf16*)base, graph_ptr, node_graph, seq_desc, num_seqs, row_stride, with_cls, N, D,
                       (gt_bf16*)h_out, (gt_bf16*)cls_out);
  GT_CHECK_LAUNCH();
  return GT_OK;
}

extern "C" size_t gt_segment_sum_workspace_bytes(int64_t N, int64_t D) {
  return (size_t)2 * gt_cdiv(N > 0 ? N : 1, SS_CH) * D * sizeof(float) + 256;
}

extern "C" int gt_segment_sum_ws(int dtype, const void* x, const void* add, const int32_t* graph_ptr, int64_t N, int64_t B,
                                 int64_t D, void* out, void* workspace, size_t workspace_bytes, gt_stream_t stream_) {
  // small inputs: the one-block-per-graph kernel is a single launch and just as fast
  if (N < 4096 || !workspace || workspace_bytes < gt_segment_sum_workspace_bytes(N, D))
    return gt_segment_sum(dtype, x, add, graph_ptr, N, B, D, out, stream_);
  int rc = check("gt_segment_sum_ws", dtype, D);
  if (rc) return rc;
  GT_CHECK_ARG(x && graph_ptr && out, "null buffer");
  if (B == 0) return GT_OK;
  GT_CHECK_ARG(B <= 0x7fffffff && N <= 0x7fffffff, "sizes beyond int32");
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t nch = gt_cdiv(N, SS_CH);
  float* head = static_cast<float*>(workspace);
  float* tail = head + nch * D;
  if (dtype == GT_F32) {
    hipLaunchKernelGGL(k_segsum_chunks<float>, dim3((unsigned)nch), dim3(SS_T), 0, stream, (const float*)x, (const float*)add,
                       graph_ptr, (int)B, N, D, (float*)out, head, tail);
    hipLaunchKernelGGL(k_segsum_fixup<float>, dim3((unsigned)B), dim3(SS_T), 0, stream, (const float*)add, graph_ptr, D,
                       (const float*)head, (const float*)tail, (float*)out);
  } else {
    hipLaunchKernelGGL(k_segsum_chunks<gt_bf16>, dim3((unsigned)nch), dim3(SS_T), 0, stream, (const gt_bf16*)x,
                       (const gt_bf16*)add, graph_ptr, (int)B, N, D, (gt_bf16*)out, head, tail);
    hipLaunchKernelGGL(k_segsum_fixup<gt_bf16>, dim3((unsigned)B), dim3(SS_T), 0, stream, (const gt_bf16*)add, graph_ptr, D,
                       (const float*)head, (const float*)tail, (gt_bf16*)out);
  }
  GT_CHECK_LAUNCH();
  return GT_OK;
}

// ---- packed token layout built on the device (graph.py:SeqLayout.packed_on_device) -----------------------------------------
// For a batch whose per-graph sizes are NOT known on the host (a bare device-resident PyG Batch): desc / last_rows / the
// attention work list from graph_ptr alone, no device->host copy.  One block; graphs are walked in chunks of 1024 with
// running prefix carries.  meta = {rows, num_work, max kv_len, S}.  Work entries past num_work are {-1, -1} (the
// attention kernels skip them): the host sizes its launches by the upper bounds rows <= N + B*cls, num_work <= B +
// rows / 64.
__global__ void __launch_bounds__(1024) k_seq_layout_packed(const int32_t* __restrict__ gptr, int B, int max_len, int cls,
                                                            int32_t* __restrict__ desc, int64_t* __restrict__ last_rows,
                                                            int32_t* __restrict__ work, int64_t work_cap, int32_t* __restrict__ meta) {
  __shared__ int sred[32];
  __shared__ int sscan[2][1024];
  __shared__ int scarry[2];
  const int t = threadIdx.x;
  // S = min(max nodes per graph, max_input_len)   (modules/utils.py:16)
  int mx = 0;
  for (int b = t; b < B; b += 1024) mx = max(mx, gptr[b + 1] - gptr[b]);
  for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
  if ((t & 63) == 0) sred[t >> 6] = mx;
  __syncthreads();
  if (t == 0) {
    int m = 0;
    for (int i = 0; i < 16; ++i) m = max(m, sred[i]);
    sred[16] = m < max_len ? m : max_len;
    scarry[0] = 0;
    scarry[1] = 0;
  }
  __syncthreads();
  const int S = sred[16];
  int max_kv = 0;
  for (int b0 = 0; b0 < B; b0 += 1024) {
    const int b = b0 + t;
    int kv = 0, tiles = 0;
    if (b < B) {
      const int n = gptr[b + 1] - gptr[b];
      kv = (n < S ? n : S) + cls;
      tiles = (kv + 63) >> 6;
      max_kv = max(max_kv, kv);
    }
    sscan[0][t] = kv;
    sscan[1][t] = tiles;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {   // inclusive Hillis-Steele scan of both columns
      int a0 = 0, a1 = 0;
      if (t >= o) { a0 = sscan[0][t - o]; a1 = sscan[1][t - o]; }
      __syncthreads();
      sscan[0][t] += a0;
      sscan[1][t] += a1;
      __syncthreads();
    }
    const int row0 = scarry[0] + sscan[0][t] - kv, w0 = scarry[1] + sscan[1][t] - tiles;
    if (b < B) {
      desc[b * 4 + 0] = row0;
      desc[b * 4 + 1] = kv;
      desc[b * 4 + 2] = 0;
      desc[b * 4 + 3] = kv;
      last_rows[b] = (int64_t)row0 + kv - 1;
      for (int i = 0; i < tiles; ++i)
        if (w0 + i < work_cap) { work[(int64_t)(w0 + i) * 2] = b; work[(int64_t)(w0 + i) * 2 + 1] = i; }
    }
    __syncthreads();
    if (t == 1023) { scarry[0] += sscan[0][1023]; scarry[1] += sscan[1][1023]; }
    __syncthreads();
  }
  for (int o = 32; o > 0; o >>= 1) max_kv = max(max_kv, __shfl_xor(max_kv, o));
  if ((t & 63) == 0) sred[t >> 6] = max_kv;
  __syncthreads();
  const int rows = scarry[0], nwork = scarry[1];
  for (int64_t w = nwork + t; w < work_cap; w += 1024) { work[w * 2] = -1; work[w * 2 + 1] = -1; }
  if (t == 0) {
    int m = 0;
    for (int i = 0; i < 16; ++i) m = max(m, sred[i]);
    meta[0] = rows; meta[1] = nwork; meta[2] = m; meta[3] = S;
  }
}

extern "C" int gt_seq_layout_packed(const int32_t* graph_ptr, int64_t B, int64_t max_input_len, int with_cls, int32_t* seq_desc,
                                    int64_t* last_rows, int32_t* work_items, int64_t work_capacity, int32_t* meta,
                                    gt_stream_t stream_) {
  GT_CHECK_ARG(graph_ptr && seq_desc && last_rows && work_items && meta, "null buffer");
  GT_CHECK_ARG(B >= 0 && B <= 65535 && max_input_len > 0 && work_capacity >= 0, "bad sizes");
  if (B == 0) return GT_OK;
  hipLaunchKernelGGL(k_seq_layout_packed, dim3(1), dim3(1024), 0, (hipStream_t)stream_, graph_ptr, (int)B,
                     (int)(max_input_len > 0x3fffffff ? 0x3fffffff : max_input_len), with_cls ? 1 : 0, seq_desc, last_rows, work_items,
                     work_capacity, meta);
  GT_CHECK_LAUNCH();
  return GT_OK;
}
