/*
 * graphtrans_hip.h — C ABI of libgraphtrans_hip.so (MI355X / gfx950 kernels for the GraphTrans
 * forward/backward hot path).
 *
 * The reference (ucbrise/graphtrans) is pure Python and has NO FFI / plugin registry; its hot path
 * bottoms out in third-party native ops.  Each entry point below names the
 * reference call site (file:line under /root/reference) and the third-party op it replaces; a
 * maintainer binds them with ctypes.
 *
 * Conventions (every function):
 *   - extern "C", plain pointers and sizes only; every pointer is a DEVICE pointer unless named
 *     `*_host`.  All buffers (inputs, outputs, workspaces) are caller-owned: the library never
 *     allocates/frees device memory, never synchronises the device and enqueues work only on the
 *     `stream` argument (a hipStream_t passed as void*; NULL = the null stream).
 *   - returns GT_OK (0) or a negative gt_status; `gt_last_error()` (thread-local) describes it.
 *   - re-entrant, no mutable global state (autograd calls backward from a second host thread).
 *   - matrices are row-major and contiguous; index arrays produced by gt_graph_prep are int32.
 *   - `dtype`: GT_F32 or GT_BF16 storage; accumulation is always fp32.
 */
#ifndef GRAPHTRANS_HIP_H
#define GRAPHTRANS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* gt_stream_t; /* hipStream_t */

enum gt_status {
  GT_OK = 0,
  GT_ERR_INVALID_ARG = -1,
  GT_ERR_UNSUPPORTED = -2,
  GT_ERR_WORKSPACE = -3,
  GT_ERR_LAUNCH = -4
};

enum gt_dtype { GT_F32 = 0, GT_BF16 = 1 };
enum gt_conv { GT_CONV_GCN = 0, GT_CONV_GIN = 1, GT_CONV_PNA = 2 /* the whole-model driver's third layer kind (gt_pna_layer) */ };
enum gt_edge_mode {
  GT_EDGE_NONE = 0,   /* edge embedding == 0          (dataset/tud.py:67-71)            */
  GT_EDGE_LINEAR = 1, /* nn.Linear(K, D) on f32 attrs (dataset/code.py:117), K <= 4     */
  GT_EDGE_TABLES = 2, /* sum of K embedding rows      (ogb BondEncoder, dataset/mol.py:84) */
  GT_EDGE_DENSE = 3   /* caller-computed (E, D) edge embedding, any edge_encoder module */
};

int gt_version(void);
const char* gt_last_error(void);

/* Events for cross-stream dependencies between entry points (thin wrappers over hipEvent, timing
 * disabled): gt_event_record marks a point on `stream`, gt_stream_wait_event makes `stream` wait for it. */
void* gt_event_create(void);
/* Side streams of the fused path (non-blocking HIP streams).  level -1 / 0 / +1 = the device's highest / default / lowest
 * stream priority: the weight-gradient GEMMs (nothing waits for them until the optimizer) run on a lowest-priority
 * stream so that the critical-path kernels of the other streams get the compute units that free up first. */
void* gt_stream_create(int level);
void gt_stream_destroy(void* stream);
int gt_stream_priority_range(int* least, int* greatest);
void gt_event_destroy(void* event);
int gt_event_record(void* event, gt_stream_t stream);
int gt_stream_wait_event(gt_stream_t stream, void* event);

/* Named runtime options: alternative implementations kept in the library as tested yardsticks (tests/test_hip_options.py runs each
 * non-default value against the oracle).  Process-wide, read at every call.  gt_option_set returns the previous value (0 / 1) or a
 * negative status for an unknown name; values are small non-negative integers (0 / 1 unless stated).
 *   "attn_f32_exact"       1: fp32 token rows run attention on the exact v_mfma_f32_16x16x4_f32 chains instead of bf16x6 products
 *   "bnstats_rows_kernel"  1: a gt_linear_bwd_bnstats request is also taken by the register-row bf16x6 dX kernel (one partial row per
 *                             128 rows; measured slower than the separate partial pass, off by default)
 *   "lin_ring"             which bf16 GEMMs with a bound fragment image (gt_w1_bind) run the LDS-ring kernel k_lin2 (csrc/linear2.h):
 *                             0 = from 65 536 rows on, and every covered shape the weight-stationary kernel does not take (default);
 *                             1 = never (k_lin1 / the tiled kernels: the yardstick); 2 = every covered shape from 2 048 rows on */
int gt_option_set(const char* name, int value);
int gt_option_get(const char* name);

/* Opt-in launch profiler: HIP events on the launch stream around the selected entry points
 * (mask: 1 aggregate, 2 attention, 4 linear).  gt_profile_enable(mask != 0) clears old records and
 * starts recording, (0) stops; after a device synchronisation gt_profile_get returns the entry
 * point name, its elapsed milliseconds and the 6 size fields it was called with. */
int gt_profile_enable(unsigned mask);
int gt_profile_resume(unsigned mask); /* change the mask, keep the records (sampling) */
int64_t gt_profile_count(void);
int gt_profile_get(int64_t index, char* name_out, int64_t name_cap, float* ms, int64_t* dims6);

/* ---------------------------------------------------------------------------------------------
 * Mini-batch assembly from an HBM-resident graph store.
 * Replaces: the per-sample `augment_edge` transform (dataset/utils.py:89-141, installed at
 * dataset/code.py:97-101) and PyG `Batch.from_data_list` behind the DataLoader (main.py:149-152).
 *
 * Store = per-graph slices of concatenated int64 arrays: node_ptr/edge_ptr [G+1]; x [Ns][x_cols];
 * node_depth [Ns] (optional); edge_src/edge_dst [Es] with node ids LOCAL to their graph; edge_attr
 * [Es][ea_cols] (optional); y [G][y_row_bytes] raw label rows (optional).  attr_rank [Ns+1] (optional) is
 * the exclusive prefix sum of (node_is_attributed == 1) over all store nodes, built once by
 * gt_attr_rank; when present the batch is AUGMENTED: per graph the output edges are
 * [ast, ast^-1, next-token, next-token^-1] and edge_attr_f32 [E][2] = (is next-token, is inverse),
 * exactly the reference's edge order and values; stored edge_attr is then ignored.
 *
 * gt_collate: graph_ids [B] (device) -> x [N][x_cols], node_depth [N], batch [N], ptr [B+1] (optional),
 * edge_index [2][E] (global ids), edge_attr_f32 or edge_attr_i64 [E][ea_cols], y [B][y_row_bytes].
 * N and E are the totals over the selected graphs (the caller owns the allocations and knows the
 * per-graph sizes: E_i = e_i, or 2 e_i + 2 max(a_i - 1, 0) when augmenting); graphs that would not
 * fit the stated N / E are skipped rather than written out of bounds.  Integer-exact.
 */
typedef struct gt_graph_store {
  const int64_t* node_ptr;
  const int64_t* edge_ptr;
  const int64_t* x;
  const int64_t* node_depth;
  const int64_t* edge_src;
  const int64_t* edge_dst;
  const int64_t* edge_attr;
  const int64_t* attr_rank;
  const void* y;
  int64_t y_row_bytes;
  int64_t num_graphs;
  int32_t x_cols;
  int32_t ea_cols;
} gt_graph_store;

typedef struct gt_collate_out {
  int64_t* x;
  int64_t* node_depth;   /* optional */
  int64_t* batch;
  int64_t* ptr;          /* optional */
  int64_t* edge_index;
  float* edge_attr_f32;  /* augmenting stores */
  int64_t* edge_attr_i64; /* optional, plain stores */
  void* y;               /* optional */
} gt_collate_out;

int gt_attr_rank(const int64_t* node_is_attributed, int64_t num_nodes, int64_t* rank, gt_stream_t stream);
size_t gt_collate_workspace_bytes(int64_t num_graphs);
int gt_collate(const gt_graph_store* store, const int64_t* graph_ids, int64_t num_graphs, int64_t num_nodes,
               int64_t num_edges, const gt_collate_out* out, void* workspace, size_t workspace_bytes,
               gt_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Graph structure, built once per collated batch.
 * Replaces: the per-layer `degree(row, N)` (modules/conv.py:57, PyG degree -> scatter_add) and the
 * unsorted atomic scatter of torch-scatter (conv.py:28,63 via MessagePassing.aggregate) with a
 * destination-sorted CSR / source-sorted CSC that is bit-identical to a stable sort of edge_index
 * (oracle/graph_struct.py); also the `batch.eq(i)` loop of pad_batch (modules/utils.py:9-13).
 *
 *   edge_index [2][E] int64 (row 0 = source `row`, row 1 = destination `col`), batch [N] int64
 *   sorted non-decreasing.  Outputs:
 *   graph_ptr[B+1]; node_graph[N] (= batch as int32); in_ptr[N+1], in_src[E], in_eid[E] (CSR by destination, ties in edge order);
 *   out_ptr[N+1], out_dst[E], out_eid[E] (CSC by source); deg[N] = 1 + out-degree (float, the
 *   GCN `deg`, conv.py:57); dis[N] = deg^-1/2 (conv.py:58).
 *   status[0] is set non-zero on device if an index is out of range (checked by the caller).
 */
size_t gt_graph_prep_workspace_bytes(int64_t num_nodes, int64_t num_edges, int64_t num_graphs);
int gt_graph_prep(const int64_t* edge_index, const int64_t* batch, int64_t num_nodes, int64_t num_edges,
                  int64_t num_graphs, int32_t* graph_ptr, int32_t* node_graph, int32_t* in_ptr, int32_t* in_src,
                  int32_t* in_eid,
                  int32_t* out_ptr, int32_t* out_dst, int32_t* out_eid, float* deg, float* dis,
                  int32_t* status, void* workspace, size_t workspace_bytes, gt_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused message + aggregate + update of GCNConv / GINConv.
 * Replaces (per layer): index_select x[row], edge_encoder GEMM / embedding gathers, add, relu,
 * mul-by-norm, torch_scatter.scatter(sum), and the self term (modules/conv.py:26-36, :50-71).
 *
 *   GCN: out[v] = dis[v] * sum_{k in in(v)} dis[src_k] * relu(h[src_k] + e_k) + relu(h[v] + root) / deg[v]
 *   GIN: out[v] = (1 + eps) * h[v] + sum_{k in in(v)} relu(h[src_k] + e_k)
 *   e_k by edge_mode: 0 | sum_j attr[k][j] * W[:, j] + b | sum_j T[tab_off[j] + attr[k][j]] | dense[k]
 *   edge_attr is indexed by ORIGINAL edge id (in_eid); GT_EDGE_LINEAR: float attr[E][K], W [D][K]
 *   (nn.Linear.weight layout), b [D]; GT_EDGE_TABLES: int64 attr[E][K], tables [rows][D],
 *   tab_off[K] int32 (host pointer); GT_EDGE_DENSE: dense [E][D] (same dtype as h).
 *   `self_param`: GCN root_emb.weight [D] (float32); GIN eps [1] (float32).  Parameters are fp32.
 */
int gt_aggregate_fwd(int conv, int edge_mode, int dtype, const void* h, int64_t num_nodes, int64_t num_edges,
                     int64_t dim, const int32_t* in_ptr, const int32_t* in_src, const int32_t* in_eid,
                     const float* deg, const float* dis, const float* self_param, const void* edge_attr,
                     int64_t edge_attr_cols, const float* edge_w, const float* edge_b,
                     const int32_t* tab_off_host, int64_t table_rows /* total rows of the tables, 0 = unknown */,
                     const void* edge_dense, void* out, gt_stream_t stream);

/* Backward of gt_aggregate_fwd (autograd of the ops above in the reference).
 *   t_k   = w_k * 1[h[row_k] + e_k > 0] * g[col_k]            (w_k = dis[row_k] dis[col_k] | 1)
 *   dh[u] = sum_{k in out(u)} t_k + (GCN: 1[h[u]+root>0] g[u]/deg[u] | GIN: (1+eps) g[u])
 *   GT_EDGE_LINEAR: d_edge_w[D][K] = sum_k t_k (x) attr_k, d_edge_b[D] = sum_k t_k
 *   GT_EDGE_TABLES: d_tables[rows][D]: row r gets sum of t_k over edges/columns that index it
 *   GT_EDGE_DENSE : d_dense[E][D] = t_k (original edge order)
 *   GCN: d_self[D] = sum_u 1[h[u]+root>0] g[u]/deg[u];  GIN: d_self[0] = sum_{u,c} g[u][c] h[u][c]
 *   (GIN: d_self must hold 1 + ceil(D/64) floats; entries past [0] are scratch)
 * Parameter gradients are fp32 and OVERWRITTEN (not accumulated); reduction order is fixed
 * (deterministic).  `table_rows` = total rows of `tables`.
 */
size_t gt_aggregate_bwd_workspace_bytes(int conv, int edge_mode, int64_t dim, int64_t edge_attr_cols,
                                        int64_t table_rows);
int gt_aggregate_bwd(int conv, int edge_mode, int dtype, const void* h, const void* grad_out, int64_t num_nodes,
                     int64_t num_edges, int64_t dim, const int32_t* out_ptr, const int32_t* out_dst,
                     const int32_t* out_eid, const float* deg, const float* dis, const float* self_param,
                     const void* edge_attr, int64_t edge_attr_cols, const float* edge_w, const float* edge_b,
                     const int32_t* tab_off_host, int64_t table_rows, const void* edge_dense, void* grad_h,
                     float* d_self, float* d_edge_w, float* d_edge_b, void* d_dense, void* workspace,
                     size_t workspace_bytes, gt_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * PNA multi-aggregator message passing (mean | max | min | std over the in-edges in one pass).
 * Replaces PyG PNAConv's per-edge Linear + 4 torch_scatter reductions (modules/pna/pna_module.py:73;
 * math stated in-tree by modules/pna_layer.py:131-167, modules/pna/aggregators.py:11-34).  The
 * per-edge message pre_nn_t([x_i || x_j]) is split into per-node terms m_k = U[i] + V[j_k]
 * (U = x A_t^T + b, V = x B_t^T computed by the caller):
 *   out[i][t][0F..1F) = U + mean_k V[j_k]   [1F..2F) = U + max_k V[j_k]   [2F..3F) = U + min_k V[j_k]
 *   out[i][t][3F..4F) = sqrt(relu(E[V^2] - E[V]^2) + 1e-5);   empty neighbourhood: 0, 0, 0, sqrt(1e-5)
 * with F = dim / towers (tower-major layout, the A-operand of the post Linear).  fp32.
 * mean_v [N][dim] and arg [N][2][dim] (original edge ids of the first max / min) are saved for the
 * backward, which returns dU and dV.  Deterministic (CSR / CSC traversal, no atomics).
 */
int gt_pna_aggregate_fwd(const float* U, const float* V, int64_t num_nodes, int64_t dim, int towers,
                         const int32_t* in_ptr, const int32_t* in_src, const int32_t* in_eid, float* out,
                         float* mean_v, int32_t* arg, gt_stream_t stream);
int gt_pna_aggregate_bwd(const float* V, const float* out, const float* mean_v, const int32_t* arg,
                         const float* grad_out, int64_t num_nodes, int64_t dim, int towers, const int32_t* in_ptr,
                         const int32_t* out_ptr, const int32_t* out_dst, const int32_t* out_eid, float* dU, float* dV,
                         gt_stream_t stream);
/* Degree scalers of PNAConv (modules/pna/scalers.py:10-31) applied to the S per-scaler output blocks of the
 * post-Linear: out[n][t][f] = sum_s Y[n][t][s][f] * scales[n][s]; bwd: dY = grad_out (x) scales (scales: no grad). */
int gt_scale_combine_fwd(const float* Y, const float* scales, int64_t num_nodes, int towers, int num_scalers, int F,
                         float* out, gt_stream_t stream);
int gt_scale_combine_bwd(const float* grad_out, const float* scales, int64_t num_nodes, int towers, int num_scalers, int F,
                         float* dY, gt_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Input node encoders: out[n] = sum_t table_t[min(idx_t[n * stride_t], clamp_t)]  (clamp_t < 0: none).
 *   Code2: ASTNodeEncoder.forward dataset/utils.py:28-30 (type + attribute + clamped depth);
 *   Molpcba: ogb AtomEncoder (dataset/mol.py:83), one table per categorical column.
 * Up to 16 tables; the *_host arrays are HOST arrays of length num_tables (device pointers inside).
 * Backward: d_table_t[r] = sum_{n: idx_t[n] = r} grad_out[n], deterministic (64-bit fixed-point
 * integer atomics scaled by 2^30 / max|grad_out|, see csrc/embed.hip); d_tables_host[t] may be NULL.
 */
int gt_embed_sum_fwd(int num_tables, const int64_t* const* idx_ptrs_host, const int64_t* idx_strides_host,
                     const int64_t* clamp_max_host, const float* const* tables_host, int64_t num_nodes, int64_t dim,
                     float* out, gt_stream_t stream);
size_t gt_embed_sum_bwd_workspace_bytes(int num_tables, const int64_t* table_rows_host, int64_t dim);
int gt_embed_sum_bwd(int num_tables, const int64_t* const* idx_ptrs_host, const int64_t* idx_strides_host,
                     const int64_t* clamp_max_host, const int64_t* table_rows_host, const float* grad_out,
                     int64_t num_nodes, int64_t dim, float* const* d_tables_host, void* workspace,
                     size_t workspace_bytes, gt_stream_t stream);
/* Sorted variant (the default of the Python layers): gt_embed_sort, once per batch, orders the node ids of every
 * table by row (stable counting sort; tables of up to 16384 rows) into `plan` (gt_embed_sort_plan_bytes, kept
 * until the backward); gt_embed_sum_bwd_sorted then sums the gradient rows of every table row in node order with
 * plain fp32 adds and no atomics: bitwise reproducible, no fixed-point rounding, insensitive to skewed
 * vocabularies.  Index values outside [0, rows) are dropped. */
size_t gt_embed_sort_plan_bytes(int num_tables, const int64_t* table_rows_host, int64_t num_nodes);
size_t gt_embed_sort_workspace_bytes(int num_tables, const int64_t* table_rows_host, int64_t num_nodes);
int gt_embed_sort(int num_tables, const int64_t* const* idx_ptrs_host, const int64_t* idx_strides_host,
                  const int64_t* clamp_max_host, const int64_t* table_rows_host, int64_t num_nodes, void* plan,
                  size_t plan_bytes, void* workspace, size_t workspace_bytes, gt_stream_t stream);
size_t gt_embed_sum_bwd_sorted_workspace_bytes(int num_tables, int64_t num_nodes, int64_t dim);
int gt_embed_sum_bwd_sorted(int num_tables, const int64_t* table_rows_host, const float* grad_out, int64_t num_nodes,
                            int64_t dim, const void* plan, float* const* d_tables_host, void* workspace,
                            size_t workspace_bytes, gt_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Per-graph segment ops on the sorted `batch` vector (virtual node + pooling).
 * gt_segment_bcast_add: out[n] = (x ? x[n] : 0) + seg[g(n)]      -- `h + vn[batch]`
 *   (modules/gnn_module.py:199); also the backward of global_add_pool.
 * gt_segment_sum:       out[g] = (add ? add[g] : 0) + sum_{n in g} x[n]   -- global_add_pool + vn
 *   (modules/gnn_module.py:219, PyG global_add_pool -> torch_scatter); also d(vn) of the bcast.
 */
int gt_segment_bcast_add(int dtype, const void* x, const void* seg, const int32_t* node_graph, int64_t num_nodes,
                         int64_t num_graphs, int64_t dim, void* out, gt_stream_t stream);
int gt_segment_sum(int dtype, const void* x, const void* add, const int32_t* graph_ptr, int64_t num_nodes,
                   int64_t num_graphs, int64_t dim, void* out, gt_stream_t stream);
/* Same result class (fixed summation order, different association) with a workspace: load-balanced over ROW chunks
 * instead of one block per graph, for batches whose largest graph would otherwise be the tail of the kernel.
 * Falls back to gt_segment_sum for small inputs or a missing workspace. */
size_t gt_segment_sum_workspace_bytes(int64_t num_nodes, int64_t dim);
int gt_segment_sum_ws(int dtype, const void* x, const void* add, const int32_t* graph_ptr, int64_t num_nodes,
                      int64_t num_graphs, int64_t dim, void* out, void* workspace, size_t workspace_bytes,
                      gt_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Sequence layout: flat node rows <-> transformer token rows.
 * Replaces pad_batch / unpad_batch (modules/utils.py:5-53) and the CLS concat
 * (modules/transformer_encoder.py:50-55).  A layout is described by seq_desc[B][4] int32 =
 * {row0, npos, kv_off, kv_len}: position p of sequence b lives at token row row0 + p*row_stride;
 * positions [kv_off, kv_off+kv_len) are real (node or CLS) tokens, the rest is padding.
 *   padded (reference) layout: row0 = b, row_stride = B, npos = S(+1), kv_off = npos - kv_len
 *   packed (fast) layout     : row0 = tok_ptr[b], row_stride = 1, npos = kv_len, kv_off = 0
 * gt_seq_gather: tokens[row(b,p)] = node row (graph_ptr[b+1] - kept_b + j) for the j-th kept node,
 *   `cls` [dim] for the CLS position (if with_cls), 0 for padding.  kept_b = kv_len - with_cls.
 * gt_seq_scatter: the adjoint (grad_h[N][dim] zero for truncated nodes; d_cls[B][dim] per-graph,
 *   reduced by the caller) and also unpad_batch when `base` (N rows kept for truncated nodes) is given.
 */
int gt_seq_gather(int dtype, const void* h, const void* cls, const int32_t* graph_ptr, const int32_t* seq_desc,
                  int64_t num_seqs, int64_t row_stride, int64_t max_npos, int with_cls, int64_t dim,
                  void* tokens, uint8_t* pad_mask /* [B][max_npos] or NULL */, gt_stream_t stream);
int gt_seq_scatter(int dtype, const void* tokens, const void* base, const int32_t* graph_ptr,
                   const int32_t* node_graph, const int32_t* seq_desc, int64_t num_seqs, int64_t row_stride,
                   int with_cls, int64_t num_nodes, int64_t dim, void* h_out, void* cls_out /* [B][dim] or NULL */,
                   gt_stream_t stream);
/* The PACKED token layout (seq_desc / last token row per sequence / attention work list of {sequence, 64-row tile}) built
 * on the device from graph_ptr alone: for batches whose per-graph sizes are not known on the host, where the reference's
 * pad_batch (modules/utils.py:9-16) synchronises B times.  meta[4] = {rows, num_work, max kv_len, S}; work entries past
 * num_work are {-1,-1}.  Callers size their launches by rows <= N + B*with_cls and num_work <= B + rows/64. */
int gt_seq_layout_packed(const int32_t* graph_ptr, int64_t B, int64_t max_input_len, int with_cls, int32_t* seq_desc,
                         int64_t* last_rows, int32_t* work_items, int64_t work_capacity, int32_t* meta, gt_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused masked multi-head self-attention over the token rows (flash-style, never materialises
 * the S x S scores).  Replaces F.multi_head_attention_forward's bmm / masked_fill(-inf) / softmax
 * / dropout / bmm (modules/transformer_encoder.py:59) between in_proj and out_proj.
 *   qkv [rows][3*d_model] (torch packed in_proj order q|k|v), ctx [rows][d_model],
 *   lse [2][nhead][rows] fp32 (row max and log2 of the row sum, saved for backward).
 *   head_dim = d_model/nhead in {8, 16, 32, 64}.
 *   Keys outside [kv_off, kv_off+kv_len) are masked (the key_padding_mask); every query position
 *   in [0, npos) is computed.  P = softmax(scale * q k^T); dropout(P, p) with a counter-based RNG
 *   keyed by (seed, seq, head, query, key) so backward replays it.
 *   dtype GT_BF16 -> v_mfma_f32_16x16x32_bf16; GT_F32 -> v_mfma_f32_16x16x4_f32 (exact fp32).
 *   work_items (optional) [num_work][2] int32 = {sequence, 64-position tile}: when given, the grid
 *   covers exactly these tiles (ragged graph sizes: most sequences need 2 of the max_npos/64 tiles);
 *   NULL -> dense grid (max tiles x heads x sequences, empty tiles exit).
 *   dense_mask [num_seqs][npos][npos] / key_valid [num_seqs][npos] (optional, fp32; every sequence
 *   must have npos == max_npos): the masks of CausalSelfAttention (modules/masked_transformer_encoder.py
 *   :44-47) — scores whose mask entry is 0 are FILLED with the finite `mask_value` (masked_fill
 *   semantics: a fully masked row becomes uniform, no gradient flows through a filled score).
 */
int gt_attn_fwd(int dtype, const void* qkv, void* ctx, float* lse, int64_t total_rows, int64_t d_model, int nhead,
                const int32_t* seq_desc, int64_t num_seqs, int64_t row_stride, int64_t max_npos,
                const int32_t* work_items, int64_t num_work, const float* dense_mask, const float* key_valid,
                float mask_value, float scale, float dropout_p, uint64_t seed, gt_stream_t stream);
/* d_qkv [rows][3*d_model] is fully written for every row belonging to a sequence position.
 * delta [nhead][rows] fp32 workspace (rowsum(dO*O)). */
int gt_attn_bwd(int dtype, const void* qkv, const void* ctx, const void* d_ctx, const float* lse, float* delta,
                void* d_qkv, int64_t total_rows, int64_t d_model, int nhead, const int32_t* seq_desc,
                int64_t num_seqs, int64_t row_stride, int64_t max_npos, const int32_t* work_items, int64_t num_work,
                const float* dense_mask, const float* key_valid, float mask_value, float scale, float dropout_p,
                uint64_t seed, gt_stream_t stream);
/* Pooled mode: after the LAST encoder layer only one row per sequence is read (cls / last pooling, models/gnn_transformer.py:113-114):
 * gt_attn_fwd_last computes the 64-row tile that holds the LAST position of every sequence only (ctx / lse rows outside those tiles
 * are not written); gt_attn_bwd_last is its backward for a d_ctx that is non-zero in those last positions only: dQ for the last
 * tiles, dK / dV for every row; d_qkv must be ZERO-FILLED by the caller; work_items = the full work list (key tiles). */
int gt_attn_fwd_last(int dtype, const void* qkv, void* ctx, float* lse, int64_t total_rows, int64_t d_model, int nhead,
                     const int32_t* seq_desc, int64_t num_seqs, int64_t row_stride, int64_t max_npos, float scale, float dropout_p,
                     uint64_t seed, gt_stream_t stream);
int gt_attn_bwd_last(int dtype, const void* qkv, const void* ctx, const void* d_ctx, const float* lse, float* delta, void* d_qkv,
                     int64_t total_rows, int64_t d_model, int nhead, const int32_t* seq_desc, int64_t num_seqs, int64_t row_stride,
                     int64_t max_npos, const int32_t* work_items, int64_t num_work, float scale, float dropout_p, uint64_t seed,
                     gt_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * BatchNorm1d over the rows of an [rows][dim] matrix (channels = columns), optional fused ReLU.
 * Replaces nn.BatchNorm1d (+ F.relu) at modules/gnn_module.py:84-90,204-209 (node BN),
 * :161-170 (virtual-node MLP) and modules/conv.py:18-20 (GIN mlp).  torch semantics: training ->
 * batch mean / biased variance, running stats updated with `momentum` (unbiased variance) and
 * num_batches_tracked += 1; eval -> running stats.  weight/bias/stat buffers are fp32.
 * save_mean/save_rstd [dim] are written for backward.  Deterministic (fixed reduction order).
 * dropout_p > 0 (training only) fuses the F.dropout that follows the layer BatchNorm in the GNN
 * (gnn_module.py:88-90,209-212,222): y = drop(bn(x)[relu]) [+ resid]; the mask is a counter hash of
 * (seed, row, column) replayed by the backward (pass the same dropout_p and seed).
 */
size_t gt_batchnorm_workspace_bytes(int64_t rows, int64_t dim);
int gt_batchnorm_fwd(int dtype, const void* x, const float* weight, const float* bias, float* running_mean,
                     float* running_var, int64_t* num_batches_tracked, float momentum, float eps, int training,
                     int relu, const void* resid /* optional: y = bn(x)[relu] + resid */, int64_t rows, int64_t dim,
                     void* y, float* save_mean, float* save_rstd, float dropout_p, uint64_t seed, void* workspace,
                     size_t workspace_bytes, gt_stream_t stream);
/* The same with `bcast[bcast_index[row]]` (rows of a [segments][dim] matrix, e.g. the NEXT layer's virtual-node
 * embedding per graph: h_list[l+1] + vn[batch], modules/gnn_module.py:199) added to every output row in the apply
 * pass.  ev_bcast_ready: optional gt_event the stream waits for before that pass (bcast produced on another stream). */
int gt_batchnorm_fwd_bcast(int dtype, const void* x, const float* weight, const float* bias, float* running_mean,
                           float* running_var, int64_t* num_batches_tracked, float momentum, float eps, int training,
                           int relu, const void* resid, const void* bcast, const int32_t* bcast_index, void* ev_bcast_ready,
                           int64_t rows, int64_t dim, void* y, float* save_mean, float* save_rstd, float dropout_p,
                           uint64_t seed, void* workspace, size_t workspace_bytes, gt_stream_t stream);
/* dx w.r.t. the BN input (the residual branch's gradient is dy itself).  The ReLU gate is recomputed
 * from x, the saved statistics, weight and bias: the forward output is not needed. */
int gt_batchnorm_bwd(int dtype, const void* x, const void* dy, const float* weight, const float* bias,
                     const float* save_mean, const float* save_rstd, int training, int relu, int64_t rows,
                     int64_t dim, void* dx, float* dweight, float* dbias, float dropout_p, uint64_t seed, void* workspace,
                     size_t workspace_bytes, gt_stream_t stream);
/* gt_batchnorm_bwd with its first pass done elsewhere: part[nparts][2][dim] = partial sums of dy' and dy' * xhat over any
 * partition of the rows (gt_linear_bwd_bnstats writes them from the dX epilogue of the GEMM that produces dy); fixed-order
 * finish + apply only.  For layers without dropout behind the BatchNorm. */
int gt_batchnorm_bwd_parts(int dtype, const void* x, const void* dy, const float* weight, const float* bias,
                           const float* save_mean, const float* save_rstd, int training, int relu, int64_t rows, int64_t dim,
                           void* dx, float* dweight, float* dbias, const float* part, int64_t nparts, gt_stream_t stream);
/* Synchronised BatchNorm statistics across data-parallel ranks (the reference normalises over its single-device
 * batch: modules/gnn_module.py:204,164,167, modules/conv.py:19).  While a hook is set for the calling host thread, every
 * training-mode gt_batchnorm_fwd* / gt_batchnorm_bwd -- including the ones inside gt_gcn_layer_*, gt_gin_layer_*, gt_vn_update_* --
 * calls it between its local pass and its apply pass:
 *   kind 0: buf[0..n) = {rows, mean[D], biased var[D]} (n = 2D+1); the hook all-gathers the ranks' n floats to buf[n .. n+world*n)
 *   kind 1: buf[0..n) = {sum dy'[D], sum dy' xhat[D], rows} (n = 2D+1); the hook all-reduces (sum) in place
 * on `stream` (stream-ordered, no host sync required); returns 0 or an error.  The collective itself is the caller's. */
typedef int (*gt_bn_sync_fn)(void* user, int kind, float* buf, int64_t n, gt_stream_t stream);
int gt_bn_sync_set(gt_bn_sync_fn fn, void* user, int world);
/* The apply passes on their own, for BatchNorm statistics synchronised over data-parallel ranks (the reference
 * normalises over the whole single-device batch, modules/gnn_module.py:204): y = drop(bn(x; mean, rstd) [relu]) [+ resid]
 * with caller-provided statistics; dx from caller-provided (all-rank) sums of dy' and dy' * xhat over `count` rows. */
int gt_batchnorm_apply(int dtype, const void* x, const float* mean, const float* rstd, const float* weight,
                       const float* bias, int relu, const void* resid, int64_t rows, int64_t dim, void* y, float dropout_p,
                       uint64_t seed, gt_stream_t stream);
int gt_batchnorm_bwd_apply(int dtype, const void* x, const void* dy, const float* weight, const float* bias,
                           const float* mean, const float* rstd, const float* sum_dy, const float* sum_dy_xhat, double count,
                           int relu, int64_t rows, int64_t dim, void* dx, float dropout_p, uint64_t seed, gt_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * y = LayerNorm(resid + dropout(x)) over the last dim of [rows][dim] token rows (dim <= 1024).
 * Replaces the `x = norm(x + dropout(sublayer(x)))` pairs of nn.TransformerEncoderLayer and the
 * norm_input / final LayerNorm (modules/transformer_encoder.py:28-32,56-59).  resid may be NULL
 * (plain LayerNorm, dropout_p = 0).  Dropout is a counter hash of (seed, row, column), replayed
 * by the backward.  save_mean/save_rstd [rows] fp32.
 * Backward: dresid = d(resid + dropout(x)), dx = dresid through the dropout mask; either may be
 * NULL.  dweight/dbias [dim] fp32, deterministic.
 */
int gt_layernorm_fwd(int dtype, const void* x, const void* resid, const float* weight, const float* bias, float eps,
                     float dropout_p, uint64_t seed, int64_t rows, int64_t dim, void* y, float* save_mean,
                     float* save_rstd, gt_stream_t stream);
size_t gt_layernorm_bwd_workspace_bytes(int64_t rows, int64_t dim);
int gt_layernorm_bwd(int dtype, const void* x, const void* resid, const void* dy, const float* weight,
                     const float* save_mean, const float* save_rstd, float dropout_p, uint64_t seed, int64_t rows,
                     int64_t dim, void* dx, void* dresid, float* dweight, float* dbias, void* workspace,
                     size_t workspace_bytes, gt_stream_t stream);
/* the column finish over `nblk` block partials [2][dim] (dweight partial | dbias partial per block), gt_layernorm_bwd's own fixed order */
int gt_layernorm_bwd_finish(const float* part, int nblk, int64_t dim, float* dweight, float* dbias, gt_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * nn.Linear on the matrix cores with fused bias / ReLU / dropout, forward and backward.
 * Replaces F.linear (+ relu, dropout, bias-grad reductions, dtype casts) at modules/conv.py:44,51
 * (GCN linear), conv.py:18-20 (GIN mlp), modules/gnn_module.py:161-170 (virtual-node MLP),
 * models/gnn_transformer.py:69-70,92 (gnn2transformer) and inside nn.TransformerEncoderLayer
 * (in_proj, out_proj, linear1+activation+dropout, linear2; modules/transformer_encoder.py:28-32).
 *   fwd: y[M][N] = dropout(act(x[M][K] weight[N][K]^T + bias))      act: 0 none | 1 relu
 *   bwd: dz = dy * 1[y > 0] / (1 - p) when y_for_mask != NULL (the forward fused relu[/dropout]),
 *        else dz = dy;  dx = dz weight [+ dx_add1 + dx_add2];  dweight = dz^T x;  dbias = colsum(dz)
 *        (any output may be NULL; dx_add* are optional [M][K] addends in x's storage type)
 * weight/bias and their gradients are fp32 (master weights are converted while staging: no cast
 * pass).  x_dtype / y_dtype: storage of x (and dx) / y (and dy); compute: GT_BF16
 * (v_mfma_f32_16x16x32_bf16) or GT_F32 (v_mfma_f32_16x16x4_f32, needs fp32 storage).
 * N % 4 == 0 and K % 4 == 0.  Fused dropout requires act == relu.  Deterministic.
 * dX of a long contraction with few output tiles (N >= 2048) is split over N into fp32 partials.
 */
int gt_linear_fwd(int x_dtype, int y_dtype, int compute, const void* x, const float* weight, const float* bias,
                  void* y, int64_t M, int64_t N, int64_t K, int act, float dropout_p, uint64_t seed,
                  gt_stream_t stream);
/* Same, with an explicit row stride ldy >= N for y / dy / y_for_mask (ldy % 4 == 0, N arbitrary): the
 * stacked 5 x 5002-way prediction heads (models/gnn_transformer.py:124-126) write N = 25010 logits per
 * graph into rows of 25012.  Columns N..ldy of y are unspecified after the forward; the backward reads
 * them in dy (multiplied by zero weights), so they must hold finite values. */
int gt_linear_fwd_ld(int x_dtype, int y_dtype, int compute, const void* x, const float* weight, const float* bias,
                     void* y, int64_t M, int64_t N, int64_t K, int64_t ldy, int act, float dropout_p, uint64_t seed,
                     gt_stream_t stream);
/* ... and with an explicit row stride ldx >= K for x / dx / the dx addends as well: column slices of a wider
 * matrix in, column slices out (the per-tower linears of PNAConv, modules/pna/pna_module.py:33-41). */
int gt_linear_fwd_ld2(int x_dtype, int y_dtype, int compute, const void* x, const float* weight, const float* bias,
                      void* y, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldy, int act, float dropout_p,
                      uint64_t seed, gt_stream_t stream);
size_t gt_linear_bwd_workspace_bytes(int compute, int64_t M, int64_t N, int64_t K);
int gt_linear_bwd(int x_dtype, int y_dtype, int compute, const void* x, const float* weight, const void* dy,
                  const void* y_for_mask, const void* dx_add1, const void* dx_add2, void* dx, float* dweight,
                  float* dbias, int64_t M, int64_t N, int64_t K, float dropout_p, void* workspace,
                  size_t workspace_bytes, gt_stream_t stream);
int gt_linear_bwd_ld(int x_dtype, int y_dtype, int compute, const void* x, const float* weight, const void* dy,
                     const void* y_for_mask, const void* dx_add1, const void* dx_add2, void* dx, float* dweight,
                     float* dbias, int64_t M, int64_t N, int64_t K, int64_t ldy, float dropout_p, void* workspace,
                     size_t workspace_bytes, gt_stream_t stream);

/* GELU (erf form: transformer_activation=gelu modules/transformer_encoder.py:17; the Block MLP of
 * masked_transformer_encoder.py:68) fused into the producing GEMM: Y = dropout(gelu(X W^T + b)).  `gmul` (nullable;
 * storage type and pitch of Y) receives gelu'(z) * dropout scale; gt_linear_bwd_mul takes it where the ReLU path takes
 * the forward output: dZ = dY * gmul, then dX / dW / db as gt_linear_bwd_ld2. */
int gt_linear_fwd_gelu(int x_dtype, int y_dtype, int compute, const void* x, const float* weight, const float* bias,
                       void* y, void* gmul, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldy, float dropout_p,
                       uint64_t seed, gt_stream_t stream);
int gt_linear_bwd_mul(int x_dtype, int y_dtype, int compute, const void* x, const float* weight, const void* dy,
                      const void* gmul, const void* dx_add1, const void* dx_add2, void* dx, float* dweight, float* dbias,
                      int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldy, void* workspace, size_t workspace_bytes,
                      gt_stream_t stream);

/* Weight gradients are off the critical path of a backward pass (only the optimizer reads them).
 * Between gt_overlap_dw_begin(main, side) and gt_overlap_dw_end() on the same host thread, every
 * gt_linear_bwd[_ld] issued on `main` that computes both dX and dW launches the dW part (and its
 * partial reduce) on `side`, ordered after everything already queued on `main`.  The caller must call
 * gt_overlap_dw_sync() (main waits for the side stream) before it overwrites a dy / x / workspace
 * buffer that such a call was given, and before it reads the weight gradients; _end syncs too.
 * gt_overlap_dw_release(workspace, bytes) is the finer join: main waits only for the forked dW GEMMs whose
 * workspace overlaps the given range (and, the side stream being in order, for those forked before them) --
 * a caller that alternates two workspaces between consecutive layers lets layer k's dW run beside layer k+1
 * and releases its workspace before layer k+2 (buffers a dW reads must live in that workspace or stay
 * unchanged until the next full sync). */
int gt_overlap_dw_begin(gt_stream_t main_stream, gt_stream_t side_stream);
/* BatchNorm-backward statistics in a dX epilogue.  When the dX of a GEMM is the dy of a BatchNorm further down the backward
 * pass (x_l = relu(BN(agg_{l-1})) feeds conv_l: gnn_module.py:199-212), gt_linear_bwd_bnstats(...) before the
 * gt_linear_bwd* call makes its epilogue also write part[gt_linear_bwd_bnstats_rows(M)][2][K]: per 64-row tile the sums of
 * dy' and dy' * xhat (dy' = dy gated by the ReLU behind that BatchNorm) -- gt_batchnorm_bwd_parts then skips its own pass
 * over dy and the BatchNorm input.  Only where gt_linear_bwd_bnstats_ok(...) says so (exact-fp32 path, fp32 rows, M >= 1024). */
int gt_linear_bwd_bnstats_ok(int compute, int x_dtype, int y_dtype, int64_t M);
int64_t gt_linear_bwd_bnstats_rows(int64_t M);
/* ... under the current image bindings of this thread: ceil(M / 128) when the register-row bf16x6 kernel (csrc/linear3r.h) takes the
 * call (bound image of W^T, fp32 rows, M >= 12288), else gt_linear_bwd_bnstats_rows(M); 0 = unsupported */
int64_t gt_linear_bwd_bnstats_rows_for(int compute, int x_dtype, int y_dtype, const float* weight, int64_t M, int64_t N, int64_t K);
int gt_linear_bwd_bnstats(const float* bn_x, int64_t ldx, const float* mean, const float* rstd, const float* w, const float* b,
                          int relu, float* part);
/* Broadcast addend of the NEXT gt_linear_bwd* call of this host thread: dx[m] += rows[idx[m]] (rows [.][K] fp32 at the dX pitch,
 * idx int32 [M]) in the dX GEMM's epilogue -- the virtual-node update's gradient d_t0[batch[m]] (modules/gnn_module.py:219)
 * without writing it out per node.  Only fp32 GEMMs with M >= 12288 on a bound W^T image take it: ask gt_linear_bwd_bcast_ok
 * (1 = yes); a call that cannot honour a pending request fails with GT_ERR_UNSUPPORTED.  Dropped after the next call. */
int gt_linear_bwd_bcast_ok(int compute, int x_dtype, int y_dtype, const float* weight, int64_t M, int64_t N, int64_t K);
int gt_linear_bwd_bcast(const float* rows, const int32_t* idx);
/* gt_linear_bwd with W^T [K][N] (fp32, gt_transpose) supplied by the caller, NULL = none: the exact-fp32 dX GEMM runs on the
 * transposed weight and otherwise transposes it in front of every call. */
int gt_linear_bwd_wt(int x_dtype, int y_dtype, int compute, const void* x, const float* weight, const float* weight_t,
                     const void* dy, const void* y_for_mask, const void* dx_add1, const void* dx_add2, void* dx,
                     float* dweight, float* dbias, int64_t M, int64_t N, int64_t K, float dropout_p, void* workspace,
                     size_t workspace_bytes, gt_stream_t stream);
int gt_transpose(const float* in /* [N][K] */, float* out /* [K][N] */, int64_t N, int64_t K, gt_stream_t stream);

/* ---- fp32-accurate GEMMs on the bf16 matrix pipe ("bf16x6", csrc/linear3x.h) -------------------------------------------
 * The reference's Linear layers are fp32 (torch.nn.Linear: modules/conv.py:44,51, modules/gnn_module.py:161-170,
 * models/gnn_transformer.py:69-70).  An fp32 value is EXACTLY the sum of three bf16 values; keeping the six operand-plane
 * products down to relative 2^-16 reproduces the fp32 GEMM to accumulation order at 2.7 x the fp32-MFMA ceiling.  A weight is
 * split once per optimizer step into an "image" (bf16 planes in LDS order); bound images reroute the big-M exact-fp32 GEMMs of
 * gt_linear_fwd* / gt_linear_bwd* (compute == GT_F32, M >= 1024, one group) to the bf16x6 kernel. */
/* JK = "cat" without its copy (torch.cat([h_list[0], h_list[-1]], 1), modules/gnn_module.py:104-105, feeding gnn2transformer,
 * models/gnn_transformer.py:92): the GEMM reads its row operand from two matrices side by side and its backward writes the two
 * gradients where their consumers read them.  Only on weights whose images are bound (ask gt_linear_cat2_ok first). */
int gt_linear_cat2_ok(int compute, const float* weight, int64_t M, int64_t N, int64_t K1, int64_t K2);
/* A row map for the NEXT gt_linear_fwd* / gt_linear_bwd* call of this host thread (consumed by that call whatever its outcome):
 * forward stores output row m at row rows[m] of y, backward reads row m of dY from row rows[m] of dy; rows[m] = -1: no such row
 * (nothing stored / zeros read).  int32 [M] on the device.  This is gnn2transformer writing / reading the Transformer's token rows in
 * place (models/gnn_transformer.py:92-96; modules/utils.py:5-29 pad_batch / unpad_batch without their pass over the node rows):
 * rows = gt_seq_token_rows.  Only the fp32-accurate big-M path with bound weight images takes it: ask gt_linear_rows_ok. */
int gt_linear_rows_ok(int compute, int x_dtype, int y_dtype, const float* weight, int64_t M, int64_t N, int64_t K);
int gt_linear_set_rows(const int32_t* rows);
/* ... with LayerNorm(ln_w, ln_b, eps) of every stored row in the same epilogue (forward only; `norm_input` behind gnn2transformer,
 * modules/transformer_encoder.py:53-57): ln_out (storage type / pitch of y) = the normalised row, ln_mean / ln_rstd = its statistics
 * (what gt_layernorm_fwd saves for gt_layernorm_bwd), all at the row the output goes to.  N must fill one column block of the
 * kernel (gt_linear_rows_layernorm_ok: 128, 160, ...). */
int gt_linear_rows_layernorm_ok(int64_t N);
int gt_linear_set_rows_layernorm(const int32_t* rows, const float* ln_w, const float* ln_b, float eps, void* ln_out, float* ln_mean,
                                 float* ln_rstd);
int gt_linear_fwd_cat2(int y_dtype, int compute, const void* x1, int64_t K1, int64_t ldx1, const void* x2, int64_t K2, int64_t ldx2,
                       const float* weight, const float* bias, void* y, int64_t M, int64_t N, int64_t ldy, gt_stream_t stream);
int gt_linear_bwd_cat2(int y_dtype, int compute, const void* x1, int64_t K1, int64_t ldx1, const void* x2, int64_t K2, int64_t ldx2,
                       const float* weight, const void* dy, void* dx1, int64_t lddx1, void* dx2, int64_t lddx2, float* dweight,
                       float* dbias, int64_t M, int64_t N, int64_t ldy, void* workspace, size_t workspace_bytes, gt_stream_t stream);
size_t gt_w3_image_bytes(int64_t rows, int64_t contraction);
/* n images in one launch per 64 jobs: weight[i] = fp32 [N[i]][K[i]]; transposed[i] == 0 -> image of W (forward), != 0 -> image of
 * W^T (dX form); image[i]: gt_w3_image_bytes(rows, contraction) bytes, 1024-byte aligned. */
int gt_w3_images(int n, const float* const* weight, const int64_t* N, const int64_t* K, const int* transposed,
                 void* const* image, gt_stream_t stream);
/* per HOST THREAD, until gt_w3_unbind(): weight[i] (pointer, N[i], K[i]) -> image_fwd[i] / image_t[i] (either may be NULL =
 * keep the exact-fp32 MFMA kernel for that direction); at most 64 entries; the images must stay valid and current while bound */
int gt_w3_bind(int n, const float* const* weight, const int64_t* N, const int64_t* K, const void* const* image_fwd,
               const void* const* image_t);
int gt_w3_unbind(void);

/* ---- bf16 images in MFMA fragment order for the encoder layers' GEMMs (weight stationary in registers; csrc/linear1.h) --------
 * Replaces, per GEMM of nn.TransformerEncoderLayer (modules/transformer_encoder.py:28-32; in_proj / out_proj / linear1 / linear2),
 * the per-block re-staging of the fp32 master weight: the image is built once per optimizer step, all weights in one launch.
 * gt_w1_image_bytes: bytes of the image of a (rows = output columns, contraction) matrix; 0 = shape not covered (rows % 64,
 * contraction % 128, contraction <= 1024 and a fragment budget of 128 VGPRs per wave) -- such GEMMs keep the tiled kernels. */
size_t gt_w1_image_bytes(int64_t rows, int64_t contraction);
/* job i: weight[i] = fp32 [N[i]][K[i]]; transposed[i] == 0 -> image of W (forward), != 0 -> image of W^T (dX); 16-byte aligned */
int gt_w1_images(int n, const float* const* weight, const int64_t* N, const int64_t* K, const int* transposed,
                 void* const* image, gt_stream_t stream);
/* per HOST THREAD, until gt_w1_unbind(): gt_linear_fwd* / gt_linear_bwd* calls with bf16 rows in and out, bf16 compute, one group,
 * act in {none, relu} whose weight (pointer, N, K) is bound run the weight-stationary kernel (either image may be NULL) */
int gt_w1_bind(int n, const float* const* weight, const int64_t* N, const int64_t* K, const void* const* image_fwd,
               const void* const* image_t);
int gt_w1_unbind(void);
/* a = x W^T + b and y = LayerNorm(resid + dropout(a)) * ln_w + ln_b in ONE launch (post-norm encoder layer: out_proj + norm1,
 * linear2 + norm2; modules/transformer_encoder.py:28-32): a_out, save_mean, save_rstd as gt_linear_fwd + gt_layernorm_fwd write them.
 * Only with a bound image and N = the LayerNorm dim in one column block: gt_linear_layernorm_fwd_ok; otherwise make the two calls. */
int gt_linear_layernorm_fwd_ok(int dtype, int compute, const float* weight, int64_t M, int64_t N, int64_t K);
int gt_linear_layernorm_fwd(int dtype, int compute, const void* x, const float* weight, const float* bias, void* a_out, int64_t M,
                            int64_t N, int64_t K, const void* resid, const float* ln_weight, const float* ln_bias, float eps,
                            float dropout_p, uint64_t seed, void* y, float* save_mean, float* save_rstd, gt_stream_t stream);
/* dX of a Linear (weight [N][K]) whose output gradient g = dy W + add1 + add2 is the gradient of the LayerNorm below it,
 * ln_out = LayerNorm(ln_resid + dropout(ln_x)) over K columns (post-norm encoder layer backward: linear1's dX -> norm1, in_proj's dX ->
 * the previous layer's norm2; modules/transformer_encoder.py:28-32), in ONE launch: g is rounded to bf16 as the two-call form stores it
 * and never written; d_sub = d ln_x and d_resid (either may be NULL) and the LayerNorm's weight / bias gradients are what
 * gt_linear_bwd (dX only) + gt_layernorm_bwd would produce (row sums in a different order).  `workspace`:
 * gt_linear_bwd_dx_layernorm_workspace_bytes.  Only with a bound image of W^T, bf16 rows, K = 128: gt_linear_bwd_dx_layernorm_ok. */
int gt_linear_bwd_dx_layernorm_ok(int dtype, int compute, const float* weight, int64_t M, int64_t N, int64_t K);
size_t gt_linear_bwd_dx_layernorm_workspace_bytes(int64_t M, int64_t N, int64_t K);
int gt_linear_bwd_dx_layernorm(int dtype, int compute, const float* weight, const void* dy, const void* dx_add1, const void* dx_add2,
                               int64_t M, int64_t N, int64_t K, const void* ln_x, const void* ln_resid, const float* ln_weight,
                               const float* save_mean, const float* save_rstd, float dropout_p, uint64_t seed, void* d_sub, void* d_resid,
                               float* ln_dweight, float* ln_dbias, void* workspace, size_t workspace_bytes, gt_stream_t stream);
/* gt_linear_bwd_ld2 with the activation gate on the dX OUTPUT: dx = gate(dy W, y_or_mul) + add1 + add2, where y_or_mul [M][ldx] is
 * the forward output of the layer below (dropout_p >= 0: * 1[y > 0] / (1 - p)) or a saved multiplier (dropout_p < 0).  dW / db use
 * dy as it is.  The encoder layer's backward writes dZ1 = d(linear1 output) straight out of linear2's dX GEMM this way (the tensor
 * both GEMMs of linear1's backward read).  Only with a bound W^T image: gt_linear_bwd_gate_out_ok says so. */
int gt_linear_bwd_gate_out_ok(int x_dtype, int y_dtype, int compute, const float* weight, int64_t M, int64_t N, int64_t K);
int gt_linear_bwd_gate_out(int x_dtype, int y_dtype, int compute, const void* x, const float* weight, const void* dy,
                           const void* y_or_mul, const void* dx_add1, const void* dx_add2, void* dx, float* dweight, float* dbias,
                           int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldy, float dropout_p, void* workspace,
                           size_t workspace_bytes, gt_stream_t stream);
/* dW / db only; inside an overlap section it still runs on the overlap stream (ordered behind what `stream` holds so far):
 * a caller can start a GEMM's weight gradient ahead of its dX GEMM. */
int gt_linear_bwd_dw_forked(int x_dtype, int y_dtype, int compute, const void* x, const float* weight, const void* dy,
                            const void* y_for_mask, float* dweight, float* dbias, int64_t M, int64_t N, int64_t K, int64_t ldx,
                            int64_t ldy, float dropout_p, void* workspace, size_t workspace_bytes, gt_stream_t stream);
int gt_linear_bwd_mul_dw_forked(int x_dtype, int y_dtype, int compute, const void* x, const float* weight, const void* dy,
                                const void* gmul, float* dweight, float* dbias, int64_t M, int64_t N, int64_t K, int64_t ldx,
                                int64_t ldy, void* workspace, size_t workspace_bytes, gt_stream_t stream);
/* Deferred partial-sum reduces.  Every weight-gradient GEMM ends in a fixed-order sum over its M-split partials, every LayerNorm
 * backward in a column finish over its block partials: launches that only produce PARAMETER gradients.  Between gt_defer_begin(arena)
 * and gt_defer_end (per host thread; the whole-model backward opens one) gt_linear_bwd* / gt_layernorm_bwd put their partials into the
 * arena and queue the sum instead of launching it; gt_defer_flush(stream) runs everything queued as ONE launch on `stream` (the
 * caller orders it behind the producers, e.g. through gt_overlap_dw_fork).  Same summation order -> the same bits.  Without a
 * section, or when the arena / the job list is full, every call reduces on the spot as before.  (take / push are what the
 * library's own producers call; listed for completeness.) */
int gt_defer_begin(void* arena, size_t bytes);
int gt_defer_limit(size_t max_take_bytes);   /* only partial buffers up to this size join the open section (0 = every size) */
void* gt_defer_take(size_t bytes);
int gt_defer_push(const float* part, int nparts, int64_t len, int64_t stride, float* out, const float* part2, int64_t len2,
                  int64_t stride2, float* out2);
int gt_defer_push_strided(const float* part, int nparts, int64_t len, int64_t stride, float* out, int64_t ostride);   /* out[i * ostride] */
int gt_defer_room(int jobs);   /* 1 when a section is open and `jobs` more sums fit its list */
int gt_defer_flush(gt_stream_t stream);
int gt_defer_end(void);
int gt_overlap_dw_sync(void);
int gt_overlap_dw_release(const void* workspace, size_t bytes);
/* the weight-gradient GEMMs forked from now on are the last work of the backward (the optimizer waits for them): they get the
 * chip-filling launch configuration instead of the one-block-per-CU configuration of an overlapped GEMM; reset by _begin */
int gt_overlap_dw_urgent(int on);
/* used by the entry points whose last launch only produces parameter gradients (gt_layernorm_bwd's column finish,
 * gt_aggregate_bwd's partial reduce): the overlap stream, ordered behind `stream`, when `stream` is the main stream of an open
 * overlap section -- else `stream`; a launch sent there is booked under the workspace it reads (gt_overlap_dw_release). */
gt_stream_t gt_overlap_dw_fork(gt_stream_t stream, unsigned profiler_category /* GT_PROF_* of the caller, 0 = none */);
void gt_overlap_dw_booked(const void* workspace, size_t bytes);
int gt_overlap_dw_end(void);

int gt_linear_bwd_ld2(int x_dtype, int y_dtype, int compute, const void* x, const float* weight, const void* dy,
                      const void* y_for_mask, const void* dx_add1, const void* dx_add2, void* dx, float* dweight,
                      float* dbias, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldy, float dropout_p,
                      void* workspace, size_t workspace_bytes, gt_stream_t stream);
/* Grouped launch: `groups` independent GEMMs of the same shape in ONE launch per kernel (grid.y = group): the T
 * towers of PNAConv's pre_nns / post_nns (modules/pna/pna_module.py:33-41).  Group g reads x + g * x_group_stride
 * (elements, row stride ldx), weight + g * N * K, bias + g * N and writes y + g * y_group_stride (row stride ldy);
 * the backward mirrors it (dweight [groups][N][K], dbias [groups][N]); workspace = groups x the plain size. */
int gt_linear_fwd_grouped(int x_dtype, int y_dtype, int compute, const void* x, const float* weight, const float* bias,
                          void* y, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldy, int groups,
                          int64_t x_group_stride, int64_t y_group_stride, int act, float dropout_p, uint64_t seed,
                          gt_stream_t stream);
size_t gt_linear_bwd_grouped_workspace_bytes(int compute, int64_t M, int64_t N, int64_t K, int groups);
int gt_linear_bwd_grouped(int x_dtype, int y_dtype, int compute, const void* x, const float* weight, const void* dy,
                          const void* y_for_mask, const void* dx_add1, const void* dx_add2, void* dx, float* dweight,
                          float* dbias, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldy, int groups,
                          int64_t x_group_stride, int64_t y_group_stride, float dropout_p, void* workspace,
                          size_t workspace_bytes, gt_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Softmax cross-entropy over the stacked prediction heads (the Code2 loss, dataset/code.py:39-45:
 * (1/L) sum_l CrossEntropyLoss()(pred_l, y_arr[:, l])).  logits[b * ld + l * C + c], fp32;
 * target[b * target_stride + l] int64, ignore_index -100 as torch's default.
 *   fwd: lse [B*L], row_loss [B*L], head_scale [L] = 1 / (L * #live rows of head l), loss [1]
 *   bwd: dlogits [B][ld] = (softmax - onehot) * head_scale[l] * grad_loss[0]; columns L*C..ld zeroed
  * `loss` points to TWO floats: loss[0] = the loss, loss[1] = status = the number of targets that are neither ignore_index
 * nor in [0, C) (they do not contribute; torch raises a device-side assert for them -- the host wrapper raises IndexError on a
 * non-zero status in validate mode). */
int gt_xent_fwd(const float* logits, int64_t B, int64_t L, int64_t C, int64_t ld, const int64_t* target,
                int64_t target_stride, float* lse, float* row_loss, float* head_scale, float* loss,
                gt_stream_t stream);
int gt_xent_bwd(const float* logits, const float* lse, const float* head_scale, const int64_t* target,
                int64_t target_stride, const float* grad_loss, int64_t B, int64_t L, int64_t C, int64_t ld,
                float* dlogits, gt_stream_t stream);

/* Masked binary cross-entropy with logits (the Molpcba loss, dataset/mol.py:24-31: BCEWithLogitsLoss over
 * the entries with y == y, i.e. not NaN).  logits [B][ld] fp32 (T <= ld columns used), target [B][target_ld]
 * fp32 with NaN = unlabelled.
 *   fwd: row_sum / row_cnt [B]; out2 = { loss = S / den, den }, den = number of labelled entries, or
 *        *den_in when given (data parallel: global count / world size, so that the rank-averaged gradient is
 *        the gradient of the global mean); no labelled entry -> NaN like torch
 *   bwd: dlogits [B][ld] = labelled ? (sigmoid(x) - y) * grad_loss[0] / den : 0 (pad columns zeroed)
 */
int gt_bce_masked_fwd(const float* logits, const float* target, int64_t B, int64_t T, int64_t ld, int64_t target_ld,
                      const float* den_in /* or NULL */, float* row_sum, float* row_cnt, float* out2,
                      gt_stream_t stream);
int gt_bce_masked_bwd(const float* logits, const float* target, const float* out2, const float* grad_loss, int64_t B,
                      int64_t T, int64_t ld, int64_t target_ld, float* dlogits, gt_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Composite layer entry points: ONE call enqueues every kernel of a layer's forward or backward
 * (the gt_* primitives above, in order, on `stream`).  Purpose: host cost — a training step is
 * ~500 kernel launches; issued from C back to back instead of through per-op Python/autograd
 * round trips.  `saved` keeps the activations the backward needs (gt_*_saved_bytes), `workspace`
 * is scratch (gt_*_workspace_bytes), `grads` is one flat fp32 buffer holding every parameter
 * gradient of the layer in the order of the descriptor's parameter fields (gt_*_grad_elems);
 * gradients are overwritten.  All parameters are fp32.
 */
typedef struct gt_encoder_layer {  /* torch nn.TransformerEncoderLayer, post-norm, ReLU or GELU FFN */
  int64_t rows, d_model, ffn;
  int32_t nhead, dtype /* token storage */, compute /* fp32 storage only: GT_F32 | GT_BF16 */, training;
  const int32_t* seq_desc;
  int64_t num_seqs, row_stride, max_npos;
  const int32_t* work_items; /* optional attention tile list, see gt_attn_fwd */
  int64_t num_work;
  float dropout_p, ln_eps;
  int32_t act /* 0 relu, 1 gelu (erf form) */, reserved_;
  uint64_t seed;
  const float *in_w, *in_b, *out_w, *out_b, *l1_w, *l1_b, *l2_w, *l2_b, *n1_w, *n1_b, *n2_w, *n2_b;
} gt_encoder_layer;
size_t gt_encoder_layer_saved_bytes(const gt_encoder_layer* layer);
size_t gt_encoder_layer_workspace_bytes(const gt_encoder_layer* layer);
int64_t gt_encoder_layer_grad_elems(const gt_encoder_layer* layer);
int gt_encoder_layer_fwd(const gt_encoder_layer* layer, const void* x, void* y, void* saved, gt_stream_t stream);
int gt_encoder_layer_bwd(const gt_encoder_layer* layer, const void* x, const void* dy, const void* saved, void* dx,
                         float* grads, void* workspace, size_t workspace_bytes, gt_stream_t stream);
/* The LAST encoder layer when only one row per sequence is read afterwards (cls / last pooling: transformer_out[-1],
 * models/gnn_transformer.py:113-114 -- the reference computes every row and drops the rest): in_proj for every row (keys and values),
 * attention for the tile of the pooled row, out_proj / norm1 / FFN / norm2 on the POOLED rows only.  pool_rows [num_seqs] = token row
 * of the last position of every sequence; y_pool / dy_pool [num_seqs][d_model] in the token dtype.  Same parameters, same gradients:
 * the skipped rows influence neither the output nor any gradient.  (The row-wise dropouts are drawn per pooled-row index.) */
size_t gt_encoder_layer_pooled_saved_bytes(const gt_encoder_layer* layer);
size_t gt_encoder_layer_pooled_workspace_bytes(const gt_encoder_layer* layer);
int gt_encoder_layer_pooled_fwd(const gt_encoder_layer* layer, const void* x, const int64_t* pool_rows, void* y_pool, void* saved,
                                gt_stream_t stream);
int gt_encoder_layer_pooled_bwd(const gt_encoder_layer* layer, const void* x, const int64_t* pool_rows, const void* dy_pool,
                                const void* saved, void* dx, float* grads, void* workspace, size_t workspace_bytes, gt_stream_t stream);

typedef struct gt_gcn_layer {  /* x = h_in [+ vn[batch]]; y = BN(GCNConv(x)) [relu] [+ x]; fp32 rows */
  int64_t N, E, B, D;
  int32_t edge_mode /* NONE | LINEAR | TABLES */, has_vn, relu, residual, training, compute;
  int64_t edge_cols, table_rows;
  int32_t tab_off[4];
  float bn_momentum, bn_eps;
  const int32_t *graph_ptr, *node_graph, *in_ptr, *in_src, *in_eid, *out_ptr, *out_dst, *out_eid;
  const float *deg, *dis;
  const void* edge_attr;
  const float *lin_w, *lin_b, *root, *edge_w, *edge_b, *bn_w, *bn_b; /* gradient order: these 7 */
  float *bn_rm, *bn_rv;
  int64_t* bn_nbt;
  /* optional gt_event handles for running the virtual-node update on a second stream beside the conv:
   * the forward records ev_x_ready once x_out is written; the backward waits for ev_dx_wait (dx_extra
   * complete) right before the dX GEMM.  NULL = no cross-stream dependency. */
  void *ev_x_ready, *ev_dx_wait;
  /* F.dropout(h, drop_ratio) after the layer BatchNorm[+ReLU] (gnn_module.py:88-90,209-212), training only */
  uint64_t seed;
  float dropout_p;
  int32_t x_has_vn;   /* 1: h_in already holds h + vn[batch] (the previous layer added it, see vn_next): no add, x_out unused */
  /* vn_next [B][D]: the NEXT layer's virtual-node rows, added to y in the BatchNorm apply pass (y then IS the next
   * layer's x); ev_vn_next: optional gt_event recorded by the stream that produces vn_next, waited for before that pass */
  const void* vn_next;
  void* ev_vn_next;
  const float* lin_wt; /* optional: lin_w transposed [D][D] (gt_transpose), for the backward's dX GEMM */
  /* backward only, both optional.  This layer's d_h_in IS the dy of the previous layer's BatchNorm: with prev_saved (that
   * layer's `saved` block: same N and D), its affine parameters and ReLU flag, the dX GEMM's epilogue writes the BatchNorm-
   * backward partial sums into prev_bn_part[gt_linear_bwd_bnstats_rows(N)][2][D] (gt_linear_bwd_bnstats; the caller checks
   * gt_linear_bwd_bnstats_ok and that no dropout follows that BatchNorm).  The previous layer's backward then gets the same
   * buffer as bn_part_in / bn_nparts_in and skips its own statistics pass (gt_batchnorm_bwd_parts). */
  const void* prev_saved;
  const float *prev_bn_w, *prev_bn_b;
  float* prev_bn_part;
  const float* bn_part_in;
  int32_t prev_relu, bn_nparts_in;
  /* forward only, optional gt_event: the graph structure (in_ptr / in_src / in_eid / deg / dis of this descriptor) is being built on
   * another stream (gt_graph_prep beside the input embedding and this layer's GEMM); the forward waits for it between its linear
   * and its aggregate.  NULL = the arrays are ready on `stream`. */
  void* ev_graph_ready;
  /* backward only, optional: d_h_in[n] += dx_bcast[dx_bcast_idx[n]] in the dX GEMM's epilogue (the virtual-node update's gradient per
   * GRAPH, modules/gnn_module.py:199,219: no N x D broadcast pass).  The caller asks gt_linear_bwd_bcast_ok first; the request is made
   * right in front of the layer's one dX GEMM and never outlives the call. */
  const float* dx_bcast;
  const int32_t* dx_bcast_idx;
} gt_gcn_layer;
size_t gt_gcn_layer_saved_bytes(const gt_gcn_layer* layer);
size_t gt_gcn_layer_workspace_bytes(const gt_gcn_layer* layer);
int64_t gt_gcn_layer_grad_elems(const gt_gcn_layer* layer);
/* x_out [N][D] receives h_in + vn[batch] when has_vn (the reference's updated h_list[layer]). */
int gt_gcn_layer_fwd(const gt_gcn_layer* layer, const void* h_in, const void* vn, void* x_out, void* y, void* saved,
                     void* workspace, size_t workspace_bytes, gt_stream_t stream);
/* x: the forward's x_out (or h_in without vn); dx_extra: optional gradient reaching x from its other
 * consumers (JK / virtual-node pooling); d_vn [B][D] = per-graph sum of d_h_in when has_vn. */
int gt_gcn_layer_bwd(const gt_gcn_layer* layer, const void* x, const void* dy, const void* dx_extra, const void* saved,
                     void* d_h_in, void* d_vn, float* grads, void* workspace, size_t workspace_bytes,
                     gt_stream_t stream);

typedef struct gt_gin_layer {  /* x = h_in [+ vn[batch]]; y = drop(BN(GINConv(x)) [relu]) [+ x]; fp32 rows.
                                * GINConv(x) = W2 ReLU(BN1(W1 ((1 + eps) x + sum_k relu(x_j + e_k))))  (conv.py:18-36) */
  int64_t N, E, B, D;
  int32_t edge_mode /* NONE | LINEAR | TABLES */, has_vn, relu, residual, training, compute;
  int64_t edge_cols, table_rows;
  int32_t tab_off[4];
  float bn_momentum, bn_eps;
  const int32_t *graph_ptr, *node_graph, *in_ptr, *in_src, *in_eid, *out_ptr, *out_dst, *out_eid;
  const void* edge_attr;
  /* gradient order: eps [20 floats: d_eps + scratch], edge_w, edge_b (LINEAR only), w1, b1, bn1_w, bn1_b, w2, b2, bn_w, bn_b */
  const float *eps, *edge_w, *edge_b, *w1, *b1, *bn1_w, *bn1_b, *w2, *b2, *bn_w, *bn_b;
  float *bn1_rm, *bn1_rv, *bn_rm, *bn_rv;
  int64_t *bn1_nbt, *bn_nbt;
  void *ev_x_ready, *ev_dx_wait; /* as in gt_gcn_layer (the backward waits right before its last add) */
  uint64_t seed;
  float dropout_p;
  int32_t x_has_vn;   /* 1: h_in already holds h + vn[batch] (the previous layer added it, see vn_next): no add, x_out unused */
  /* vn_next [B][D]: the NEXT layer's virtual-node rows, added to y in the BatchNorm apply pass (y then IS the next
   * layer's x); ev_vn_next: optional gt_event recorded by the stream that produces vn_next, waited for before that pass */
  const void* vn_next;
  void* ev_vn_next;
  const float *w1_t, *w2_t; /* optional: w1 / w2 transposed (gt_transpose), for the backward's dX GEMMs */
} gt_gin_layer;
size_t gt_gin_layer_saved_bytes(const gt_gin_layer* layer);
size_t gt_gin_layer_workspace_bytes(const gt_gin_layer* layer);
int64_t gt_gin_layer_grad_elems(const gt_gin_layer* layer);
int gt_gin_layer_fwd(const gt_gin_layer* layer, const void* h_in, const void* vn, void* x_out, void* y, void* saved,
                     void* workspace, size_t workspace_bytes, gt_stream_t stream);
int gt_gin_layer_bwd(const gt_gin_layer* layer, const void* x, const void* dy, const void* dx_extra, const void* saved,
                     void* d_h_in, void* d_vn, float* grads, void* workspace, size_t workspace_bytes,
                     gt_stream_t stream);

typedef struct gt_vn_update {  /* vn_out = drop(ReLU(BN(W2 ReLU(BN(W1 (pool(x) + vn)))))) [+ vn]; fp32 */
  int64_t N, B, D;
  int32_t residual, training, compute, pad_;
  float bn_momentum, bn_eps;
  const int32_t *graph_ptr, *node_graph, *identity_graph /* [B] = 0..B-1 */;
  const float *w1, *b1, *bn1_w, *bn1_b, *w2, *b2, *bn2_w, *bn2_b; /* gradient order: these 8 */
  float *bn1_rm, *bn1_rv, *bn2_rm, *bn2_rv;
  int64_t *bn1_nbt, *bn2_nbt;
  uint64_t seed;     /* F.dropout on the MLP output (gnn_module.py:222), training only */
  float dropout_p;
  int32_t pad2_;
  void* ev_dx_done;  /* optional gt_event: recorded by gt_vn_update_bwd once d_x and d_vn are enqueued -- BEFORE its two weight-
                        gradient GEMMs, which nothing on the critical path waits for */
} gt_vn_update;
size_t gt_vn_update_saved_bytes(const gt_vn_update* layer);
size_t gt_vn_update_workspace_bytes(const gt_vn_update* layer);
int64_t gt_vn_update_grad_elems(const gt_vn_update* layer);
int gt_vn_update_fwd(const gt_vn_update* layer, const void* x, const void* vn, void* vn_out, void* saved,
                     void* workspace, size_t workspace_bytes, gt_stream_t stream);
/* d_x [N][D] = d(pooled)[graph(n)] (+ d_x_add [N][D] when not NULL), d_vn [B][D]. */
int gt_vn_update_bwd(const gt_vn_update* layer, const void* d_vn_out, const void* saved, const void* d_x_add, void* d_x,
                     void* d_vn, float* grads, void* workspace, size_t workspace_bytes, gt_stream_t stream);
/* d_t0 [B][D] (fp32) inside `workspace` of a gt_vn_update_bwd call.  With d_x == NULL that call skips its broadcast pass
 * d_x[n] = d_t0[node_graph[n]] (+ d_x_add[n]) over the N node rows; the caller adds the rows where d_x is consumed -- in the
 * epilogue of the layer's dX GEMM (gt_linear_bwd_bcast).  Valid until the workspace is reused. */
const float* gt_vn_update_bwd_dt0(const gt_vn_update* L, void* workspace);

/* PNA layer (modules/pna/pna_module.py:57-78 around PyG's PNAConv(towers = T, divide_input = True, no edge features); math stated
 * in-tree by modules/pna_layer.py:131-167, modules/pna/aggregators.py:11-34, modules/pna/scalers.py:10-31):
 *   x' = relu(BN(lin(post(cat[x, scaled aggregates of pre([x_i || x_j])])))) + x   [then F.dropout when dropout_p > 0]
 * on re-stacked IMAGES of the tower weights (built by the caller with gt_gather_f32 once per step, all layers in one launch):
 *   pre_w  [T][2F][F] = [A_t ; B_t] (pre_nns[t].weight [F][2F] = [A_t | B_t]: U = x_t A_t^T + b_t is the target-role term, V = x_t B_t^T the
 *                       source-role term of the per-edge Linear), pre_b [T][2F] = [b_t | 0]
 *   post_w [T][S Fo][5F]: block s = the columns of scaler s over the kernel's [x | mean | max | min | std] operand (x and the bias in
 *                       block 0 only, zeros for unused aggregators), post_b [T][S Fo]; scales [N][S] = the per-node degree scalers.
 * One grouped GEMM gives [U | V] for all towers, the aggregate kernel writes [x | mean | max | min | std] where the grouped post-GEMM
 * reads it, its S output blocks are combined with the scalers, then lin + BatchNorm + ReLU + residual.  `grads` (gradient order):
 * lin_w [D][D], lin_b [D], bn_w [D], bn_b [D]; the image gradients go to d_pre_w / d_pre_b / d_post_w / d_post_b (same shapes as the
 * images; the caller maps them back onto the parameters with gt_gather_f32 through the inverse map). */
typedef struct gt_pna_layer {
  int64_t N, E, D;
  int32_t T, S, training, compute;
  float bn_momentum, bn_eps;
  const int32_t *in_ptr, *in_src, *in_eid, *out_ptr, *out_dst, *out_eid;
  const float* scales;
  const float *pre_w, *pre_b, *post_w, *post_b;   /* images */
  const float *lin_w, *lin_b, *bn_w, *bn_b;
  float *bn_rm, *bn_rv;
  int64_t* bn_nbt;
  float *d_pre_w, *d_pre_b, *d_post_w, *d_post_b; /* image gradients (backward) */
  uint64_t seed;
  float dropout_p;
  int32_t pad_;
} gt_pna_layer;
size_t gt_pna_layer_saved_bytes(const gt_pna_layer* layer);
size_t gt_pna_layer_workspace_bytes(const gt_pna_layer* layer);
int64_t gt_pna_layer_grad_elems(const gt_pna_layer* layer);
int gt_pna_layer_fwd(const gt_pna_layer* layer, const void* x, void* y, void* saved, void* workspace, size_t workspace_bytes,
                     gt_stream_t stream);
int gt_pna_layer_bwd(const gt_pna_layer* layer, const void* x, const void* dy, const void* saved, void* dx, float* grads,
                     void* workspace, size_t workspace_bytes, gt_stream_t stream);
/* per-node degree scalers of PNAConv for one batch: scales[n][s], s over `kinds` (0 identity / none, 1 amplification
 * log(d+1)/avg_log, 2 attenuation avg_log/log(d+1) (1 at d = 0), 3 linear d/avg_lin, 4 inverse_linear avg_lin/d (1 at d = 0)),
 * d = in-degree from in_ptr (modules/pna/scalers.py:10-31). */
int gt_pna_scales(const int32_t* in_ptr, int64_t num_nodes, int num_scalers, const int32_t* kinds_host, float avg_log, float avg_lin,
                  float* scales, gt_stream_t stream);
/* dst[i] = map[i] < 0 ? 0 : src[map[i]] (fp32): re-stacked weight images and, through the inverse map, their gradients */
int gt_gather_f32(float* dst, const float* src, const int32_t* map, int64_t n, gt_stream_t stream);
/* gt_pna_aggregate_* on the fused layer's layouts: UV [N][T][2F] = [U_t | V_t], in5 [N][T][5F] = [x_t | mean | max | min | std]
 * (x copied on the way in), backward from d_in5 to dUV [N][T][2F] and the x block's gradient dxpart [N][D]. */
int gt_pna_aggregate_fwd_uv(const float* UV, const float* x, int64_t num_nodes, int64_t dim, int towers, const int32_t* in_ptr,
                            const int32_t* in_src, const int32_t* in_eid, float* in5, float* mean_v, int32_t* arg,
                            gt_stream_t stream);
int gt_pna_aggregate_bwd_uv(const float* UV, const float* in5, const float* mean_v, const int32_t* arg, const float* d_in5,
                            int64_t num_nodes, int64_t dim, int towers, const int32_t* in_ptr, const int32_t* out_ptr,
                            const int32_t* out_dst, const int32_t* out_eid, float* dUV, float* dxpart, gt_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Whole-model driver: ONE C call per direction for the training step's forward and backward.
 * Replaces the per-step Python sequencing of the composites above (the reference's step is
 * `pred = model(batch); loss.backward()`, trainers/base_trainer.py:29-36, over models/gnn_transformer.py:90-128,
 * modules/gnn_module.py:60-107 / :172-241, modules/transformer_encoder.py:42-61): graph structure, token layout, input
 * embedding, L message-passing layers with the virtual-node chain on a second stream, gnn2transformer, token rows + CLS,
 * norm_input, the encoder layers, the final norm, cls / last pooling and the (stacked) prediction heads; the backward in
 * three stages (heads .. gnn2transformer | message passing | input encoder) so that a data-parallel caller can put each
 * finished range of the flat gradient buffer on the wire between two calls.
 *
 * gt_model is filled ONCE per model (static pointers, sizes, gradient offsets, streams, events); gt_model_batch once per
 * step (the collated batch); `ctx` (gt_model_ctx_bytes() of HOST memory, caller-owned, one per forward, kept until its
 * backward) carries everything the backward needs.  All device memory is the caller's: gt_model_prepare (host only) says
 * how many bytes of arena the forward and the backward need.
 */
#define GT_MODEL_MAX_LAYERS 16
#define GT_MODEL_MAX_TABLES 16

typedef struct gt_image_set {   /* images of a weight list (gt_w3_* or gt_w1_*): build jobs and bind table, HOST arrays */
  int32_t n_jobs, n_bind;
  const float* const* job_w;
  const int64_t *job_N, *job_K;
  const int* job_T;
  void* const* job_img;
  const float* const* bind_w;
  const int64_t *bind_N, *bind_K;
  const void* const* bind_f;
  const void* const* bind_t;
} gt_image_set;

typedef struct gt_stage_ring {   /* pinned HOST staging slots for the token layout's H2D copy (one per device) */
  void* base;                    /* slots x slot_bytes of pinned memory */
  int64_t slot_bytes;
  int32_t slots, next;
  void* events[64];              /* gt_event per slot (recorded behind the copy that read the slot); slots <= 64 */
} gt_stage_ring;

typedef struct gt_model {
  int32_t conv /* GT_CONV_GCN | GT_CONV_GIN | GT_CONV_PNA */, L, n_enc, has_vn, jk_cat, residual;
  int32_t embed_kind /* 0 one table per column of x (AtomEncoder), 1 nn.Linear, 2 ASTNodeEncoder (x[:,0], x[:,1], node_depth) */, n_tables, vn0_in_embed, embed_sorted;
  int32_t with_cls, vn_defer_dw;
  int64_t D, d, Nh, ldy, ne_K, max_input_len, dw_overlap_min_elems;
  void* conv_layers;        /* gt_gcn_layer[L] | gt_gin_layer[L] | gt_pna_layer[L]: static fields filled by the caller */
  gt_vn_update* vn;         /* [L-1] or NULL */
  gt_encoder_layer* enc;    /* [n_enc] */
  const float* tables[GT_MODEL_MAX_TABLES];
  int64_t table_rows[GT_MODEL_MAX_TABLES], table_clamp[GT_MODEL_MAX_TABLES];
  const float *vn_emb, *ne_w, *ne_b, *g2t_w, *g2t_b, *cls, *nin_w, *nin_b, *nout_w, *nout_b, *head_w, *head_b;
  const int64_t* zero_i64;  /* one device int64 0 (index column of the virtual-node row in the embedding sum) */
  float nin_eps, nout_eps;
  /* offsets (floats) into the flat gradient buffer; -1 = absent */
  int64_t off_tables[GT_MODEL_MAX_TABLES], off_ne_w, off_ne_b, off_vn_emb, off_conv[GT_MODEL_MAX_LAYERS],
      off_vn[GT_MODEL_MAX_LAYERS], off_g2t_w, off_g2t_b, off_cls, off_nin_w, off_nin_b, off_enc[GT_MODEL_MAX_LAYERS],
      off_nout_w, off_nout_b, off_head_w, off_head_b, grad_total;
  /* streams (gt_stream_create; NULL = that overlap is off) and events (gt_event_create) */
  void *st_vn, *st_dw, *st_prep;
  void *ev_x[GT_MODEL_MAX_LAYERS], *ev_vn[GT_MODEL_MAX_LAYERS], *ev_dvn[GT_MODEL_MAX_LAYERS], *ev_extra[GT_MODEL_MAX_LAYERS],
      *ev_pool[GT_MODEL_MAX_LAYERS];
  void *ev_vnemb, *ev_sort[2], *ev_wt[2], *ev_prep_begin, *ev_graph, *ev_w1;
  gt_image_set w3, w3_enc, w1;   /* bf16x3 images without / with the encoder weights, fragment-order encoder images */
  /* conv == GT_CONV_PNA (conv_layers = gt_pna_layer[L]; no virtual node, JK = last, residual): every tower weight of every layer
   * lives in ONE flat fp32 buffer pna_src [pna_n_src]; the forward rebuilds the re-stacked images pna_img [pna_n_img] with one
   * gt_gather_f32 through pna_map, the backward maps the image gradients back onto the flat gradient buffer at off_pna_src through
   * pna_inv [pna_n_src].  pna_img_off[l] = offsets (floats) of layer l's {pre_w, pre_b, post_w, post_b} inside the image. */
  const float* pna_src;
  float* pna_img;
  const int32_t *pna_map, *pna_inv;
  int64_t pna_n_img, pna_n_src, off_pna_src;
  int64_t pna_img_off[GT_MODEL_MAX_LAYERS][4];
  int32_t pna_kinds[8];          /* degree scalers of the S output blocks (gt_pna_scales) */
  float pna_avg_log, pna_avg_lin;
} gt_model;

typedef struct gt_model_batch {
  int64_t N, E, B;
  /* the collated batch (PyG layout): edge_index [2][E] int64, batch [N] int64 sorted */
  const int64_t *edge_index, *batch;
  const int64_t* sizes_host;   /* HOST per-graph node counts or NULL (then the token layout is built on the device) */
  /* optional, already built by the caller (gt_graph_prep outputs): graph_ptr != NULL skips the in-driver prep */
  const int32_t *graph_ptr, *node_graph, *in_ptr, *in_src, *in_eid, *out_ptr, *out_dst, *out_eid;
  const float *deg, *dis;
  /* optional, already built token layout: seq_desc != NULL */
  const int32_t* seq_desc;
  const int64_t* last_rows;
  const int32_t* work_items;
  int64_t rows, max_npos, num_work;
  int32_t lay_exact, pad0_;
  /* inputs: tables: x [N][>= n_tables] int64 (element strides), ASTNodeEncoder: + node_depth; linear: x [N][ne_K] fp32 */
  const void* x;
  int64_t x_stride0, x_stride1;
  const int64_t* node_depth;
  int64_t depth_stride;
  const void* edge_attr;
  const int32_t *zeros_B, *ident_B, *ptr01;   /* [B] zeros, [B] 0..B-1, {0, B} (device, int32) */
  int32_t training, compute /* GT_F32 | GT_BF16: the fp32-stored GEMMs */, tdt /* token rows */, will_bwd;
  int32_t use_w3 /* 0 none, 1 gt_model.w3, 2 gt_model.w3_enc */, use_w1, sync_bn /* gt_bn_sync_set hook installed */, pad2_;
  float gnn_p, enc_p;
  uint64_t gnn_seed, enc_seed;
  gt_stage_ring* ring;
} gt_model_batch;

typedef struct gt_model_sizes {
  int64_t rows, max_npos, num_work, arena_bytes, barena_bytes;
  int32_t exact /* 0: the arenas must be zero-filled by the caller */, pad_;
} gt_model_sizes;

size_t gt_model_ctx_bytes(void);
int gt_model_prepare(const gt_model* model, const gt_model_batch* batch, void* ctx, gt_model_sizes* sizes);
/* logits [B][ldy] fp32 */
int gt_model_forward(const gt_model* model, void* ctx, void* arena, float* logits, gt_stream_t stream);
/* dlogits [B][ldy] fp32 (pad columns zero); grads: the flat gradient buffer [grad_total] (overwritten);
 * stages: bit 0 heads .. gnn2transformer, bit 1 message passing, bit 2 input encoder + final joins; in order, each once. */
int gt_model_backward(const gt_model* model, void* ctx, const float* dlogits, float* grads, void* barena, int stages,
                      gt_stream_t stream);
/* first / one-past-last float of the gradient range each stage completes: {g2t..total, gnn_lo..g2t, 0..gnn_lo} */
int gt_model_grad_ranges(const gt_model* model, int64_t* lo3, int64_t* hi3);
/* out4 = sizeof {gt_model, gt_model_batch, gt_image_set, gt_stage_ring}: a binding checks its mirror of the layouts */
int gt_model_abi_sizes(int64_t* out4);
/* The host half of the packed token layout (what gt_model_prepare stages for its H2D copy) on its own: seq_desc [B][4] at offset 0,
 * last_rows [B] int64 at meta6[3], the attention work list [num_work][2] at meta6[4]; meta6 = {rows, max_npos, num_work, offset of
 * last_rows, offset of the work list, total bytes}.  out_host == NULL: sizes only.  Pure host code (no GPU needed). */
int gt_seq_layout_packed_host(const int64_t* sizes_host, int64_t B, int64_t max_input_len, int with_cls, void* out_host,
                              size_t out_bytes, int64_t* meta6);
/* gt_seq_gather with the CLS row given in fp32 whatever the token dtype (converted while it is written) */
int gt_seq_gather_cls32(int dtype, const void* h, const float* cls32, const int32_t* graph_ptr, const int32_t* seq_desc,
                        int64_t num_seqs, int64_t row_stride, int64_t max_npos, int with_cls, int64_t dim,
                        void* tokens, gt_stream_t stream);
/* out[c] (fp32) = sum_r x[r][c] in a fixed order (the CLS embedding's gradient from its per-graph rows) */
int gt_colsum_f32(int dtype, const void* x, int64_t rows, int64_t dim, float* out, gt_stream_t stream);
/* ... over the rows row_idx[0 .. rows) of x only (the CLS rows of the token matrix) */
int gt_colsum_rows_f32(int dtype, const void* x, const int64_t* row_idx, int64_t rows, int64_t D, float* out, gt_stream_t stream);
/* The token row of every node -- the row map of gt_linear_set_rows -- and the CLS rows of the token matrix, for token layouts WITHOUT
 * pad rows (the packed layout of gt_seq_layout_packed: kv_off = 0, npos = kv_len): rows[r] (int32 [N]) = the row gt_seq_gather would
 * copy node r to, -1 for the leading nodes a truncated graph drops (modules/utils.py:17-21 keeps the last max_num_nodes);
 * tokens[cls row of sequence b] = cls32 (fp32 [D]).  With gt_linear_set_rows this replaces gt_seq_gather / gt_seq_scatter. */
int gt_seq_token_rows(int dtype, const float* cls32, const int32_t* graph_ptr, const int32_t* node_graph, const int32_t* seq_desc,
                      int64_t num_seqs, int64_t row_stride, int with_cls, int64_t N, int64_t D, void* tokens, int32_t* rows,
                      gt_stream_t stream);
/* ... whose CLS rows also get LayerNorm(ln_w, ln_b, eps): ln_out / ln_mean / ln_rstd as in gt_linear_set_rows_layernorm (which
 * covers the node rows) */
int gt_seq_token_rows_layernorm(int dtype, const float* cls32, const int32_t* graph_ptr, const int32_t* node_graph,
                                const int32_t* seq_desc, int64_t num_seqs, int64_t row_stride, int with_cls, int64_t N, int64_t D,
                                void* tokens, int32_t* rows, const float* ln_w, const float* ln_b, float ln_eps, void* ln_out,
                                float* ln_mean, float* ln_rstd, gt_stream_t stream);

/* Stand-alone dropout (F.dropout / nn.Dropout with no producing kernel to carry it: masked_transformer_encoder.py:54,75,
 * pna/pna_module.py:78): y[i] = keep(i, seed) ? x[i] / (1 - p) : 0 over n elements (n % 4 == 0; x == y allowed).  The same
 * call on the gradient is the backward (the mask is a function of (i, seed) only). */
int gt_dropout(int dtype, const void* x, void* y, int64_t n, float p, uint64_t seed, gt_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Data movement of the fused model path (graphtrans_amd/engine.py).
 * gt_copy2d: dst[r][0..width) = src[r][0..width) with byte pitches (all multiples of 16): the two
 *   column slabs of JK = "cat" (torch.cat([h_list[0], h_list[-1]], 1), modules/gnn_module.py:104-105)
 *   and their split in the backward.
 * gt_rows_gather: out[i] (fp32) = x[idx[i]] -- transformer_out[-1] of every sequence for the cls / last
 *   pooling (models/gnn_transformer.py:113-114); gt_rows_scatter: its adjoint into a zeroed
 *   [total_rows][dim] buffer of the given dtype (idx must not repeat).
 * gt_repitch: dst[r][c] = c < src_cols ? src[r][c] : 0 for c < dst_cols (contiguous rows, 2- or 4-byte elements):
 *   zero-pads / truncates the columns of a feature matrix or weight whose width is not a multiple of the
 *   16-byte chunk the GEMM needs (the 37-feature TU node encoder nn.Linear(F, D), dataset/tud.py:65).
 */
int gt_repitch(void* dst, int64_t dst_cols, const void* src, int64_t src_cols, int64_t rows, int elt_bytes,
               gt_stream_t stream);
int gt_copy2d(void* dst, int64_t dst_pitch_bytes, const void* src, int64_t src_pitch_bytes, int64_t width_bytes,
              int64_t rows, gt_stream_t stream);
int gt_add3(const float* a, const float* b, const float* c /* or NULL */, int64_t n, float* out, gt_stream_t stream);
int gt_rows_gather(int dtype, const void* x, const int64_t* idx, int64_t n, int64_t dim, float* out, gt_stream_t stream);
/* rows moved in their storage type (the pooled rows of the last encoder layer): take out[i] = x[idx[i]]; put out[idx[i]] = x[i] into
 * a zero-filled [total_rows][dim] matrix; add out[idx[i]] += x[i] (idx must not repeat) */
int gt_rows_take(int dtype, const void* x, const int64_t* idx, int64_t n, int64_t dim, void* out, gt_stream_t stream);
int gt_rows_put(int dtype, const void* x, const int64_t* idx, int64_t n, int64_t total_rows, int64_t dim, void* out, gt_stream_t stream);
int gt_rows_add(int dtype, const void* x, const int64_t* idx, int64_t n, int64_t dim, void* out, gt_stream_t stream);
int gt_rows_scatter(int dtype, const float* grad, const int64_t* idx, int64_t n, int64_t total_rows, int64_t dim,
                    void* out, gt_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * AdamW (decoupled weight decay, no amsgrad) over many fp32 tensors in one launch: the optimizer step
 * of the training loop (optim.AdamW at main.py:178, optimizer.step() at trainers/base_trainer.py:36).
 * table [T] (device) describes every tensor; chunk c (gt_adamw_chunk_elems() elements) belongs to
 * tensor chunk_tensor[c] and is its chunk_local[c]-th chunk (both device, built once).  One call
 * updates tensors [tensor_begin, tensor_begin + num_tensors) = chunks [chunk_begin, +num_chunks);
 * grads_host[i] is the DEVICE pointer of tensor tensor_begin + i's gradient (HOST array; NULL skips
 * the tensor, as torch skips parameters without a gradient).  step = 1, 2, ... (bias correction).
 */
#define GT_ADAMW_MAX_TENSORS 384
typedef struct gt_adamw_tensor {
  float* param;
  float* exp_avg;
  float* exp_avg_sq;
  int64_t numel;
} gt_adamw_tensor;
int gt_adamw_chunk_elems(void);
int gt_adamw_step(const gt_adamw_tensor* table, const int32_t* chunk_tensor, const int32_t* chunk_local,
                  int64_t chunk_begin, int64_t num_chunks, int tensor_begin, int num_tensors,
                  const float* const* grads_host, float lr, float beta1, float beta2, float eps, float weight_decay,
                  int64_t step, const float* grad_scale /* device scalar or NULL */, gt_stream_t stream);
/* Gradient clipping by global norm (torch.nn.utils.clip_grad_norm_(model.parameters(), args.grad_clip),
 * trainers/base_trainer.py:34-35) without touching the gradients: gt_grad_sqnorm writes one partial sum of
 * squares per chunk of the same chunk map (partial [total chunks], indexed by absolute chunk id),
 * gt_grad_clip_coef reduces them in fixed order to out2 = { total_norm, min(1, max_norm / (total_norm + 1e-6)) };
 * &out2[1] is then passed to gt_adamw_step as grad_scale. */
int gt_grad_sqnorm(const gt_adamw_tensor* table, const int32_t* chunk_tensor, const int32_t* chunk_local,
                   int64_t chunk_begin, int64_t num_chunks, int tensor_begin, int num_tensors,
                   const float* const* grads_host, float* partial, gt_stream_t stream);
int gt_grad_clip_coef(const float* partial, int64_t num_partials, float max_norm, float* out2, gt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GRAPHTRANS_HIP_H */
