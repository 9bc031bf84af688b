// model.hip — whole-model driver: ONE C call per direction (gt_model_forward / gt_model_backward).
//
// The reference's training step is `pred = model(batch); loss = calc_loss(pred, batch); loss.backward()`
// (trainers/base_trainer.py:29-36) over GNNTransformer.forward (models/gnn_transformer.py:90-128): node encoder ->
// L x {virtual-node add, conv, BatchNorm, ReLU, dropout, virtual-node update} (modules/gnn_module.py:60-107, :172-241) ->
// JK -> gnn2transformer -> pad + CLS + norm_input -> encoder layers -> final norm (modules/transformer_encoder.py:42-61) ->
// pooling -> heads.  Rounds 1-3 sequenced the composites of layers.hip from Python (~60 ctypes calls, descriptor refresh,
// arena book-keeping: 2.3 ms of interpreter time per Code2 step against ~1 ms of launches).  Here the same calls are issued
// from C in the same order on the same three streams; the caller supplies one arena per direction.
// Pure host code + nothing allocated or synchronised: every buffer is the caller's.
#include <algorithm>
#include <cstring>
#include <vector>

#include "gt_common.h"

namespace {

#define GT_TRY(call)                \
  do {                              \
    int rc__ = (call);              \
    if (rc__ != GT_OK) return rc__; \
  } while (0)

constexpr int MAXL = GT_MODEL_MAX_LAYERS;
constexpr int MAXT = GT_MODEL_MAX_TABLES;
constexpr int64_t PREP_MIN_EDGES = 400000;   // gt_graph_prep beside the first kernels only for batches this large (see gt_model_prepare)
constexpr uint64_t SEED_STEP = 0x9E3779B97F4A7C15ULL;   // graphtrans_amd/modules/gnn_module.py:layer_seed / vn_seed

struct Bump {
  size_t off = 0;
  size_t take(size_t bytes) {
    size_t o = off;
    off = (o + bytes + 255) / 256 * 256;
    return o;
  }
};
inline size_t c4(size_t n) { return (n + 3) / 4 * 4; }
inline size_t esz(int dtype) { return dtype == GT_BF16 ? 2 : 4; }

// ---- per-forward context (host memory of the caller, opaque to it) -----------------------------------------------------------
struct Ctx {
  uint32_t magic;
  int prepared, forwarded, stages_done;
  gt_model_batch in;
  int64_t rows, max_npos, num_work;
  int exact, lay_host;                 // lay_host: the layout was built into a staging slot by gt_model_prepare
  int compute, tdt;
  size_t tsz;
  int use_prep, build_graph, build_layout_dev;
  // staging slot of the host-built layout
  void* stage_ptr;
  void* stage_event;
  size_t lay_bytes, lay_o_last, lay_o_work;
  // batch descriptors
  gt_gcn_layer gcn[MAXL];
  gt_gin_layer gin[MAXL];
  gt_pna_layer pna[MAXL];
  gt_vn_update vn[MAXL];
  gt_encoder_layer enc[MAXL];
  // forward arena
  size_t o_rowmap;   // int32 [N]: the token row of every node (gt_seq_token_rows), when gnn2transformer writes the token rows itself
  int fuse_rows;
  size_t o_h[MAXL + 1], o_x0, o_vn[MAXL], o_vn_saved[MAXL], o_conv_saved[MAXL], o_cat, o_hn, o_tok, o_xin, o_st0, o_xe[MAXL],
      o_enc_saved[MAXL], o_hgin, o_sto, o_eplan, o_esort_ws, o_ne_x, o_ne_w, o_hg, o_ws, o_ws2, o_wt[MAXL], o_g2t_wt;
  size_t o_scales, q_dimg;
  size_t o_graph_ptr, o_node_graph, o_in_ptr, o_out_ptr, o_idx, o_dd, o_status, o_prep_ws, o_lay, o_lay_meta;
  size_t ws_bytes, ws2_bytes, eplan_bytes, esort_ws_bytes, prep_ws_bytes, arena_bytes;
  int cat2, want_wt, esort, late_wait;
  int64_t Kc;
  char* base;
  // resolved pointers
  const int32_t *graph_ptr, *node_graph, *seq_desc, *work_items;
  const int64_t* last_rows;
  const void *xptr[MAXL], *enc_in[MAXL], *pre_out, *first, *node_rep, *h_last;
  const float *g2t_wt, *ne_x, *ne_w;
  int T;
  const int64_t* e_idx[MAXT + 1];
  int64_t e_str[MAXT + 1], e_clamp[MAXT + 1], e_rows[MAXT + 1];
  // backward arena
  size_t q_d_hgin, q_dyp;
  size_t q_defer, defer_bytes, defer_small;   // defer_small: only partial buffers up to this size join the section (0 = all)   // the arena of the backward's deferred partial-sum reduces (gt_defer_begin)
  size_t q_d_hg, q_dtok[2], q_d_hn, q_d_cls, q_d_rep, q_dA, q_dB, q_dC, q_dJ, q_dvn[4], q_ne_dw, q_bnpart[MAXL], q_heads_ws,
      q_ws[2], q_ws2, q_ws3;
  size_t bws_bytes, heads_ws_bytes, seg_ws_bytes, barena_bytes;
  int fuse_bn, bn_rows, ov;
  // backward running state (between stages)
  char* bb;
  int slot;
  void *dy, *d_vn_next, *d_h0;
};
constexpr uint32_t CTX_MAGIC = 0x67744d31u;

inline gt_gcn_layer* gcn_static(const gt_model* m) { return (gt_gcn_layer*)m->conv_layers; }
inline gt_gin_layer* gin_static(const gt_model* m) { return (gt_gin_layer*)m->conv_layers; }
inline gt_pna_layer* pna_static(const gt_model* m) { return (gt_pna_layer*)m->conv_layers; }

int model_check(const char* fn, const gt_model* m) {
  if (!m) { gt_set_error("%s: null model", fn); return GT_ERR_INVALID_ARG; }
  // (n_enc == 0 is legal in the reference -- it pools the token rows themselves -- but not built here: the callers route it to the module path)
  if (m->L < 1 || m->L > MAXL || m->n_enc < 1 || m->n_enc > MAXL || m->n_tables < 0 || m->n_tables > MAXT) {
    gt_set_error("%s: layer / table counts out of range", fn);
    return GT_ERR_INVALID_ARG;
  }
  if (!m->conv_layers || (m->n_enc && !m->enc) || (m->has_vn && m->L > 1 && !m->vn)) { gt_set_error("%s: null descriptor array", fn); return GT_ERR_INVALID_ARG; }
  if (m->D <= 0 || m->D % 4 || m->d <= 0 || m->d % 8) { gt_set_error("%s: bad widths", fn); return GT_ERR_INVALID_ARG; }
  if (m->conv != GT_CONV_GCN && m->conv != GT_CONV_GIN && m->conv != GT_CONV_PNA) { gt_set_error("%s: bad conv kind", fn); return GT_ERR_INVALID_ARG; }
  if (m->conv == GT_CONV_PNA && (m->has_vn || m->jk_cat || !m->residual || !m->pna_src || !m->pna_img || !m->pna_map || !m->pna_inv)) {
    gt_set_error("%s: the PNA stack runs without a virtual node, with JK = last, the residual connection and its weight images", fn);
    return GT_ERR_INVALID_ARG;
  }
  return GT_OK;
}

// ---- the packed token layout on the host (graphtrans_amd/graph.py:SeqLayout; modules/utils.py:5-29 + transformer_encoder.py:50-55)
// desc[b] = {row0, npos, kv_off, kv_len}; last_rows[b]; attention work list {sequence, 64-position tile}: sequences dealt to the
// eight XCD slices by length rank (longest first inside each slice), slices padded with {-1, 0} to equal size.
struct HostLayout {
  int64_t rows, max_npos, num_work;
  size_t o_last, o_work, bytes;
};
HostLayout layout_sizes(const int64_t* n, int64_t B, int64_t max_input_len, int cls, std::vector<int64_t>* kvlen_out) {
  int64_t S = 0;
  for (int64_t b = 0; b < B; ++b) S = std::max(S, n[b]);
  S = std::min(S, max_input_len);
  HostLayout h{};
  int64_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  std::vector<int64_t>& kv = *kvlen_out;
  kv.resize((size_t)B);
  for (int64_t b = 0; b < B; ++b) {
    kv[(size_t)b] = std::min(n[b], S) + cls;
    h.rows += kv[(size_t)b];
    h.max_npos = std::max(h.max_npos, kv[(size_t)b]);
  }
  // tiles per eighth need the length ranks: computed by the builder below; here an exact count through the same ranking
  std::vector<int32_t> order((size_t)B);
  for (int64_t b = 0; b < B; ++b) order[(size_t)b] = (int32_t)b;
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t c) { return kv[(size_t)a] > kv[(size_t)c]; });
  for (int64_t r = 0; r < B; ++r) cnt[r % 8] += (kv[(size_t)order[(size_t)r]] + 63) / 64;
  int64_t wpx = 0;
  for (int x = 0; x < 8; ++x) wpx = std::max(wpx, cnt[x]);
  h.num_work = B ? 8 * wpx : 0;
  const size_t nd = (size_t)B * 16, nl = (size_t)B * 8, nw = (size_t)h.num_work * 8;
  h.o_last = (nd + 15) / 16 * 16;
  h.o_work = (h.o_last + nl + 15) / 16 * 16;
  h.bytes = std::max(h.o_work + nw, (size_t)16);
  return h;
}
void layout_fill(const std::vector<int64_t>& kv, int64_t B, const HostLayout& h, char* dst) {
  int32_t* desc = (int32_t*)dst;
  int64_t* last = (int64_t*)(dst + h.o_last);
  int32_t* work = (int32_t*)(dst + h.o_work);
  int64_t row = 0;
  for (int64_t b = 0; b < B; ++b) {
    const int64_t k = kv[(size_t)b];
    desc[b * 4 + 0] = (int32_t)row;
    desc[b * 4 + 1] = (int32_t)k;
    desc[b * 4 + 2] = 0;
    desc[b * 4 + 3] = (int32_t)k;
    last[b] = row + k - 1;
    row += k;
  }
  if (!B) return;
  std::vector<int32_t> order((size_t)B);
  for (int64_t b = 0; b < B; ++b) order[(size_t)b] = (int32_t)b;
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t c) { return kv[(size_t)a] > kv[(size_t)c]; });
  const int64_t wpx = h.num_work / 8;
  for (int64_t i = 0; i < h.num_work; ++i) { work[2 * i] = -1; work[2 * i + 1] = 0; }
  for (int x = 0; x < 8; ++x) {
    int64_t pos = (int64_t)x * wpx;
    for (int64_t r = x; r < B; r += 8) {
      const int32_t s = order[(size_t)r];
      const int64_t t = (kv[(size_t)s] + 63) / 64;
      for (int64_t j = 0; j < t; ++j) { work[2 * pos] = s; work[2 * pos + 1] = (int32_t)j; ++pos; }
    }
  }
}

int bind_images(const gt_model* m, const Ctx* c) {
  if (c->in.use_w3) {
    const gt_image_set& s = c->in.use_w3 == 2 ? m->w3_enc : m->w3;
    if (s.n_bind > 0) GT_TRY(gt_w3_bind(s.n_bind, s.bind_w, s.bind_N, s.bind_K, s.bind_f, s.bind_t));
  }
  if (c->in.use_w1 && m->w1.n_bind > 0)
    GT_TRY(gt_w1_bind(m->w1.n_bind, m->w1.bind_w, m->w1.bind_N, m->w1.bind_K, m->w1.bind_f, m->w1.bind_t));
  return GT_OK;
}
struct BindGuard {   // the bind tables are per host thread: always undone on the way out
  bool dw = false;
  bool defer_abort = false;   // an error return must not leave the deferred-reduce section open on this thread (its arena dies with the step)
  ~BindGuard() {
    gt_w3_unbind();
    gt_w1_unbind();
    if (dw) gt_overlap_dw_end();
    if (defer_abort) (void)gt_defer_begin(nullptr, 0);
  }
};

size_t conv_saved_bytes(const gt_model* m, const Ctx* c, int l) {
  if (m->conv == GT_CONV_PNA) return gt_pna_layer_saved_bytes(&c->pna[l]);
  return m->conv == GT_CONV_GIN ? gt_gin_layer_saved_bytes(&c->gin[l]) : gt_gcn_layer_saved_bytes(&c->gcn[l]);
}
size_t conv_ws_bytes(const gt_model* m, const Ctx* c, int l) {
  if (m->conv == GT_CONV_PNA) return gt_pna_layer_workspace_bytes(&c->pna[l]);
  return m->conv == GT_CONV_GIN ? gt_gin_layer_workspace_bytes(&c->gin[l]) : gt_gcn_layer_workspace_bytes(&c->gcn[l]);
}

}  // namespace

extern "C" size_t gt_model_ctx_bytes(void) { return sizeof(Ctx); }

extern "C" int gt_model_grad_ranges(const gt_model* m, int64_t* lo3, int64_t* hi3) {
  GT_TRY(model_check("gt_model_grad_ranges", m));
  GT_CHECK_ARG(lo3 && hi3, "null output");
  const int64_t gnn_lo = m->has_vn ? m->off_vn_emb : m->off_conv[0];
  lo3[0] = m->off_g2t_w; hi3[0] = m->grad_total;
  lo3[1] = gnn_lo;       hi3[1] = m->off_g2t_w;
  lo3[2] = 0;            hi3[2] = gnn_lo;
  return GT_OK;
}

// =================================================================================================================================
extern "C" int gt_model_prepare(const gt_model* m, const gt_model_batch* b, void* ctx_, gt_model_sizes* out) {
  GT_TRY(model_check("gt_model_prepare", m));
  GT_CHECK_ARG(b && ctx_ && out, "null argument");
  GT_CHECK_ARG(b->N > 0 && b->B > 0 && b->E >= 0, "empty batch");
  GT_CHECK_ARG(b->x && ((b->edge_index && b->batch) || b->graph_ptr), "null batch arrays");
  Ctx* c = (Ctx*)ctx_;
  memset(c, 0, sizeof(Ctx));
  c->magic = CTX_MAGIC;
  c->in = *b;
  const int64_t N = b->N, E = b->E, B = b->B, D = m->D, d = m->d;
  const int L = m->L, nenc = m->n_enc;
  c->compute = b->compute;
  c->tdt = b->tdt;
  c->tsz = esz(b->tdt);
  const size_t tsz = c->tsz;
  const int cls = m->with_cls ? 1 : 0;

  // ---- token layout
  c->build_graph = b->graph_ptr == nullptr;
  // the prep stream pays for its stream switches and events (~0.4 ms of host time per step) only when the structure kernels are long:
  // measured r4, Code2 / Molpcba / PNA (E <= 1e5): same step time with and without, host 0.15-0.4 ms cheaper without; the
  // Erdos-Renyi stress (E = 1.05 M): 1 % faster with
  c->use_prep = (m->st_prep && b->sizes_host && c->build_graph && E >= PREP_MIN_EDGES) ? 1 : 0;
  if (b->seq_desc) {
    c->rows = b->rows; c->max_npos = b->max_npos; c->num_work = b->num_work; c->exact = b->lay_exact;
  } else if (b->sizes_host) {
    std::vector<int64_t> kv;
    const HostLayout h = layout_sizes(b->sizes_host, B, m->max_input_len, cls, &kv);
    c->rows = h.rows; c->max_npos = h.max_npos; c->num_work = h.num_work; c->exact = 1;
    c->lay_host = 1;
    c->lay_bytes = h.bytes; c->lay_o_last = h.o_last; c->lay_o_work = h.o_work;
    gt_stage_ring* ring = b->ring;
    GT_CHECK_ARG(ring && ring->base && ring->slots > 0 && ring->slots <= 64, "host-built layout needs a staging ring");
    if ((int64_t)h.bytes > ring->slot_bytes) { gt_set_error("gt_model_prepare: layout of %zu bytes exceeds the staging slot", h.bytes); return GT_ERR_WORKSPACE; }
    const int i = ring->next;
    ring->next = (i + 1) % ring->slots;
    if (ring->events[i]) (void)hipEventSynchronize((hipEvent_t)ring->events[i]);   // the copy that last read this slot has completed
    c->stage_ptr = (char*)ring->base + (size_t)i * (size_t)ring->slot_bytes;
    c->stage_event = ring->events[i];
    layout_fill(kv, B, h, (char*)c->stage_ptr);
  } else {   // sizes unknown on the host: gt_seq_layout_packed on the device, upper bounds here
    c->build_layout_dev = 1;
    c->rows = N + B * cls;
    c->max_npos = std::min<int64_t>(m->max_input_len, N) + cls;
    c->num_work = B + c->rows / 64;
    c->exact = 0;
  }
  const int64_t rows = c->rows;

  // ---- batch descriptors
  const int training = b->training ? 1 : 0;
  for (int l = 0; l < L; ++l) {
    if (m->conv == GT_CONV_GCN) {
      gt_gcn_layer& g = c->gcn[l];
      g = gcn_static(m)[l];
      g.N = N; g.E = E; g.B = B;
      g.has_vn = m->has_vn; g.relu = l != L - 1; g.residual = m->residual;
      g.training = training; g.compute = c->compute;
      g.dropout_p = training ? b->gnn_p : 0.f;
      g.seed = b->gnn_seed + SEED_STEP * (uint64_t)(2 * l + 1);
      g.edge_attr = (g.edge_mode != GT_EDGE_NONE && E > 0) ? b->edge_attr : nullptr;
      g.ev_x_ready = nullptr;
      g.ev_dx_wait = (m->st_vn && m->has_vn && l < L - 1) ? m->ev_extra[l] : nullptr;
      g.x_has_vn = m->has_vn;
      g.vn_next = nullptr; g.ev_vn_next = nullptr; g.lin_wt = nullptr;
      g.prev_saved = nullptr; g.prev_bn_w = g.prev_bn_b = nullptr; g.prev_bn_part = nullptr; g.bn_part_in = nullptr;
      g.prev_relu = 0; g.bn_nparts_in = 0; g.ev_graph_ready = nullptr;
      g.dx_bcast = nullptr; g.dx_bcast_idx = nullptr;
    } else if (m->conv == GT_CONV_PNA) {
      gt_pna_layer& g = c->pna[l];
      g = pna_static(m)[l];
      g.N = N; g.E = E;
      g.training = training; g.compute = c->compute;
      g.dropout_p = training ? b->gnn_p : 0.f;
      g.seed = b->gnn_seed + SEED_STEP * (uint64_t)(2 * l + 1);
    } else {
      gt_gin_layer& g = c->gin[l];
      g = gin_static(m)[l];
      g.N = N; g.E = E; g.B = B;
      g.has_vn = m->has_vn; g.relu = l != L - 1; g.residual = m->residual;
      g.training = training; g.compute = c->compute;
      g.dropout_p = training ? b->gnn_p : 0.f;
      g.seed = b->gnn_seed + SEED_STEP * (uint64_t)(2 * l + 1);
      g.edge_attr = (g.edge_mode != GT_EDGE_NONE && E > 0) ? b->edge_attr : nullptr;
      g.ev_x_ready = nullptr;
      g.ev_dx_wait = (m->st_vn && m->has_vn && l < L - 1) ? m->ev_extra[l] : nullptr;
      g.x_has_vn = m->has_vn;
      g.vn_next = nullptr; g.ev_vn_next = nullptr; g.w1_t = g.w2_t = nullptr;
    }
  }
  const int nvn = m->has_vn ? L - 1 : 0;
  for (int l = 0; l < nvn; ++l) {
    gt_vn_update& v = c->vn[l];
    v = m->vn[l];
    v.N = N; v.B = B;
    v.residual = m->residual; v.training = training; v.compute = c->compute;
    v.dropout_p = training ? b->gnn_p : 0.f;
    v.seed = b->gnn_seed + SEED_STEP * (uint64_t)(2 * l + 2);
    v.identity_graph = b->ident_B;
    v.ev_dx_done = (m->st_vn && m->vn_defer_dw) ? m->ev_extra[l] : nullptr;
  }
  for (int i = 0; i < nenc; ++i) {
    gt_encoder_layer& e = c->enc[i];
    e = m->enc[i];
    e.rows = rows;
    e.dtype = c->tdt; e.compute = c->compute; e.training = training;
    e.num_seqs = B; e.row_stride = 1; e.max_npos = c->max_npos;
    e.num_work = c->num_work;
    e.dropout_p = training ? b->enc_p : 0.f;
    e.seed = b->enc_seed + SEED_STEP * (uint64_t)(i + 1);
  }

  // ---- forward arena
  Bump a;
  const size_t ND4 = (size_t)N * D * 4;
  for (int l = 0; l <= L; ++l) c->o_h[l] = a.take(ND4);
  if (m->has_vn) {
    c->o_x0 = a.take(m->vn0_in_embed ? 0 : ND4);
    for (int l = 0; l < L; ++l) c->o_vn[l] = a.take((size_t)B * D * 4);
    for (int l = 0; l < nvn; ++l) c->o_vn_saved[l] = a.take(gt_vn_update_saved_bytes(&c->vn[l]));
  }
  for (int l = 0; l < L; ++l) c->o_conv_saved[l] = a.take(conv_saved_bytes(m, c, l));
  const int64_t Kc = m->jk_cat ? 2 * D : D;
  c->Kc = Kc;
  // JK = "cat" without its copy needs the bound image of gnn2transformer's weight (thread-local table: bound for the question)
  c->cat2 = 0;
  if (m->jk_cat && b->use_w3) {
    BindGuard guard;
    GT_TRY(bind_images(m, c));
    c->cat2 = gt_linear_cat2_ok(c->compute, m->g2t_w, N, d, D, D) ? 1 : 0;
  }
  c->o_cat = a.take((m->jk_cat && !c->cat2) ? (size_t)N * Kc * 4 : 0);
  c->o_hn = a.take((size_t)N * d * tsz);
  c->o_rowmap = a.take((size_t)N * 4);
  c->o_tok = a.take((size_t)rows * d * tsz);
  if (m->nin_w) {
    c->o_xin = a.take((size_t)rows * d * tsz);
    c->o_st0 = a.take((size_t)2 * rows * 4);
  }
  // the LAST encoder layer is the pooled variant (gt_encoder_layer_pooled_*): cls / last pooling reads one row per sequence of it
  for (int i = 0; i < nenc; ++i) c->o_xe[i] = a.take((size_t)(i == nenc - 1 ? B : rows) * d * tsz);
  for (int i = 0; i < nenc; ++i)
    c->o_enc_saved[i] = a.take(i == nenc - 1 ? gt_encoder_layer_pooled_saved_bytes(&c->enc[i]) : gt_encoder_layer_saved_bytes(&c->enc[i]));
  if (m->nout_w) {   // the final norm runs on the POOLED rows only (cls / last pooling reads nothing else of it: gnn_transformer.py:113-114)
    c->o_hgin = a.take((size_t)B * d * 4);
    c->o_sto = a.take((size_t)2 * B * 4);
  }
  c->T = m->embed_kind != 1 ? m->n_tables : 0;
  for (int t = 0; t < c->T; ++t) c->e_rows[t] = m->table_rows[t];
  c->esort = (b->will_bwd && m->embed_sorted && m->embed_kind != 1) ? 1 : 0;
  if (c->esort) {
    c->eplan_bytes = gt_embed_sort_plan_bytes(c->T, c->e_rows, N);
    c->esort_ws_bytes = gt_embed_sort_workspace_bytes(c->T, c->e_rows, N);
    c->o_eplan = a.take(c->eplan_bytes);
    c->o_esort_ws = a.take(c->esort_ws_bytes);
  }
  const int64_t Kp = (int64_t)c4((size_t)m->ne_K);
  if (m->embed_kind == 1 && Kp != m->ne_K) {
    c->o_ne_x = a.take((size_t)N * Kp * 4);
    c->o_ne_w = a.take((size_t)D * Kp * 4);
  }
  c->o_hg = a.take((size_t)B * d * 4);
  if (m->conv == GT_CONV_PNA) c->o_scales = a.take((size_t)N * c->pna[0].S * 4);
  size_t ws = 256, ws2 = 256;
  for (int l = 0; l < L; ++l) ws = std::max(ws, conv_ws_bytes(m, c, l));
  for (int l = 0; l < nvn; ++l) ws2 = std::max(ws2, gt_vn_update_workspace_bytes(&c->vn[l]));
  ws = std::max(ws, ws2);
  c->ws_bytes = ws; c->ws2_bytes = ws2;
  c->o_ws = a.take(ws);
  c->o_ws2 = a.take(ws2);
  const bool have_imgs = b->use_w3 != 0;
  c->want_wt = (m->conv != GT_CONV_PNA && b->will_bwd && c->compute == GT_F32 && N >= 1024 && (!have_imgs || (!m->has_vn && m->conv == GT_CONV_GCN))) ? 1 : 0;
  if (c->want_wt) {
    for (int l = 0; l < L; ++l) c->o_wt[l] = a.take((size_t)(m->conv == GT_CONV_GIN ? 2 : 1) * 2 * D * D * 4);
    c->o_g2t_wt = a.take((size_t)d * Kc * 4);
  }
  if (c->build_graph) {
    c->o_graph_ptr = a.take((size_t)(B + 1) * 4);
    c->o_node_graph = a.take((size_t)N * 4);
    c->o_in_ptr = a.take((size_t)(N + 1) * 4);
    c->o_out_ptr = a.take((size_t)(N + 1) * 4);
    c->o_idx = a.take((size_t)4 * std::max<int64_t>(E, 1) * 4);
    c->o_dd = a.take((size_t)2 * N * 4);
    c->o_status = a.take(4);
    c->prep_ws_bytes = gt_graph_prep_workspace_bytes(N, E, B);
    c->o_prep_ws = a.take(c->prep_ws_bytes);
  }
  if (c->lay_host) c->o_lay = a.take(c->lay_bytes);
  if (c->build_layout_dev) {
    const size_t nd = (size_t)B * 16, nl = (size_t)B * 8, nw = (size_t)std::max<int64_t>(c->num_work, 1) * 8;
    c->lay_o_last = (nd + 15) / 16 * 16;
    c->lay_o_work = (c->lay_o_last + nl + 15) / 16 * 16;
    c->lay_bytes = c->lay_o_work + nw;
    c->o_lay = a.take(c->lay_bytes);
    c->o_lay_meta = a.take(16);
  }
  c->arena_bytes = std::max(a.off, (size_t)256);

  // ---- backward arena
  Bump q;
  c->q_d_hg = q.take((size_t)B * d * 4);
  c->q_d_hgin = q.take(m->nout_w ? (size_t)B * d * 4 : 0);
  c->q_dyp = q.take((size_t)B * d * tsz);
  c->q_dtok[0] = q.take((size_t)rows * d * tsz);
  c->q_dtok[1] = q.take((size_t)rows * d * tsz);
  c->q_d_hn = q.take((size_t)N * d * tsz);
  c->q_d_cls = q.take((size_t)B * d * tsz);
  c->q_d_rep = q.take((size_t)N * Kc * 4);
  c->q_dA = q.take(ND4); c->q_dB = q.take(ND4); c->q_dC = q.take(ND4);
  c->q_dJ = q.take(m->jk_cat ? ND4 : 0);
  for (int i = 0; i < 4; ++i) c->q_dvn[i] = q.take((size_t)B * D * 4);
  size_t enc_ws = 256;
  for (int i = 0; i < nenc; ++i)
    enc_ws = std::max(enc_ws, i == nenc - 1 ? gt_encoder_layer_pooled_workspace_bytes(&c->enc[i]) : gt_encoder_layer_workspace_bytes(&c->enc[i]));
  const size_t ln_ws = std::max(gt_layernorm_bwd_workspace_bytes(rows, d), gt_layernorm_bwd_workspace_bytes(B, d));
  const size_t lin_ws = std::max(gt_linear_bwd_workspace_bytes(c->compute, B, m->Nh, d), gt_linear_bwd_workspace_bytes(c->compute, N, d, Kc));
  size_t emb_ws;
  if (m->embed_kind == 1) {
    emb_ws = gt_linear_bwd_workspace_bytes(c->compute, N, D, Kp);
    c->q_ne_dw = q.take(Kp != m->ne_K ? (size_t)D * Kp * 4 : 0);
  } else if (c->esort) {
    emb_ws = gt_embed_sum_bwd_sorted_workspace_bytes(c->T, N, D);
  } else {
    emb_ws = gt_embed_sum_bwd_workspace_bytes(c->T, c->e_rows, D);
  }
  c->bws_bytes = std::max(std::max(std::max(c->ws_bytes, enc_ws), std::max(ln_ws, lin_ws)), emb_ws);
  // BatchNorm-backward statistics summed in the dX epilogue of the layer above (GCN, exact-fp32 GEMMs, no GNN dropout, no virtual node)
  c->fuse_bn = 0;
  c->bn_rows = 0;
  // (not beyond 64 k rows: the statistics ride in the EXACT-fp32 dX kernel -- at the Erdos-Renyi stress' 131 k x 256 x 256 that GEMM is
  // 219 us against 116 us for the bf16x6 kernel + a 50-us partial pass: 17.6 k -> 17.9 k graphs/s without)
  if (L > 1 && !b->sync_bn && m->conv == GT_CONV_GCN && training && c->gcn[0].dropout_p == 0.f &&
      gt_linear_bwd_bnstats_ok(c->compute, GT_F32, GT_F32, N)) {
    // which kernel will run the layers' dX GEMMs: the register-row bf16x6 kernel (bound images, N >= 12288: one partial row per 128
    // rows, any N, with or without a virtual node -- its epilogue holds the complete d x_l, virtual-node rows included) or the
    // exact-fp32 one (64-row tiles; without a virtual node and up to 64 k rows only, see above)
    int64_t rows = gt_linear_bwd_bnstats_rows(N);
    bool on_rows_kernel = false;
    if (b->use_w3) {
      BindGuard guard;
      GT_TRY(bind_images(m, c));
      rows = gt_linear_bwd_bnstats_rows_for(c->compute, GT_F32, GT_F32, c->gcn[1].lin_w, N, D, D);
      on_rows_kernel = rows != gt_linear_bwd_bnstats_rows(N);
    }
    // (the register-row kernel answers only with gt_option_set("bnstats_rows_kernel", 1) -- measured r5, Code2 b256: statistics in its epilogue 74.0 k graphs/s,
    // the separate partial pass 74.85 k; ER 19.07 k against 19.14 k: the epilogue's second row read + 320 lane shuffles per wave
    // cost more than the 23-us pass they replace)
    if (rows > 0 && (on_rows_kernel || (!m->has_vn && N <= 65536))) {
      c->fuse_bn = 1;
      c->bn_rows = (int)rows;
    }
  }
  for (int l = 0; l + 1 < L; ++l) c->q_bnpart[l] = q.take(c->fuse_bn ? (size_t)c->bn_rows * 2 * D * 4 : 0);
  c->heads_ws_bytes = gt_linear_bwd_workspace_bytes(c->compute, B, m->Nh, d);
  c->q_heads_ws = q.take(c->heads_ws_bytes);
  c->q_ws[0] = q.take(c->bws_bytes);
  c->q_ws[1] = q.take(c->bws_bytes);
  c->q_ws2 = q.take(c->ws2_bytes);
  c->seg_ws_bytes = m->has_vn ? gt_segment_sum_workspace_bytes(N, D) : 0;
  c->q_ws3 = q.take(c->seg_ws_bytes);
  c->q_dimg = q.take(m->conv == GT_CONV_PNA ? (size_t)m->pna_n_img * 4 : 0);
  {   // room for every weight-gradient GEMM's partials and every LayerNorm's block partials of one backward (an upper bound: a producer
      // that finds the arena full reduces on the spot)
    size_t need = lin_ws + emb_ws + (size_t)(2 * nenc + 2) * ln_ws;
    for (int l = 0; l < L; ++l) need += (m->conv == GT_CONV_GIN ? 2 : 1) * (m->conv == GT_CONV_PNA ? 0 : conv_ws_bytes(m, c, l));
    for (int i = 0; i < nenc; ++i) {
      const gt_encoder_layer& e = c->enc[i];
      const int ec = e.dtype == GT_BF16 ? GT_BF16 : e.compute;
      need += gt_linear_bwd_workspace_bytes(ec, e.rows, 3 * e.d_model, e.d_model) + gt_linear_bwd_workspace_bytes(ec, e.rows, e.d_model, e.d_model) +
              gt_linear_bwd_workspace_bytes(ec, e.rows, e.ffn, e.d_model) + gt_linear_bwd_workspace_bytes(ec, e.rows, e.d_model, e.ffn);
    }
    constexpr int64_t defer_max = 6000000;
    // (see gt_model_backward: the big batches keep the GEMMs' immediate reduces and defer the LayerNorms' column sums only)
    c->defer_small = N * D > defer_max && nenc > 0 ? ln_ws : 0;
    c->defer_bytes = N * D <= defer_max ? need + 64 * 256 : (c->defer_small ? (size_t)(2 * nenc + 2) * (ln_ws + 256) + 64 * 256 : 0);
    c->q_defer = q.take(c->defer_bytes);
  }
  c->barena_bytes = std::max(q.off, (size_t)256);

  c->prepared = 1;
  out->rows = c->rows; out->max_npos = c->max_npos; out->num_work = c->num_work;
  out->arena_bytes = (int64_t)c->arena_bytes; out->barena_bytes = (int64_t)c->barena_bytes;
  out->exact = c->exact; out->pad_ = 0;
  return GT_OK;
}

// =================================================================================================================================
extern "C" int gt_model_forward(const gt_model* m, void* ctx_, void* arena, float* logits, gt_stream_t st) {
  GT_TRY(model_check("gt_model_forward", m));
  Ctx* c = (Ctx*)ctx_;
  GT_CHECK_ARG(c && c->magic == CTX_MAGIC && c->prepared && !c->forwarded, "context not prepared (or used twice)");
  GT_CHECK_ARG(arena && logits, "null buffer");
  const gt_model_batch& b = c->in;
  const int64_t N = b.N, E = b.E, B = b.B, D = m->D, d = m->d, rows = c->rows;
  const int L = m->L, nenc = m->n_enc, tdt = c->tdt, compute = c->compute;
  const int nvn = m->has_vn ? L - 1 : 0;
  char* base = (char*)arena;
  c->base = base;
  c->forwarded = 1;
  auto P = [&](size_t off) -> void* { return base + off; };
  BindGuard guard;
  GT_TRY(bind_images(m, c));

  // ---- graph structure (gt_graph_prep) beside the first kernels, on the prep stream
  gt_stream_t pst = c->use_prep ? m->st_prep : st;
  void* ev_graph = nullptr;
  const int32_t *graph_ptr, *node_graph, *in_ptr, *in_src, *in_eid, *out_ptr, *out_dst, *out_eid;
  const float *deg, *dis;
  if (c->build_graph) {
    int32_t* idx = (int32_t*)P(c->o_idx);
    const int64_t Ep = std::max<int64_t>(E, 1);
    float* dd = (float*)P(c->o_dd);
    graph_ptr = (int32_t*)P(c->o_graph_ptr); node_graph = (int32_t*)P(c->o_node_graph);
    in_ptr = (int32_t*)P(c->o_in_ptr); out_ptr = (int32_t*)P(c->o_out_ptr);
    in_src = idx; in_eid = idx + Ep; out_dst = idx + 2 * Ep; out_eid = idx + 3 * Ep;
    deg = dd; dis = dd + N;
    if (c->use_prep) {
      GT_TRY(gt_event_record(m->ev_prep_begin, st));
      GT_TRY(gt_stream_wait_event(pst, m->ev_prep_begin));
    }
    GT_TRY(gt_graph_prep(b.edge_index, b.batch, N, E, B, (int32_t*)graph_ptr, (int32_t*)node_graph, (int32_t*)in_ptr, (int32_t*)in_src,
                         (int32_t*)in_eid, (int32_t*)out_ptr, (int32_t*)out_dst, (int32_t*)out_eid, (float*)deg, (float*)dis,
                         (int32_t*)P(c->o_status), P(c->o_prep_ws), c->prep_ws_bytes, pst));
    if (c->use_prep) {
      GT_TRY(gt_event_record(m->ev_graph, pst));
      ev_graph = m->ev_graph;
    }
  } else {
    graph_ptr = b.graph_ptr; node_graph = b.node_graph; in_ptr = b.in_ptr; in_src = b.in_src; in_eid = b.in_eid;
    out_ptr = b.out_ptr; out_dst = b.out_dst; out_eid = b.out_eid; deg = b.deg; dis = b.dis;
  }
  c->graph_ptr = graph_ptr; c->node_graph = node_graph;
  // ---- token layout
  if (b.seq_desc) {
    c->seq_desc = b.seq_desc; c->last_rows = b.last_rows; c->work_items = b.work_items;
  } else {
    char* lay = (char*)P(c->o_lay);
    c->seq_desc = (const int32_t*)lay;
    c->last_rows = (const int64_t*)(lay + c->lay_o_last);
    c->work_items = c->num_work ? (const int32_t*)(lay + c->lay_o_work) : nullptr;
    if (c->lay_host) {
      if (hipMemcpyAsync(lay, c->stage_ptr, c->lay_bytes, hipMemcpyHostToDevice, (hipStream_t)st) != hipSuccess) {
        gt_set_error("gt_model_forward: layout copy failed");
        return GT_ERR_LAUNCH;
      }
      if (c->stage_event) GT_TRY(gt_event_record(c->stage_event, st));
    } else {
      GT_TRY(gt_seq_layout_packed(graph_ptr, B, m->max_input_len, m->with_cls ? 1 : 0, (int32_t*)lay, (int64_t*)(lay + c->lay_o_last),
                                  (int32_t*)(lay + c->lay_o_work), c->num_work, (int32_t*)P(c->o_lay_meta), st));
    }
  }
  for (int i = 0; i < nenc; ++i) { c->enc[i].seq_desc = c->seq_desc; c->enc[i].work_items = c->work_items; }
  if (m->conv == GT_CONV_PNA)   // the towers' re-stacked weights (the optimizer changed them: one launch for all layers); their
    GT_TRY(gt_gather_f32(m->pna_img, m->pna_src, m->pna_map, m->pna_n_img, st));   // bf16x3 images are built from these just below
  // ---- weight images (the weights changed since the last step: one launch each)
  if (b.use_w3) {
    const gt_image_set& s = b.use_w3 == 2 ? m->w3_enc : m->w3;
    if (s.n_jobs > 0) GT_TRY(gt_w3_images(s.n_jobs, s.job_w, s.job_N, s.job_K, s.job_T, s.job_img, st));
  }
  void* ev_w1 = nullptr;
  if (b.use_w1 && m->w1.n_jobs > 0) {
    GT_TRY(gt_w1_images(m->w1.n_jobs, m->w1.job_w, m->w1.job_N, m->w1.job_K, m->w1.job_T, m->w1.job_img, pst));
    if (c->use_prep) {
      GT_TRY(gt_event_record(m->ev_w1, pst));
      ev_w1 = m->ev_w1;
    }
  }
  // ---- graph pointers into the descriptors
  for (int l = 0; l < L; ++l) {
    if (m->conv == GT_CONV_GCN) {
      gt_gcn_layer& g = c->gcn[l];
      g.graph_ptr = graph_ptr; g.node_graph = node_graph; g.in_ptr = in_ptr; g.in_src = in_src; g.in_eid = in_eid;
      g.out_ptr = out_ptr; g.out_dst = out_dst; g.out_eid = out_eid; g.deg = deg; g.dis = dis;
    } else if (m->conv == GT_CONV_PNA) {
      gt_pna_layer& g = c->pna[l];
      g.in_ptr = in_ptr; g.in_src = in_src; g.in_eid = in_eid; g.out_ptr = out_ptr; g.out_dst = out_dst; g.out_eid = out_eid;
      g.scales = (const float*)P(c->o_scales);
    } else {
      gt_gin_layer& g = c->gin[l];
      g.graph_ptr = graph_ptr; g.node_graph = node_graph; g.in_ptr = in_ptr; g.in_src = in_src; g.in_eid = in_eid;
      g.out_ptr = out_ptr; g.out_dst = out_dst; g.out_eid = out_eid;
    }
  }
  for (int l = 0; l < nvn; ++l) { c->vn[l].graph_ptr = graph_ptr; c->vn[l].node_graph = node_graph; }

  gt_stream_t side = m->has_vn ? m->st_vn : nullptr;
  // ---- transposed weights for the backward's exact-fp32 dX GEMMs, written beside the forward
  c->g2t_wt = nullptr;
  if (c->want_wt) {
    gt_stream_t tst = st;
    if (m->st_dw) {
      tst = m->st_dw;
      GT_TRY(gt_event_record(m->ev_wt[0], st));
      GT_TRY(gt_stream_wait_event(tst, m->ev_wt[0]));
    }
    for (int l = 0; l < L; ++l) {
      float* wt = (float*)P(c->o_wt[l]);
      if (m->conv == GT_CONV_GIN) {
        gt_gin_layer& g = c->gin[l];
        g.w1_t = wt; g.w2_t = wt + 2 * D * D;
        GT_TRY(gt_transpose(g.w1, wt, 2 * D, D, tst));
        GT_TRY(gt_transpose(g.w2, wt + 2 * D * D, D, 2 * D, tst));
      } else {
        c->gcn[l].lin_wt = wt;
        GT_TRY(gt_transpose(c->gcn[l].lin_w, wt, D, D, tst));
      }
    }
    c->g2t_wt = (float*)P(c->o_g2t_wt);
    GT_TRY(gt_transpose(m->g2t_w, (float*)P(c->o_g2t_wt), d, c->Kc, tst));
    if (m->st_dw) GT_TRY(gt_event_record(m->ev_wt[1], tst));
  }
  // ---- input encoder   (dataset/utils.py:28-30 / ogb AtomEncoder / nn.Linear(F, D) dataset/tud.py:65)
  const int T = c->T;
  c->ne_x = c->ne_w = nullptr;
  if (m->embed_kind == 1) {
    const int64_t K = m->ne_K, Kp = (int64_t)c4((size_t)K);
    const float *nx = (const float*)b.x, *nw = m->ne_w;
    if (Kp != K) {
      GT_TRY(gt_repitch(P(c->o_ne_x), Kp, b.x, K, N, 4, st));
      GT_TRY(gt_repitch(P(c->o_ne_w), Kp, m->ne_w, K, D, 4, st));
      nx = (const float*)P(c->o_ne_x); nw = (const float*)P(c->o_ne_w);
    }
    c->ne_x = nx; c->ne_w = nw;
    GT_TRY(gt_linear_fwd(GT_F32, GT_F32, compute, nx, nw, m->ne_b, P(c->o_h[0]), N, D, Kp, 0, 0.f, 0, st));
  } else {
    const int64_t* x = (const int64_t*)b.x;
    if (m->embed_kind == 2) {   // ASTNodeEncoder: type, attribute, clamped depth
      GT_CHECK_ARG(T == 3 && b.node_depth, "ASTNodeEncoder has three tables and needs node_depth");
      c->e_idx[0] = x; c->e_str[0] = b.x_stride0;
      c->e_idx[1] = x + b.x_stride1; c->e_str[1] = b.x_stride0;
      c->e_idx[2] = b.node_depth; c->e_str[2] = b.depth_stride;
    } else {
      for (int t = 0; t < T; ++t) { c->e_idx[t] = x + (int64_t)t * b.x_stride1; c->e_str[t] = b.x_stride0; }
    }
    const float* tabs[MAXT + 1];
    for (int t = 0; t < T; ++t) { c->e_clamp[t] = m->table_clamp[t]; tabs[t] = m->tables[t]; }
    int Tn = T;
    if (m->vn0_in_embed) {   // + virtualnode_embedding.weight[0] for every node (gnn_module.py:195,199): one more table, stride-0 index
      c->e_idx[T] = m->zero_i64; c->e_str[T] = 0; c->e_clamp[T] = -1; tabs[T] = m->vn_emb;
      Tn = T + 1;
    }
    GT_TRY(gt_embed_sum_fwd(Tn, c->e_idx, c->e_str, c->e_clamp, tabs, N, D, (float*)P(c->o_h[0]), st));
    if (c->esort) {   // node ids per table row for the backward: beside the forward, only the index columns are read
      gt_stream_t sst = st;
      if (m->st_dw) {
        sst = m->st_dw;
        GT_TRY(gt_event_record(m->ev_sort[0], st));
        GT_TRY(gt_stream_wait_event(sst, m->ev_sort[0]));
      }
      GT_TRY(gt_embed_sort(T, c->e_idx, c->e_str, c->e_clamp, c->e_rows, N, P(c->o_eplan), c->eplan_bytes, P(c->o_esort_ws),
                           c->esort_ws_bytes, sst));
      if (m->st_dw) GT_TRY(gt_event_record(m->ev_sort[1], sst));
    }
  }
  // ---- the structure may still be in the making on the prep stream: GCN layer 0 waits between its GEMM and its aggregate,
  // the virtual-node stream before its first segment sum; every other configuration right here
  const bool late_wait = ev_graph && m->conv == GT_CONV_GCN && (!m->has_vn || m->vn0_in_embed);
  if (m->conv == GT_CONV_GCN) c->gcn[0].ev_graph_ready = late_wait ? ev_graph : nullptr;
  if (ev_graph) {
    if (!late_wait) GT_TRY(gt_stream_wait_event(st, ev_graph));
    else if (side) GT_TRY(gt_stream_wait_event(side, ev_graph));
  }
  // ---- message passing   (modules/gnn_module.py:181-224)
  auto X = [&](int l) -> void* {
    if (!m->has_vn) return P(c->o_h[l]);
    return (l == 0 && !m->vn0_in_embed) ? P(c->o_x0) : P(c->o_h[l]);
  };
  if (m->conv == GT_CONV_PNA)   // per-node degree scalers of this batch (in-degrees: the structure is ready on this stream here)
    GT_TRY(gt_pna_scales(in_ptr, N, c->pna[0].S, m->pna_kinds, m->pna_avg_log, m->pna_avg_lin, (float*)P(c->o_scales), st));
  auto conv_fwd = [&](int l, const void* h_in, const void* vn, void* y) -> int {
    if (m->conv == GT_CONV_PNA) return gt_pna_layer_fwd(&c->pna[l], h_in, y, P(c->o_conv_saved[l]), P(c->o_ws), c->ws_bytes, st);
    if (m->conv == GT_CONV_GIN)
      return gt_gin_layer_fwd(&c->gin[l], h_in, vn, nullptr, y, P(c->o_conv_saved[l]), P(c->o_ws), c->ws_bytes, st);
    return gt_gcn_layer_fwd(&c->gcn[l], h_in, vn, nullptr, y, P(c->o_conv_saved[l]), P(c->o_ws), c->ws_bytes, st);
  };
  if (m->has_vn)
    GT_TRY(gt_segment_bcast_add(GT_F32, nullptr, m->vn_emb, b.zeros_B, B, 1, D, P(c->o_vn[0]), st));
  for (int l = 0; l < L; ++l) {
    if (m->has_vn) {
      const bool last = l == L - 1;
      const void* vn_next = last ? nullptr : P(c->o_vn[l + 1]);
      void* ev_next = (!last && side) ? m->ev_vn[l] : nullptr;
      if (m->conv == GT_CONV_GIN) { c->gin[l].vn_next = vn_next; c->gin[l].ev_vn_next = ev_next; }
      else { c->gcn[l].vn_next = vn_next; c->gcn[l].ev_vn_next = ev_next; }
      if (l == 0 && !m->vn0_in_embed)   // Linear node encoder: x_0 = h_0 + vn_0[batch] as its own pass
        GT_TRY(gt_segment_bcast_add(GT_F32, P(c->o_h[0]), P(c->o_vn[0]), node_graph, N, B, D, P(c->o_x0), st));
      if (!last && side) {
        // vn_{l+1} beside layer l's GEMM / aggregate on the second stream; layer l's apply pass waits for it (ev_vn_next)
        GT_TRY(gt_event_record(m->ev_x[l], st));
        GT_TRY(gt_stream_wait_event(side, m->ev_x[l]));
        GT_TRY(gt_vn_update_fwd(&c->vn[l], X(l), P(c->o_vn[l]), P(c->o_vn[l + 1]), P(c->o_vn_saved[l]), P(c->o_ws2), c->ws2_bytes, side));
        GT_TRY(gt_event_record(m->ev_vn[l], side));
      } else if (!last) {
        GT_TRY(gt_vn_update_fwd(&c->vn[l], X(l), P(c->o_vn[l]), P(c->o_vn[l + 1]), P(c->o_vn_saved[l]), P(c->o_ws), c->ws_bytes, st));
      }
      GT_TRY(conv_fwd(l, X(l), P(c->o_vn[l]), P(c->o_h[l + 1])));
    } else {
      GT_TRY(conv_fwd(l, P(c->o_h[l]), nullptr, P(c->o_h[l + 1])));
    }
    c->xptr[l] = X(l);
  }
  c->first = X(0);   // h_list[0] after the in-place virtual-node add
  c->h_last = P(c->o_h[L]);
  if (ev_w1) GT_TRY(gt_stream_wait_event(st, ev_w1));   // the encoder's weight images were built on the prep stream
  // ---- gnn2transformer + token rows + encoder   (models/gnn_transformer.py:92-114)
  // the GEMM's epilogue writes the token rows itself (row map; CLS rows by the map's kernel) when the layout is the packed one built
  // here and the bf16x6 kernel runs it: no pad pass over the node rows (modules/utils.py:5-29), no [N][d] intermediate
  c->fuse_rows = (!b.seq_desc && gt_linear_rows_ok(compute, GT_F32, tdt, m->g2t_w, N, d, c->Kc)) ? 1 : 0;
  void* g2t_out = P(c->o_hn);
  // ... and norm_input (transformer_encoder.py:53-57) in the same epilogue when a token row fills one column block of the kernel
  const bool fuse_nin = c->fuse_rows && m->nin_w && gt_linear_rows_layernorm_ok(d);
  if (c->fuse_rows) {
    int32_t* rmap = (int32_t*)P(c->o_rowmap);
    if (fuse_nin) {
      float* st0 = (float*)P(c->o_st0);
      GT_TRY(gt_seq_token_rows_layernorm(tdt, m->cls, graph_ptr, node_graph, c->seq_desc, B, 1, m->with_cls ? 1 : 0, N, d, P(c->o_tok), rmap, m->nin_w,
                                         m->nin_b, m->nin_eps, P(c->o_xin), st0, st0 + rows, st));
      GT_TRY(gt_linear_set_rows_layernorm(rmap, m->nin_w, m->nin_b, m->nin_eps, P(c->o_xin), st0, st0 + rows));
    } else {
      GT_TRY(gt_seq_token_rows(tdt, m->cls, graph_ptr, node_graph, c->seq_desc, B, 1, m->with_cls ? 1 : 0, N, d, P(c->o_tok), rmap, st));
      GT_TRY(gt_linear_set_rows(rmap));
    }
    g2t_out = P(c->o_tok);
  }
  if (c->cat2) {
    c->node_rep = nullptr;
    GT_TRY(gt_linear_fwd_cat2(tdt, compute, c->first, D, D, c->h_last, D, D, m->g2t_w, m->g2t_b, g2t_out, N, d, d, st));
  } else {
    if (m->jk_cat) {   // torch.cat([h_list[0], h_list[-1]], 1)   (gnn_module.py:104-105)
      GT_TRY(gt_copy2d(P(c->o_cat), c->Kc * 4, c->first, D * 4, D * 4, N, st));
      GT_TRY(gt_copy2d((char*)P(c->o_cat) + D * 4, c->Kc * 4, c->h_last, D * 4, D * 4, N, st));
      c->node_rep = P(c->o_cat);
    } else {
      c->node_rep = c->h_last;
    }
    GT_TRY(gt_linear_fwd(GT_F32, tdt, compute, c->node_rep, m->g2t_w, m->g2t_b, g2t_out, N, d, c->Kc, 0, 0.f, 0, st));
  }
  if (!c->fuse_rows)
    GT_TRY(gt_seq_gather_cls32(tdt, P(c->o_hn), m->cls, graph_ptr, c->seq_desc, B, 1, c->max_npos, m->with_cls ? 1 : 0, d, P(c->o_tok), st));
  const void* cur = P(c->o_tok);
  if (fuse_nin) {
    cur = P(c->o_xin);
  } else if (m->nin_w) {
    GT_TRY(gt_layernorm_fwd(tdt, cur, nullptr, m->nin_w, m->nin_b, m->nin_eps, 0.f, 0, rows, d, P(c->o_xin), (float*)P(c->o_st0),
                            (float*)P(c->o_st0) + rows, st));
    cur = P(c->o_xin);
  }
  for (int i = 0; i < nenc; ++i) {
    c->enc_in[i] = cur;
    if (i == nenc - 1) {   // only the pooled row of every sequence is read behind the last layer: y = [B][d]
      GT_TRY(gt_encoder_layer_pooled_fwd(&c->enc[i], cur, c->last_rows, P(c->o_xe[i]), P(c->o_enc_saved[i]), st));
    } else {
      GT_TRY(gt_encoder_layer_fwd(&c->enc[i], cur, P(c->o_xe[i]), P(c->o_enc_saved[i]), st));
    }
    cur = P(c->o_xe[i]);
  }
  c->pre_out = cur;   // [B][d]: the pooled rows
  if (m->nout_w) {
    // transformer.norm (transformer_encoder.py:28-32) is row-wise: on the pooled rows (fp32 copies), forward and backward
    GT_TRY(gt_rows_gather(tdt, cur, nullptr, B, d, (float*)P(c->o_hgin), st));
    GT_TRY(gt_layernorm_fwd(GT_F32, P(c->o_hgin), nullptr, m->nout_w, m->nout_b, m->nout_eps, 0.f, 0, B, d, P(c->o_hg), (float*)P(c->o_sto),
                            (float*)P(c->o_sto) + B, st));
  } else {
    GT_TRY(gt_rows_gather(tdt, cur, nullptr, B, d, (float*)P(c->o_hg), st));
  }
  // ---- prediction heads as one GEMM over the stacked weights   (gnn_transformer.py:120-126)
  GT_TRY(gt_linear_fwd_ld(GT_F32, GT_F32, compute, P(c->o_hg), m->head_w, m->head_b, logits, B, m->Nh, d, m->ldy, 0, 0.f, 0, st));
  // the side streams wrote into this arena: join them before anything can hand it back to the allocator
  if (c->esort && m->st_dw) GT_TRY(gt_stream_wait_event(st, m->ev_sort[1]));
  else if (c->want_wt && m->st_dw) GT_TRY(gt_stream_wait_event(st, m->ev_wt[1]));
  (void)E;
  return GT_OK;
}

// =================================================================================================================================
extern "C" int gt_model_backward(const gt_model* m, void* ctx_, const float* dlogits, float* grads, void* barena, int stages,
                                 gt_stream_t st) {
  GT_TRY(model_check("gt_model_backward", m));
  Ctx* c = (Ctx*)ctx_;
  GT_CHECK_ARG(c && c->magic == CTX_MAGIC && c->forwarded, "context without a forward");
  GT_CHECK_ARG(dlogits && grads && barena, "null buffer");
  GT_CHECK_ARG(stages > 0 && stages < 8 && !(stages & c->stages_done), "bad stage mask");
  GT_CHECK_ARG(!(stages & 2) || ((stages | c->stages_done) & 1), "stage order");
  GT_CHECK_ARG(!(stages & 4) || ((stages | c->stages_done) & 2), "stage order");
  const gt_model_batch& b = c->in;
  const int64_t N = b.N, B = b.B, D = m->D, d = m->d, rows = c->rows, Kc = c->Kc;
  const int L = m->L, nenc = m->n_enc, tdt = c->tdt, compute = c->compute;
  char* base = c->base;
  char* bb = (char*)barena;
  GT_CHECK_ARG(!c->stages_done || bb == c->bb, "the stages of one backward share one arena");
  c->bb = bb;
  auto P = [&](size_t off) -> void* { return base + off; };
  auto Q = [&](size_t off) -> void* { return bb + off; };
  float* G = grads;
  gt_stream_t side = m->has_vn ? m->st_vn : nullptr;
  const size_t ws_bytes = c->bws_bytes;

  BindGuard guard;
  GT_TRY(bind_images(m, c));
  // the overlap stream pays for itself only when the kernels are long enough to hide its extra stream operations
  const bool ov = m->st_dw && N * D >= m->dw_overlap_min_elems;
  c->ov = ov;
  if (ov) {
    if (!c->stages_done) GT_TRY(gt_overlap_dw_begin(st, m->st_dw));
    guard.dw = (stages & 4) != 0;   // the section stays open between the stage calls of one backward (same host thread)
  }
  // the sums over weight-gradient / LayerNorm partials of a stage: queued by their producers, ONE launch at the end of the stage
  // (section open across the stage calls like the overlap section; the arena is not reused inside one backward)
  // (up to ~6 M node-row elements: where the step is made of launches it gains them -- Molpcba +1 %, NCI1 +2 %, PNA b128 +1.2 % --;
  // at Code2 b256 (9.5 M) the arena copies of the partials are cold memory where the reused workspaces stay in the Infinity Cache:
  // -0.9 %, so the big batches keep the immediate reduces)
  if (!c->stages_done) {
    GT_TRY(gt_defer_begin(c->defer_bytes ? Q(c->q_defer) : nullptr, c->defer_bytes));
    if (c->defer_bytes && c->defer_small) GT_TRY(gt_defer_limit(c->defer_small));
  }
  guard.defer_abort = true;   // cleared on the successful way out
  // `behind`: an event of a stream OTHER than the main one whose producers queued partials too (the virtual-node update's weight
  // gradients run on the second stream and the main stream joins that stream only in stage 4)
  auto flush = [&](void* behind = nullptr) -> int {
    gt_stream_t fs = ov ? gt_overlap_dw_fork(st, 0) : st;   // behind everything queued on the main stream, on the overlap stream
    if (behind) GT_TRY(gt_stream_wait_event(fs, behind));
    GT_TRY(gt_defer_flush(fs));
    if (fs != st) gt_overlap_dw_booked(Q(c->q_defer), c->defer_bytes);
    return GT_OK;
  };
  // (an error return in a middle stage leaves the section open on this thread: the next gt_overlap_dw_begin resets it)

  // the next stage's workspace: the two slots alternate, so the weight-gradient GEMMs a stage forks onto the third stream (their
  // partials and the dy they read live in the stage's slot) run beside the NEXT stage; the main stream waits only for the
  // GEMMs that used this slot two stages ago
  auto W = [&]() -> void* {
    c->slot ^= 1;
    void* p = Q(c->q_ws[c->slot]);
    if (ov) gt_overlap_dw_release(p, ws_bytes);
    return p;
  };
  auto conv_saved = [&](int l) -> void* { return P(c->o_conv_saved[l]); };

  if (stages & 1) {
    // ---- heads: the weight gradient first, forked onto the overlap stream beside the heads' own dX GEMM
    GT_TRY(gt_linear_bwd_dw_forked(GT_F32, GT_F32, compute, P(c->o_hg), m->head_w, dlogits, nullptr, G + m->off_head_w, G + m->off_head_b, B,
                                   m->Nh, d, d, m->ldy, 0.f, Q(c->q_heads_ws), c->heads_ws_bytes, st));
    GT_TRY(gt_linear_bwd_ld(GT_F32, GT_F32, compute, P(c->o_hg), m->head_w, dlogits, nullptr, nullptr, nullptr, Q(c->q_d_hg), nullptr, nullptr,
                            B, m->Nh, d, m->ldy, 0.f, W(), ws_bytes, st));
    // ---- pooled rows -> token rows
    void *dcur = Q(c->q_dtok[0]), *dnext = Q(c->q_dtok[1]);
    const float* d_pool = (const float*)Q(c->q_d_hg);
    if (m->nout_w) {   // the final norm on the pooled rows
      GT_TRY(gt_layernorm_bwd(GT_F32, P(c->o_hgin), nullptr, Q(c->q_d_hg), m->nout_w, (float*)P(c->o_sto), (float*)P(c->o_sto) + B, 0.f, 0, B, d,
                              Q(c->q_d_hgin), nullptr, G + m->off_nout_w, G + m->off_nout_b, W(), ws_bytes, st));
      d_pool = (const float*)Q(c->q_d_hgin);
    }
    GT_TRY(gt_rows_scatter(tdt, d_pool, nullptr, B, B, d, Q(c->q_dyp), st));   // the pooled rows' gradient in the token dtype
    for (int i = nenc - 1; i >= 0; --i) {
      if (i == nenc - 1)
        GT_TRY(gt_encoder_layer_pooled_bwd(&c->enc[i], c->enc_in[i], c->last_rows, Q(c->q_dyp), P(c->o_enc_saved[i]), dnext, G + m->off_enc[i],
                                           W(), ws_bytes, st));
      else
        GT_TRY(gt_encoder_layer_bwd(&c->enc[i], c->enc_in[i], dcur, P(c->o_enc_saved[i]), dnext, G + m->off_enc[i], W(), ws_bytes, st));
      std::swap(dcur, dnext);
    }
    if (m->nin_w) {
      GT_TRY(gt_layernorm_bwd(tdt, P(c->o_tok), nullptr, dcur, m->nin_w, (float*)P(c->o_st0), (float*)P(c->o_st0) + rows, 0.f, 0, rows, d,
                              dnext, nullptr, G + m->off_nin_w, G + m->off_nin_b, W(), ws_bytes, st));
      std::swap(dcur, dnext);
    }
    // ---- token rows -> node rows (+ the CLS gradient)
    const void* d_hn = Q(c->q_d_hn);
    if (c->fuse_rows) {   // the GEMMs read the token-row gradient through the row map; the cls gradient = column sums of the CLS rows
      if (m->cls) GT_TRY(gt_colsum_rows_f32(tdt, dcur, c->last_rows, B, d, G + m->off_cls, st));
      GT_TRY(gt_linear_set_rows((const int32_t*)P(c->o_rowmap)));
      d_hn = dcur;
    } else {
      GT_TRY(gt_seq_scatter(tdt, dcur, nullptr, c->graph_ptr, c->node_graph, c->seq_desc, B, 1, m->with_cls ? 1 : 0, N, d, Q(c->q_d_hn),
                            m->cls ? Q(c->q_d_cls) : nullptr, st));
      if (m->cls) GT_TRY(gt_colsum_f32(tdt, Q(c->q_d_cls), B, d, G + m->off_cls, st));
    }
    if (c->g2t_wt && m->st_dw) GT_TRY(gt_stream_wait_event(st, m->ev_wt[1]));   // W^T was written on the overlap stream beside the forward
    if (c->cat2) {   // d h_list[0] -> dJ, d h_list[-1] -> dA straight from the GEMM
      GT_TRY(gt_linear_bwd_cat2(tdt, compute, c->first, D, D, c->h_last, D, D, m->g2t_w, d_hn, Q(c->q_dJ), D, Q(c->q_dA), D,
                                G + m->off_g2t_w, G + m->off_g2t_b, N, d, d, W(), ws_bytes, st));
      c->dy = Q(c->q_dA);
    } else {
      GT_TRY(gt_linear_bwd_wt(GT_F32, tdt, compute, c->node_rep, m->g2t_w, c->g2t_wt, d_hn, nullptr, nullptr, nullptr, Q(c->q_d_rep),
                              G + m->off_g2t_w, G + m->off_g2t_b, N, d, Kc, 0.f, W(), ws_bytes, st));
      if (m->jk_cat) c->dy = nullptr;   // split below (stage 2 prologue)
      else c->dy = Q(c->q_d_rep);
    }
    GT_TRY(flush());
    c->stages_done |= 1;
  }

  if (stages & 2) {
    // ---- message passing, last layer first.  dy = d h_list[l+1]; "extra" = gradient reaching x_l from its consumers other than
    // conv_l: the JK slab (l = 0) and the virtual-node update's pooling (l < L-1)
    void* dy = c->dy;
    if (!c->cat2 && m->jk_cat) {
      GT_TRY(gt_copy2d(Q(c->q_dA), D * 4, (char*)Q(c->q_d_rep) + D * 4, Kc * 4, D * 4, N, st));   // d h_list[-1]
      GT_TRY(gt_copy2d(Q(c->q_dJ), D * 4, Q(c->q_d_rep), Kc * 4, D * 4, N, st));                   // d h_list[0]
      dy = Q(c->q_dA);
    }
    if (c->fuse_bn) {
      for (int l = 1; l < L; ++l) {
        gt_gcn_layer &up = c->gcn[l], &dn = c->gcn[l - 1];
        up.prev_saved = conv_saved(l - 1);
        up.prev_bn_w = dn.bn_w; up.prev_bn_b = dn.bn_b; up.prev_relu = dn.relu;
        up.prev_bn_part = (float*)Q(c->q_bnpart[l - 1]);
        dn.bn_part_in = (const float*)Q(c->q_bnpart[l - 1]);
        dn.bn_nparts_in = c->bn_rows;
      }
    }
    void* d_vn_next = nullptr;
    for (int l = L - 1; l >= 0; --l) {
      const void* extra = (l == 0 && m->jk_cat) ? Q(c->q_dJ) : nullptr;
      const bool upd = m->has_vn && l < L - 1;
      void* d_vn_upd = Q(c->q_dvn[2]);   // d vn_l through update l (its pooled + residual inputs)
      const float* dt0 = nullptr;   // the update's gradient per GRAPH, added per node in the dX GEMM's epilogue (no N x D broadcast pass)
      if (upd) {   // vn_{l+1} = update(x_l, vn_l): d x_l = pooled gradient (+ the JK slab at l = 0)
        const bool bc = m->conv == GT_CONV_GCN && gt_linear_bwd_bcast_ok(compute, GT_F32, GT_F32, c->gcn[l].lin_w, N, D, D);
        void* dxl = bc ? nullptr : Q(c->q_dC);
        if (side) {   // beside layer l's BatchNorm / aggregate backward; joined before its dX GEMM (ev_dx_wait)
          GT_TRY(gt_event_record(m->ev_dvn[l], st));
          GT_TRY(gt_stream_wait_event(side, m->ev_dvn[l]));
          // without a residual branch d vn_l IS d_t0: read where the update's last GEMM wrote it (no copy launch)
          d_vn_upd = m->residual ? Q(c->q_dvn[2]) : (void*)gt_vn_update_bwd_dt0(&c->vn[l], Q(c->q_ws2));
          GT_TRY(gt_vn_update_bwd(&c->vn[l], d_vn_next, P(c->o_vn_saved[l]), extra, dxl, d_vn_upd, G + m->off_vn[l], Q(c->q_ws2),
                                  c->ws2_bytes, side));
          if (!m->vn_defer_dw) GT_TRY(gt_event_record(m->ev_extra[l], side));
          if (bc) dt0 = gt_vn_update_bwd_dt0(&c->vn[l], Q(c->q_ws2));
        } else {
          void* vws = W();
          GT_TRY(gt_vn_update_bwd(&c->vn[l], d_vn_next, P(c->o_vn_saved[l]), extra, dxl, Q(c->q_dvn[2]), G + m->off_vn[l], vws, ws_bytes, st));
          if (bc) dt0 = gt_vn_update_bwd_dt0(&c->vn[l], vws);
        }
        if (!bc) extra = Q(c->q_dC);
      }
      void* out = dy == Q(c->q_dA) ? Q(c->q_dB) : Q(c->q_dA);
      if (l == 0 && ov) gt_overlap_dw_urgent(1);   // layer 0's weight gradients are the last: nothing left to overlap them with
      const bool pool_on_side = m->has_vn && side;
      void* d_vn = (m->has_vn && !pool_on_side) ? Q(c->q_dvn[3]) : nullptr;
      if (m->conv == GT_CONV_PNA) {
        gt_pna_layer& g = c->pna[l];
        float* dimg = (float*)Q(c->q_dimg);
        g.d_pre_w = dimg + m->pna_img_off[l][0]; g.d_pre_b = dimg + m->pna_img_off[l][1];
        g.d_post_w = dimg + m->pna_img_off[l][2]; g.d_post_b = dimg + m->pna_img_off[l][3];
        GT_TRY(gt_pna_layer_bwd(&g, c->xptr[l], dy, conv_saved(l), out, G + m->off_conv[l], W(), ws_bytes, st));
      } else if (m->conv == GT_CONV_GIN)
        GT_TRY(gt_gin_layer_bwd(&c->gin[l], c->xptr[l], dy, extra, conv_saved(l), out, d_vn, G + m->off_conv[l], W(), ws_bytes, st));
      else {
        c->gcn[l].dx_bcast = dt0;   // (requested inside the layer, right in front of its one dX GEMM)
        c->gcn[l].dx_bcast_idx = dt0 ? c->node_graph : nullptr;
        GT_TRY(gt_gcn_layer_bwd(&c->gcn[l], c->xptr[l], dy, extra, conv_saved(l), out, d_vn, G + m->off_conv[l], W(), ws_bytes, st));
      }
      if (m->has_vn) {   // d vn_l = per-graph sum of d x_l (+ update l's pooled + residual inputs): off the main chain
        gt_stream_t vst = pool_on_side ? side : st;
        void* tgt = Q(c->q_dvn[l % 2]);
        if (pool_on_side) {   // the pooled gradient + update l's share in ONE pass (the sum's `add` rows): no bcast_add / copy launch
          GT_TRY(gt_event_record(m->ev_pool[l], st));
          GT_TRY(gt_stream_wait_event(side, m->ev_pool[l]));
          GT_TRY(gt_segment_sum_ws(GT_F32, out, upd ? d_vn_upd : nullptr, c->graph_ptr, N, B, D, tgt, Q(c->q_ws3), c->seg_ws_bytes, side));
        } else if (upd) {
          GT_TRY(gt_segment_bcast_add(GT_F32, Q(c->q_dvn[3]), d_vn_upd, b.ident_B, B, B, D, tgt, vst));
        } else {
          GT_TRY(gt_copy2d(tgt, D * 4, Q(c->q_dvn[3]), D * 4, D * 4, B, vst));
        }
        d_vn_next = tgt;
      }
      dy = out;
    }
    c->d_h0 = dy;
    if (m->conv == GT_CONV_PNA) {   // image gradients -> the tower parameters' gradients (every source element sits in the images at most once)
      if (ov) gt_overlap_dw_sync();
      GT_TRY(gt_gather_f32(G + m->off_pna_src, (const float*)Q(c->q_dimg), m->pna_inv, m->pna_n_src, st));
    }
    if (m->has_vn) {
      gt_stream_t vst = side ? side : st;
      GT_TRY(gt_segment_sum(GT_F32, d_vn_next, nullptr, b.ptr01, B, 1, D, G + m->off_vn_emb, vst));
      if (side) GT_TRY(gt_event_record(m->ev_vnemb, side));
    }
    // the virtual-node updates' weight-gradient GEMMs queued their partials from the second stream: the sum runs behind its tail
    GT_TRY(flush((m->has_vn && side) ? m->ev_vnemb : nullptr));
    c->stages_done |= 2;
  }

  if (stages & 4) {
    // ---- input encoder
    void* d_h0 = c->d_h0;
    if (m->embed_kind == 1) {   // dW = d_h0^T x, db = colsum(d_h0); the features need no gradient
      const int64_t K = m->ne_K, Kp = (int64_t)c4((size_t)K);
      float* dw = Kp == K ? G + m->off_ne_w : (float*)Q(c->q_ne_dw);
      GT_TRY(gt_linear_bwd(GT_F32, GT_F32, compute, c->ne_x, c->ne_w, d_h0, nullptr, nullptr, nullptr, nullptr, dw, G + m->off_ne_b, N, D, Kp,
                           0.f, W(), ws_bytes, st));
      if (Kp != K) GT_TRY(flush());   // the re-pitch below reads the summed gradient
      if (ov) gt_overlap_dw_sync();
      if (Kp != K) GT_TRY(gt_repitch(G + m->off_ne_w, K, dw, Kp, D, 4, st));
    } else {
      float* d_tabs[MAXT];
      for (int t = 0; t < c->T; ++t) d_tabs[t] = G + m->off_tables[t];
      if (c->esort)
        GT_TRY(gt_embed_sum_bwd_sorted(c->T, c->e_rows, (const float*)d_h0, N, D, P(c->o_eplan), d_tabs, W(), ws_bytes, st));
      else
        GT_TRY(gt_embed_sum_bwd(c->T, c->e_idx, c->e_str, c->e_clamp, c->e_rows, (const float*)d_h0, N, D, d_tabs, W(), ws_bytes, st));
    }
    // the virtual-node chain's tail (d vn_0 reduced on the second stream) joins here; the overlap section closes in the guard
    if (m->has_vn && side) GT_TRY(gt_stream_wait_event(st, m->ev_vnemb));
    GT_TRY(flush());
    GT_TRY(gt_defer_end());
    c->stages_done |= 4;
  }
  guard.defer_abort = false;   // (between the stage calls of one backward the section stays open, like the overlap section)
  return GT_OK;
}

// sizes of the caller-filled structs (a binding checks its mirror of the layouts against these)
extern "C" int gt_model_abi_sizes(int64_t* out4) {
  GT_CHECK_ARG(out4, "null output");
  out4[0] = (int64_t)sizeof(gt_model);
  out4[1] = (int64_t)sizeof(gt_model_batch);
  out4[2] = (int64_t)sizeof(gt_image_set);
  out4[3] = (int64_t)sizeof(gt_stage_ring);
  return GT_OK;
}

// The host half of the packed token layout on its own (what gt_model_prepare writes into its staging slot): meta6 = {rows,
// max_npos, num_work, offset of last_rows, offset of the work list, total bytes}; out_host == NULL only sizes it.
extern "C" int gt_seq_layout_packed_host(const int64_t* sizes_host, int64_t B, int64_t max_input_len, int with_cls, void* out_host,
                                         size_t out_bytes, int64_t* meta6) {
  GT_CHECK_ARG(sizes_host && meta6 && B >= 0, "null argument");
  std::vector<int64_t> kv;
  const HostLayout h = layout_sizes(sizes_host, B, max_input_len, with_cls ? 1 : 0, &kv);
  meta6[0] = h.rows; meta6[1] = h.max_npos; meta6[2] = h.num_work;
  meta6[3] = (int64_t)h.o_last; meta6[4] = (int64_t)h.o_work; meta6[5] = (int64_t)h.bytes;
  if (!out_host) return GT_OK;
  if (out_bytes < h.bytes) { gt_set_error("gt_seq_layout_packed_host: buffer too small"); return GT_ERR_WORKSPACE; }
  layout_fill(kv, B, h, (char*)out_host);
  return GT_OK;
}
