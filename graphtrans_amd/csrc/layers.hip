// layers.hip — composite layer entry points: one C call enqueues every kernel of a GraphTrans layer
// (forward or backward), so the host issues ~10 launches back to back instead of going through
// ~10 Python/autograd round trips per layer (the step was host-bound at ~530 launches,
// profiles/r01d).  Pure host code: it only sequences the gt_* primitives of this library on the
// caller's stream and carves caller-owned buffers; nothing is allocated or synchronised here.
//
// Reference structure being sequenced (paths under /root/reference):
//   encoder layer  torch nn.TransformerEncoderLayer (post-norm) via modules/transformer_encoder.py:28-32,59
//   GCN layer      modules/gnn_module.py:199-212 (vn add, conv, batch_norm, relu, residual) + modules/conv.py:50-71
//   VN update      modules/gnn_module.py:217-229 (global_add_pool + vn -> MLP)
#include "gt_common.h"

namespace {

struct Bump {
  char* base;
  size_t off;
  explicit Bump(void* p) : base(reinterpret_cast<char*>(p)), off(0) {}
  void* take(size_t bytes) {
    void* p = base ? base + off : nullptr;
    off += (bytes + 255) & ~size_t(255);
    return p;
  }
};

size_t elt(int dtype) { return dtype == GT_BF16 ? 2 : 4; }

#define GT_TRY(call)            \
  do {                          \
    int rc__ = (call);          \
    if (rc__ != GT_OK) return rc__; \
  } while (0)

// ---------------------------------------------------------------- encoder layer
struct EncSaved {
  void *qkv, *ctx, *a, *x1, *f1, *f2, *g1;   // g1: gelu only, the FFN activation's gradient multiplier
  float *lse, *st1, *st2;
  size_t bytes;
};
EncSaved enc_saved(const gt_encoder_layer* L, void* p) {
  Bump b(p);
  const size_t e = elt(L->dtype);
  EncSaved s;
  s.qkv = b.take((size_t)L->rows * 3 * L->d_model * e);
  s.ctx = b.take((size_t)L->rows * L->d_model * e);
  s.a = b.take((size_t)L->rows * L->d_model * e);
  s.x1 = b.take((size_t)L->rows * L->d_model * e);
  s.f1 = b.take((size_t)L->rows * L->ffn * e);
  s.f2 = b.take((size_t)L->rows * L->d_model * e);
  s.g1 = b.take(L->act == 1 ? (size_t)L->rows * L->ffn * e : 0);
  s.lse = (float*)b.take((size_t)2 * L->nhead * L->rows * 4);
  s.st1 = (float*)b.take((size_t)2 * L->rows * 4);
  s.st2 = (float*)b.take((size_t)2 * L->rows * 4);
  s.bytes = b.off;
  return s;
}
struct EncWork {
  void *d_f2, *d_x1, *d_f1, *d_a, *d_ctx, *d_qkv, *lin_ws, *ln_ws, *ln_ws1;
  float* delta;
  size_t lin_ws_bytes, ln_ws_bytes, bytes;
};
EncWork enc_work(const gt_encoder_layer* L, void* p) {
  Bump b(p);
  const size_t e = elt(L->dtype);
  EncWork w;
  w.d_f2 = b.take((size_t)L->rows * L->d_model * e);
  w.d_x1 = b.take((size_t)L->rows * L->d_model * e);
  w.d_f1 = b.take((size_t)L->rows * L->ffn * e);
  w.d_a = b.take((size_t)L->rows * L->d_model * e);
  w.d_ctx = b.take((size_t)L->rows * L->d_model * e);
  w.d_qkv = b.take((size_t)L->rows * 3 * L->d_model * e);
  w.delta = (float*)b.take((size_t)L->nhead * L->rows * 4);
  size_t m = 0;
  const int64_t d = L->d_model, F = L->ffn, R = L->rows;
  const int c = L->dtype == GT_BF16 ? GT_BF16 : L->compute;
  size_t q;
  q = gt_linear_bwd_workspace_bytes(c, R, 3 * d, d); m = q > m ? q : m;
  q = gt_linear_bwd_workspace_bytes(c, R, d, d); m = q > m ? q : m;
  q = gt_linear_bwd_workspace_bytes(c, R, F, d); m = q > m ? q : m;
  q = gt_linear_bwd_workspace_bytes(c, R, d, F); m = q > m ? q : m;
  w.lin_ws_bytes = m;
  w.lin_ws = b.take(m);
  w.ln_ws_bytes = gt_layernorm_bwd_workspace_bytes(R, d);
  w.ln_ws = b.take(w.ln_ws_bytes);
  w.ln_ws1 = b.take(w.ln_ws_bytes);   // norm1's own: norm2's column finish (overlap stream) may still read ln_ws
  w.bytes = b.off;
  return w;
}
// flat gradient layout (floats), parameter order of the header
struct EncGrads {
  float *in_w, *in_b, *out_w, *out_b, *l1_w, *l1_b, *l2_w, *l2_b, *n1_w, *n1_b, *n2_w, *n2_b;
};
EncGrads enc_grads(const gt_encoder_layer* L, float* g) {
  const int64_t d = L->d_model, F = L->ffn;
  EncGrads r;
  r.in_w = g; g += 3 * d * d;
  r.in_b = g; g += 3 * d;
  r.out_w = g; g += d * d;
  r.out_b = g; g += d;
  r.l1_w = g; g += F * d;
  r.l1_b = g; g += F;
  r.l2_w = g; g += d * F;
  r.l2_b = g; g += d;
  r.n1_w = g; g += d;
  r.n1_b = g; g += d;
  r.n2_w = g; g += d;
  r.n2_b = g; g += d;
  return r;
}

int enc_check(const char* fn, const gt_encoder_layer* L) {
  if (!L) { gt_set_error("%s: null descriptor", fn); return GT_ERR_INVALID_ARG; }
  if (L->dtype != GT_F32 && L->dtype != GT_BF16) { gt_set_error("%s: bad dtype", fn); return GT_ERR_INVALID_ARG; }
  if (L->rows < 0 || L->d_model <= 0 || L->ffn <= 0 || L->nhead <= 0) { gt_set_error("%s: bad sizes", fn); return GT_ERR_INVALID_ARG; }
  if (L->d_model % 8 || L->ffn % 8) { gt_set_error("%s: d_model and ffn must be multiples of 8", fn); return GT_ERR_UNSUPPORTED; }
  return GT_OK;
}

// ---------------------------------------------------------------- GCN layer
struct GcnSaved {
  void *lin, *agg;
  float* stats;
  size_t bytes;
};
GcnSaved gcn_saved(const gt_gcn_layer* L, void* p) {
  Bump b(p);
  GcnSaved s;
  s.lin = b.take((size_t)L->N * L->D * 4);
  s.agg = b.take((size_t)L->N * L->D * 4);
  s.stats = (float*)b.take((size_t)2 * L->D * 4);
  s.bytes = b.off;
  return s;
}
int64_t gcn_edge_w_elems(const gt_gcn_layer* L) {
  if (L->edge_mode == GT_EDGE_LINEAR) return L->D * L->edge_cols;
  if (L->edge_mode == GT_EDGE_TABLES) return L->table_rows * L->D;
  return 0;
}
struct GcnGrads {
  float *lin_w, *lin_b, *root, *edge_w, *edge_b, *bn_w, *bn_b;
};
GcnGrads gcn_grads(const gt_gcn_layer* L, float* g) {
  GcnGrads r;
  r.lin_w = g; g += L->D * L->D;
  r.lin_b = g; g += L->D;
  r.root = g; g += L->D;
  r.edge_w = g; g += gcn_edge_w_elems(L);
  r.edge_b = g; g += (L->edge_mode == GT_EDGE_LINEAR ? L->D : 0);
  r.bn_w = g; g += L->D;
  r.bn_b = g; g += L->D;
  return r;
}
struct GcnWork {
  void *d_agg, *d_lin, *bn_ws, *agg_ws, *lin_ws, *seg_ws;
  size_t bn_ws_bytes, agg_ws_bytes, lin_ws_bytes, seg_ws_bytes, bytes;
};
GcnWork gcn_work(const gt_gcn_layer* L, void* p) {
  Bump b(p);
  GcnWork w;
  w.d_agg = b.take((size_t)L->N * L->D * 4);
  w.d_lin = b.take((size_t)L->N * L->D * 4);
  w.bn_ws_bytes = gt_batchnorm_workspace_bytes(L->N, L->D);
  w.bn_ws = b.take(w.bn_ws_bytes);
  w.agg_ws_bytes = gt_aggregate_bwd_workspace_bytes(GT_CONV_GCN, L->edge_mode, L->D, L->edge_cols, L->table_rows);
  w.agg_ws = b.take(w.agg_ws_bytes);
  w.lin_ws_bytes = gt_linear_bwd_workspace_bytes(L->compute, L->N, L->D, L->D);
  w.lin_ws = b.take(w.lin_ws_bytes);
  w.seg_ws_bytes = L->has_vn ? gt_segment_sum_workspace_bytes(L->N, L->D) : 0;   // d vn = pooled d x (its own slot: the
  w.seg_ws = b.take(w.seg_ws_bytes);                                             // dW GEMM may still use lin_ws)
  w.bytes = b.off;
  return w;
}
int gcn_check(const char* fn, const gt_gcn_layer* L) {
  if (!L) { gt_set_error("%s: null descriptor", fn); return GT_ERR_INVALID_ARG; }
  if (L->N < 0 || L->D <= 0 || L->D % 4) { gt_set_error("%s: bad sizes", fn); return GT_ERR_INVALID_ARG; }
  if (L->edge_mode == GT_EDGE_DENSE) { gt_set_error("%s: dense edge embeddings use the un-fused ops", fn); return GT_ERR_UNSUPPORTED; }
  return GT_OK;
}

// ---------------------------------------------------------------- virtual-node update
struct VnSaved {
  void *t0, *z1, *a1, *z2;
  float *st1, *st2;
  size_t bytes;
};
VnSaved vn_saved(const gt_vn_update* L, void* p) {
  Bump b(p);
  VnSaved s;
  s.t0 = b.take((size_t)L->B * L->D * 4);
  s.z1 = b.take((size_t)L->B * 2 * L->D * 4);
  s.a1 = b.take((size_t)L->B * 2 * L->D * 4);
  s.z2 = b.take((size_t)L->B * L->D * 4);
  s.st1 = (float*)b.take((size_t)2 * 2 * L->D * 4);
  s.st2 = (float*)b.take((size_t)2 * L->D * 4);
  s.bytes = b.off;
  return s;
}
struct VnGrads {
  float *w1, *b1, *bn1_w, *bn1_b, *w2, *b2, *bn2_w, *bn2_b;
};
VnGrads vn_grads(const gt_vn_update* L, float* g) {
  const int64_t D = L->D;
  VnGrads r;
  r.w1 = g; g += 2 * D * D;
  r.b1 = g; g += 2 * D;
  r.bn1_w = g; g += 2 * D;
  r.bn1_b = g; g += 2 * D;
  r.w2 = g; g += 2 * D * D;
  r.b2 = g; g += D;
  r.bn2_w = g; g += D;
  r.bn2_b = g; g += D;
  return r;
}
struct VnWork {
  void *d_z2, *d_a1, *d_z1, *d_t0, *bn_ws, *lin_ws, *seg_ws;
  size_t bn_ws_bytes, lin_ws_bytes, seg_ws_bytes, bytes;
};
VnWork vn_work(const gt_vn_update* L, void* p) {
  Bump b(p);
  VnWork w;
  w.d_z2 = b.take((size_t)L->B * L->D * 4);
  w.d_a1 = b.take((size_t)L->B * 2 * L->D * 4);
  w.d_z1 = b.take((size_t)L->B * 2 * L->D * 4);
  w.d_t0 = b.take((size_t)L->B * L->D * 4);
  w.bn_ws_bytes = gt_batchnorm_workspace_bytes(L->B, 2 * L->D);
  w.bn_ws = b.take(w.bn_ws_bytes);
  size_t a = gt_linear_bwd_workspace_bytes(L->compute, L->B, 2 * L->D, L->D);
  size_t c = gt_linear_bwd_workspace_bytes(L->compute, L->B, L->D, 2 * L->D);
  w.lin_ws_bytes = a > c ? a : c;
  w.lin_ws = b.take(w.lin_ws_bytes);
  w.seg_ws_bytes = gt_segment_sum_workspace_bytes(L->N, L->D);   // the pooling of the forward
  w.seg_ws = b.take(w.seg_ws_bytes);
  w.bytes = b.off;
  return w;
}

}  // namespace

// =================================================================================================
extern "C" size_t gt_encoder_layer_saved_bytes(const gt_encoder_layer* L) { return L ? enc_saved(L, nullptr).bytes : 0; }
extern "C" size_t gt_encoder_layer_workspace_bytes(const gt_encoder_layer* L) { return L ? enc_work(L, nullptr).bytes : 0; }
extern "C" int64_t gt_encoder_layer_grad_elems(const gt_encoder_layer* L) {
  if (!L) return 0;
  const int64_t d = L->d_model, F = L->ffn;
  return 3 * d * d + 3 * d + d * d + d + F * d + F + d * F + d + 4 * d;
}

extern "C" int gt_encoder_layer_fwd(const gt_encoder_layer* L, const void* x, void* y, void* saved, gt_stream_t st) {
  GT_TRY(enc_check("gt_encoder_layer_fwd", L));
  GT_CHECK_ARG(x && y && saved, "null buffer");
  if (L->rows == 0) return GT_OK;
  const EncSaved s = enc_saved(L, saved);
  const int t = L->dtype, c = t == GT_BF16 ? GT_BF16 : L->compute;
  const int64_t R = L->rows, d = L->d_model, F = L->ffn;
  const float p = L->training ? L->dropout_p : 0.f;
  const float scale = 1.0f / sqrtf((float)(d / L->nhead));
  GT_TRY(gt_linear_fwd(t, t, c, x, L->in_w, L->in_b, s.qkv, R, 3 * d, d, 0, 0.f, 0, st));
  GT_TRY(gt_attn_fwd(t, s.qkv, s.ctx, s.lse, R, d, L->nhead, L->seq_desc, L->num_seqs, L->row_stride, L->max_npos,
                     L->work_items, L->num_work, nullptr, nullptr, 0.f, scale, p, L->seed, st));
  if (gt_linear_layernorm_fwd_ok(t, c, L->out_w, R, d, d)) {   // out_proj + residual + dropout + norm1 as one launch (linear1.h)
    GT_TRY(gt_linear_layernorm_fwd(t, c, s.ctx, L->out_w, L->out_b, s.a, R, d, d, x, L->n1_w, L->n1_b, L->ln_eps, p,
                                   L->seed ^ 0x5851F42D4C957F2DULL, s.x1, s.st1, s.st1 + R, st));
  } else {
    GT_TRY(gt_linear_fwd(t, t, c, s.ctx, L->out_w, L->out_b, s.a, R, d, d, 0, 0.f, 0, st));
    GT_TRY(gt_layernorm_fwd(t, s.a, x, L->n1_w, L->n1_b, L->ln_eps, p, L->seed ^ 0x5851F42D4C957F2DULL, R, d, s.x1, s.st1,
                            s.st1 + R, st));
  }
  if (L->act == 1)   // f1 = drop(gelu(x1 W1^T + b1)), multiplier saved for the backward
    GT_TRY(gt_linear_fwd_gelu(t, t, c, s.x1, L->l1_w, L->l1_b, s.f1, s.g1, R, F, d, d, F, p, L->seed ^ 0x2545F4914F6CDD1DULL, st));
  else
    GT_TRY(gt_linear_fwd(t, t, c, s.x1, L->l1_w, L->l1_b, s.f1, R, F, d, 1, p, L->seed ^ 0x2545F4914F6CDD1DULL, st));
  if (gt_linear_layernorm_fwd_ok(t, c, L->l2_w, R, d, F)) {    // linear2 + residual + dropout + norm2
    GT_TRY(gt_linear_layernorm_fwd(t, c, s.f1, L->l2_w, L->l2_b, s.f2, R, d, F, s.x1, L->n2_w, L->n2_b, L->ln_eps, p,
                                   L->seed ^ 0x14057B7EF767814FULL, y, s.st2, s.st2 + R, st));
  } else {
    GT_TRY(gt_linear_fwd(t, t, c, s.f1, L->l2_w, L->l2_b, s.f2, R, d, F, 0, 0.f, 0, st));
    GT_TRY(gt_layernorm_fwd(t, s.f2, s.x1, L->n2_w, L->n2_b, L->ln_eps, p, L->seed ^ 0x14057B7EF767814FULL, R, d, y, s.st2,
                            s.st2 + R, st));
  }
  return GT_OK;
}

extern "C" int gt_encoder_layer_bwd(const gt_encoder_layer* L, const void* x, const void* dy, const void* saved, void* dx,
                                    float* grads, void* workspace, size_t workspace_bytes, gt_stream_t st) {
  GT_TRY(enc_check("gt_encoder_layer_bwd", L));
  GT_CHECK_ARG(x && dy && saved && dx && grads && workspace, "null buffer");
  const EncWork w = enc_work(L, workspace);
  if (workspace_bytes < w.bytes) { gt_set_error("gt_encoder_layer_bwd: workspace too small"); return GT_ERR_WORKSPACE; }
  if (L->rows == 0) return GT_OK;
  const EncSaved s = enc_saved(L, const_cast<void*>(saved));
  const EncGrads g = enc_grads(L, grads);
  const int t = L->dtype, c = t == GT_BF16 ? GT_BF16 : L->compute;
  const int64_t R = L->rows, d = L->d_model, F = L->ffn;
  const float p = L->training ? L->dropout_p : 0.f;
  const float scale = 1.0f / sqrtf((float)(d / L->nhead));
  // x2 = LN2(x1 + drop(f2))
  GT_TRY(gt_layernorm_bwd(t, s.f2, s.x1, dy, L->n2_w, s.st2, s.st2 + R, p, L->seed ^ 0x14057B7EF767814FULL, R, d, w.d_f2,
                          w.d_x1, g.n2_w, g.n2_b, w.ln_ws, w.ln_ws_bytes, st));
  // f2 = f1 W2^T + b2 ; f1 = drop(act(x1 W1^T + b1)) ; d_x1 += ...
  bool norm1_done = false;
  if (gt_linear_bwd_gate_out_ok(t, t, c, L->l2_w, R, d, F)) {
    // weight-stationary path (linear1.h): linear2's dX GEMM writes the GATED gradient dZ1 = (dF2 W2) * act'(.) * dropout scale, the
    // tensor both GEMMs of linear1's backward read (the tiled kernels gate d_f1 while they stage it, twice)
    GT_TRY(gt_linear_bwd_gate_out(t, t, c, s.f1, L->l2_w, w.d_f2, L->act == 1 ? s.g1 : s.f1, nullptr, nullptr, w.d_f1, g.l2_w, g.l2_b, R, d, F,
                                  F, d, L->act == 1 ? -1.f : p, w.lin_ws, w.lin_ws_bytes, st));
    if (gt_linear_bwd_dx_layernorm_ok(t, c, L->l1_w, R, F, d) && w.ln_ws_bytes >= gt_linear_bwd_dx_layernorm_workspace_bytes(R, F, d)) {
      // linear1's dX GEMM ends in norm1's backward (x1 = LN1(x + drop(a))): d_x1 + dZ1 W1 never reaches memory (linear1.h, LNB epilogue)
      // (the weight gradient is forked BEHIND the dX launch, as gt_linear_bwd does: beside it, it slowed the critical kernel)
      GT_TRY(gt_linear_bwd_dx_layernorm(t, c, L->l1_w, w.d_f1, w.d_x1, nullptr, R, F, d, s.a, x, L->n1_w, s.st1, s.st1 + R, p,
                                        L->seed ^ 0x5851F42D4C957F2DULL, w.d_a, dx, g.n1_w, g.n1_b, w.ln_ws1, w.ln_ws_bytes, st));
      GT_TRY(gt_linear_bwd_dw_forked(t, t, c, s.x1, L->l1_w, w.d_f1, nullptr, g.l1_w, g.l1_b, R, F, d, d, F, 0.f, w.lin_ws, w.lin_ws_bytes, st));
      norm1_done = true;
    } else
    GT_TRY(gt_linear_bwd(t, t, c, s.x1, L->l1_w, w.d_f1, nullptr, w.d_x1, nullptr, w.d_x1, g.l1_w, g.l1_b, R, F, d, 0.f, w.lin_ws,
                         w.lin_ws_bytes, st));
  } else {
    GT_TRY(gt_linear_bwd(t, t, c, s.f1, L->l2_w, w.d_f2, nullptr, nullptr, nullptr, w.d_f1, g.l2_w, g.l2_b, R, d, F, 0.f,
                         w.lin_ws, w.lin_ws_bytes, st));
    if (L->act == 1)
      GT_TRY(gt_linear_bwd_mul(t, t, c, s.x1, L->l1_w, w.d_f1, s.g1, w.d_x1, nullptr, w.d_x1, g.l1_w, g.l1_b, R, F, d, d, F, w.lin_ws,
                               w.lin_ws_bytes, st));
    else
      GT_TRY(gt_linear_bwd(t, t, c, s.x1, L->l1_w, w.d_f1, s.f1, w.d_x1, nullptr, w.d_x1, g.l1_w, g.l1_b, R, F, d, p, w.lin_ws,
                           w.lin_ws_bytes, st));
  }
  // x1 = LN1(x + drop(a))
  if (!norm1_done)
    GT_TRY(gt_layernorm_bwd(t, s.a, x, w.d_x1, L->n1_w, s.st1, s.st1 + R, p, L->seed ^ 0x5851F42D4C957F2DULL, R, d, w.d_a, dx,
                            g.n1_w, g.n1_b, w.ln_ws1, w.ln_ws_bytes, st));
  // a = ctx Wo^T + bo
  GT_TRY(gt_linear_bwd(t, t, c, s.ctx, L->out_w, w.d_a, nullptr, nullptr, nullptr, w.d_ctx, g.out_w, g.out_b, R, d, d, 0.f,
                       w.lin_ws, w.lin_ws_bytes, st));
  GT_TRY(gt_attn_bwd(t, s.qkv, s.ctx, w.d_ctx, s.lse, w.delta, w.d_qkv, R, d, L->nhead, L->seq_desc, L->num_seqs,
                     L->row_stride, L->max_npos, L->work_items, L->num_work, nullptr, nullptr, 0.f, scale, p, L->seed, st));
  // qkv = x Win^T + bin ; dx += ...   The weight gradient first: forked onto the overlap stream it starts beside this layer's own
  // dX GEMM, not together with the next stage's first kernel (a LayerNorm backward)
  GT_TRY(gt_linear_bwd_dw_forked(t, t, c, x, L->in_w, w.d_qkv, nullptr, g.in_w, g.in_b, R, 3 * d, d, d, 3 * d, 0.f, w.lin_ws,
                                 w.lin_ws_bytes, st));
  GT_TRY(gt_linear_bwd(t, t, c, x, L->in_w, w.d_qkv, nullptr, dx, nullptr, dx, nullptr, nullptr, R, 3 * d, d, 0.f, w.lin_ws,
                       w.lin_ws_bytes, st));
  return GT_OK;
}

// ---------------------------------------------------------------- GIN layer
struct GinSaved {
  void *agg, *z1, *a1, *z2;
  float *st1, *st;
  size_t bytes;
};
GinSaved gin_saved(const gt_gin_layer* L, void* p) {
  Bump b(p);
  GinSaved s;
  s.agg = b.take((size_t)L->N * L->D * 4);
  s.z1 = b.take((size_t)L->N * 2 * L->D * 4);
  s.a1 = b.take((size_t)L->N * 2 * L->D * 4);
  s.z2 = b.take((size_t)L->N * L->D * 4);
  s.st1 = (float*)b.take((size_t)2 * 2 * L->D * 4);
  s.st = (float*)b.take((size_t)2 * L->D * 4);
  s.bytes = b.off;
  return s;
}
int64_t gin_edge_w_elems(const gt_gin_layer* L) {
  if (L->edge_mode == GT_EDGE_LINEAR) return L->D * L->edge_cols;
  if (L->edge_mode == GT_EDGE_TABLES) return L->table_rows * L->D;
  return 0;
}
constexpr int64_t GIN_EPS_SLOT = 20;  // d_eps + the aggregate backward's column-tile scratch (1 + ceil(1024/64))
struct GinGrads {
  float *eps, *edge_w, *edge_b, *w1, *b1, *bn1_w, *bn1_b, *w2, *b2, *bn_w, *bn_b;
};
GinGrads gin_grads(const gt_gin_layer* L, float* g) {
  const int64_t D = L->D;
  GinGrads r;
  r.eps = g; g += GIN_EPS_SLOT;
  r.edge_w = g; g += gin_edge_w_elems(L);
  r.edge_b = g; g += (L->edge_mode == GT_EDGE_LINEAR ? D : 0);
  r.w1 = g; g += 2 * D * D;
  r.b1 = g; g += 2 * D;
  r.bn1_w = g; g += 2 * D;
  r.bn1_b = g; g += 2 * D;
  r.w2 = g; g += 2 * D * D;
  r.b2 = g; g += D;
  r.bn_w = g; g += D;
  r.bn_b = g; g += D;
  return r;
}
struct GinWork {
  void *d_z2, *d_a1, *d_z1, *d_agg, *d_x, *bn_ws, *agg_ws, *lin_ws, *lin_ws1, *seg_ws;
  size_t bn_ws_bytes, agg_ws_bytes, lin_ws_bytes, seg_ws_bytes, bytes;
};
GinWork gin_work(const gt_gin_layer* L, void* p) {
  Bump b(p);
  GinWork w;
  const int64_t N = L->N, D = L->D;
  w.d_z2 = b.take((size_t)N * D * 4);
  w.d_a1 = b.take((size_t)N * 2 * D * 4);
  w.d_z1 = b.take((size_t)N * 2 * D * 4);
  w.d_agg = b.take((size_t)N * D * 4);
  w.d_x = b.take((size_t)N * D * 4);
  w.bn_ws_bytes = gt_batchnorm_workspace_bytes(N, 2 * D);
  w.bn_ws = b.take(w.bn_ws_bytes);
  w.agg_ws_bytes = gt_aggregate_bwd_workspace_bytes(GT_CONV_GIN, L->edge_mode, D, L->edge_cols, L->table_rows);
  w.agg_ws = b.take(w.agg_ws_bytes);
  size_t a = gt_linear_bwd_workspace_bytes(L->compute, N, 2 * D, D), c = gt_linear_bwd_workspace_bytes(L->compute, N, D, 2 * D);
  w.lin_ws_bytes = a > c ? a : c;
  w.lin_ws = b.take(w.lin_ws_bytes);
  w.lin_ws1 = b.take(w.lin_ws_bytes);   // the first Linear's own: the second one's dW GEMM (side stream) still uses lin_ws
  w.seg_ws_bytes = gt_segment_sum_workspace_bytes(L->N, L->D);   // the pooling of the forward
  w.seg_ws = b.take(w.seg_ws_bytes);
  w.bytes = b.off;
  return w;
}
int gin_check(const char* fn, const gt_gin_layer* L) {
  if (!L) { gt_set_error("%s: null descriptor", fn); return GT_ERR_INVALID_ARG; }
  if (L->N < 0 || L->D <= 0 || L->D % 4 || L->D > 1024) { gt_set_error("%s: bad sizes", fn); return GT_ERR_INVALID_ARG; }
  if (L->edge_mode == GT_EDGE_DENSE) { gt_set_error("%s: dense edge embeddings use the un-fused ops", fn); return GT_ERR_UNSUPPORTED; }
  return GT_OK;
}

// =================================================================================================
extern "C" size_t gt_gcn_layer_saved_bytes(const gt_gcn_layer* L) { return L ? gcn_saved(L, nullptr).bytes : 0; }
extern "C" size_t gt_gcn_layer_workspace_bytes(const gt_gcn_layer* L) { return L ? gcn_work(L, nullptr).bytes : 0; }
extern "C" int64_t gt_gcn_layer_grad_elems(const gt_gcn_layer* L) {
  if (!L) return 0;
  return L->D * L->D + 2 * L->D + gcn_edge_w_elems(L) + (L->edge_mode == GT_EDGE_LINEAR ? L->D : 0) + 2 * L->D;
}

extern "C" int gt_gcn_layer_fwd(const gt_gcn_layer* L, const void* h_in, const void* vn, void* x_out, void* y, void* saved,
                                void* workspace, size_t workspace_bytes, gt_stream_t st) {
  GT_TRY(gcn_check("gt_gcn_layer_fwd", L));
  GT_CHECK_ARG(h_in && y && saved && workspace, "null buffer");
  GT_CHECK_ARG(!L->has_vn || L->x_has_vn || (vn && x_out), "virtual-node layer needs vn and x_out");
  const GcnWork w = gcn_work(L, workspace);
  if (workspace_bytes < w.bytes) { gt_set_error("gt_gcn_layer_fwd: workspace too small"); return GT_ERR_WORKSPACE; }
  if (L->N == 0) return GT_OK;
  const GcnSaved s = gcn_saved(L, saved);
  const void* x = h_in;
  if (L->has_vn && !L->x_has_vn) {  // h_list[layer] = h_list[layer] + vn[batch]   (gnn_module.py:199)
    GT_TRY(gt_segment_bcast_add(GT_F32, h_in, vn, L->node_graph, L->N, L->B, L->D, x_out, st));
    x = x_out;
  }
  if (L->has_vn && L->ev_x_ready) GT_TRY(gt_event_record(L->ev_x_ready, st));   // x (with its virtual-node add) is complete
  GT_TRY(gt_linear_fwd(GT_F32, GT_F32, L->compute, x, L->lin_w, L->lin_b, s.lin, L->N, L->D, L->D, 0, 0.f, 0, st));
  if (L->ev_graph_ready) GT_TRY(gt_stream_wait_event(st, L->ev_graph_ready));   // gt_graph_prep ran beside everything up to here
  GT_TRY(gt_aggregate_fwd(GT_CONV_GCN, L->edge_mode, GT_F32, s.lin, L->N, L->E, L->D, L->in_ptr, L->in_src, L->in_eid, L->deg,
                          L->dis, L->root, L->edge_attr, L->edge_cols, L->edge_w, L->edge_b, L->tab_off, L->table_rows, nullptr, s.agg, st));
  // h = batch_norm(h) [relu] [+ h_list[layer]]   (gnn_module.py:204-212; dropout p = 0 or eval here)
  // (+ vn_next[batch]: the next layer's virtual-node add, folded into this layer's apply pass)
  GT_TRY(gt_batchnorm_fwd_bcast(GT_F32, s.agg, L->bn_w, L->bn_b, L->bn_rm, L->bn_rv, L->training ? L->bn_nbt : nullptr,
                                L->bn_momentum, L->bn_eps, L->training, L->relu, L->residual ? x : nullptr, L->vn_next,
                                L->vn_next ? L->node_graph : nullptr, L->vn_next ? L->ev_vn_next : nullptr, L->N, L->D, y, s.stats,
                                s.stats + L->D, L->dropout_p, L->seed, w.bn_ws, w.bn_ws_bytes, st));
  return GT_OK;
}

extern "C" int gt_gcn_layer_bwd(const gt_gcn_layer* L, const void* x, const void* dy, const void* dx_extra,
                                const void* saved, void* d_h_in, void* d_vn, float* grads, void* workspace,
                                size_t workspace_bytes, gt_stream_t st) {
  GT_TRY(gcn_check("gt_gcn_layer_bwd", L));
  GT_CHECK_ARG(x && dy && saved && d_h_in && grads && workspace, "null buffer");
  const GcnWork w = gcn_work(L, workspace);   // d_vn may be NULL: the caller pools d_h_in itself (e.g. on another stream)
  if (workspace_bytes < w.bytes) { gt_set_error("gt_gcn_layer_bwd: workspace too small"); return GT_ERR_WORKSPACE; }
  if (L->N == 0) return GT_OK;
  const GcnSaved s = gcn_saved(L, const_cast<void*>(saved));
  const GcnGrads g = gcn_grads(L, grads);
  if (L->bn_part_in && L->bn_nparts_in > 0 && L->dropout_p == 0.f)   // statistics already summed in the dX epilogue that produced dy
    GT_TRY(gt_batchnorm_bwd_parts(GT_F32, s.agg, dy, L->bn_w, L->bn_b, s.stats, s.stats + L->D, L->training, L->relu, L->N, L->D,
                                  w.d_agg, g.bn_w, g.bn_b, L->bn_part_in, L->bn_nparts_in, st));
  else
    GT_TRY(gt_batchnorm_bwd(GT_F32, s.agg, dy, L->bn_w, L->bn_b, s.stats, s.stats + L->D, L->training, L->relu, L->N, L->D,
                            w.d_agg, g.bn_w, g.bn_b, L->dropout_p, L->seed, w.bn_ws, w.bn_ws_bytes, st));
  GT_TRY(gt_aggregate_bwd(GT_CONV_GCN, L->edge_mode, GT_F32, s.lin, w.d_agg, L->N, L->E, L->D, L->out_ptr, L->out_dst,
                          L->out_eid, L->deg, L->dis, L->root, L->edge_attr, L->edge_cols, L->edge_w, L->edge_b, L->tab_off,
                          L->table_rows, nullptr, w.d_lin, g.root, g.edge_w, g.edge_b, nullptr, w.agg_ws, w.agg_ws_bytes, st));
  // d_x = d_lin W (+ grads reaching x from its other consumers) (+ dy through the residual branch)
  if (L->ev_dx_wait) GT_TRY(gt_stream_wait_event(st, L->ev_dx_wait));
  if (L->prev_saved && L->prev_bn_part) {   // d_h_in is the dy of the previous layer's BatchNorm: its statistics ride in this epilogue
    const GcnSaved ps = gcn_saved(L, const_cast<void*>(L->prev_saved));
    GT_TRY(gt_linear_bwd_bnstats((const float*)ps.agg, L->D, ps.stats, ps.stats + L->D, L->prev_bn_w, L->prev_bn_b, L->prev_relu,
                                 L->prev_bn_part));
  }
  // (the broadcast request is consumed -- or dropped -- by the very next gt_linear_bwd* call of this thread: this one)
  if (L->dx_bcast) GT_TRY(gt_linear_bwd_bcast(L->dx_bcast, L->dx_bcast_idx));
  GT_TRY(gt_linear_bwd_wt(GT_F32, GT_F32, L->compute, x, L->lin_w, L->lin_wt, w.d_lin, nullptr, dx_extra, L->residual ? dy : nullptr,
                          d_h_in, g.lin_w, g.lin_b, L->N, L->D, L->D, 0.f, w.lin_ws, w.lin_ws_bytes, st));
  if (L->has_vn && d_vn)
    GT_TRY(gt_segment_sum_ws(GT_F32, d_h_in, nullptr, L->graph_ptr, L->N, L->B, L->D, d_vn, w.seg_ws, w.seg_ws_bytes, st));
  return GT_OK;
}

// =================================================================================================
extern "C" size_t gt_vn_update_saved_bytes(const gt_vn_update* L) { return L ? vn_saved(L, nullptr).bytes : 0; }
extern "C" size_t gt_vn_update_workspace_bytes(const gt_vn_update* L) { return L ? vn_work(L, nullptr).bytes : 0; }
extern "C" int64_t gt_vn_update_grad_elems(const gt_vn_update* L) { return L ? 4 * L->D * L->D + 9 * L->D : 0; }

extern "C" int gt_vn_update_fwd(const gt_vn_update* L, const void* x, const void* vn, void* vn_out, void* saved,
                                void* workspace, size_t workspace_bytes, gt_stream_t st) {
  GT_CHECK_ARG(L && x && vn && vn_out && saved && workspace, "null buffer");
  GT_CHECK_ARG(L->D > 0 && L->D % 4 == 0 && L->B > 0, "bad sizes");
  const VnWork w = vn_work(L, workspace);
  if (workspace_bytes < w.bytes) { gt_set_error("gt_vn_update_fwd: workspace too small"); return GT_ERR_WORKSPACE; }
  const VnSaved s = vn_saved(L, saved);
  const int64_t B = L->B, D = L->D;
  // global_add_pool(h_list[layer], batch) + vn   (gnn_module.py:219)
  GT_TRY(gt_segment_sum_ws(GT_F32, x, vn, L->graph_ptr, L->N, B, D, s.t0, w.seg_ws, w.seg_ws_bytes, st));
  // mlp_virtualnode_list[layer]: Linear(D,2D) BN ReLU Linear(2D,D) BN ReLU   (gnn_module.py:161-170)
  GT_TRY(gt_linear_fwd(GT_F32, GT_F32, L->compute, s.t0, L->w1, L->b1, s.z1, B, 2 * D, D, 0, 0.f, 0, st));
  GT_TRY(gt_batchnorm_fwd(GT_F32, s.z1, L->bn1_w, L->bn1_b, L->bn1_rm, L->bn1_rv, L->training ? L->bn1_nbt : nullptr,
                          L->bn_momentum, L->bn_eps, L->training, 1, nullptr, B, 2 * D, s.a1, s.st1, s.st1 + 2 * D, 0.f, 0,
                          w.bn_ws, w.bn_ws_bytes, st));
  GT_TRY(gt_linear_fwd(GT_F32, GT_F32, L->compute, s.a1, L->w2, L->b2, s.z2, B, D, 2 * D, 0, 0.f, 0, st));
  GT_TRY(gt_batchnorm_fwd(GT_F32, s.z2, L->bn2_w, L->bn2_b, L->bn2_rm, L->bn2_rv, L->training ? L->bn2_nbt : nullptr,
                          L->bn_momentum, L->bn_eps, L->training, 1, L->residual ? vn : nullptr, B, D, vn_out, s.st2,
                          s.st2 + D, L->dropout_p, L->seed, w.bn_ws, w.bn_ws_bytes, st));  // vn (+)= drop(mlp(t))   (:222)
  return GT_OK;
}

extern "C" int gt_vn_update_bwd(const gt_vn_update* L, const void* d_vn_out, const void* saved, const void* d_x_add,
                                void* d_x, void* d_vn, float* grads, void* workspace, size_t workspace_bytes,
                                gt_stream_t st) {
  GT_CHECK_ARG(L && d_vn_out && saved && d_vn && grads && workspace, "null buffer");   // d_x == NULL: see gt_vn_update_bwd_dt0
  const VnWork w = vn_work(L, workspace);
  if (workspace_bytes < w.bytes) { gt_set_error("gt_vn_update_bwd: workspace too small"); return GT_ERR_WORKSPACE; }
  const VnSaved s = vn_saved(L, const_cast<void*>(saved));
  const VnGrads g = vn_grads(L, grads);
  const int64_t B = L->B, D = L->D;
  // with ev_dx_done: the dX chain first (d_x / d_vn are what the next layer's backward waits for), the event, then the two
  // weight gradients; without it each GEMM's dW follows its dX directly
  const bool defer = L->ev_dx_done != nullptr;
  GT_TRY(gt_batchnorm_bwd(GT_F32, s.z2, d_vn_out, L->bn2_w, L->bn2_b, s.st2, s.st2 + D, L->training, 1, B, D, w.d_z2, g.bn2_w,
                          g.bn2_b, L->dropout_p, L->seed, w.bn_ws, w.bn_ws_bytes, st));
  GT_TRY(gt_linear_bwd(GT_F32, GT_F32, L->compute, s.a1, L->w2, w.d_z2, nullptr, nullptr, nullptr, w.d_a1, defer ? nullptr : g.w2,
                       defer ? nullptr : g.b2, B, D, 2 * D, 0.f, w.lin_ws, w.lin_ws_bytes, st));
  GT_TRY(gt_batchnorm_bwd(GT_F32, s.z1, w.d_a1, L->bn1_w, L->bn1_b, s.st1, s.st1 + 2 * D, L->training, 1, B, 2 * D, w.d_z1,
                          g.bn1_w, g.bn1_b, 0.f, 0, w.bn_ws, w.bn_ws_bytes, st));
  // d_t0 = d_z1 W1 ; d_vn = d_t0 (+ d_vn_out through the residual branch)
  GT_TRY(gt_linear_bwd(GT_F32, GT_F32, L->compute, s.t0, L->w1, w.d_z1, nullptr, nullptr, nullptr, w.d_t0, defer ? nullptr : g.w1,
                       defer ? nullptr : g.b1, B, 2 * D, D, 0.f, w.lin_ws, w.lin_ws_bytes, st));
  // d_x[n] = d_t0[graph(n)] (+ d_x_add[n]: gradient reaching x from its other consumers)
  if (d_x) GT_TRY(gt_segment_bcast_add(GT_F32, d_x_add, w.d_t0, L->node_graph, L->N, B, D, d_x, st));
  if (L->residual)
    GT_TRY(gt_segment_bcast_add(GT_F32, w.d_t0, d_vn_out, L->identity_graph, B, B, D, d_vn, st));
  else if (d_vn != w.d_t0)   // (a caller that passes gt_vn_update_bwd_dt0() as d_vn reads d_t0 where it lies: no copy launch)
    (void)hipMemcpyAsync(d_vn, w.d_t0, (size_t)B * D * 4, hipMemcpyDeviceToDevice, (hipStream_t)st);
  if (defer) {
    GT_TRY(gt_event_record(L->ev_dx_done, st));
    GT_TRY(gt_linear_bwd(GT_F32, GT_F32, L->compute, s.a1, L->w2, w.d_z2, nullptr, nullptr, nullptr, nullptr, g.w2, g.b2, B, D,
                         2 * D, 0.f, w.lin_ws, w.lin_ws_bytes, st));
    GT_TRY(gt_linear_bwd(GT_F32, GT_F32, L->compute, s.t0, L->w1, w.d_z1, nullptr, nullptr, nullptr, nullptr, g.w1, g.b1, B,
                         2 * D, D, 0.f, w.lin_ws, w.lin_ws_bytes, st));
  }
  return GT_OK;
}

// d_t0 [B][D] inside the workspace of a gt_vn_update_bwd call: with d_x == NULL that call does not broadcast d_t0 over the nodes
// (an N x D pass) -- the caller adds d_t0[node_graph[n]] where d_x is consumed (gt_linear_bwd_bcast: the dX GEMM's epilogue).
extern "C" const float* gt_vn_update_bwd_dt0(const gt_vn_update* L, void* workspace) {
  if (!L || !workspace) return nullptr;
  return (const float*)vn_work(L, workspace).d_t0;
}

// =================================================================================================
extern "C" size_t gt_gin_layer_saved_bytes(const gt_gin_layer* L) { return L ? gin_saved(L, nullptr).bytes : 0; }
extern "C" size_t gt_gin_layer_workspace_bytes(const gt_gin_layer* L) { return L ? gin_work(L, nullptr).bytes : 0; }
extern "C" int64_t gt_gin_layer_grad_elems(const gt_gin_layer* L) {
  if (!L) return 0;
  const int64_t D = L->D;
  return GIN_EPS_SLOT + gin_edge_w_elems(L) + (L->edge_mode == GT_EDGE_LINEAR ? D : 0) + 4 * D * D + 9 * D;
}

extern "C" int gt_gin_layer_fwd(const gt_gin_layer* L, const void* h_in, const void* vn, void* x_out, void* y, void* saved,
                                void* workspace, size_t workspace_bytes, gt_stream_t st) {
  GT_TRY(gin_check("gt_gin_layer_fwd", L));
  GT_CHECK_ARG(h_in && y && saved && workspace, "null buffer");
  GT_CHECK_ARG(!L->has_vn || L->x_has_vn || (vn && x_out), "virtual-node layer needs vn and x_out");
  const GinWork w = gin_work(L, workspace);
  if (workspace_bytes < w.bytes) { gt_set_error("gt_gin_layer_fwd: workspace too small"); return GT_ERR_WORKSPACE; }
  if (L->N == 0) return GT_OK;
  const GinSaved s = gin_saved(L, saved);
  const int64_t N = L->N, D = L->D;
  const void* x = h_in;
  if (L->has_vn && !L->x_has_vn) {  // h_list[layer] = h_list[layer] + vn[batch]   (gnn_module.py:199)
    GT_TRY(gt_segment_bcast_add(GT_F32, h_in, vn, L->node_graph, N, L->B, D, x_out, st));
    x = x_out;
  }
  if (L->has_vn && L->ev_x_ready) GT_TRY(gt_event_record(L->ev_x_ready, st));   // x (with its virtual-node add) is complete
  // GINConv: mlp((1 + eps) x + sum_k relu(x_j + e_k))   (conv.py:26-36)
  GT_TRY(gt_aggregate_fwd(GT_CONV_GIN, L->edge_mode, GT_F32, x, N, L->E, D, L->in_ptr, L->in_src, L->in_eid, nullptr, nullptr,
                          L->eps, L->edge_attr, L->edge_cols, L->edge_w, L->edge_b, L->tab_off, L->table_rows, nullptr, s.agg, st));
  GT_TRY(gt_linear_fwd(GT_F32, GT_F32, L->compute, s.agg, L->w1, L->b1, s.z1, N, 2 * D, D, 0, 0.f, 0, st));
  GT_TRY(gt_batchnorm_fwd(GT_F32, s.z1, L->bn1_w, L->bn1_b, L->bn1_rm, L->bn1_rv, L->training ? L->bn1_nbt : nullptr,
                          L->bn_momentum, L->bn_eps, L->training, 1, nullptr, N, 2 * D, s.a1, s.st1, s.st1 + 2 * D, 0.f, 0,
                          w.bn_ws, w.bn_ws_bytes, st));
  GT_TRY(gt_linear_fwd(GT_F32, GT_F32, L->compute, s.a1, L->w2, L->b2, s.z2, N, D, 2 * D, 0, 0.f, 0, st));
  // h = drop(batch_norm(h) [relu]) [+ h_list[layer]]   (gnn_module.py:204-212)
  GT_TRY(gt_batchnorm_fwd_bcast(GT_F32, s.z2, L->bn_w, L->bn_b, L->bn_rm, L->bn_rv, L->training ? L->bn_nbt : nullptr, L->bn_momentum,
                                L->bn_eps, L->training, L->relu, L->residual ? x : nullptr, L->vn_next,
                                L->vn_next ? L->node_graph : nullptr, L->vn_next ? L->ev_vn_next : nullptr, N, D, y, s.st, s.st + D,
                                L->dropout_p, L->seed, w.bn_ws, w.bn_ws_bytes, st));
  return GT_OK;
}

extern "C" int gt_gin_layer_bwd(const gt_gin_layer* L, const void* x, const void* dy, const void* dx_extra,
                                const void* saved, void* d_h_in, void* d_vn, float* grads, void* workspace,
                                size_t workspace_bytes, gt_stream_t st) {
  GT_TRY(gin_check("gt_gin_layer_bwd", L));
  GT_CHECK_ARG(x && dy && saved && d_h_in && grads && workspace, "null buffer");
  const GinWork w = gin_work(L, workspace);   // d_vn may be NULL: the caller pools d_h_in itself
  if (workspace_bytes < w.bytes) { gt_set_error("gt_gin_layer_bwd: workspace too small"); return GT_ERR_WORKSPACE; }
  if (L->N == 0) return GT_OK;
  const GinSaved s = gin_saved(L, const_cast<void*>(saved));
  const GinGrads g = gin_grads(L, grads);
  const int64_t N = L->N, D = L->D;
  GT_TRY(gt_batchnorm_bwd(GT_F32, s.z2, dy, L->bn_w, L->bn_b, s.st, s.st + D, L->training, L->relu, N, D, w.d_z2, g.bn_w, g.bn_b,
                          L->dropout_p, L->seed, w.bn_ws, w.bn_ws_bytes, st));
  GT_TRY(gt_linear_bwd_wt(GT_F32, GT_F32, L->compute, s.a1, L->w2, L->w2_t, w.d_z2, nullptr, nullptr, nullptr, w.d_a1, g.w2, g.b2, N, D,
                          2 * D, 0.f, w.lin_ws, w.lin_ws_bytes, st));
  GT_TRY(gt_batchnorm_bwd(GT_F32, s.z1, w.d_a1, L->bn1_w, L->bn1_b, s.st1, s.st1 + 2 * D, L->training, 1, N, 2 * D, w.d_z1,
                          g.bn1_w, g.bn1_b, 0.f, 0, w.bn_ws, w.bn_ws_bytes, st));
  GT_TRY(gt_linear_bwd_wt(GT_F32, GT_F32, L->compute, s.agg, L->w1, L->w1_t, w.d_z1, nullptr, nullptr, nullptr, w.d_agg, g.w1, g.b1, N,
                          2 * D, D, 0.f, w.lin_ws1, w.lin_ws_bytes, st));
  const bool adds = dx_extra || L->residual;
  void* dx_conv = adds ? w.d_x : d_h_in;
  GT_TRY(gt_aggregate_bwd(GT_CONV_GIN, L->edge_mode, GT_F32, x, w.d_agg, N, L->E, D, L->out_ptr, L->out_dst, L->out_eid, nullptr,
                          nullptr, L->eps, L->edge_attr, L->edge_cols, L->edge_w, L->edge_b, L->tab_off, L->table_rows, nullptr,
                          dx_conv, g.eps, g.edge_w, g.edge_b, nullptr, w.agg_ws, w.agg_ws_bytes, st));
  // d_x = conv gradient (+ grads reaching x from its other consumers) (+ dy through the residual branch)
  if (L->ev_dx_wait) GT_TRY(gt_stream_wait_event(st, L->ev_dx_wait));
  if (adds) {
    const float* e1 = dx_extra ? (const float*)dx_extra : (const float*)dy;
    const float* e2 = (dx_extra && L->residual) ? (const float*)dy : nullptr;
    GT_TRY(gt_add3((const float*)w.d_x, e1, e2, N * D, (float*)d_h_in, st));
  }
  if (L->has_vn && d_vn) GT_TRY(gt_segment_sum_ws(GT_F32, d_h_in, nullptr, L->graph_ptr, N, L->B, D, d_vn, w.seg_ws, w.seg_ws_bytes, st));
  return GT_OK;
}

// =================================================================================================
// The LAST encoder layer under cls / last pooling: only one row per sequence is read afterwards (transformer_out[-1],
// models/gnn_transformer.py:113-114; the reference computes every row and drops the rest).  Keys and values are needed for every row,
// everything behind the attention only for the pooled row: in_proj on all rows, attention for the tile of the pooled row, then
// out_proj / norm1 / FFN / norm2 on B rows.  The skipped rows influence neither the output nor any gradient.
namespace {
struct EncPoolSaved {
  void *qkv, *ctx, *xp, *ctxp, *a, *x1, *f1, *f2, *g1;
  float *lse, *st1, *st2;
  size_t bytes;
};
EncPoolSaved encp_saved(const gt_encoder_layer* L, void* p) {
  Bump b(p);
  const size_t e = elt(L->dtype);
  const size_t R = (size_t)L->rows, B = (size_t)L->num_seqs, d = (size_t)L->d_model, F = (size_t)L->ffn;
  EncPoolSaved s;
  s.qkv = b.take(R * 3 * d * e);
  s.ctx = b.take(R * d * e);
  s.lse = (float*)b.take((size_t)2 * L->nhead * R * 4);
  s.xp = b.take(B * d * e);
  s.ctxp = b.take(B * d * e);
  s.a = b.take(B * d * e);
  s.x1 = b.take(B * d * e);
  s.f1 = b.take(B * F * e);
  s.f2 = b.take(B * d * e);
  s.g1 = b.take(L->act == 1 ? B * F * e : 0);
  s.st1 = (float*)b.take(2 * B * 4);
  s.st2 = (float*)b.take(2 * B * 4);
  s.bytes = b.off;
  return s;
}
struct EncPoolWork {
  void *d_f2, *d_x1, *d_f1, *d_a, *d_ctxp, *d_xp, *d_ctx, *d_qkv, *lin_ws, *lin_ws_in, *ln_ws, *ln_ws1;
  float* delta;
  size_t lin_ws_bytes, lin_ws_in_bytes, ln_ws_bytes, bytes;
};
EncPoolWork encp_work(const gt_encoder_layer* L, void* p) {
  Bump b(p);
  const size_t e = elt(L->dtype);
  const int64_t R = L->rows, B = L->num_seqs, d = L->d_model, F = L->ffn;
  const int c = L->dtype == GT_BF16 ? GT_BF16 : L->compute;
  EncPoolWork w;
  w.d_f2 = b.take((size_t)B * d * e);
  w.d_x1 = b.take((size_t)B * d * e);
  w.d_f1 = b.take((size_t)B * F * e);
  w.d_a = b.take((size_t)B * d * e);
  w.d_ctxp = b.take((size_t)B * d * e);
  w.d_xp = b.take((size_t)B * d * e);
  w.d_ctx = b.take((size_t)R * d * e);
  w.d_qkv = b.take((size_t)R * 3 * d * e);
  w.delta = (float*)b.take((size_t)L->nhead * R * 4);
  size_t m = 0, q;
  q = gt_linear_bwd_workspace_bytes(c, B, d, d); m = q > m ? q : m;
  q = gt_linear_bwd_workspace_bytes(c, B, F, d); m = q > m ? q : m;
  q = gt_linear_bwd_workspace_bytes(c, B, d, F); m = q > m ? q : m;
  w.lin_ws_bytes = m;
  w.lin_ws = b.take(m);
  w.lin_ws_in_bytes = gt_linear_bwd_workspace_bytes(c, R, 3 * d, d);
  w.lin_ws_in = b.take(w.lin_ws_in_bytes);
  w.ln_ws_bytes = gt_layernorm_bwd_workspace_bytes(B, d);
  w.ln_ws = b.take(w.ln_ws_bytes);
  w.ln_ws1 = b.take(w.ln_ws_bytes);
  w.bytes = b.off;
  return w;
}
}  // namespace

extern "C" size_t gt_encoder_layer_pooled_saved_bytes(const gt_encoder_layer* L) { return L ? encp_saved(L, nullptr).bytes : 0; }
extern "C" size_t gt_encoder_layer_pooled_workspace_bytes(const gt_encoder_layer* L) { return L ? encp_work(L, nullptr).bytes : 0; }

extern "C" int gt_encoder_layer_pooled_fwd(const gt_encoder_layer* L, const void* x, const int64_t* pool_rows, void* y_pool, void* saved,
                                           gt_stream_t st) {
  GT_TRY(enc_check("gt_encoder_layer_pooled_fwd", L));
  GT_CHECK_ARG(x && pool_rows && y_pool && saved, "null buffer");
  if (L->rows == 0 || L->num_seqs == 0) return GT_OK;
  const EncPoolSaved s = encp_saved(L, saved);
  const int t = L->dtype, c = t == GT_BF16 ? GT_BF16 : L->compute;
  const int64_t R = L->rows, B = L->num_seqs, d = L->d_model, F = L->ffn;
  const float p = L->training ? L->dropout_p : 0.f;
  const float scale = 1.0f / sqrtf((float)(d / L->nhead));
  GT_TRY(gt_linear_fwd(t, t, c, x, L->in_w, L->in_b, s.qkv, R, 3 * d, d, 0, 0.f, 0, st));
  GT_TRY(gt_attn_fwd_last(t, s.qkv, s.ctx, s.lse, R, d, L->nhead, L->seq_desc, L->num_seqs, L->row_stride, L->max_npos, scale, p, L->seed, st));
  GT_TRY(gt_rows_take(t, x, pool_rows, B, d, s.xp, st));
  GT_TRY(gt_rows_take(t, s.ctx, pool_rows, B, d, s.ctxp, st));
  // from here on: the same launches as gt_encoder_layer_fwd on B rows
  if (gt_linear_layernorm_fwd_ok(t, c, L->out_w, B, d, d)) {
    GT_TRY(gt_linear_layernorm_fwd(t, c, s.ctxp, L->out_w, L->out_b, s.a, B, d, d, s.xp, L->n1_w, L->n1_b, L->ln_eps, p,
                                   L->seed ^ 0x5851F42D4C957F2DULL, s.x1, s.st1, s.st1 + B, st));
  } else {
    GT_TRY(gt_linear_fwd(t, t, c, s.ctxp, L->out_w, L->out_b, s.a, B, d, d, 0, 0.f, 0, st));
    GT_TRY(gt_layernorm_fwd(t, s.a, s.xp, L->n1_w, L->n1_b, L->ln_eps, p, L->seed ^ 0x5851F42D4C957F2DULL, B, d, s.x1, s.st1, s.st1 + B, st));
  }
  if (L->act == 1)
    GT_TRY(gt_linear_fwd_gelu(t, t, c, s.x1, L->l1_w, L->l1_b, s.f1, s.g1, B, F, d, d, F, p, L->seed ^ 0x2545F4914F6CDD1DULL, st));
  else
    GT_TRY(gt_linear_fwd(t, t, c, s.x1, L->l1_w, L->l1_b, s.f1, B, F, d, 1, p, L->seed ^ 0x2545F4914F6CDD1DULL, st));
  if (gt_linear_layernorm_fwd_ok(t, c, L->l2_w, B, d, F)) {
    GT_TRY(gt_linear_layernorm_fwd(t, c, s.f1, L->l2_w, L->l2_b, s.f2, B, d, F, s.x1, L->n2_w, L->n2_b, L->ln_eps, p,
                                   L->seed ^ 0x14057B7EF767814FULL, y_pool, s.st2, s.st2 + B, st));
  } else {
    GT_TRY(gt_linear_fwd(t, t, c, s.f1, L->l2_w, L->l2_b, s.f2, B, d, F, 0, 0.f, 0, st));
    GT_TRY(gt_layernorm_fwd(t, s.f2, s.x1, L->n2_w, L->n2_b, L->ln_eps, p, L->seed ^ 0x14057B7EF767814FULL, B, d, y_pool, s.st2, s.st2 + B, st));
  }
  return GT_OK;
}

extern "C" int gt_encoder_layer_pooled_bwd(const gt_encoder_layer* L, const void* x, const int64_t* pool_rows, const void* dy_pool,
                                           const void* saved, void* dx, float* grads, void* workspace, size_t workspace_bytes,
                                           gt_stream_t st) {
  GT_TRY(enc_check("gt_encoder_layer_pooled_bwd", L));
  GT_CHECK_ARG(x && pool_rows && dy_pool && saved && dx && grads && workspace, "null buffer");
  const EncPoolWork w = encp_work(L, workspace);
  if (workspace_bytes < w.bytes) { gt_set_error("gt_encoder_layer_pooled_bwd: workspace too small"); return GT_ERR_WORKSPACE; }
  if (L->rows == 0 || L->num_seqs == 0) return GT_OK;
  const EncPoolSaved s = encp_saved(L, const_cast<void*>(saved));
  const EncGrads g = enc_grads(L, grads);
  const int t = L->dtype, c = t == GT_BF16 ? GT_BF16 : L->compute;
  const int64_t R = L->rows, B = L->num_seqs, d = L->d_model, F = L->ffn;
  const float p = L->training ? L->dropout_p : 0.f;
  const float scale = 1.0f / sqrtf((float)(d / L->nhead));
  // ---- the row-wise half on the B pooled rows (gt_encoder_layer_bwd's launches)
  GT_TRY(gt_layernorm_bwd(t, s.f2, s.x1, dy_pool, L->n2_w, s.st2, s.st2 + B, p, L->seed ^ 0x14057B7EF767814FULL, B, d, w.d_f2, w.d_x1, g.n2_w,
                          g.n2_b, w.ln_ws, w.ln_ws_bytes, st));
  if (gt_linear_bwd_gate_out_ok(t, t, c, L->l2_w, B, d, F)) {
    GT_TRY(gt_linear_bwd_gate_out(t, t, c, s.f1, L->l2_w, w.d_f2, L->act == 1 ? s.g1 : s.f1, nullptr, nullptr, w.d_f1, g.l2_w, g.l2_b, B, d, F,
                                  F, d, L->act == 1 ? -1.f : p, w.lin_ws, w.lin_ws_bytes, st));
    GT_TRY(gt_linear_bwd(t, t, c, s.x1, L->l1_w, w.d_f1, nullptr, w.d_x1, nullptr, w.d_x1, g.l1_w, g.l1_b, B, F, d, 0.f, w.lin_ws,
                         w.lin_ws_bytes, st));
  } else {
    GT_TRY(gt_linear_bwd(t, t, c, s.f1, L->l2_w, w.d_f2, nullptr, nullptr, nullptr, w.d_f1, g.l2_w, g.l2_b, B, d, F, 0.f, w.lin_ws,
                         w.lin_ws_bytes, st));
    if (L->act == 1)
      GT_TRY(gt_linear_bwd_mul(t, t, c, s.x1, L->l1_w, w.d_f1, s.g1, w.d_x1, nullptr, w.d_x1, g.l1_w, g.l1_b, B, F, d, d, F, w.lin_ws,
                               w.lin_ws_bytes, st));
    else
      GT_TRY(gt_linear_bwd(t, t, c, s.x1, L->l1_w, w.d_f1, s.f1, w.d_x1, nullptr, w.d_x1, g.l1_w, g.l1_b, B, F, d, p, w.lin_ws,
                           w.lin_ws_bytes, st));
  }
  GT_TRY(gt_layernorm_bwd(t, s.a, s.xp, w.d_x1, L->n1_w, s.st1, s.st1 + B, p, L->seed ^ 0x5851F42D4C957F2DULL, B, d, w.d_a, w.d_xp, g.n1_w,
                          g.n1_b, w.ln_ws1, w.ln_ws_bytes, st));
  GT_TRY(gt_linear_bwd(t, t, c, s.ctxp, L->out_w, w.d_a, nullptr, nullptr, nullptr, w.d_ctxp, g.out_w, g.out_b, B, d, d, 0.f, w.lin_ws,
                       w.lin_ws_bytes, st));
  // ---- attention: d_ctx is zero outside the pooled rows; dQ exists for the last tiles only, dK / dV for every row
  GT_TRY(gt_rows_put(t, w.d_ctxp, pool_rows, B, R, d, w.d_ctx, st));
  if (hipMemsetAsync(w.d_qkv, 0, (size_t)R * 3 * d * elt(t), (hipStream_t)st) != hipSuccess) {
    gt_set_error("gt_encoder_layer_pooled_bwd: memset failed");
    return GT_ERR_LAUNCH;
  }
  GT_TRY(gt_attn_bwd_last(t, s.qkv, s.ctx, w.d_ctx, s.lse, w.delta, w.d_qkv, R, d, L->nhead, L->seq_desc, L->num_seqs, L->row_stride,
                          L->max_npos, L->work_items, L->num_work, scale, p, L->seed, st));
  // ---- in_proj on every row; the residual branch's gradient exists in the pooled rows only
  GT_TRY(gt_linear_bwd_dw_forked(t, t, c, x, L->in_w, w.d_qkv, nullptr, g.in_w, g.in_b, R, 3 * d, d, d, 3 * d, 0.f, w.lin_ws_in,
                                 w.lin_ws_in_bytes, st));
  GT_TRY(gt_linear_bwd(t, t, c, x, L->in_w, w.d_qkv, nullptr, nullptr, nullptr, dx, nullptr, nullptr, R, 3 * d, d, 0.f, w.lin_ws_in,
                       w.lin_ws_in_bytes, st));
  GT_TRY(gt_rows_add(t, w.d_xp, pool_rows, B, d, dx, st));
  return GT_OK;
}

// =================================================================================================
// PNA layer (modules/pna/pna_module.py:57-78; PNAConv math: modules/pna_layer.py:131-167)
namespace {
struct PnaSaved {
  float *UV, *in5, *mean_v, *out, *z, *stats;
  int32_t* arg;
  size_t bytes;
};
PnaSaved pna_saved(const gt_pna_layer* L, void* p) {
  Bump b(p);
  PnaSaved s;
  const size_t ND = (size_t)L->N * L->D;
  s.UV = (float*)b.take(ND * 2 * 4);
  s.in5 = (float*)b.take(ND * 5 * 4);
  s.mean_v = (float*)b.take(ND * 4);
  s.arg = (int32_t*)b.take(ND * 2 * 4);
  s.out = (float*)b.take(ND * 4);
  s.z = (float*)b.take(ND * 4);
  s.stats = (float*)b.take((size_t)2 * L->D * 4);
  s.bytes = b.off;
  return s;
}
struct PnaWork {
  float *Y, *g, *d_z, *d_out, *d_in5, *dUV, *dxpart;
  void *bn_ws, *lin_ws, *post_ws, *pre_ws;   // one GEMM workspace each: a forked weight-gradient GEMM still uses its own while the next runs
  size_t bn_ws_bytes, lin_ws_bytes, post_ws_bytes, pre_ws_bytes, bytes;
};
PnaWork pna_work(const gt_pna_layer* L, void* p) {
  Bump b(p);
  PnaWork w;
  const size_t ND = (size_t)L->N * L->D;
  const int64_t F = L->D / L->T;
  w.Y = (float*)b.take(ND * L->S * 4);        // forward: the post-GEMM's S output blocks; backward: their gradient
  w.g = (float*)b.take(ND * 4);
  w.d_z = (float*)b.take(ND * 4);
  w.d_out = (float*)b.take(ND * 4);
  w.d_in5 = (float*)b.take(ND * 5 * 4);
  w.dUV = (float*)b.take(ND * 2 * 4);
  w.dxpart = (float*)b.take(ND * 4);
  w.bn_ws_bytes = gt_batchnorm_workspace_bytes(L->N, L->D);
  w.bn_ws = b.take(w.bn_ws_bytes);
  w.lin_ws_bytes = gt_linear_bwd_workspace_bytes(L->compute, L->N, L->D, L->D);
  w.lin_ws = b.take(w.lin_ws_bytes);
  w.post_ws_bytes = gt_linear_bwd_grouped_workspace_bytes(L->compute, L->N, L->S * F, 5 * F, L->T);
  w.post_ws = b.take(w.post_ws_bytes);
  w.pre_ws_bytes = gt_linear_bwd_grouped_workspace_bytes(L->compute, L->N, 2 * F, F, L->T);
  w.pre_ws = b.take(w.pre_ws_bytes);
  w.bytes = b.off;
  return w;
}
int pna_layer_check(const char* fn, const gt_pna_layer* L) {
  if (!L) { gt_set_error("%s: null descriptor", fn); return GT_ERR_INVALID_ARG; }
  if (L->N < 0 || L->D <= 0 || L->T <= 0 || L->D % L->T || (L->D / L->T) % 4 || L->D > 1024 || L->S < 1 || L->S > 8) {
    gt_set_error("%s: bad sizes (need D %% T == 0, (D / T) %% 4 == 0, D <= 1024, 1 <= S <= 8)", fn);
    return GT_ERR_INVALID_ARG;
  }
  if (!(L->pre_w && L->pre_b && L->post_w && L->post_b && L->lin_w && L->lin_b && L->bn_w && L->bn_b && L->scales && L->in_ptr)) {
    gt_set_error("%s: null parameter / structure pointer", fn);
    return GT_ERR_INVALID_ARG;
  }
  return GT_OK;
}
}  // namespace

extern "C" size_t gt_pna_layer_saved_bytes(const gt_pna_layer* L) { return L ? pna_saved(L, nullptr).bytes : 0; }
extern "C" size_t gt_pna_layer_workspace_bytes(const gt_pna_layer* L) { return L ? pna_work(L, nullptr).bytes : 0; }
extern "C" int64_t gt_pna_layer_grad_elems(const gt_pna_layer* L) { return L ? L->D * L->D + 3 * L->D : 0; }

extern "C" int gt_pna_layer_fwd(const gt_pna_layer* L, const void* x, void* y, void* saved, void* workspace, size_t workspace_bytes,
                                gt_stream_t st) {
  GT_TRY(pna_layer_check("gt_pna_layer_fwd", L));
  GT_CHECK_ARG(x && y && saved && workspace, "null buffer");
  const PnaWork w = pna_work(L, workspace);
  if (workspace_bytes < w.bytes) { gt_set_error("gt_pna_layer_fwd: workspace too small"); return GT_ERR_WORKSPACE; }
  if (L->N == 0) return GT_OK;
  const PnaSaved s = pna_saved(L, saved);
  const int64_t N = L->N, D = L->D, F = D / L->T;
  const int T = L->T, S = L->S;
  // [U_t | V_t] = x_t [A_t ; B_t]^T + [b_t | 0]: the per-edge Linear(2F, F) split into its target-role and source-role halves
  GT_TRY(gt_linear_fwd_grouped(GT_F32, GT_F32, L->compute, x, L->pre_w, L->pre_b, s.UV, N, 2 * F, F, D, 2 * D, T, F, 2 * F, 0, 0.f, 0, st));
  GT_TRY(gt_pna_aggregate_fwd_uv(s.UV, (const float*)x, N, D, T, L->in_ptr, L->in_src, L->in_eid, s.in5, s.mean_v, s.arg, st));
  // the post-Linear once on [x | agg]; its S per-scaler output blocks are combined with the degree scalers
  GT_TRY(gt_linear_fwd_grouped(GT_F32, GT_F32, L->compute, s.in5, L->post_w, L->post_b, w.Y, N, S * F, 5 * F, 5 * D, S * D, T, 5 * F, S * F, 0,
                               0.f, 0, st));
  GT_TRY(gt_scale_combine_fwd(w.Y, L->scales, N, T, S, (int)F, s.out, st));
  GT_TRY(gt_linear_fwd(GT_F32, GT_F32, L->compute, s.out, L->lin_w, L->lin_b, s.z, N, D, D, 0, 0.f, 0, st));
  // h = relu(batch_norm(conv(x))); x = h + x; x = dropout(x)   (pna_module.py:73-78: the dropout follows the residual add)
  GT_TRY(gt_batchnorm_fwd(GT_F32, s.z, L->bn_w, L->bn_b, L->bn_rm, L->bn_rv, L->training ? L->bn_nbt : nullptr, L->bn_momentum, L->bn_eps,
                          L->training, 1, x, N, D, y, s.stats, s.stats + D, 0.f, 0, w.bn_ws, w.bn_ws_bytes, st));
  if (L->training && L->dropout_p > 0.f) GT_TRY(gt_dropout(GT_F32, y, y, N * D, L->dropout_p, L->seed, st));
  return GT_OK;
}

extern "C" int gt_pna_layer_bwd(const gt_pna_layer* L, const void* x, const void* dy, const void* saved, void* dx, float* grads,
                                void* workspace, size_t workspace_bytes, gt_stream_t st) {
  GT_TRY(pna_layer_check("gt_pna_layer_bwd", L));
  GT_CHECK_ARG(x && dy && saved && dx && grads && workspace, "null buffer");
  GT_CHECK_ARG(L->d_pre_w && L->d_pre_b && L->d_post_w && L->d_post_b && L->out_ptr, "null image-gradient / structure pointer");
  const PnaWork w = pna_work(L, workspace);
  if (workspace_bytes < w.bytes) { gt_set_error("gt_pna_layer_bwd: workspace too small"); return GT_ERR_WORKSPACE; }
  if (L->N == 0) return GT_OK;
  const PnaSaved s = pna_saved(L, const_cast<void*>(saved));
  const int64_t N = L->N, D = L->D, F = D / L->T;
  const int T = L->T, S = L->S;
  float *g_lin_w = grads, *g_lin_b = grads + D * D, *g_bn_w = g_lin_b + D, *g_bn_b = g_bn_w + D;
  const void* g = dy;
  if (L->training && L->dropout_p > 0.f) {   // the mask is a function of (element, seed): the same call on the gradient
    GT_TRY(gt_dropout(GT_F32, dy, w.g, N * D, L->dropout_p, L->seed, st));
    g = w.g;
  }
  GT_TRY(gt_batchnorm_bwd(GT_F32, s.z, g, L->bn_w, L->bn_b, s.stats, s.stats + D, L->training, 1, N, D, w.d_z, g_bn_w, g_bn_b, 0.f, 0, w.bn_ws,
                          w.bn_ws_bytes, st));
  GT_TRY(gt_linear_bwd(GT_F32, GT_F32, L->compute, s.out, L->lin_w, w.d_z, nullptr, nullptr, nullptr, w.d_out, g_lin_w, g_lin_b, N, D, D, 0.f,
                       w.lin_ws, w.lin_ws_bytes, st));
  GT_TRY(gt_scale_combine_bwd(w.d_out, L->scales, N, T, S, (int)F, w.Y, st));
  GT_TRY(gt_linear_bwd_grouped(GT_F32, GT_F32, L->compute, s.in5, L->post_w, w.Y, nullptr, nullptr, nullptr, w.d_in5, L->d_post_w, L->d_post_b,
                               N, S * F, 5 * F, 5 * D, S * D, T, 5 * F, S * F, 0.f, w.post_ws, w.post_ws_bytes, st));
  GT_TRY(gt_pna_aggregate_bwd_uv(s.UV, s.in5, s.mean_v, s.arg, w.d_in5, N, D, T, L->in_ptr, L->out_ptr, L->out_dst, L->out_eid, w.dUV,
                                 w.dxpart, st));
  // dx = dUV [A ; B] + (the x block of the post-Linear's operand) + (the residual branch)
  GT_TRY(gt_linear_bwd_grouped(GT_F32, GT_F32, L->compute, x, L->pre_w, w.dUV, nullptr, w.dxpart, g, dx, L->d_pre_w, L->d_pre_b, N, 2 * F, F, D,
                               2 * D, T, F, 2 * F, 0.f, w.pre_ws, w.pre_ws_bytes, st));
  return GT_OK;
}
