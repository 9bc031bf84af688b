// pna.hip — PNA multi-aggregator message passing (mean / max / min / std in ONE pass over the in-edges).
//
// Reference (paths under /root/reference): modules/pna/pna_module.py:43-51,73 instantiates PyG's
// PNAConv(towers=4, divide_input=True, no edge features) — third-party code that is not in the tree;
// its math is stated in-tree by modules/pna_layer.py:131-167 (message / aggregate / scalers),
// modules/pna/aggregators.py:11-34 and modules/pna/scalers.py:10-31:
//     m_k   = pre_nn_t([x_i || x_j])            per edge k = (j -> i) and tower t      (Linear(2F, F))
//     agg_i = [mean_k m_k | max_k m_k | min_k m_k | std_k m_k],  std = sqrt(relu(E[m^2]-E[m]^2) + 1e-5)
// The per-edge Linear is split algebraically: m_k = U[i] + V[j] with U = x A_t^T + b, V = x B_t^T
// (two per-node GEMMs instead of an E x 2F GEMM), so
//     mean_i = U[i] + mean_k V[j_k]   max_i = U[i] + max_k V[j_k]   min_i = U[i] + min_k V[j_k]
//     std_i  = sqrt(relu(E[V^2] - E[V]^2) + 1e-5)                 (the shift U[i] cancels)
// and empty neighbourhoods give 0 / 0 / 0 / sqrt(1e-5) (torch-scatter fills empty segments with 0).
// Forward: one wave-tile per destination node over the destination-sorted CSR (no atomics).
// Backward: one wave-tile per SOURCE node over the CSC; the max/min winners are identified by the
// original edge id stored in the forward (first maximum in edge order).
#include "gt_common.h"

namespace {

constexpr int PT = 256;

struct PnaArgs {
  const float* U;
  const float* V;
  const int32_t* ptr;   // fwd: in_ptr ; bwd: out_ptr
  const int32_t* nbr;   // fwd: in_src ; bwd: out_dst
  const int32_t* eid;   // fwd: in_eid ; bwd: out_eid
  const int32_t* in_ptr;  // bwd: in-degree of the destination
  float* out;           // fwd: [N][T][4F]
  float* mean_v;        // [N][D]  mean_k V[j_k] (saved)
  int32_t* arg;         // [N][2][D] original edge ids of the max / min winners (saved)
  const float* g;       // bwd: grad wrt out [N][T][4F]
  float* dU;
  float* dV;
  int64_t N, D;
  int T, F;
  // layouts (elements): U / V / dU / dV element (n, t, f) at n * ldu + t * tsu + f; out / g aggregator block a of (n, t) at
  // n * ldo + t * tso + ao + a * F.  The stand-alone entry points use ldu = D, tsu = F, ldo = 4 D, tso = 4 F, ao = 0.
  int64_t ldu, ldo;
  int tsu, tso, ao;
  const float* xcopy;   // fwd, optional [N][D]: also written to out (n, t) columns [0, F) (the [x | agg] operand of the post-Linear)
  float* dxpart;        // bwd, optional [N][D]: receives g (n, t) columns [0, F) (the gradient of that copy)
};

// column chunk c (4 floats at columns col..col+3) of aggregator `agg` lives at tower-major offset:
__device__ __forceinline__ int64_t out_off(const PnaArgs& a, int64_t n, int col, int agg) {
  const int t = col / a.F, f = col % a.F;
  return n * a.ldo + (int64_t)t * a.tso + a.ao + agg * a.F + f;
}
__device__ __forceinline__ int64_t uv_off(const PnaArgs& a, int64_t n, int col) {
  const int t = col / a.F, f = col % a.F;
  return n * a.ldu + (int64_t)t * a.tsu + f;
}

template <int LPN, int NCH>
__global__ void __launch_bounds__(PT) k_pna_fwd(PnaArgs a) {
  constexpr int NPW = 64 / LPN;
  const int lane = threadIdx.x & 63;
  const int sub = lane / LPN, sl = lane % LPN;
  const int64_t v = ((int64_t)blockIdx.x * (PT / 64) + (threadIdx.x >> 6)) * NPW + sub;
  if (v >= a.N) return;
  const int D = (int)a.D;
  const int beg = a.ptr[v], end = a.ptr[v + 1];
  const float deg = (float)(end - beg);
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int col = (sl + j * 64) * 4;
    if (col >= D) continue;
    // Welford running mean / M2 per column: E[V^2] - E[V]^2 cancels catastrophically when the
    // neighbours' values nearly agree (variance ~1e-6), exactly where the std gradient 1/std is largest
    float4 mv = gt_zero4(), m2 = gt_zero4();
    float cnt = 0.f;
    float4 mx = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    float4 mn = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
    int4 amx = make_int4(-1, -1, -1, -1), amn = make_int4(-1, -1, -1, -1);
    for (int p = beg; p < end; ++p) {
      const int src = a.nbr[p], e = a.eid[p];
      const float4 x = *reinterpret_cast<const float4*>(a.V + uv_off(a, src, col));
      cnt += 1.f;
      const float ik = 1.0f / cnt;
      const float4 dl = make_float4(x.x - mv.x, x.y - mv.y, x.z - mv.z, x.w - mv.w);
      mv = gt_fma4(dl, ik, mv);
      m2 = make_float4(fmaf(dl.x, x.x - mv.x, m2.x), fmaf(dl.y, x.y - mv.y, m2.y), fmaf(dl.z, x.z - mv.z, m2.z),
                       fmaf(dl.w, x.w - mv.w, m2.w));
      if (x.x > mx.x) { mx.x = x.x; amx.x = e; }
      if (x.y > mx.y) { mx.y = x.y; amx.y = e; }
      if (x.z > mx.z) { mx.z = x.z; amx.z = e; }
      if (x.w > mx.w) { mx.w = x.w; amx.w = e; }
      if (x.x < mn.x) { mn.x = x.x; amn.x = e; }
      if (x.y < mn.y) { mn.y = x.y; amn.y = e; }
      if (x.z < mn.z) { mn.z = x.z; amn.z = e; }
      if (x.w < mn.w) { mn.w = x.w; amn.w = e; }
    }
    float4 mean = gt_zero4(), omax = gt_zero4(), omin = gt_zero4(), ostd;
    if (end > beg) {
      const float inv = 1.0f / deg;
      const float4 u = *reinterpret_cast<const float4*>(a.U + uv_off(a, v, col));
      mean = gt_add4(u, mv);
      omax = gt_add4(u, mx);
      omin = gt_add4(u, mn);
      ostd = make_float4(sqrtf(m2.x * inv + 1e-5f), sqrtf(m2.y * inv + 1e-5f), sqrtf(m2.z * inv + 1e-5f),
                         sqrtf(m2.w * inv + 1e-5f));
    } else {
      const float e0 = sqrtf(1e-5f);
      ostd = make_float4(e0, e0, e0, e0);
    }
    *reinterpret_cast<float4*>(a.out + out_off(a, v, col, 0)) = mean;
    *reinterpret_cast<float4*>(a.out + out_off(a, v, col, 1)) = omax;
    *reinterpret_cast<float4*>(a.out + out_off(a, v, col, 2)) = omin;
    *reinterpret_cast<float4*>(a.out + out_off(a, v, col, 3)) = ostd;
    if (a.xcopy) *reinterpret_cast<float4*>(a.out + out_off(a, v, col, 0) - a.ao) = *reinterpret_cast<const float4*>(a.xcopy + v * D + col);
    *reinterpret_cast<float4*>(a.mean_v + v * D + col) = mv;
    *reinterpret_cast<int4*>(a.arg + (v * 2 + 0) * D + col) = amx;
    *reinterpret_cast<int4*>(a.arg + (v * 2 + 1) * D + col) = amn;
  }
}

// backward: per source node u
//   dV[u] = sum_{k=(u->d)} g_mean[d]/deg_d + [eid_k == argmax_d] g_max[d] + [eid_k == argmin_d] g_min[d]
//                          + g_std[d] * 1[var_d > 0] * (V[u] - meanV[d]) / (deg_d * std_d)
//   dU[u] = (deg_u > 0) ? g_mean[u] + g_max[u] + g_min[u] : 0
template <int LPN, int NCH>
__global__ void __launch_bounds__(PT) k_pna_bwd(PnaArgs a) {
  constexpr int NPW = 64 / LPN;
  const int lane = threadIdx.x & 63;
  const int sub = lane / LPN, sl = lane % LPN;
  const int64_t u = ((int64_t)blockIdx.x * (PT / 64) + (threadIdx.x >> 6)) * NPW + sub;
  if (u >= a.N) return;
  const int D = (int)a.D;
  const int beg = a.ptr[u], end = a.ptr[u + 1];
  const bool has_in = a.in_ptr[u + 1] > a.in_ptr[u];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int col = (sl + j * 64) * 4;
    if (col >= D) continue;
    const float4 vu = *reinterpret_cast<const float4*>(a.V + uv_off(a, u, col));
    float4 acc = gt_zero4();
    for (int p = beg; p < end; ++p) {
      const int d = a.nbr[p], e = a.eid[p];
      const float degd = (float)(a.in_ptr[d + 1] - a.in_ptr[d]);
      const float inv = 1.0f / degd;
      const float4 gm = *reinterpret_cast<const float4*>(a.g + out_off(a, d, col, 0));
      const float4 gx = *reinterpret_cast<const float4*>(a.g + out_off(a, d, col, 1));
      const float4 gn = *reinterpret_cast<const float4*>(a.g + out_off(a, d, col, 2));
      const float4 gs = *reinterpret_cast<const float4*>(a.g + out_off(a, d, col, 3));
      const float4 sd = *reinterpret_cast<const float4*>(a.out + out_off(a, d, col, 3));
      const float4 mv = *reinterpret_cast<const float4*>(a.mean_v + (int64_t)d * D + col);
      const int4 ax = *reinterpret_cast<const int4*>(a.arg + ((int64_t)d * 2 + 0) * D + col);
      const int4 an = *reinterpret_cast<const int4*>(a.arg + ((int64_t)d * 2 + 1) * D + col);
      // relu'(var): var > 0  <=>  std^2 > 1e-5 (strictly)
      auto term = [&](float g_mean, float g_max, float g_min, float g_std, float s, float m, int amx, int amn, float x) {
        float t = g_mean * inv;
        if (amx == e) t += g_max;
        if (amn == e) t += g_min;
        if (s * s > 1e-5f) t += g_std * (x - m) * inv / s;
        return t;
      };
      acc.x += term(gm.x, gx.x, gn.x, gs.x, sd.x, mv.x, ax.x, an.x, vu.x);
      acc.y += term(gm.y, gx.y, gn.y, gs.y, sd.y, mv.y, ax.y, an.y, vu.y);
      acc.z += term(gm.z, gx.z, gn.z, gs.z, sd.z, mv.z, ax.z, an.z, vu.z);
      acc.w += term(gm.w, gx.w, gn.w, gs.w, sd.w, mv.w, ax.w, an.w, vu.w);
    }
    *reinterpret_cast<float4*>(a.dV + uv_off(a, u, col)) = acc;
    float4 du = gt_zero4();
    if (has_in) {
      du = gt_add4(gt_add4(*reinterpret_cast<const float4*>(a.g + out_off(a, u, col, 0)),
                           *reinterpret_cast<const float4*>(a.g + out_off(a, u, col, 1))),
                   *reinterpret_cast<const float4*>(a.g + out_off(a, u, col, 2)));
    }
    *reinterpret_cast<float4*>(a.dU + uv_off(a, u, col)) = du;
    if (a.dxpart) *reinterpret_cast<float4*>(a.dxpart + u * D + col) = *reinterpret_cast<const float4*>(a.g + out_off(a, u, col, 0) - a.ao);
  }
}

template <bool BWD>
void pna_launch(const PnaArgs& a, hipStream_t stream) {
#define GT_PNA(LPN, NCH)                                                                                  \
  do {                                                                                                    \
    int64_t waves = gt_cdiv(a.N, 64 / (LPN));                                                             \
    dim3 grid((unsigned)gt_cdiv(waves, PT / 64));                                                         \
    if constexpr (BWD) hipLaunchKernelGGL((k_pna_bwd<LPN, NCH>), grid, dim3(PT), 0, stream, a);           \
    else hipLaunchKernelGGL((k_pna_fwd<LPN, NCH>), grid, dim3(PT), 0, stream, a);                          \
  } while (0)
  if (a.D <= 64) GT_PNA(16, 1);
  else if (a.D <= 128) GT_PNA(32, 1);
  else if (a.D <= 256) GT_PNA(64, 1);
  else if (a.D <= 512) GT_PNA(64, 2);
  else if (a.D <= 768) GT_PNA(64, 3);
  else GT_PNA(64, 4);
#undef GT_PNA
}

int pna_check(const char* fn, int64_t N, int64_t D, int towers) {
  if (N < 0 || D <= 0 || towers <= 0 || D % towers != 0) { gt_set_error("%s: bad sizes", fn); return GT_ERR_INVALID_ARG; }
  if ((D / towers) % 4 != 0 || D > 1024) { gt_set_error("%s: need (dim / towers) %% 4 == 0 and dim <= 1024", fn); return GT_ERR_UNSUPPORTED; }
  return GT_OK;
}

// ---- degree scalers applied to the post-Linear's S output blocks (modules/pna/scalers.py:10-31):
//   out[n][t][f] = sum_s Y[n][t][s][f] * sc[n][s]        bwd: dY[n][t][s][f] = dout[n][t][f] * sc[n][s]
// (the post-Linear runs once on [x | agg]; scaling its per-scaler output blocks equals scaling the aggregates first)
__global__ void __launch_bounds__(256) k_scale_combine_fwd(const float* __restrict__ Y, const float* __restrict__ sc, int64_t N,
                                                           int T, int S, int F, float* __restrict__ out) {
  const int64_t F4 = F / 4, total = N * T * F4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i / (T * F4), r = i % (T * F4);
    const int t = (int)(r / F4), f = (int)(r % F4) * 4;
    float4 acc = gt_zero4();
    for (int s = 0; s < S; ++s)
      acc = gt_fma4(*reinterpret_cast<const float4*>(Y + ((n * T + t) * S + s) * F + f), sc[n * S + s], acc);
    *reinterpret_cast<float4*>(out + (n * T + t) * F + f) = acc;
  }
}

__global__ void __launch_bounds__(256) k_scale_combine_bwd(const float* __restrict__ dout, const float* __restrict__ sc, int64_t N,
                                                           int T, int S, int F, float* __restrict__ dY) {
  const int64_t F4 = F / 4, total = N * T * F4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i / (T * F4), r = i % (T * F4);
    const int t = (int)(r / F4), f = (int)(r % F4) * 4;
    const float4 g = *reinterpret_cast<const float4*>(dout + (n * T + t) * F + f);
    for (int s = 0; s < S; ++s)
      *reinterpret_cast<float4*>(dY + ((n * T + t) * S + s) * F + f) = gt_scale4(g, sc[n * S + s]);
  }
}

// per-node degree scalers (modules/pna/scalers.py:10-31): d = in-degree
struct ScaleKinds { int k[8]; };
__global__ void __launch_bounds__(256) k_pna_scales(const int32_t* __restrict__ in_ptr, int64_t N, int S, ScaleKinds kinds, float avg_log,
                                                    float avg_lin, float* __restrict__ sc) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const float d = (float)(in_ptr[n + 1] - in_ptr[n]);
  const float lg = logf(d + 1.f);
  for (int s = 0; s < S; ++s) {
    float v = 1.f;
    switch (kinds.k[s]) {
      case 1: v = lg / avg_log; break;
      case 2: v = d == 0.f ? 1.f : avg_log / lg; break;
      case 3: v = d / avg_lin; break;
      case 4: v = d == 0.f ? 1.f : avg_lin / d; break;
      default: break;
    }
    sc[n * S + s] = v;
  }
}

}  // namespace

extern "C" int gt_pna_scales(const int32_t* in_ptr, int64_t N, int S, const int32_t* kinds_host, float avg_log, float avg_lin, float* scales,
                             gt_stream_t stream_) {
  GT_CHECK_ARG(N >= 0 && S >= 1 && S <= 8 && kinds_host, "bad arguments");
  if (N == 0) return GT_OK;
  GT_CHECK_ARG(in_ptr && scales, "null buffer");
  ScaleKinds k{};
  for (int s = 0; s < S; ++s) k.k[s] = kinds_host[s];
  hipLaunchKernelGGL(k_pna_scales, dim3((unsigned)gt_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream_, in_ptr, N, S, k, avg_log, avg_lin, scales);
  GT_CHECK_LAUNCH();
  return GT_OK;
}

static void pna_classic(PnaArgs& a) {
  a.ldu = a.D; a.tsu = a.F; a.ldo = 4 * a.D; a.tso = 4 * a.F; a.ao = 0;
}

extern "C" int gt_pna_aggregate_fwd(const float* U, const float* V, int64_t N, int64_t D, int towers, const int32_t* in_ptr,
                                    const int32_t* in_src, const int32_t* in_eid, float* out, float* mean_v, int32_t* arg,
                                    gt_stream_t stream_) {
  int rc = pna_check("gt_pna_aggregate_fwd", N, D, towers);
  if (rc) return rc;
  GT_CHECK_ARG(U && V && in_ptr && out && mean_v && arg, "null buffer");
  if (N == 0) return GT_OK;
  PnaArgs a{};
  a.U = U; a.V = V; a.ptr = in_ptr; a.nbr = in_src; a.eid = in_eid; a.out = out; a.mean_v = mean_v; a.arg = arg;
  a.N = N; a.D = D; a.T = towers; a.F = (int)(D / towers);
  pna_classic(a);
  GtProfScope prof__(GT_PROF_AGGREGATE, "gt_pna_aggregate_fwd", stream_, {N, in_ptr ? 0 : 0, D, 4, 0, 0});
  pna_launch<false>(a, (hipStream_t)stream_);
  GT_CHECK_LAUNCH();
  return GT_OK;
}

extern "C" int gt_pna_aggregate_bwd(const float* V, const float* out, const float* mean_v, const int32_t* arg,
                                    const float* grad_out, int64_t N, int64_t D, int towers, const int32_t* in_ptr,
                                    const int32_t* out_ptr, const int32_t* out_dst, const int32_t* out_eid, float* dU,
                                    float* dV, gt_stream_t stream_) {
  int rc = pna_check("gt_pna_aggregate_bwd", N, D, towers);
  if (rc) return rc;
  GT_CHECK_ARG(V && out && mean_v && arg && grad_out && in_ptr && out_ptr && dU && dV, "null buffer");
  if (N == 0) return GT_OK;
  PnaArgs a{};
  a.V = V; a.out = const_cast<float*>(out); a.mean_v = const_cast<float*>(mean_v); a.arg = const_cast<int32_t*>(arg);
  a.g = grad_out; a.ptr = out_ptr; a.nbr = out_dst; a.eid = out_eid; a.in_ptr = in_ptr; a.dU = dU; a.dV = dV;
  a.N = N; a.D = D; a.T = towers; a.F = (int)(D / towers);
  pna_classic(a);
  GtProfScope prof__(GT_PROF_AGGREGATE, "gt_pna_aggregate_bwd", stream_, {N, 0, D, 4, 0, 0});
  pna_launch<true>(a, (hipStream_t)stream_);
  GT_CHECK_LAUNCH();
  return GT_OK;
}

// Layout of the fused PNA layer (gt_pna_layer_*, layers.hip): UV [N][T][2F] = [U_t | V_t] out of ONE grouped pre-GEMM, the
// aggregates written straight into the post-Linear's operand in5 [N][T][5F] = [x_t | mean | max | min | std] (x copied on the way),
// the backward reading its gradient from d_in5 [N][T][5F], writing dUV [N][T][2F] and the x block's gradient to dxpart [N][D].
extern "C" int gt_pna_aggregate_fwd_uv(const float* UV, const float* x, int64_t N, int64_t D, int towers, const int32_t* in_ptr,
                                       const int32_t* in_src, const int32_t* in_eid, float* in5, float* mean_v, int32_t* arg,
                                       gt_stream_t stream_) {
  int rc = pna_check("gt_pna_aggregate_fwd_uv", N, D, towers);
  if (rc) return rc;
  GT_CHECK_ARG(UV && x && in_ptr && in5 && mean_v && arg, "null buffer");
  if (N == 0) return GT_OK;
  PnaArgs a{};
  a.N = N; a.D = D; a.T = towers; a.F = (int)(D / towers);
  a.U = UV; a.V = UV + a.F; a.ptr = in_ptr; a.nbr = in_src; a.eid = in_eid; a.out = in5; a.mean_v = mean_v; a.arg = arg;
  a.ldu = 2 * D; a.tsu = 2 * a.F; a.ldo = 5 * D; a.tso = 5 * a.F; a.ao = a.F; a.xcopy = x;
  GtProfScope prof__(GT_PROF_AGGREGATE, "gt_pna_aggregate_fwd", stream_, {N, 0, D, 4, 0, 0});
  pna_launch<false>(a, (hipStream_t)stream_);
  GT_CHECK_LAUNCH();
  return GT_OK;
}

extern "C" int gt_pna_aggregate_bwd_uv(const float* UV, const float* in5, const float* mean_v, const int32_t* arg, const float* d_in5,
                                       int64_t N, int64_t D, int towers, const int32_t* in_ptr, const int32_t* out_ptr,
                                       const int32_t* out_dst, const int32_t* out_eid, float* dUV, float* dxpart, gt_stream_t stream_) {
  int rc = pna_check("gt_pna_aggregate_bwd_uv", N, D, towers);
  if (rc) return rc;
  GT_CHECK_ARG(UV && in5 && mean_v && arg && d_in5 && in_ptr && out_ptr && dUV && dxpart, "null buffer");
  if (N == 0) return GT_OK;
  PnaArgs a{};
  a.N = N; a.D = D; a.T = towers; a.F = (int)(D / towers);
  a.V = UV + a.F; a.out = const_cast<float*>(in5); a.mean_v = const_cast<float*>(mean_v); a.arg = const_cast<int32_t*>(arg);
  a.g = d_in5; a.ptr = out_ptr; a.nbr = out_dst; a.eid = out_eid; a.in_ptr = in_ptr; a.dU = dUV; a.dV = dUV + a.F;
  a.ldu = 2 * D; a.tsu = 2 * a.F; a.ldo = 5 * D; a.tso = 5 * a.F; a.ao = a.F; a.dxpart = dxpart;
  GtProfScope prof__(GT_PROF_AGGREGATE, "gt_pna_aggregate_bwd", stream_, {N, 0, D, 4, 0, 0});
  pna_launch<true>(a, (hipStream_t)stream_);
  GT_CHECK_LAUNCH();
  return GT_OK;
}

extern "C" int gt_scale_combine_fwd(const float* Y, const float* scales, int64_t N, int towers, int num_scalers, int F,
                                    float* out, gt_stream_t stream_) {
  GT_CHECK_ARG(N >= 0 && towers > 0 && num_scalers > 0 && F > 0 && F % 4 == 0, "F must be a positive multiple of 4");
  if (N == 0) return GT_OK;
  GT_CHECK_ARG(Y && scales && out, "null buffer");
  const int64_t items = N * towers * (F / 4);
  const int grid = (int)(gt_cdiv(items, 256) < 4096 ? gt_cdiv(items, 256) : 4096);
  hipLaunchKernelGGL(k_scale_combine_fwd, dim3(grid), dim3(256), 0, (hipStream_t)stream_, Y, scales, N, towers, num_scalers, F, out);
  GT_CHECK_LAUNCH();
  return GT_OK;
}

extern "C" int gt_scale_combine_bwd(const float* grad_out, const float* scales, int64_t N, int towers, int num_scalers, int F,
                                    float* dY, gt_stream_t stream_) {
  GT_CHECK_ARG(N >= 0 && towers > 0 && num_scalers > 0 && F > 0 && F % 4 == 0, "F must be a positive multiple of 4");
  if (N == 0) return GT_OK;
  GT_CHECK_ARG(grad_out && scales && dY, "null buffer");
  const int64_t items = N * towers * (F / 4);
  const int grid = (int)(gt_cdiv(items, 256) < 4096 ? gt_cdiv(items, 256) : 4096);
  hipLaunchKernelGGL(k_scale_combine_bwd, dim3(grid), dim3(256), 0, (hipStream_t)stream_, grad_out, scales, N, towers, num_scalers, F,
                     dY);
  GT_CHECK_LAUNCH();
  return GT_OK;
}
