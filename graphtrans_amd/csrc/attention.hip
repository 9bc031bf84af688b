// attention.hip — fused masked multi-head self-attention over graph-node token rows
// (flash-style forward + backward on the CDNA4 matrix cores; the S x S scores never reach memory).
//
// Reference semantics (paths under /root/reference):
//   modules/transformer_encoder.py:59  nn.TransformerEncoder -> F.multi_head_attention_forward:
//       q,k,v = in_proj(x); q *= hd^-1/2; att = bmm(q,k^T); masked_fill(key_padding_mask, -inf);
//       softmax; dropout(p); bmm(att, v)                       [B*nhead, S, S] fp32 x3 per layer
//   modules/utils.py:5-29              left-padded sequences: valid keys are a suffix
// This file replaces everything between in_proj and out_proj.  The padding mask is a per-sequence
// key range [kv_off, kv_off+kv_len) compared in-register; fully masked key tiles are skipped.
//
// Orientation ("everything transposed"): a wave owns 16 queries (fwd / dQ) or 16 keys (dK,dV) as
// the N (column) index of 16x16 MFMA tiles, so that per-query softmax statistics live in the lane
// that owns column n = lane & 15 and the score tile S^T[key][query] leaves the MFMA already in
// the B-operand layout of the following P.V product:
//     S^T = K Q^T      (A = K rows from LDS, B = Q from registers)
//     O^T += V^T P^T   (A = V^T via LDS transpose-read, B = P^T straight from the S^T registers)
// The 16 rows of an S^T tile are mapped to keys g*8 + t*4 + r (pi permutation on the A rows), so
// the two 16-key tiles of a 32-key step give each lane exactly its 8 consecutive k-slots.
//
// dtype GT_BF16: v_mfma_f32_16x16x32_bf16, operands bf16, accumulate fp32.
// dtype GT_F32 : v_mfma_f32_16x16x4_f32 x8 per 32-deep step: exact fp32 (fma chain), the parity mode.
#include <cstdlib>
#include <cstring>
#include "gt_common.h"
#include "mfma_frag.h"

namespace {
using namespace gtf;

constexpr int ATT_THREADS = 256;
constexpr int TILE = 32;    // keys (fwd, dQ) or queries (dK/dV) per inner step
constexpr int BLOCK_N = 64; // queries (fwd, dQ) or keys (dK/dV) per block: 4 waves x 16
constexpr float LOG2E = 1.4426950408889634f;

// v_exp_f32 directly: the arguments here are <= 0 (scores minus the running max), where the library
// exp2f's denormal-range rescaling (compare, select, two multiplies per call) buys nothing; results
// below 2^-126 flush to zero, which is what a softmax weight that small contributes anyway.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// ---- dropout RNG: counter-based hash of (seed, seq*nhead+head, query pos, key pos) ---------------
// The pre-mix is an XOR of a query/head part and a key part, so a lane that walks keys for one query (fwd, dQ)
// or queries for one key (dK/dV) hoists its fixed part and advances the other by one add (no per-pair multiply
// before the finalizer; 32-bit integer multiplies are quarter rate).
constexpr uint32_t RNG_CQ = 0x9E3779B1u, RNG_CK = 0x85EBCA77u, RNG_CH = 0xC2B2AE3Du;
__device__ __forceinline__ uint32_t rng_qpart(uint32_t s1, uint32_t bh, uint32_t q) { return (q * RNG_CQ) ^ (bh * RNG_CH + s1); }
__device__ __forceinline__ uint32_t rng_kpart(uint32_t s0, uint32_t k) { return k * RNG_CK + s0; }
__device__ __forceinline__ uint32_t rng_mix(uint32_t qpart, uint32_t kpart) {
  uint32_t x = qpart ^ kpart;
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t rng_hash(uint32_t s0, uint32_t s1, uint32_t bh, uint32_t q, uint32_t k) {
  return rng_mix(rng_qpart(s1, bh, q), rng_kpart(s0, k));
}

struct AttnArgs {
  const void* qkv;
  const void* ctx;    // fwd: output ; bwd: saved output
  const void* d_ctx;  // bwd
  float* lse;         // [2][nhead][rows]: running max m and log2(sum exp2(s - m)), log2 domain of the scaled scores
  float* delta;       // [nhead][rows]
  void* out;          // fwd: ctx ; bwd: d_qkv
  const int32_t* desc;
  const int32_t* work;  // optional [num_work][2] = {sequence, 64-row tile}: 1-D grid over real tiles only
  int num_work, work_per_xcd;  // work-list launches: see block_item
  // optional dense masks of CausalSelfAttention (modules/masked_transformer_encoder.py:44-47): entries == 0 are
  // FILLED with mask_fill (masked_fill semantics: finite value, no gradient through the score)
  const float* dense_mask;  // [num_seqs][npos][npos]
  const float* key_valid;   // [num_seqs][npos]
  float mask_fill2;         // mask_value * log2(e)
  int64_t rows, d_model, row_stride;
  int nhead;
  float scale_log2;   // scale * log2(e)
  float scale;
  float inv_keep;     // 1/(1-p)
  uint32_t drop_thr;  // keep iff the key's 16-bit half of its pair's hash >= thr ; 0 -> no dropout
  uint32_t seed0, seed1;
  // pooled mode (cls / last pooling reads ONE row per sequence after the last encoder layer, models/gnn_transformer.py:113-114):
  //   last_only   fwd / dQ: a block per (sequence, head) computes the 64-query tile that holds the sequence's LAST position only
  //   q_last_only dK / dV: every key tile, but the query loop starts at the 32-query step that holds the last position
  int last_only, q_last_only;
};

// head dims 8 / 16 are zero-padded to one 32-deep MFMA step (HDP); LDS rows are HDP wide + 16 B pad
__device__ __forceinline__ bool dense_masked(const AttnArgs& a, int seq, int qpos, int kp, int npos) {
  if (a.key_valid && a.key_valid[(int64_t)seq * npos + kp] == 0.f) return true;
  if (a.dense_mask && a.dense_mask[((int64_t)seq * npos + qpos) * npos + kp] == 0.f) return true;
  return false;
}

// (sequence, 64-row tile, head) of this block.  Work-list launches are 1-D and XCD-aware: workgroup b
// runs on XCD b % 8 (own L2); XCD x takes the x-th contiguous eighth of the (sequence-sorted) work list
// and walks it head-fastest, so every tile and head of a sequence -- which all read that sequence's K
// and V rows -- hit the same L2 (PMC before: 95 MB fetched per forward launch for 24.5 MB of qkv).
__device__ __forceinline__ bool block_item(const AttnArgs& a, int& seq, int& tile, int& head) {
  if (a.work && !a.last_only) {
    const int b = blockIdx.x;
    const int slot = b / 8;
    const int w = (b % 8) * a.work_per_xcd + slot / a.nhead;
    if (slot / a.nhead >= a.work_per_xcd || w >= a.num_work) return false;
    head = slot % a.nhead;
    seq = a.work[w * 2];
    tile = a.work[w * 2 + 1];
    if (seq < 0) return false;   // padding entry of a device-built work list (gt_seq_layout_packed)
  } else {
    head = blockIdx.y;
    seq = blockIdx.z;
    tile = blockIdx.x;
  }
  return true;
}

template <int HD, typename T>
struct Lds {
  static constexpr int HDP = HD < 32 ? 32 : HD;
  static constexpr int LD = HDP + (sizeof(T) == 2 ? 8 : 4);
};

// zero the LDS columns [HD, LD) that load_tile never writes (only needed when HD < 32)
template <typename T, int HD>
__device__ __forceinline__ void zero_pad_cols(T* lds) {
  constexpr int LD = Lds<HD, T>::LD;
  if constexpr (HD < 32) {
    for (int i = threadIdx.x; i < TILE * LD; i += ATT_THREADS) lds[i] = (T)0;
  }
}

// operand row fragment from global memory: dims [c0, c0+8) of a head row, zero beyond HD
template <typename T, int HD>
__device__ __forceinline__ Frag<T> frag_load_head(const T* head_row, int c0, bool valid) {
  return (valid && c0 < HD) ? frag_load(head_row + c0) : frag_zero<T>();
}

// cooperative load of a [TILE][HD] tile of rows (pos0 + r) into LDS, zero-filled outside [lo, hi)
template <typename T, int HD>
__device__ __forceinline__ void load_tile(T* lds, const T* src, int64_t src_ld, int64_t row0, int64_t row_stride,
                                          int pos0, int lo, int hi) {
  constexpr int LD = Lds<HD, T>::LD;
  constexpr int CH = HD / 8;
  for (int c = threadIdx.x; c < TILE * CH; c += ATT_THREADS) {
    int r = c / CH, col = (c % CH) * 8;
    int pos = pos0 + r;
    Frag<T> f = frag_zero<T>();
    if (pos >= lo && pos < hi) f = frag_load(src + (row0 + (int64_t)pos * row_stride) * src_ld + col);
    frag_store_lds(lds + r * LD + col, f);
  }
}

// Register-staged pair of [TILE][HD] tiles (K and V, or Q and dO): load() issues this thread's two 16-byte chunks of
// the NEXT key / query tile before the current tile's MFMAs, store() parks them in the other LDS buffer behind them --
// one barrier per 32-position step and the global latency off the step (the step of a 1000-token sequence is walked 32
// times in a row by one block: with a synchronous load between two barriers per step that block WAS the kernel time).
template <typename T, int HD>
struct TilePair {
  static constexpr int LD = Lds<HD, T>::LD;
  static constexpr int CH = HD / 8;
  Frag<T> fa, fb;
  int r, col;
  bool has, ok;
  const T* pa;   // this thread's chunk of the tile at position 0 of the sequence
  const T* pb;
  const T* za;   // this thread's chunk of a row that always exists: what a lane outside [lo, hi) fetches (and then zeroes)
  const T* zb;
  int64_t sa, sb;   // elements per position (wave-uniform)
  __device__ __forceinline__ void init(const T* srcA, int64_t ldA, const T* srcB, int64_t ldB, int64_t row0, int64_t row_stride, int safe_pos) {
    const int c = threadIdx.x;
    has = c < TILE * CH;
    r = c / CH;
    col = (c % CH) * 8;
    pa = srcA + (row0 + (int64_t)r * row_stride) * ldA + col;
    pb = srcB + (row0 + (int64_t)r * row_stride) * ldB + col;
    za = srcA + (row0 + (int64_t)safe_pos * row_stride) * ldA + col;
    zb = srcB + (row0 + (int64_t)safe_pos * row_stride) * ldB + col;
    sa = row_stride * ldA;
    sb = row_stride * ldB;
    ok = false;
  }
  // The tile at positions [pos0, pos0 + TILE): the per-thread row address is base + pos0 * (row_stride * ld) -- pos0 and the
  // stride are wave-uniform, so the 64-bit product is scalar work and the vector side is one 64-bit add per operand (the
  // per-lane (row0 + pos * row_stride) * ld of the first version was five quarter-rate multiplies per operand and step).
  // BRANCH-FREE: a lane whose position lies outside [lo, hi) loads the safe row instead and store() writes zeros for it.  With the
  // loads under an exec-masked branch (`ok ? load : zero`) hipcc's wait-count pass lost track of them across the loop and put
  // s_waitcnt vmcnt(0) in front of the step's FIRST MFMA -- every step waited out the latency of the prefetch it had just issued
  // (forward and dK/dV: 0.7-1.1 us per 32-position step).
  __device__ __forceinline__ void load(int pos0, int lo, int hi) {
    const int pos = pos0 + r;
    ok = has && pos >= lo && pos < hi;
    fa = frag_load(ok ? pa + (int64_t)pos0 * sa : za);
    fb = frag_load(ok ? pb + (int64_t)pos0 * sb : zb);
  }
  template <typename TO, bool PIN = false>
  __device__ __forceinline__ void store(typename TO::LT* sA, typename TO::LT* sB) const {
    // PIN (dK/dV): nothing of the store moves up across this point -- hipcc's scheduler otherwise hoists the zero-selects of fa / fb to
    // right behind the loads, and the wait for the loads with them, in front of the step's MFMAs
    if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
    if (has) {
      TO::store(sA, r, col, ok ? fa : frag_zero<T>());
      TO::store(sB, r, col, ok ? fb : frag_zero<T>());
    }
  }
};

// What a kernel does with its LDS tiles and MFMA operands, by storage type T and arithmetic:
//   plain (SP = false): tiles of T, operands Frag<T>, products by mma (bf16 MFMA; for fp32 rows the exact v_mfma_f32_16x16x4_f32 chains);
//   SP (fp32 rows, head dims 32 / 64): fp32-accurate products on the bf16 pipe ("bf16x6", mfma_frag.h: the arithmetic of the fp32
//     contract mode's GEMMs).  A tile is split ONCE, when it is staged, into three bf16 planes [plane][TILE][HDP + 8] -- so the row
//     fragments are three ds_read_b128 and the transposed ones come from ds_read_b64_tr_b16 like the bf16 kernels' (the fp32 tiles'
//     transposed fragment was eight strided ds_read_b32) --, the per-query / per-key register operands (Q, dO, K, V) once at load
//     time, the softmax weights / score gradients once per step.
template <typename T, int HD, bool SP>
struct TileOps {
  using LT = T;
  using Op = Frag<T>;
  static constexpr int LD = Lds<HD, T>::LD;
  static constexpr int ELEMS = TILE * LD;
  static __device__ __forceinline__ Op reg(const Frag<T>& f) { return f; }
  static __device__ __forceinline__ Op from(const float* x) { return frag_from_f32<T>(x); }
  static __device__ __forceinline__ Op row(const LT* tile, int r, int c) { return frag_load(tile + r * LD + c); }
  static __device__ __forceinline__ Op tr(const LT* tile, int col0, int n, int g) { return frag_load_tr(tile, LD, 0, col0, n, g); }
  static __device__ __forceinline__ void store(LT* tile, int r, int c, const Frag<T>& f) { frag_store_lds(tile + r * LD + c, f); }
  static __device__ __forceinline__ f32x4 mm(const Op& a, const Op& b, f32x4 c) { return mma(a, b, c); }
};
template <int HD>
struct TileOps<float, HD, true> {
  static_assert(HD >= 32, "whole 32-deep steps");
  using LT = gt_bf16;
  using Op = Frag3;
  static constexpr int LD = HD + 8;             // bf16 elements per row of a plane
  static constexpr int PLANE = TILE * LD;
  static constexpr int ELEMS = 3 * PLANE;
  static __device__ __forceinline__ Op reg(const Frag<float>& f) { return frag_split3(f); }
  static __device__ __forceinline__ Op from(const float* x) { return frag_split3(frag_from_f32<float>(x)); }
  static __device__ __forceinline__ Op row(const LT* tile, int r, int c) {
    Op o;
    o.p1 = *reinterpret_cast<const uint4*>(tile + r * LD + c);
    o.p2 = *reinterpret_cast<const uint4*>(tile + PLANE + r * LD + c);
    o.p3 = *reinterpret_cast<const uint4*>(tile + 2 * PLANE + r * LD + c);
    return o;
  }
  static __device__ __forceinline__ Op tr(const LT* tile, int col0, int n, int g) {
    Op o;
    o.p1 = frag_load_tr(tile, LD, 0, col0, n, g).v;
    o.p2 = frag_load_tr(tile + PLANE, LD, 0, col0, n, g).v;
    o.p3 = frag_load_tr(tile + 2 * PLANE, LD, 0, col0, n, g).v;
    return o;
  }
  static __device__ __forceinline__ void store(LT* tile, int r, int c, const Frag<float>& f) {
    const Op o = frag_split3(f);
    *reinterpret_cast<uint4*>(tile + r * LD + c) = o.p1;
    *reinterpret_cast<uint4*>(tile + PLANE + r * LD + c) = o.p2;
    *reinterpret_cast<uint4*>(tile + 2 * PLANE + r * LD + c) = o.p3;
  }
  static __device__ __forceinline__ f32x4 mm(const Op& a, const Op& b, f32x4 c) { return mma3(a, b, c); }
};

// =================================================================================================
// forward
// =================================================================================================
template <typename T, int HD, bool DENSE, bool SP = false>
__global__ void __launch_bounds__(ATT_THREADS, 2) k_attn_fwd(AttnArgs a) {
  using TO = TileOps<T, HD, SP>;
  using LT = typename TO::LT;
  using Op = typename TO::Op;
  constexpr int KK = Lds<HD, T>::HDP / 32;  // 32-deep steps over head_dim
  constexpr int DT = (HD + 15) / 16;        // 16-wide output dim tiles
  __shared__ __attribute__((aligned(16))) LT sKb[2][TO::ELEMS];
  __shared__ __attribute__((aligned(16))) LT sVb[2][TO::ELEMS];
  int seq, tile_, head;
  if (!block_item(a, seq, tile_, head)) return;
  const int row0 = a.desc[seq * 4 + 0], npos = a.desc[seq * 4 + 1], kv_off = a.desc[seq * 4 + 2],
            kv_len = a.desc[seq * 4 + 3];
  if (a.last_only) tile_ = npos > 0 ? (npos - 1) / BLOCK_N : 0;
  const int q_base = tile_ * BLOCK_N;
  if (q_base >= npos) return;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int qp = q_base + wid * 16 + n;
  const bool qvalid = qp < npos;
  const int64_t ld3 = 3 * a.d_model;
  const T* qkv = reinterpret_cast<const T*>(a.qkv);
  const int64_t qrow = row0 + (int64_t)qp * a.row_stride;

  Op bq[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk)
    bq[kk] = TO::reg(frag_load_head<T, HD>(qkv + qrow * ld3 + head * HD, kk * 32 + g * 8, qvalid));
  if constexpr (!SP) zero_pad_cols<T, HD>(sKb[0]);
  if constexpr (!SP) zero_pad_cols<T, HD>(sVb[0]);
  if constexpr (!SP) zero_pad_cols<T, HD>(sKb[1]);
  if constexpr (!SP) zero_pad_cols<T, HD>(sVb[1]);

  float m = -INFINITY, lsum = 0.f;
  f32x4 acc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int kv_end = kv_off + kv_len;
  const uint32_t bh = (uint32_t)(seq * a.nhead + head);
  const uint32_t hq = rng_qpart(a.seed1, bh, (uint32_t)qp);
  const uint32_t thr16 = a.drop_thr << 16;
  TilePair<T, HD> stg;
  const T* srcK = qkv + a.d_model + head * HD;
  const T* srcV = qkv + 2 * a.d_model + head * HD;
  stg.init(srcK, ld3, srcV, ld3, row0, a.row_stride, kv_off < npos ? kv_off : npos - 1);
  const int k_first = (kv_off / TILE) * TILE;
  if (HD < 32) __syncthreads();   // the zero fill above and the first store touch the same rows
  stg.load(k_first, kv_off, kv_end);
  stg.template store<TO>(sKb[0], sVb[0]);
  __syncthreads();
  int cur = 0;
  for (int k0 = k_first; k0 < kv_end; k0 += TILE, cur ^= 1) {
    const bool more = k0 + TILE < kv_end;
    if (more) stg.load(k0 + TILE, kv_off, kv_end);
    const LT* sK = sKb[cur];
    const LT* sV = sVb[cur];
    // S^T: two 16-key tiles; row m of tile t <-> key (m>>2)*8 + t*4 + (m&3)
    float s[8];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
      const int krow = (n >> 2) * 8 + t * 4 + (n & 3);
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) c = TO::mm(TO::row(sK, krow, kk * 32 + g * 8), bq[kk], c);
#pragma unroll
      for (int r = 0; r < 4; ++r) s[t * 4 + r] = c[r];
    }
    float mt = -INFINITY;
    bool full_tile = false;
    if constexpr (!DENSE) full_tile = k0 >= kv_off && k0 + TILE <= kv_end;  // block-uniform: every key valid
    // !DENSE: the running max is taken on the RAW scores (scale > 0) and the scale rides in the exponent's fma -- one multiply per score
    // less; masked_fill instantiations keep the scaled domain (their fill value lives there)
    if constexpr (!DENSE) {
      if (!full_tile) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int kp = k0 + g * 8 + i;
          s[i] = (kp >= kv_off && kp < kv_end) ? s[i] : -INFINITY;
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) mt = fmaxf(mt, s[i]);
      mt *= a.scale_log2;
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int kp = k0 + g * 8 + i;
        s[i] = (kp >= kv_off && kp < kv_end) ? s[i] * a.scale_log2 : -INFINITY;
        if (qvalid && kp < npos && dense_masked(a, seq, qp, kp, npos)) s[i] = a.mask_fill2;
        mt = fmaxf(mt, s[i]);
      }
    }
    mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m, mt);  // finite: every tile in range holds >= 1 valid key
    const float alpha = fast_exp2(m - m_new);
    m = m_new;
    float p[8], psum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (!DENSE) p[i] = fast_exp2(fmaf(s[i], a.scale_log2, -m_new));
      else p[i] = fast_exp2(s[i] - m_new);
      psum += p[i];
    }
    lsum = lsum * alpha + psum;
    if (a.drop_thr) {   // kept weights stay UNSCALED here: 1 / keep is folded into the final 1 / l (the P.V product is linear in it)
      const uint32_t kpart0 = rng_kpart(a.seed0, (uint32_t)(k0 + g * 8));
#pragma unroll
      for (int i = 0; i < 8; i += 2) {   // one hash decides a PAIR of adjacent keys (16 bits each)
        const uint32_t h = rng_mix(hq, kpart0 + (uint32_t)i * RNG_CK);
        p[i] = (h << 16) >= thr16 ? p[i] : 0.f;      // == (h & 0xffff) >= thr
        p[i + 1] = h >= thr16 ? p[i + 1] : 0.f;      // == (h >> 16) >= thr
      }
    }
    const Op bp = TO::from(p);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      f32x4 o = acc[dt];
      o[0] *= alpha; o[1] *= alpha; o[2] *= alpha; o[3] *= alpha;
      acc[dt] = TO::mm(TO::tr(sV, dt * 16, n, g), bp, o);
    }
    if (more) stg.template store<TO>(sKb[cur ^ 1], sVb[cur ^ 1]);
    __syncthreads();
  }
  lsum += __shfl_xor(lsum, 16, 64);
  lsum += __shfl_xor(lsum, 32, 64);
  if (!qvalid) return;
  const float inv_l = (a.drop_thr ? a.inv_keep : 1.0f) / lsum;
  T* ctx = reinterpret_cast<T*>(a.out);
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    f32x4 o = acc[dt];
    o[0] *= inv_l; o[1] *= inv_l; o[2] *= inv_l; o[3] *= inv_l;
    if (dt * 16 + g * 4 < HD) store4<T>(ctx + qrow * a.d_model + head * HD + dt * 16 + g * 4, o);
  }
  if (g == 0) {  // running max and log2(sum) kept apart: m may be -1.44e6 (masked_fill rows), where m + log2(l) loses l
    a.lse[(int64_t)head * a.rows + qrow] = m;
    a.lse[((int64_t)a.nhead + head) * a.rows + qrow] = log2f(lsum);
  }
}

// =================================================================================================
// backward, pass 1: delta = rowsum(dO * O) and dQ      (block = 64 queries, loop over key tiles)
// =================================================================================================
template <typename T, int HD, bool DENSE, bool SP = false>
__global__ void __launch_bounds__(ATT_THREADS, 2) k_attn_bwd_dq(AttnArgs a) {
  using TO = TileOps<T, HD, SP>;
  using LT = typename TO::LT;
  using Op = typename TO::Op;
  constexpr int KK = Lds<HD, T>::HDP / 32;
  constexpr int DT = (HD + 15) / 16;
  __shared__ __attribute__((aligned(16))) LT sKb[2][TO::ELEMS];
  __shared__ __attribute__((aligned(16))) LT sVb[2][TO::ELEMS];
  int seq, tile_, head;
  if (!block_item(a, seq, tile_, head)) return;
  const int row0 = a.desc[seq * 4 + 0], npos = a.desc[seq * 4 + 1], kv_off = a.desc[seq * 4 + 2],
            kv_len = a.desc[seq * 4 + 3];
  if (a.last_only) tile_ = npos > 0 ? (npos - 1) / BLOCK_N : 0;
  const int q_base = tile_ * BLOCK_N;
  if (q_base >= npos) return;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int qp = q_base + wid * 16 + n;
  const bool qvalid = qp < npos;
  const int64_t ld3 = 3 * a.d_model;
  const T* qkv = reinterpret_cast<const T*>(a.qkv);
  const T* dctx = reinterpret_cast<const T*>(a.d_ctx);
  const T* ctx = reinterpret_cast<const T*>(a.ctx);
  const int64_t qrow = row0 + (int64_t)qp * a.row_stride;

  Op bq[KK], bdo[KK];
  float delta = 0.f;
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    if (qvalid && kk * 32 + g * 8 < HD) {
      bq[kk] = TO::reg(frag_load(qkv + qrow * ld3 + head * HD + kk * 32 + g * 8));
      bdo[kk] = TO::reg(frag_load(dctx + qrow * a.d_model + head * HD + kk * 32 + g * 8));
      // delta partial over this lane's 8 dims
      const T* po = ctx + qrow * a.d_model + head * HD + kk * 32 + g * 8;
      const T* pd = dctx + qrow * a.d_model + head * HD + kk * 32 + g * 8;
      float4 o0 = gt_load4<T>(po), o1 = gt_load4<T>(po + 4), d0 = gt_load4<T>(pd), d1 = gt_load4<T>(pd + 4);
      delta += o0.x * d0.x + o0.y * d0.y + o0.z * d0.z + o0.w * d0.w + o1.x * d1.x + o1.y * d1.y + o1.z * d1.z +
               o1.w * d1.w;
    } else {
      bq[kk] = TO::reg(frag_zero<T>());
      bdo[kk] = TO::reg(frag_zero<T>());
    }
  }
  delta += __shfl_xor(delta, 16, 64);
  delta += __shfl_xor(delta, 32, 64);
  const float lse = qvalid ? a.lse[(int64_t)head * a.rows + qrow] : 0.f;
  const float logl = qvalid ? a.lse[((int64_t)a.nhead + head) * a.rows + qrow] : 0.f;
  const float negl = -(lse + logl);
  const uint32_t thr16 = a.drop_thr << 16;
  if (qvalid && g == 0) a.delta[(int64_t)head * a.rows + qrow] = delta;
  if constexpr (!SP) zero_pad_cols<T, HD>(sKb[0]);
  if constexpr (!SP) zero_pad_cols<T, HD>(sVb[0]);
  if constexpr (!SP) zero_pad_cols<T, HD>(sKb[1]);
  if constexpr (!SP) zero_pad_cols<T, HD>(sVb[1]);

  f32x4 acc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int kv_end = kv_off + kv_len;
  const uint32_t bh = (uint32_t)(seq * a.nhead + head);
  const uint32_t hq = rng_qpart(a.seed1, bh, (uint32_t)qp);
  TilePair<T, HD> stg;
  const T* srcK = qkv + a.d_model + head * HD;
  const T* srcV = qkv + 2 * a.d_model + head * HD;
  stg.init(srcK, ld3, srcV, ld3, row0, a.row_stride, kv_off < npos ? kv_off : npos - 1);
  const int k_first = (kv_off / TILE) * TILE;
  if (HD < 32) __syncthreads();
  stg.load(k_first, kv_off, kv_end);
  stg.template store<TO>(sKb[0], sVb[0]);
  __syncthreads();
  int cur = 0;
  for (int k0 = k_first; k0 < kv_end; k0 += TILE, cur ^= 1) {
    const bool more = k0 + TILE < kv_end;
    if (more) stg.load(k0 + TILE, kv_off, kv_end);
    const LT* sK = sKb[cur];
    const LT* sV = sVb[cur];
    float ds[8];
    const uint32_t kpart0 = rng_kpart(a.seed0, (uint32_t)(k0 + g * 8));
    bool full_tile = false;
    if constexpr (!DENSE) full_tile = k0 >= kv_off && k0 + TILE <= kv_end;  // block-uniform: every key valid
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x4 c = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
      const int krow = (n >> 2) * 8 + t * 4 + (n & 3);
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        c = TO::mm(TO::row(sK, krow, kk * 32 + g * 8), bq[kk], c);
        dp = TO::mm(TO::row(sV, krow, kk * 32 + g * 8), bdo[kk], dp);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = t * 4 + r;
        const int kp = k0 + g * 8 + i;
        float dd;
        if (a.drop_thr) {   // the pair's hash (even key of the pair), this key's 16-bit half: the compiler shares it between r, r + 1
          const uint32_t h = rng_mix(hq, kpart0 + (uint32_t)(i & ~1) * RNG_CK);
          const uint32_t hs = (i & 1) ? h : h << 16;
          dd = hs >= thr16 ? fmaf(dp[r], a.inv_keep, -delta) : -delta;
        } else {
          dd = dp[r] - delta;
        }
        if constexpr (!DENSE) {
          float e = fmaf(c[r], a.scale_log2, negl);   // negl = -(max + log2 sum): one fma per score
          if (!full_tile) e = (kp >= kv_off && kp < kv_end) ? e : -INFINITY;   // (block-uniform branch)
          ds[i] = fast_exp2(e) * dd;
        } else {
          const bool kvalid = kp >= kv_off && kp < kv_end;
          const bool filled = qvalid && kvalid && kp < npos && dense_masked(a, seq, qp, kp, npos);
          const float p = kvalid ? fast_exp2(((filled ? a.mask_fill2 : c[r] * a.scale_log2) - lse) - logl) : 0.f;
          ds[i] = filled ? 0.f : p * dd;  // masked_fill: no gradient through a filled score
        }
      }
    }
    const Op bds = TO::from(ds);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt] = TO::mm(TO::tr(sK, dt * 16, n, g), bds, acc[dt]);
    if (more) stg.template store<TO>(sKb[cur ^ 1], sVb[cur ^ 1]);
    __syncthreads();
  }
  if (!qvalid) return;
  T* dqkv = reinterpret_cast<T*>(a.out);
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    f32x4 o = acc[dt];
    o[0] *= a.scale; o[1] *= a.scale; o[2] *= a.scale; o[3] *= a.scale;
    if (dt * 16 + g * 4 < HD) store4<T>(dqkv + qrow * ld3 + head * HD + dt * 16 + g * 4, o);
  }
}

// =================================================================================================
// backward, pass 2: dK, dV                              (block = 64 keys, loop over query tiles)
// =================================================================================================
template <typename T, int HD, bool DENSE, bool SP = false>
__global__ void __launch_bounds__(ATT_THREADS, 2) k_attn_bwd_dkv(AttnArgs a) {
  using TO = TileOps<T, HD, SP>;
  using LT = typename TO::LT;
  using Op = typename TO::Op;
  constexpr int KK = Lds<HD, T>::HDP / 32;
  constexpr int DT = (HD + 15) / 16;
  __shared__ __attribute__((aligned(16))) LT sQb[2][TO::ELEMS];
  __shared__ __attribute__((aligned(16))) LT sDOb[2][TO::ELEMS];
  __shared__ __attribute__((aligned(16))) float sAux[2][3 * TILE];   // per query of the tile: running max, log2(sum), delta
  int seq, tile_, head;
  if (!block_item(a, seq, tile_, head)) return;
  const int row0 = a.desc[seq * 4 + 0], npos = a.desc[seq * 4 + 1], kv_off = a.desc[seq * 4 + 2],
            kv_len = a.desc[seq * 4 + 3];
  const int k_base = tile_ * BLOCK_N;
  if (k_base >= npos) return;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int kp = k_base + wid * 16 + n;
  const int kv_end = kv_off + kv_len;
  const bool kin = kp < npos;                       // row exists
  const bool kvalid = kp >= kv_off && kp < kv_end;  // key is not padding
  const int64_t ld3 = 3 * a.d_model;
  const T* qkv = reinterpret_cast<const T*>(a.qkv);
  const T* dctx = reinterpret_cast<const T*>(a.d_ctx);
  const int64_t krow = row0 + (int64_t)kp * a.row_stride;

  Op bk[KK], bv[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    bk[kk] = TO::reg(frag_load_head<T, HD>(qkv + krow * ld3 + a.d_model + head * HD, kk * 32 + g * 8, kvalid));
    bv[kk] = TO::reg(frag_load_head<T, HD>(qkv + krow * ld3 + 2 * a.d_model + head * HD, kk * 32 + g * 8, kvalid));
  }
  if constexpr (!SP) zero_pad_cols<T, HD>(sQb[0]);
  if constexpr (!SP) zero_pad_cols<T, HD>(sDOb[0]);
  if constexpr (!SP) zero_pad_cols<T, HD>(sQb[1]);
  if constexpr (!SP) zero_pad_cols<T, HD>(sDOb[1]);
  f32x4 dk[DT], dv[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const uint32_t bh = (uint32_t)(seq * a.nhead + head);
  const uint32_t kpart = rng_kpart(a.seed0, (uint32_t)(kp & ~1));   // dropout decisions come in pairs of adjacent keys
  const uint32_t hb = bh * RNG_CH + a.seed1;
  const uint32_t thr16 = a.drop_thr << 16;
  const uint32_t half_shift = (kp & 1) ? 0u : 16u;  // masked_fill masks are a separate instantiation: the common kernel carries none of it
  // a block whose 64 keys are all padding only writes zeros
  const bool any_valid = (k_base < kv_end) && (k_base + BLOCK_N > kv_off);
  TilePair<T, HD> stg;
  const T* srcQ = qkv + head * HD;
  const T* srcDO = dctx + head * HD;
  stg.init(srcQ, ld3, srcDO, a.d_model, row0, a.row_stride, 0);
  // threads 0..95 also carry one statistic of one query of the tile (0: running max, 1: log2(sum), 2: delta).  Branch-free, and the
  // two loads of slot 0 are COMBINED WHEN THEY ARE STORED: with `aux = -(m + lse2[..])` computed at load time the step waited for both
  // loads (and with them for the Q / dO prefetch issued in front of them) before its first MFMA.
  const int aux_r = threadIdx.x & (TILE - 1), aux_which = (threadIdx.x / TILE) % 3;
  const float* aux_src = aux_which == 0 ? a.lse : (aux_which == 1 ? a.lse + (int64_t)a.nhead * a.rows : a.delta);
  const float* aux_src2 = a.lse + (int64_t)a.nhead * a.rows;
  float aux_v = 0.f, aux_w = 0.f;
  bool aux_ok = false;
  auto load_aux = [&](int q0) {
    const int pos = q0 + aux_r;
    aux_ok = pos < npos;
    const int64_t o = (int64_t)head * a.rows + row0 + (int64_t)(aux_ok ? pos : 0) * a.row_stride;
    aux_v = aux_src[o];
    aux_w = aux_src2[o];
  };
  auto aux_value = [&]() {   // slot 0 carries -(max + log2 sum), the addend of the exponent's fma, unless the masks are dense (slot 1 is read then)
    float v = aux_v;
    if constexpr (!DENSE) v = aux_which == 0 ? -(aux_v + aux_w) : aux_v;
    return aux_ok ? v : 0.f;
  };
  const int q_begin = (a.q_last_only && npos > 0) ? ((npos - 1) / TILE) * TILE : 0;   // pooled mode: only the last position's gradient is non-zero
  if (HD < 32) __syncthreads();
  // (unconditional, also for a block whose keys are all padding: every path into the loop passes this store and with it the wait for
  // the bk / bv loads above -- behind `if (any_valid)` the wait-count pass kept them pending on the other path and put vmcnt(0) in
  // front of every step's first MFMA)
  stg.load(q_begin, 0, npos);
  load_aux(q_begin);
  stg.template store<TO, true>(sQb[0], sDOb[0]);
  if (threadIdx.x < 3 * TILE) sAux[0][threadIdx.x] = aux_value();
  __syncthreads();
  int cur = 0;
  for (int q0 = q_begin; any_valid && q0 < npos; q0 += TILE, cur ^= 1) {
    const bool more = q0 + TILE < npos;
    if (more) {
      stg.load(q0 + TILE, 0, npos);
      load_aux(q0 + TILE);
    }
    const LT* sQ = sQb[cur];
    const LT* sDO = sDOb[cur];
    const float* sLse = sAux[cur];
    const float* sLogl = sAux[cur] + TILE;
    const float* sDelta = sAux[cur] + 2 * TILE;
    float pd[8], ds[8];
    // (Sharing the pair's hash between lanes n and n ^ 1 -- each computes four of the eight and takes the rest by DPP -- was built
    // and measured: 45 us against 43 us for this form on the Code2 batch; the kernel is not bound by its VALU instruction count.)
    const uint32_t qmul0 = (uint32_t)(q0 + g * 8) * RNG_CQ;
    // this lane group's 8 queries' statistics as four 16-byte LDS reads (they were sixteen ds_read_b32)
    float a0v[8], dlv[8];
    *reinterpret_cast<float4*>(a0v) = *reinterpret_cast<const float4*>(sLse + g * 8);
    *reinterpret_cast<float4*>(a0v + 4) = *reinterpret_cast<const float4*>(sLse + g * 8 + 4);
    *reinterpret_cast<float4*>(dlv) = *reinterpret_cast<const float4*>(sDelta + g * 8);
    *reinterpret_cast<float4*>(dlv + 4) = *reinterpret_cast<const float4*>(sDelta + g * 8 + 4);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x4 c = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
      const int qr = (n >> 2) * 8 + t * 4 + (n & 3);
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        c = TO::mm(TO::row(sQ, qr, kk * 32 + g * 8), bk[kk], c);
        dp = TO::mm(TO::row(sDO, qr, kk * 32 + g * 8), bv[kk], dp);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = t * 4 + r;
        const int qi = g * 8 + i;  // query row inside the tile owned by this slot
        const int qpos = q0 + qi;
        const bool ok = kvalid && qpos < npos;
        const float a0 = a0v[i], dlt = dlv[i];
        float p;
        bool filled = false;
        if constexpr (!DENSE) {
          p = ok ? fast_exp2(fmaf(c[r], a.scale_log2, a0)) : 0.f;   // slot 0 holds -(max + log2 sum) here
        } else {
          filled = ok && dense_masked(a, seq, qpos, kp, npos);
          p = ok ? fast_exp2(((filled ? a.mask_fill2 : c[r] * a.scale_log2) - a0) - sLogl[qi]) : 0.f;
        }
        float dd, pdrop = p;   // kept weights stay unscaled in pd: 1 / keep is applied to dV once, at the end
        if (a.drop_thr) {
          const uint32_t h = rng_mix((qmul0 + (uint32_t)i * RNG_CQ) ^ hb, kpart);  // == rng_qpart(seed1, bh, qpos)
          const bool keep = (h << half_shift) >= thr16;   // this key's 16-bit half of its pair's hash
          dd = keep ? fmaf(dp[r], a.inv_keep, -dlt) : -dlt;
          pdrop = keep ? p : 0.f;
        } else {
          dd = dp[r] - dlt;
        }
        pd[i] = pdrop;
        ds[i] = filled ? 0.f : p * dd;
      }
    }
    const Op bp = TO::from(pd);
    const Op bds = TO::from(ds);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      dv[dt] = TO::mm(TO::tr(sDO, dt * 16, n, g), bp, dv[dt]);
      dk[dt] = TO::mm(TO::tr(sQ, dt * 16, n, g), bds, dk[dt]);
    }
    if (more) {
      stg.template store<TO, true>(sQb[cur ^ 1], sDOb[cur ^ 1]);
      if (threadIdx.x < 3 * TILE) sAux[cur ^ 1][threadIdx.x] = aux_value();
    }
    __syncthreads();
  }
  if (!kin) return;
  T* dqkv = reinterpret_cast<T*>(a.out);
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    f32x4 o = dk[dt];
    o[0] *= a.scale; o[1] *= a.scale; o[2] *= a.scale; o[3] *= a.scale;
    f32x4 ov = dv[dt];
    if (a.drop_thr) { ov[0] *= a.inv_keep; ov[1] *= a.inv_keep; ov[2] *= a.inv_keep; ov[3] *= a.inv_keep; }
    if (dt * 16 + g * 4 < HD) {
      store4<T>(dqkv + krow * ld3 + a.d_model + head * HD + dt * 16 + g * 4, o);
      store4<T>(dqkv + krow * ld3 + 2 * a.d_model + head * HD + dt * 16 + g * 4, ov);
    }
  }
}

int check_attn(const char* fn, int dtype, int64_t d_model, int nhead, int64_t num_seqs, int64_t max_npos,
               float dropout_p) {
  if (dtype != GT_F32 && dtype != GT_BF16) { gt_set_error("%s: bad dtype", fn); return GT_ERR_INVALID_ARG; }
  if (nhead <= 0 || d_model <= 0 || d_model % nhead != 0) { gt_set_error("%s: bad d_model/nhead", fn); return GT_ERR_INVALID_ARG; }
  int64_t hd = d_model / nhead;
  if (hd != 8 && hd != 16 && hd != 32 && hd != 64) { gt_set_error("%s: head_dim %lld unsupported (8, 16, 32 or 64)", fn, (long long)hd); return GT_ERR_UNSUPPORTED; }
  if (num_seqs < 0 || num_seqs > 65535) { gt_set_error("%s: num_seqs out of range", fn); return GT_ERR_INVALID_ARG; }
  if (max_npos < 0) { gt_set_error("%s: bad max_npos", fn); return GT_ERR_INVALID_ARG; }
  if (!(dropout_p >= 0.f && dropout_p < 1.f)) { gt_set_error("%s: dropout_p must be in [0,1)", fn); return GT_ERR_INVALID_ARG; }
  return GT_OK;
}

AttnArgs make_args(const void* qkv, const void* ctx, const void* d_ctx, float* lse, float* delta, void* out,
                   int64_t rows, int64_t d_model, int nhead, const int32_t* desc, const int32_t* work,
                   int64_t row_stride, float scale, float dropout_p, uint64_t seed, const float* dense_mask,
                   const float* key_valid, float mask_value) {
  AttnArgs a{};
  a.dense_mask = dense_mask; a.key_valid = key_valid; a.mask_fill2 = mask_value * LOG2E;
  a.qkv = qkv; a.ctx = ctx; a.d_ctx = d_ctx; a.lse = lse; a.delta = delta; a.out = out; a.desc = desc; a.work = work;
  a.rows = rows; a.d_model = d_model; a.row_stride = row_stride; a.nhead = nhead;
  a.scale = scale; a.scale_log2 = scale * LOG2E;
  a.inv_keep = 1.0f / (1.0f - dropout_p);
  double thr = (double)dropout_p * 65536.0 + 0.5;   // 16-bit decisions: two keys per hash; p is resolved to 1.5e-5
  a.drop_thr = dropout_p > 0.f ? (uint32_t)(thr > 65535.0 ? 65535.0 : (thr < 1.0 ? 1.0 : thr)) : 0u;
  a.seed0 = (uint32_t)seed; a.seed1 = (uint32_t)(seed >> 32);
  return a;
}

// fp32 rows: bf16x6 products unless gt_option_set("attn_f32_exact", 1) -- the exact v_mfma_f32_16x16x4_f32 chains stay the parity
// yardstick (graphtrans_amd/w3.py sets it together with the exact fp32 GEMMs: GT_F32_GEMM=exact; tests/test_hip_options.py)
bool attn_f32_split() { return !gt_opt(GT_OPT_ATTN_F32_EXACT); }

}  // namespace

static int attn_fwd_impl(int pooled, int dtype, const void* qkv, void* ctx, float* lse, int64_t total_rows, int64_t d_model,
                           int nhead, const int32_t* seq_desc, int64_t num_seqs, int64_t row_stride,
                           int64_t max_npos, const int32_t* work_items, int64_t num_work, const float* dense_mask,
                           const float* key_valid, float mask_value, float scale, float dropout_p, uint64_t seed,
                           gt_stream_t stream_) {
  int rc = check_attn("gt_attn_fwd", dtype, d_model, nhead, num_seqs, max_npos, dropout_p);
  if (rc) return rc;
  GT_CHECK_ARG(qkv && ctx && lse && seq_desc, "null buffer");
  if (num_seqs == 0 || max_npos == 0) return GT_OK;
  GtProfScope prof__(GT_PROF_ATTENTION, "gt_attn_fwd", stream_, {total_rows, d_model, nhead, dtype == GT_F32 ? 4 : 2, num_seqs, num_work});
  hipStream_t stream = (hipStream_t)stream_;
  if (work_items && num_work == 0) return GT_OK;
  AttnArgs a = make_args(qkv, nullptr, nullptr, lse, nullptr, ctx, total_rows, d_model, nhead, seq_desc, work_items,
                         row_stride, scale, dropout_p, seed, dense_mask, key_valid, mask_value);
  a.num_work = (int)num_work;
  a.work_per_xcd = (int)gt_cdiv(num_work, 8);
  a.last_only = pooled;
  dim3 grid = (work_items && !pooled) ? dim3((unsigned)(8 * a.work_per_xcd * nhead), 1, 1)
                                      : dim3(pooled ? 1u : (unsigned)gt_cdiv(max_npos, BLOCK_N), (unsigned)nhead, (unsigned)num_seqs);
  const int hd = (int)(d_model / nhead);
  const bool dense_launch = dense_mask != nullptr || key_valid != nullptr;
#define GT_LAUNCH(T, HD)                                                                                     \
  do {                                                                                                       \
    if (dense_launch) hipLaunchKernelGGL((k_attn_fwd<T, HD, true>), grid, dim3(ATT_THREADS), 0, stream, a);  \
    else hipLaunchKernelGGL((k_attn_fwd<T, HD, false>), grid, dim3(ATT_THREADS), 0, stream, a);              \
  } while (0)
  if (dtype == GT_F32 && !dense_launch && attn_f32_split() && (hd == 32 || hd == 64)) {   // bf16x6 products (TileOps<float, HD, true>)
    if (hd == 32) hipLaunchKernelGGL((k_attn_fwd<float, 32, false, true>), grid, dim3(ATT_THREADS), 0, stream, a);
    else hipLaunchKernelGGL((k_attn_fwd<float, 64, false, true>), grid, dim3(ATT_THREADS), 0, stream, a);
  } else if (dtype == GT_F32) {
    if (hd == 8) GT_LAUNCH(float, 8); else if (hd == 16) GT_LAUNCH(float, 16);
    else if (hd == 32) GT_LAUNCH(float, 32); else GT_LAUNCH(float, 64);
  } else {
    if (hd == 8) GT_LAUNCH(gt_bf16, 8); else if (hd == 16) GT_LAUNCH(gt_bf16, 16);
    else if (hd == 32) GT_LAUNCH(gt_bf16, 32); else GT_LAUNCH(gt_bf16, 64);
  }
#undef GT_LAUNCH
  GT_CHECK_LAUNCH();
  return GT_OK;
}

extern "C" int gt_attn_fwd(int dtype, const void* qkv, void* ctx, float* lse, int64_t total_rows, int64_t d_model,
                           int nhead, const int32_t* seq_desc, int64_t num_seqs, int64_t row_stride,
                           int64_t max_npos, const int32_t* work_items, int64_t num_work, const float* dense_mask,
                           const float* key_valid, float mask_value, float scale, float dropout_p, uint64_t seed,
                           gt_stream_t stream_) {
  return attn_fwd_impl(0, dtype, qkv, ctx, lse, total_rows, d_model, nhead, seq_desc, num_seqs, row_stride, max_npos, work_items, num_work,
                       dense_mask, key_valid, mask_value, scale, dropout_p, seed, stream_);
}
// the 64-row tile that holds the LAST position of every sequence only (ctx / lse rows outside those tiles are not written)
extern "C" int gt_attn_fwd_last(int dtype, const void* qkv, void* ctx, float* lse, int64_t total_rows, int64_t d_model,
                                int nhead, const int32_t* seq_desc, int64_t num_seqs, int64_t row_stride, int64_t max_npos,
                                float scale, float dropout_p, uint64_t seed, gt_stream_t stream_) {
  return attn_fwd_impl(1, dtype, qkv, ctx, lse, total_rows, d_model, nhead, seq_desc, num_seqs, row_stride, max_npos, nullptr, 0,
                       nullptr, nullptr, 0.f, scale, dropout_p, seed, stream_);
}

static int attn_bwd_impl(int pooled, int dtype, const void* qkv, const void* ctx, const void* d_ctx, const float* lse,
                           float* delta, void* d_qkv, int64_t total_rows, int64_t d_model, int nhead,
                           const int32_t* seq_desc, int64_t num_seqs, int64_t row_stride, int64_t max_npos,
                           const int32_t* work_items, int64_t num_work, const float* dense_mask,
                           const float* key_valid, float mask_value, float scale, float dropout_p, uint64_t seed,
                           gt_stream_t stream_) {
  int rc = check_attn("gt_attn_bwd", dtype, d_model, nhead, num_seqs, max_npos, dropout_p);
  if (rc) return rc;
  GT_CHECK_ARG(qkv && ctx && d_ctx && lse && delta && d_qkv && seq_desc, "null buffer");
  if (num_seqs == 0 || max_npos == 0) return GT_OK;
  GtProfScope prof__(GT_PROF_ATTENTION, "gt_attn_bwd", stream_, {total_rows, d_model, nhead, dtype == GT_F32 ? 4 : 2, num_seqs, num_work});
  hipStream_t stream = (hipStream_t)stream_;
  if (work_items && num_work == 0) return GT_OK;
  AttnArgs a = make_args(qkv, ctx, d_ctx, const_cast<float*>(lse), delta, d_qkv, total_rows, d_model, nhead, seq_desc,
                         work_items, row_stride, scale, dropout_p, seed, dense_mask, key_valid, mask_value);
  a.num_work = (int)num_work;
  a.work_per_xcd = (int)gt_cdiv(num_work, 8);
  dim3 grid = work_items ? dim3((unsigned)(8 * a.work_per_xcd * nhead), 1, 1)
                         : dim3((unsigned)gt_cdiv(max_npos, BLOCK_N), (unsigned)nhead, (unsigned)num_seqs);
  // pooled: dQ for the last tile of every sequence (a block per sequence and head), dK / dV for every key tile with the query loop cut
  // to the last step; the dQ rows outside the last tiles are zero (the caller zero-fills d_qkv)
  AttnArgs aq = a;
  dim3 grid_q = grid;
  if (pooled) {
    aq.last_only = 1;
    grid_q = dim3(1u, (unsigned)nhead, (unsigned)num_seqs);
    a.q_last_only = 1;
  }
  const int hd = (int)(d_model / nhead);
  const bool dense_launch = dense_mask != nullptr || key_valid != nullptr;
#define GT_LAUNCH(T, HD)                                                                        \
  do {                                                                                          \
    if (dense_launch) {                                                                         \
      hipLaunchKernelGGL((k_attn_bwd_dq<T, HD, true>), grid_q, dim3(ATT_THREADS), 0, stream, aq);  \
      hipLaunchKernelGGL((k_attn_bwd_dkv<T, HD, true>), grid, dim3(ATT_THREADS), 0, stream, a); \
    } else {                                                                                    \
      hipLaunchKernelGGL((k_attn_bwd_dq<T, HD, false>), grid_q, dim3(ATT_THREADS), 0, stream, aq); \
      hipLaunchKernelGGL((k_attn_bwd_dkv<T, HD, false>), grid, dim3(ATT_THREADS), 0, stream, a);\
    }                                                                                           \
  } while (0)
  if (dtype == GT_F32 && !dense_launch && attn_f32_split() && (hd == 32 || hd == 64)) {   // bf16x6 products (TileOps<float, HD, true>)
    if (hd == 32) {
      hipLaunchKernelGGL((k_attn_bwd_dq<float, 32, false, true>), grid_q, dim3(ATT_THREADS), 0, stream, aq);
      hipLaunchKernelGGL((k_attn_bwd_dkv<float, 32, false, true>), grid, dim3(ATT_THREADS), 0, stream, a);
    } else {
      hipLaunchKernelGGL((k_attn_bwd_dq<float, 64, false, true>), grid_q, dim3(ATT_THREADS), 0, stream, aq);
      hipLaunchKernelGGL((k_attn_bwd_dkv<float, 64, false, true>), grid, dim3(ATT_THREADS), 0, stream, a);
    }
  } else if (dtype == GT_F32) {
    if (hd == 8) GT_LAUNCH(float, 8); else if (hd == 16) GT_LAUNCH(float, 16);
    else if (hd == 32) GT_LAUNCH(float, 32); else GT_LAUNCH(float, 64);
  } else {
    if (hd == 8) GT_LAUNCH(gt_bf16, 8); else if (hd == 16) GT_LAUNCH(gt_bf16, 16);
    else if (hd == 32) GT_LAUNCH(gt_bf16, 32); else GT_LAUNCH(gt_bf16, 64);
  }
#undef GT_LAUNCH
  GT_CHECK_LAUNCH();
  return GT_OK;
}

extern "C" int gt_attn_bwd(int dtype, const void* qkv, const void* ctx, const void* d_ctx, const float* lse,
                           float* delta, void* d_qkv, int64_t total_rows, int64_t d_model, int nhead,
                           const int32_t* seq_desc, int64_t num_seqs, int64_t row_stride, int64_t max_npos,
                           const int32_t* work_items, int64_t num_work, const float* dense_mask,
                           const float* key_valid, float mask_value, float scale, float dropout_p, uint64_t seed,
                           gt_stream_t stream_) {
  return attn_bwd_impl(0, dtype, qkv, ctx, d_ctx, lse, delta, d_qkv, total_rows, d_model, nhead, seq_desc, num_seqs, row_stride, max_npos,
                       work_items, num_work, dense_mask, key_valid, mask_value, scale, dropout_p, seed, stream_);
}
// backward of gt_attn_fwd_last: d_ctx is non-zero only in the last position of every sequence; d_qkv must be ZERO-FILLED by the caller
// (dQ is written for the last 64-row tiles only); work_items: the full attention work list (key tiles)
extern "C" int gt_attn_bwd_last(int dtype, const void* qkv, const void* ctx, const void* d_ctx, const float* lse, float* delta,
                                void* d_qkv, int64_t total_rows, int64_t d_model, int nhead, const int32_t* seq_desc, int64_t num_seqs,
                                int64_t row_stride, int64_t max_npos, const int32_t* work_items, int64_t num_work, float scale,
                                float dropout_p, uint64_t seed, gt_stream_t stream_) {
  return attn_bwd_impl(1, dtype, qkv, ctx, d_ctx, lse, delta, d_qkv, total_rows, d_model, nhead, seq_desc, num_seqs, row_stride, max_npos,
                       work_items, num_work, nullptr, nullptr, 0.f, scale, dropout_p, seed, stream_);
}
