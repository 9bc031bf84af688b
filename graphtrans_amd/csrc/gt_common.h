// gt_common.h — shared helpers for the gfx950 kernels (device code + C-ABI plumbing).
#pragma once
#include <hip/hip_runtime.h>
#ifdef GT_DEBUG_SKIP   // tools/skip_probe_build.sh: a what-if build, every launch behind GT_SKIP (never the shipped library)
#include <cstdlib>
#include <cstring>
static inline bool gt_dbg_skip(const char* kernel) {
  static const char* e = getenv("GT_SKIP");
  if (!e || !*e) return false;
  const char* p = e;
  while (*p) {
    const char* q = strchr(p, ',');
    const size_t n = q ? (size_t)(q - p) : strlen(p);
    char buf[128];
    if (n > 0 && n < sizeof(buf)) {
      memcpy(buf, p, n);
      buf[n] = 0;
      if (strstr(kernel, buf)) return true;
    }
    if (!q) break;
    p = q + 1;
  }
  return false;
}
// GT_EMPTY=1: every launch becomes ONE empty 64-thread block on the same stream -- the step's exact stream / event topology with no
// work in it: the floor of the launch chain (tools/launch_floor.sh)
static __global__ void gt_dbg_empty_kernel() {}
static inline bool gt_dbg_empty() {
  static const bool on = [] { const char* e = getenv("GT_EMPTY"); return e && *e && *e != '0'; }();
  return on;
}
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(k, g, b, l, s, ...)                                                              \
  do {                                                                                                      \
    if (gt_dbg_empty()) hipLaunchKernelGGLInternal(gt_dbg_empty_kernel, dim3(1), dim3(64), 0, (s));        \
    else if (!gt_dbg_skip(#k)) hipLaunchKernelGGLInternal((k), (g), (b), (l), (s), __VA_ARGS__);           \
  } while (0)
#endif
#include <stdint.h>
#include <stdio.h>

#include <initializer_list>

#include "../../include/graphtrans_hip.h"

#define GT_WAVE 64

// thread-local last-error string (C-ABI contract: never throws, never exits)
void gt_set_error(const char* fmt, ...);

#define GT_CHECK_ARG(cond, msg)                     \
  do {                                              \
    if (!(cond)) {                                  \
      gt_set_error("%s: %s", __func__, msg);        \
      return GT_ERR_INVALID_ARG;                    \
    }                                               \
  } while (0)

#define GT_CHECK_LAUNCH()                                                        \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      gt_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__));   \
      return GT_ERR_LAUNCH;                                                      \
    }                                                                            \
  } while (0)

// opt-in launch profiler (common.hip): categories for gt_profile_enable's mask
enum { GT_PROF_AGGREGATE = 1, GT_PROF_ATTENTION = 2, GT_PROF_LINEAR = 4, GT_PROF_NORM = 8, GT_PROF_SEGMENT = 16,
       // the GEMM KERNELS one by one, on the stream each is launched on (the overlap stream included): unlike GT_PROF_LINEAR
       // (whole entry points, which keeps the weight-gradient GEMMs on the caller's stream while it brackets them) this
       // category times the schedule the un-profiled step runs -- bench.py's roofline uses it
       GT_PROF_GEMM_KERNEL = 32 };
unsigned gt_prof_mask();
// named runtime options (gt_option_set / gt_option_get, csrc/common.hip): alternative implementations that stay in the library as
// TESTED yardsticks (tests/test_hip_options.py runs each non-default value against the oracle).  Process-wide, read at every call.
enum { GT_OPT_ATTN_F32_EXACT = 0, GT_OPT_BNSTATS_ROWS_KERNEL = 1, GT_OPT_LIN_RING = 2, GT_OPT_COUNT };
int gt_opt(int id);
int64_t gt_prof_begin(const char* name, hipStream_t stream, const int64_t* dims, int ndims);
void gt_prof_end(int64_t id, hipStream_t stream);
struct GtProfScope {
  int64_t id;
  hipStream_t stream;
  GtProfScope(unsigned cat, const char* name, gt_stream_t st, std::initializer_list<int64_t> dims)
      : id(-1), stream((hipStream_t)st) {
    if (gt_prof_mask() & cat) id = gt_prof_begin(name, stream, dims.begin(), (int)dims.size());
  }
  ~GtProfScope() { gt_prof_end(id, stream); }
};

static inline int64_t gt_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- bf16 <-> f32 (storage type is a raw 16-bit pattern) -------------------------------------
typedef uint16_t gt_bf16;

__device__ __forceinline__ float gt_bf16_to_f32(gt_bf16 v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ gt_bf16 gt_f32_to_bf16(float f) {  // round-to-nearest-even, NaN preserved
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (gt_bf16)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (gt_bf16)(u >> 16);
}

// two fp32 -> packed bf16x2 (lo in bits 0..15) in ONE instruction: v_cvt_pk_bf16_f32 (gfx950, RNE).
// The software sequence above costs ~7 VALU per element and made the GEMM staging VALU-bound.
// Through the compiler's own vector conversion, NOT inline asm: an asm statement is opaque to the scheduler and the hazard
// recognizer, and with it the LayerNorm backward wrote a wrong row once in a few hundred passes beside the overlap stream's
// GEMMs -- with this form the same schedule ran 65 000 passes bitwise identical.
typedef __bf16 gt_v2bf __attribute__((ext_vector_type(2)));
typedef float gt_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t gt_pack_bf16(float lo, float hi) {
  const gt_v2f f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, gt_v2bf));
}

// ---- activations fused into GEMM epilogues / gradient loads ---------------------------------------------------------
// GELU in torch's default (erf) form and its derivative (modules/transformer_encoder.py:17, masked_transformer_encoder.py:68)
__device__ __forceinline__ void gt_gelu(float z, float& y, float& dydz) {
  const float cdf = 0.5f * (1.f + erff(z * 0.70710678118654752f));
  y = z * cdf;
  dydz = cdf + z * 0.3989422804014327f * __expf(-0.5f * z * z);
}
// Gradient gate of a fused activation, applied while dY is loaded: `m` is the saved tensor's element.
//   inv_keep > 0: m = forward output of relu(+dropout): dZ = dY * 1[m > 0] * inv_keep
//   inv_keep == 0 (sentinel): m = saved multiplier d act / dz * dropout scale (GELU): dZ = dY * m
__device__ __forceinline__ float gt_gate(float dy, float m, float inv_keep) {
  return inv_keep == 0.f ? dy * m : (m > 0.f ? dy * inv_keep : 0.f);
}

struct gt_f4 {
  float x, y, z, w;
};

// 4-element vector load/store of a row chunk, converting storage <-> fp32.
template <typename T>
__device__ __forceinline__ float4 gt_load4(const T* p);
template <>
__device__ __forceinline__ float4 gt_load4<float>(const float* p) {
  return *reinterpret_cast<const float4*>(p);
}
template <>
__device__ __forceinline__ float4 gt_load4<gt_bf16>(const gt_bf16* p) {
  uint2 u = *reinterpret_cast<const uint2*>(p);
  float4 r;
  r.x = __uint_as_float(u.x << 16);
  r.y = __uint_as_float(u.x & 0xffff0000u);
  r.z = __uint_as_float(u.y << 16);
  r.w = __uint_as_float(u.y & 0xffff0000u);
  return r;
}
template <typename T>
__device__ __forceinline__ void gt_store4(T* p, float4 v);
template <>
__device__ __forceinline__ void gt_store4<float>(float* p, float4 v) {
  *reinterpret_cast<float4*>(p) = v;
}
template <>
__device__ __forceinline__ void gt_store4<gt_bf16>(gt_bf16* p, float4 v) {
  uint2 u;
  u.x = gt_pack_bf16(v.x, v.y);
  u.y = gt_pack_bf16(v.z, v.w);
  *reinterpret_cast<uint2*>(p) = u;
}

__device__ __forceinline__ float4 gt_zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 gt_add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 gt_scale4(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 gt_fma4(float4 a, float s, float4 c) {
  return make_float4(fmaf(a.x, s, c.x), fmaf(a.y, s, c.y), fmaf(a.z, s, c.z), fmaf(a.w, s, c.w));
}
__device__ __forceinline__ float4 gt_relu4(float4 a) {
  return make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f));
}

