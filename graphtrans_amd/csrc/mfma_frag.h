// mfma_frag.h — 8-slot MFMA operand fragments shared by the attention and linear kernels.
//
// One "32-deep step" contracts 32 k-slots; lane (n = lane & 15, g = lane >> 4) holds the 8 slots
// g*8 .. g*8+7 of row/column n.  bf16: one v_mfma_f32_16x16x32_bf16; fp32: eight
// v_mfma_f32_16x16x4_f32 (slot i of every lane group in step i) = exact fp32 fma chains.
// Accumulator layout (both): c[r] = C[row = g*4 + r][col = n].
#pragma once
#include "gt_common.h"

namespace gtf {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;


// ---- 8-slot operand fragments ------------------------------------------------------------------
template <typename T>
struct Frag;
template <>
struct Frag<gt_bf16> {
  uint4 v;  // 8 x bf16
};
template <>
struct Frag<float> {
  float v[8];
};

template <typename T>
__device__ __forceinline__ Frag<T> frag_zero();
template <>
__device__ __forceinline__ Frag<gt_bf16> frag_zero<gt_bf16>() {
  Frag<gt_bf16> f;
  f.v = make_uint4(0, 0, 0, 0);
  return f;
}
template <>
__device__ __forceinline__ Frag<float> frag_zero<float>() {
  Frag<float> f;
#pragma unroll
  for (int i = 0; i < 8; ++i) f.v[i] = 0.f;
  return f;
}

// 8 contiguous elements (row operand), global or LDS
__device__ __forceinline__ Frag<gt_bf16> frag_load(const gt_bf16* p) {
  Frag<gt_bf16> f;
  f.v = *reinterpret_cast<const uint4*>(p);
  return f;
}
__device__ __forceinline__ Frag<float> frag_load(const float* p) {
  Frag<float> f;
  float4 a = *reinterpret_cast<const float4*>(p);
  float4 b = *reinterpret_cast<const float4*>(p + 4);
  f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
  f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
  return f;
}

__device__ __forceinline__ void frag_store_lds(gt_bf16* p, const Frag<gt_bf16>& f) {
  *reinterpret_cast<uint4*>(p) = f.v;
}
__device__ __forceinline__ void frag_store_lds(float* p, const Frag<float>& f) {
  *reinterpret_cast<float4*>(p) = make_float4(f.v[0], f.v[1], f.v[2], f.v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(f.v[4], f.v[5], f.v[6], f.v[7]);
}

// transposed operand: slot i = tile[(k0 + g*8 + i) * ld + col0 + n]   (n = lane&15, g = lane>>4)
__device__ __forceinline__ Frag<gt_bf16> frag_load_tr(const gt_bf16* tile, int ld, int k0, int col0, int n, int g) {
  // ds_read_b64_tr_b16: each lane supplies the address of 4 contiguous bf16; within a 16-lane
  // group the 16x4 block is returned transposed: lane n gets rows 0..3 of column n.
  const gt_bf16* p0 = tile + (k0 + g * 8 + (n >> 2)) * ld + col0 + (n & 3) * 4;
  const gt_bf16* p1 = p0 + 4 * ld;
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
  Frag<gt_bf16> f;
  f.v.x = (uint32_t)(uint16_t)a[0] | ((uint32_t)(uint16_t)a[1] << 16);
  f.v.y = (uint32_t)(uint16_t)a[2] | ((uint32_t)(uint16_t)a[3] << 16);
  f.v.z = (uint32_t)(uint16_t)b[0] | ((uint32_t)(uint16_t)b[1] << 16);
  f.v.w = (uint32_t)(uint16_t)b[2] | ((uint32_t)(uint16_t)b[3] << 16);
  return f;
}
__device__ __forceinline__ Frag<float> frag_load_tr(const float* tile, int ld, int k0, int col0, int n, int g) {
  Frag<float> f;
  const float* p = tile + (k0 + g * 8) * ld + col0 + n;
#pragma unroll
  for (int i = 0; i < 8; ++i) f.v[i] = p[i * ld];
  return f;
}

// fp32 values -> operand fragment
template <typename T>
__device__ __forceinline__ Frag<T> frag_from_f32(const float* x);
template <>
__device__ __forceinline__ Frag<gt_bf16> frag_from_f32<gt_bf16>(const float* x) {
  Frag<gt_bf16> f;
  f.v.x = gt_pack_bf16(x[0], x[1]);
  f.v.y = gt_pack_bf16(x[2], x[3]);
  f.v.z = gt_pack_bf16(x[4], x[5]);
  f.v.w = gt_pack_bf16(x[6], x[7]);
  return f;
}
template <>
__device__ __forceinline__ Frag<float> frag_from_f32<float>(const float* x) {
  Frag<float> f;
#pragma unroll
  for (int i = 0; i < 8; ++i) f.v[i] = x[i];
  return f;
}

// C[16x16] += A[16 x 32slots] B[32slots x 16]
__device__ __forceinline__ f32x4 mma(const Frag<gt_bf16>& a, const Frag<gt_bf16>& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a.v), __builtin_bit_cast(bf16x8_t, b.v),
                                                 c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma(const Frag<float>& a, const Frag<float>& b, f32x4 c) {
#pragma unroll
  for (int i = 0; i < 8; ++i) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[i], b.v[i], c, 0, 0, 0);
  return c;
}

// ---- fp32-accurate product on the bf16 pipe ("bf16x6", as linear3x.h): both 8-slot fp32 fragments split EXACTLY into three bf16
// planes (8 + 8 + 8 significand bits), the six products of order <= 2^-16 kept, small terms first.  6 x 16 cycles instead of the
// 8 dependent v_mfma_f32_16x16x4_f32 (8 x 32 cycles) of the exact form, at ~14 VALU per split pair.
struct Frag3 {
  uint4 p1, p2, p3;
};
__device__ __forceinline__ Frag3 frag_split3(const Frag<float>& f) {
  Frag3 r;
  uint32_t u1[4], u2[4], u3[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float lo = f.v[2 * q], hi = f.v[2 * q + 1];
    u1[q] = gt_pack_bf16(lo, hi);
    const float rl = lo - __uint_as_float(u1[q] << 16), rh = hi - __uint_as_float(u1[q] & 0xffff0000u);   // exact
    u2[q] = gt_pack_bf16(rl, rh);
    u3[q] = gt_pack_bf16(rl - __uint_as_float(u2[q] << 16), rh - __uint_as_float(u2[q] & 0xffff0000u));
  }
  r.p1 = make_uint4(u1[0], u1[1], u1[2], u1[3]);
  r.p2 = make_uint4(u2[0], u2[1], u2[2], u2[3]);
  r.p3 = make_uint4(u3[0], u3[1], u3[2], u3[3]);
  return r;
}
__device__ __forceinline__ f32x4 mma3(const Frag3& a, const Frag3& b, f32x4 c) {
#define GT_MMA3(X, Y) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a.X), __builtin_bit_cast(bf16x8_t, b.Y), c, 0, 0, 0)
  GT_MMA3(p3, p1); GT_MMA3(p1, p3); GT_MMA3(p2, p2); GT_MMA3(p2, p1); GT_MMA3(p1, p2); GT_MMA3(p1, p1);
#undef GT_MMA3
  return c;
}

template <typename T>
__device__ __forceinline__ void store4(T* p, f32x4 v) {
  gt_store4<T>(p, make_float4(v[0], v[1], v[2], v[3]));
}


}  // namespace gtf
