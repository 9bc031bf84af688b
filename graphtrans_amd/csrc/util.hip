// util.hip — small data-movement kernels of the fused model path (engine.py): column-slab copies
// for the JK concatenation and row gathers / scatters for the sequence pooling.
//
// Reference (paths under /root/reference): torch.cat([h_list[0], h_list[-1]], dim=1) for JK = "cat"
// (modules/gnn_module.py:104-105) and transformer_out[-1] for the "cls" / "last" pooling
// (models/gnn_transformer.py:113-114).  Pure HBM copies, 16-byte vectorised.
#include "gt_common.h"

namespace {

constexpr int UT = 256;

// dst[r][0..width) = src[r][0..width), byte pitches; width, pitches and bases multiples of 16
__global__ void __launch_bounds__(UT) k_copy2d(char* __restrict__ dst, int64_t dst_pitch, const char* __restrict__ src,
                                               int64_t src_pitch, int64_t width16, int64_t rows) {
  const int64_t total = rows * width16;
  for (int64_t i = (int64_t)blockIdx.x * UT + threadIdx.x; i < total; i += (int64_t)gridDim.x * UT) {
    const int64_t r = i / width16, c = i % width16;
    *reinterpret_cast<uint4*>(dst + r * dst_pitch + c * 16) = *reinterpret_cast<const uint4*>(src + r * src_pitch + c * 16);
  }
}

template <typename T>
__global__ void __launch_bounds__(UT) k_rows_gather(const T* __restrict__ x, const int64_t* __restrict__ idx, int64_t n,
                                                    int64_t dim, float* __restrict__ out) {
  const int64_t C = dim / 4, total = n * C;
  for (int64_t i = (int64_t)blockIdx.x * UT + threadIdx.x; i < total; i += (int64_t)gridDim.x * UT) {
    const int64_t r = i / C, c = (i % C) * 4;
    *reinterpret_cast<float4*>(out + r * dim + c) = gt_load4<T>(x + (idx ? idx[r] : r) * dim + c);
  }
}

template <typename T>
__global__ void __launch_bounds__(UT) k_rows_scatter(const float* __restrict__ g, const int64_t* __restrict__ idx, int64_t n,
                                                     int64_t dim, T* __restrict__ out) {
  const int64_t C = dim / 4, total = n * C;
  for (int64_t i = (int64_t)blockIdx.x * UT + threadIdx.x; i < total; i += (int64_t)gridDim.x * UT) {
    const int64_t r = i / C, c = (i % C) * 4;
    gt_store4<T>(out + (idx ? idx[r] : r) * dim + c, *reinterpret_cast<const float4*>(g + r * dim + c));
  }
}

// out = a + b (+ c), fp32, n4 float4 chunks
__global__ void __launch_bounds__(UT) k_add3(const float* __restrict__ a, const float* __restrict__ b,
                                             const float* __restrict__ c, int64_t n4, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * UT + threadIdx.x; i < n4; i += (int64_t)gridDim.x * UT) {
    float4 v = gt_add4(*reinterpret_cast<const float4*>(a + i * 4), *reinterpret_cast<const float4*>(b + i * 4));
    if (c) v = gt_add4(v, *reinterpret_cast<const float4*>(c + i * 4));
    *reinterpret_cast<float4*>(out + i * 4) = v;
  }
}

// rows moved in their storage type (the pooled rows of the last encoder layer): mode 0 out[i] = x[idx[i]] (gather), 1 out[idx[i]] = x[i]
// (scatter into a zero-filled matrix), 2 out[idx[i]] += x[i] (fp32 add, idx must not repeat)
template <typename T>
__global__ void __launch_bounds__(UT) k_rows_move(const T* __restrict__ x, const int64_t* __restrict__ idx, int64_t n, int64_t dim,
                                                  T* __restrict__ out, int mode) {
  const int64_t C = dim / 4, total = n * C;
  for (int64_t i = (int64_t)blockIdx.x * UT + threadIdx.x; i < total; i += (int64_t)gridDim.x * UT) {
    const int64_t r = i / C, c = (i % C) * 4;
    if (mode == 0) {
      gt_store4<T>(out + r * dim + c, gt_load4<T>(x + idx[r] * dim + c));
    } else {
      T* o = out + idx[r] * dim + c;
      float4 v = gt_load4<T>(x + r * dim + c);
      if (mode == 2) v = gt_add4(v, gt_load4<T>(o));
      gt_store4<T>(o, v);
    }
  }
}

// dst[i] = map[i] < 0 ? 0 : src[map[i]]: re-stacked weight images (and, through the inverse map, their gradients)
__global__ void __launch_bounds__(UT) k_gather_f32(float* __restrict__ dst, const float* __restrict__ src, const int32_t* __restrict__ map,
                                                   int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * UT + threadIdx.x; i < n; i += (int64_t)gridDim.x * UT) {
    const int32_t j = map[i];
    dst[i] = j < 0 ? 0.f : src[j];
  }
}

// dst[r][c] = c < src_cols ? src[r][c] : 0, contiguous rows of dst_cols / src_cols elements (any counts)
template <typename U>
__global__ void __launch_bounds__(UT) k_repitch(U* __restrict__ dst, int64_t dst_cols, const U* __restrict__ src,
                                                int64_t src_cols, int64_t rows) {
  const int64_t total = rows * dst_cols;
  for (int64_t i = (int64_t)blockIdx.x * UT + threadIdx.x; i < total; i += (int64_t)gridDim.x * UT) {
    const int64_t r = i / dst_cols, c = i % dst_cols;
    dst[i] = c < src_cols ? src[r * src_cols + c] : (U)0;
  }
}

int grid_for(int64_t items) {
  int64_t g = gt_cdiv(items, UT);
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" int gt_copy2d(void* dst, int64_t dst_pitch_bytes, const void* src, int64_t src_pitch_bytes, int64_t width_bytes,
                         int64_t rows, gt_stream_t stream_) {
  GT_CHECK_ARG(rows >= 0 && width_bytes >= 0, "bad sizes");
  if (rows == 0 || width_bytes == 0) return GT_OK;
  GT_CHECK_ARG(dst && src, "null buffer");
  GT_CHECK_ARG(((uintptr_t)dst | (uintptr_t)src | (uintptr_t)dst_pitch_bytes | (uintptr_t)src_pitch_bytes |
                (uintptr_t)width_bytes) % 16 == 0, "pointers, pitches and width must be multiples of 16 bytes");
  hipLaunchKernelGGL(k_copy2d, dim3(grid_for(rows * (width_bytes / 16))), dim3(UT), 0, (hipStream_t)stream_, (char*)dst,
                     dst_pitch_bytes, (const char*)src, src_pitch_bytes, width_bytes / 16, rows);
  GT_CHECK_LAUNCH();
  return GT_OK;
}

static int rows_move(const char* fn, int mode, int dtype, const void* x, const int64_t* idx, int64_t n, int64_t total_rows, int64_t dim,
                     void* out, gt_stream_t stream_) {
  if (dtype != GT_F32 && dtype != GT_BF16) { gt_set_error("%s: bad dtype", fn); return GT_ERR_INVALID_ARG; }
  if (n < 0 || dim <= 0 || dim % 4) { gt_set_error("%s: dim must be a positive multiple of 4", fn); return GT_ERR_INVALID_ARG; }
  hipStream_t stream = (hipStream_t)stream_;
  if (mode == 1) {
    if (total_rows == 0) return GT_OK;
    if (!out) { gt_set_error("%s: null buffer", fn); return GT_ERR_INVALID_ARG; }
    (void)hipMemsetAsync(out, 0, (size_t)total_rows * dim * (dtype == GT_F32 ? 4 : 2), stream);
  }
  if (n == 0) return GT_OK;
  if (!(x && idx && out)) { gt_set_error("%s: null buffer", fn); return GT_ERR_INVALID_ARG; }
  if (dtype == GT_F32)
    hipLaunchKernelGGL(k_rows_move<float>, dim3(grid_for(n * dim / 4)), dim3(UT), 0, stream, (const float*)x, idx, n, dim, (float*)out, mode);
  else
    hipLaunchKernelGGL(k_rows_move<gt_bf16>, dim3(grid_for(n * dim / 4)), dim3(UT), 0, stream, (const gt_bf16*)x, idx, n, dim, (gt_bf16*)out,
                       mode);
  GT_CHECK_LAUNCH();
  return GT_OK;
}
extern "C" int gt_rows_take(int dtype, const void* x, const int64_t* idx, int64_t n, int64_t dim, void* out, gt_stream_t stream) {
  return rows_move("gt_rows_take", 0, dtype, x, idx, n, 0, dim, out, stream);
}
extern "C" int gt_rows_put(int dtype, const void* x, const int64_t* idx, int64_t n, int64_t total_rows, int64_t dim, void* out,
                           gt_stream_t stream) {
  return rows_move("gt_rows_put", 1, dtype, x, idx, n, total_rows, dim, out, stream);
}
extern "C" int gt_rows_add(int dtype, const void* x, const int64_t* idx, int64_t n, int64_t dim, void* out, gt_stream_t stream) {
  return rows_move("gt_rows_add", 2, dtype, x, idx, n, 0, dim, out, stream);
}

extern "C" int gt_gather_f32(float* dst, const float* src, const int32_t* map, int64_t n, gt_stream_t stream_) {
  GT_CHECK_ARG(n >= 0, "bad size");
  if (n == 0) return GT_OK;
  GT_CHECK_ARG(dst && src && map, "null buffer");
  hipLaunchKernelGGL(k_gather_f32, dim3(grid_for(n)), dim3(UT), 0, (hipStream_t)stream_, dst, src, map, n);
  GT_CHECK_LAUNCH();
  return GT_OK;
}

extern "C" int gt_repitch(void* dst, int64_t dst_cols, const void* src, int64_t src_cols, int64_t rows, int elt_bytes,
                          gt_stream_t stream_) {
  GT_CHECK_ARG(rows >= 0 && dst_cols >= 0 && src_cols >= 0, "bad sizes");
  GT_CHECK_ARG(elt_bytes == 2 || elt_bytes == 4, "elt_bytes must be 2 or 4");
  if (rows == 0 || dst_cols == 0) return GT_OK;
  GT_CHECK_ARG(dst && (src || src_cols == 0), "null buffer");
  if (elt_bytes == 4)
    hipLaunchKernelGGL(k_repitch<uint32_t>, dim3(grid_for(rows * dst_cols)), dim3(UT), 0, (hipStream_t)stream_, (uint32_t*)dst,
                       dst_cols, (const uint32_t*)src, src_cols, rows);
  else
    hipLaunchKernelGGL(k_repitch<uint16_t>, dim3(grid_for(rows * dst_cols)), dim3(UT), 0, (hipStream_t)stream_, (uint16_t*)dst,
                       dst_cols, (const uint16_t*)src, src_cols, rows);
  GT_CHECK_LAUNCH();
  return GT_OK;
}

extern "C" int gt_rows_gather(int dtype, const void* x, const int64_t* idx, int64_t n, int64_t dim, float* out,
                              gt_stream_t stream_) {
  GT_CHECK_ARG(dtype == GT_F32 || dtype == GT_BF16, "bad dtype");
  GT_CHECK_ARG(n >= 0 && dim > 0 && dim % 4 == 0, "dim must be a positive multiple of 4");
  if (n == 0) return GT_OK;
  GT_CHECK_ARG(x && out, "null buffer");   // idx == NULL: rows 0 .. n-1 (a storage-type -> fp32 conversion of n rows)
  if (dtype == GT_F32)
    hipLaunchKernelGGL(k_rows_gather<float>, dim3(grid_for(n * dim / 4)), dim3(UT), 0, (hipStream_t)stream_, (const float*)x, idx,
                       n, dim, out);
  else
    hipLaunchKernelGGL(k_rows_gather<gt_bf16>, dim3(grid_for(n * dim / 4)), dim3(UT), 0, (hipStream_t)stream_,
                       (const gt_bf16*)x, idx, n, dim, out);
  GT_CHECK_LAUNCH();
  return GT_OK;
}

extern "C" int gt_rows_scatter(int dtype, const float* grad, const int64_t* idx, int64_t n, int64_t total_rows, int64_t dim,
                               void* out, gt_stream_t stream_) {
  GT_CHECK_ARG(dtype == GT_F32 || dtype == GT_BF16, "bad dtype");
  GT_CHECK_ARG(n >= 0 && total_rows >= 0 && dim > 0 && dim % 4 == 0, "dim must be a positive multiple of 4");
  if (total_rows == 0) return GT_OK;
  GT_CHECK_ARG(out && (n == 0 || grad), "null buffer");   // idx == NULL: rows 0 .. n-1
  hipStream_t stream = (hipStream_t)stream_;
  (void)hipMemsetAsync(out, 0, (size_t)total_rows * dim * (dtype == GT_F32 ? 4 : 2), stream);
  if (n == 0) return GT_OK;
  if (dtype == GT_F32)
    hipLaunchKernelGGL(k_rows_scatter<float>, dim3(grid_for(n * dim / 4)), dim3(UT), 0, stream, grad, idx, n, dim, (float*)out);
  else
    hipLaunchKernelGGL(k_rows_scatter<gt_bf16>, dim3(grid_for(n * dim / 4)), dim3(UT), 0, stream, grad, idx, n, dim,
                       (gt_bf16*)out);
  GT_CHECK_LAUNCH();
  return GT_OK;
}

// ---- stand-alone dropout (F.dropout / nn.Dropout where no producing kernel can carry it: the residual dropouts of the masked
// encoder, masked_transformer_encoder.py:54,75; PNANodeEmbedding, pna/pna_module.py:78): y = keep ? x / (1 - p) : 0 with the
// counter hash of the fused dropouts (element index, seed); the backward replays the same mask on the gradient ---------------
namespace {
__device__ __forceinline__ uint32_t drop_hash(uint32_t s0, uint32_t s1, uint32_t idx) {
  uint32_t x = (idx * 0x9E3779B1u) ^ s0;
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x ^ s1;
}
template <typename T>
__global__ void __launch_bounds__(UT) k_dropout(const T* __restrict__ x, T* __restrict__ y, int64_t n4, uint32_t thr, float inv_keep,
                                                uint32_t s0, uint32_t s1) {
  for (int64_t i = (int64_t)blockIdx.x * UT + threadIdx.x; i < n4; i += (int64_t)gridDim.x * UT) {
    float4 v = gt_load4<T>(x + i * 4);
    float* f = reinterpret_cast<float*>(&v);
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = drop_hash(s0, s1, (uint32_t)(i * 4 + e)) >= thr ? f[e] * inv_keep : 0.f;
    gt_store4<T>(y + i * 4, v);
  }
}
}  // namespace

extern "C" int gt_dropout(int dtype, const void* x, void* y, int64_t n, float p, uint64_t seed, gt_stream_t stream_) {
  GT_CHECK_ARG(x && y && n >= 0 && n % 4 == 0 && n < (int64_t)1 << 32, "n must be a multiple of 4 below 2^32");
  GT_CHECK_ARG(p >= 0.f && p < 1.f, "p must be in [0,1)");
  GT_CHECK_ARG(dtype == GT_F32 || dtype == GT_BF16, "bad dtype");
  if (n == 0) return GT_OK;
  const double t = (double)p * 4294967296.0;
  const uint32_t thr = p > 0.f ? (uint32_t)(t > 4294967295.0 ? 4294967295.0 : (t < 1.0 ? 1.0 : t)) : 0u;
  const int64_t n4 = n / 4;
  const unsigned grid = (unsigned)(gt_cdiv(n4, UT) < 4096 ? gt_cdiv(n4, UT) : 4096);
  if (dtype == GT_F32)
    hipLaunchKernelGGL(k_dropout<float>, dim3(grid), dim3(UT), 0, (hipStream_t)stream_, (const float*)x, (float*)y, n4, thr,
                       1.0f / (1.0f - p), (uint32_t)seed, (uint32_t)(seed >> 32));
  else
    hipLaunchKernelGGL(k_dropout<gt_bf16>, dim3(grid), dim3(UT), 0, (hipStream_t)stream_, (const gt_bf16*)x, (gt_bf16*)y, n4, thr,
                       1.0f / (1.0f - p), (uint32_t)seed, (uint32_t)(seed >> 32));
  GT_CHECK_LAUNCH();
  return GT_OK;
}

extern "C" int gt_add3(const float* a, const float* b, const float* c, int64_t n, float* out, gt_stream_t stream_) {
  GT_CHECK_ARG(n >= 0 && n % 4 == 0, "element count must be a multiple of 4");
  if (n == 0) return GT_OK;
  GT_CHECK_ARG(a && b && out, "null buffer");
  hipLaunchKernelGGL(k_add3, dim3(grid_for(n / 4)), dim3(UT), 0, (hipStream_t)stream_, a, b, c, n / 4, out);
  GT_CHECK_LAUNCH();
  return GT_OK;
}
