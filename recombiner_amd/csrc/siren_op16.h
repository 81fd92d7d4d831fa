// 16-bit operand helpers shared by the width-32 kernel (siren_mlp_bf16.hip) and the wide kernel (siren_mlp_wide.hip).
#pragma once
#include "siren_common.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));

// 16-bit operand traits: bf16 (8-bit mantissa) or f16 (11-bit mantissa, gradients pre-scaled by 2^10
// so that they stay in f16's normal range; everything downstream is linear in them and unscaled in fp32)
template <typename T> struct Op16;
template <> struct Op16<__bf16> {
  typedef __bf16 v8 __attribute__((ext_vector_type(8)));
  typedef __bf16 v4 __attribute__((ext_vector_type(4)));
  typedef __bf16 v2 __attribute__((ext_vector_type(2)));
  static constexpr float GRAD_SCALE = 1.0f;
  static constexpr float W_SCALE = 1.0f;
  static constexpr bool IS_BF16 = true;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ float dot2(v2 a, v2 b, float c) { return __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false); }
};
template <> struct Op16<_Float16> {
  typedef _Float16 v8 __attribute__((ext_vector_type(8)));
  typedef _Float16 v4 __attribute__((ext_vector_type(4)));
  typedef _Float16 v2 __attribute__((ext_vector_type(2)));
  static constexpr bool IS_BF16 = false;
  static constexpr float GRAD_SCALE = 1024.0f;
  // effective INR weights (h_w @ A) are ~1e-4: scaled by 2^10 into f16's normal range; the factor is
  // folded into the sine argument / cosine multipliers, biases are pre-scaled in the accumulator
  static constexpr float W_SCALE = 1024.0f;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ float dot2(v2 a, v2 b, float c) { return __builtin_amdgcn_fdot2(a, b, c, false); }
};

namespace rcb {
namespace op16 {

__host__ __device__ constexpr int cmax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ constexpr int fk(int s, int h, int j) { return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3); }

// 16-bit input rows as operand bits (IN16 instances).  Half-wave 0 loads the coordinate features from the caller's 16-bit copy of
// xf, which is already in the operand format T (rows padded to a multiple of 8 features: rcb_siren_desc.xf_bf16); half-wave 1
// loads the upsampled features, stored as bf16 by their producer: operand bits as they are when T is bf16, widened and rounded
// to f16 otherwise (the value the fp32-input path produces: (T)(float)bf16).
template <typename T>
__device__ __forceinline__ typename Op16<T>::v8 in16_operand(uint4 u, int h) {
  union { uint4 u; typename Op16<T>::v8 v; } bits, cv;
  bits.u = u;
  if constexpr (Op16<T>::IS_BF16) {
    return bits.v;
  } else {
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      cv.v[2 * i] = (T)__uint_as_float(w[i] << 16);
      cv.v[2 * i + 1] = (T)__uint_as_float(w[i] & 0xffff0000u);
    }
    cv.u.x = h ? cv.u.x : bits.u.x;
    cv.u.y = h ? cv.u.y : bits.u.y;
    cv.u.z = h ? cv.u.z : bits.u.z;
    cv.u.w = h ? cv.u.w : bits.u.w;
    return cv.v;
  }
}

template <typename T>
__device__ __forceinline__ float sum8_16(typename Op16<T>::v8 v, float acc) {
  const typename Op16<T>::v2 ones = {(T)1.0f, (T)1.0f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    typename Op16<T>::v2 pr = {v[2 * i], v[2 * i + 1]};
    acc = Op16<T>::dot2(pr, ones, acc);
  }
  return acc;
}

// pack registers 8s..8s+7 of an accumulator tile into the 16-bit B-operand of k-step s
template <typename T>
__device__ __forceinline__ typename Op16<T>::v8 pack8(const f32x16& v, int s) {
  typename Op16<T>::v8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (T)v[8 * s + j];
  return o;
}

// element offset of feature f of pixel row `pix` in a swizzled [pixel][feature] image: the 8-byte chunk
// index (f >> 2) has its low 3 bits XORed with (pix >> 1) & 7  -> conflict-free 8-byte row writes
// (16 consecutive pixels, same chunk) and conflict-free ds_read_b64_tr_b16 (4 rows x 8 chunks)
__device__ __forceinline__ int swz(int pix, int f, int stride) {
  const int c = f >> 2;
  return pix * stride + ((((c & 7) ^ ((pix >> 1) & 7)) | (c & ~7)) << 2) + (f & 3);
}

// transposed operand read: 8 pixels (16 s + 8 h + 0..7) of feature column (lane & 31) from a
// [pixel][feature] image with row stride `stride` elements, feature block offset `fcol`
template <typename T>
__device__ __forceinline__ typename Op16<T>::v8 read_tr(const T* img, int stride, int s, int lane, int fcol) {
  const int h = lane >> 5, fb = (lane >> 4) & 1, i = lane & 15, q4 = i >> 2, p4 = i & 3;
  union { s16x4 v[2]; typename Op16<T>::v8 b; } u;
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    const T* ptr = img + swz(16 * s + 8 * h + 4 * w + q4, fcol + 16 * fb + 4 * p4, stride);
    u.v[w] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)ptr);
  }
  return u.b;
}

}  // namespace op16
}  // namespace rcb
