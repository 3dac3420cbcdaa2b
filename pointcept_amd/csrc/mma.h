// mma.h -- MFMA operand / accumulator types shared by the implicit-GEMM kernels (spconv.hip,
// wgrad2.hip).  gfx950: v_mfma_f32_16x16x32_{bf16,f16} (8 contraction values per lane) and the
// exact-f32 v_mfma_f32_16x16x4_f32 (4 steps per 16 channels).
#pragma once
#include "ptc_common.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static constexpr int KS = 32;   // channels per super-step
  static constexpr int EPL = 8;   // elements per lane per super-step (16 bytes)
  using frag = s16x8;
  static __device__ __forceinline__ frag zero() { frag z = {0, 0, 0, 0, 0, 0, 0, 0}; return z; }
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma<f16_t> {
  static constexpr int KS = 32;
  static constexpr int EPL = 8;
  using frag = h16x8;
  static __device__ __forceinline__ frag zero() { frag z = {0, 0, 0, 0, 0, 0, 0, 0}; return z; }
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static constexpr int KS = 16;
  static constexpr int EPL = 4;
  using frag = f32x4;
  static __device__ __forceinline__ frag zero() { frag z = {0.f, 0.f, 0.f, 0.f}; return z; }
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
    return c;
  }
};

template <typename T>
__device__ __forceinline__ typename Mma<T>::frag ld_frag(const T* p) {
  return *reinterpret_cast<const typename Mma<T>::frag*>(p);
}

