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



// ---- predicated 16-byte loads without branches: raw buffer loads -------------------------------------------
// `buffer_load_dwordx4 ... offen` range-checks the per-lane byte offset against the resource's num_records and
// returns ZEROS for out-of-range lanes without touching memory.  Gathers through a sparse table ("no neighbour":
// ~2/3 of the entries) become ONE unconditional instruction: no exec-masked branch (which made the compiler wait
// vmcnt(0) at every use and serialised every prefetch pipeline in this library, r01_aj), and no stand-in fetch
// for absent rows (r01_ak: clamped unconditional loads cost conv3 7 % because every lane then fetched).
// Tensors must be < 2 GiB (PTC_BUF_OOB is the out-of-range offset); callers fall back to the v1 kernels above that.
#define PTC_BUF_MAX_BYTES 0x7fffffffull
#define PTC_BUF_OOB 0x80000000u
typedef int ptc_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ptc_buf(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 ptc_buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  const ptc_i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
  return make_uint4((uint32_t)v[0], (uint32_t)v[1], (uint32_t)v[2], (uint32_t)v[3]);
}
__device__ __forceinline__ int32_t ptc_buf_load4(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0);
}
template <typename T>
__device__ __forceinline__ typename Mma<T>::frag ld_frag_buf(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  const uint4 v = ptc_buf_load16(r, byte_off);
  typename Mma<T>::frag f;
  __builtin_memcpy(&f, &v, sizeof(f));
  return f;
}
