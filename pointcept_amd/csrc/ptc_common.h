// ptc_common.h -- shared helpers for the gfx950 kernels of libptcore.so.
// Wave = 64 lanes everywhere (CDNA4); no warp-32 idioms, no CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/ptcore.h"

#define PTC_WAVE 64

// ---- error plumbing -------------------------------------------------------------------------
void ptc_set_error(const char* fmt, ...);

#define PTC_REQUIRE(cond, code, ...)      \
  do {                                    \
    if (!(cond)) {                        \
      ptc_set_error(__VA_ARGS__);         \
      return (code);                      \
    }                                     \
  } while (0)

#define PTC_CHECK_LAUNCH(name)                                                        \
  do {                                                                                \
    hipError_t e__ = hipGetLastError();                                               \
    if (e__ != hipSuccess) {                                                          \
      ptc_set_error("%s: launch failed: %s", (name), hipGetErrorString(e__));         \
      return PTC_EHIP;                                                                \
    }                                                                                 \
  } while (0)

#define PTC_HIP(call)                                                                 \
  do {                                                                                \
    hipError_t e__ = (call);                                                          \
    if (e__ != hipSuccess) {                                                          \
      ptc_set_error("%s failed: %s", #call, hipGetErrorString(e__));                  \
      return PTC_EHIP;                                                                \
    }                                                                                 \
  } while (0)

static inline int64_t ptc_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t ptc_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- storage types --------------------------------------------------------------------------
// bf16 / f16 are carried as raw 16-bit patterns; arithmetic is always fp32.
struct bf16_t { uint16_t x; };
struct f16_t { _Float16 x; };

__device__ __forceinline__ float ptc_to_float(float v) { return v; }
__device__ __forceinline__ float ptc_to_float(bf16_t v) { return __uint_as_float(((uint32_t)v.x) << 16); }
__device__ __forceinline__ float ptc_to_float(f16_t v) { return (float)v.x; }

// round-to-nearest-even fp32 -> bf16 (NaN kept quiet), same rounding as torch's .to(bfloat16):
// the hardware conversion v_cvt_pk_bf16_f32 of gfx950 (one instruction per two values)
__device__ __forceinline__ uint32_t ptc_pack_bf16x2(float lo, float hi) {
  typedef float ptc_f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 ptc_bf16x2 __attribute__((ext_vector_type(2)));
  ptc_f32x2 f = {lo, hi};
  ptc_bf16x2 h = __builtin_convertvector(f, ptc_bf16x2);
  return *reinterpret_cast<uint32_t*>(&h);
}
__device__ __forceinline__ uint16_t ptc_f32_to_bf16_bits(float f) { return (uint16_t)(ptc_pack_bf16x2(f, 0.f) & 0xffffu); }
template <typename T> __device__ __forceinline__ T ptc_from_float(float v);
template <> __device__ __forceinline__ float ptc_from_float<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t ptc_from_float<bf16_t>(float v) { bf16_t r; r.x = ptc_f32_to_bf16_bits(v); return r; }
template <> __device__ __forceinline__ f16_t ptc_from_float<f16_t>(float v) { f16_t r; r.x = (_Float16)v; return r; }

// dispatch a templated launcher on the ptc_dtype tag
#define PTC_DISPATCH_DTYPE(dtype, T, ...)                 \
  switch (dtype) {                                        \
    case PTC_F32: { using T = float; __VA_ARGS__; } break; \
    case PTC_F16: { using T = f16_t; __VA_ARGS__; } break; \
    case PTC_BF16: { using T = bf16_t; __VA_ARGS__; } break; \
    default: ptc_set_error("bad dtype %d", (int)(dtype)); return PTC_EINVAL; \
  }

static inline size_t ptc_dtype_size(int dtype) { return dtype == PTC_F32 ? 4 : 2; }

// XOR swizzle of the 16-byte pieces of a 128-byte row in the block-staged LDS images (blocks.hip writes it into the table entries,
// conv7.h / wgrad7.h apply it on the source side of their DMA): piece p of the row in slot s sits at position p ^ PTC_SWZ64(s).
// Bits 0-1 = bits 2-3 of the slot, bit 2 = bit 1 of the slot -- a bijection of (s >> 1) & 7, so 16 consecutive slots x one piece index
// still cover the 16 bank quads once (conv7's ds_read_b128 gathers), AND the 64-byte half of four consecutive slots covers the four
// 64-byte bank quarters once (wgrad7's ds_read_b64_tr_b16 gathers: a 32-lane pass reads 4 rows x 64 bytes; with the round-3 swizzle
// (s >> 1) & 7 slots s and s + 2 met in one quarter: 2-way conflicts on every gather of the weight-gradient kernel, r04_b).
#define PTC_SWZ64(s) (((((s) >> 1) & 1) << 2) | (((s) >> 2) & 3))

// ---- wave helpers ---------------------------------------------------------------------------
__device__ __forceinline__ int ptc_lane() { return threadIdx.x & 63; }

// f(ptc_int<0>{}), f(ptc_int<1>{}), ...: an unrolled loop whose index is a TYPE (a register array indexed by it never turns into a
// runtime-indexed -- i.e. scratch -- access, whatever the control flow inside f)
template <int I> struct ptc_int { static constexpr int value = I; };
template <int N, int I = 0, typename F> __device__ __forceinline__ void ptc_static_for(F&& f) {
  if constexpr (I < N) {
    f(ptc_int<I>{});
    ptc_static_for<N, I + 1>(f);
  }
}

// One rotary pair (u, v) <- (u cos - v sin, v cos + u sin) with its operation order FIXED (one multiply + one fused multiply-add per
// output): rope.hip's pass and the rotation fused into the attention kernels (attention_hd.h) must round identically, and the compiler's
// own contraction of `u * cs - v * sn` picks either product for the fma depending on the surrounding code.
__host__ __device__ __forceinline__ void ptc_rope_pair(float u, float v, float cs, float sn, float& ru, float& rv) {
  ru = fmaf(u, cs, -(v * sn));
  rv = fmaf(v, cs, u * sn);
}

// ---- nn.GELU() (erf form) in fp32, branch-free (round 6) ------------------------------------------------------------------------------
// Phi(z) = 0.5 erfc(-z / sqrt 2).  libm's erff is two exec-masked branches (|x| < 1: odd polynomial; else 1 - exp(-p(|x|))), both taken
// by every wave of real activations: 38 vector instructions per GELU, 42 per GELU' -- at 819200 x 256 hidden values per stage-0 MLP the
// epilogues of fc1 / fc2's input gradient were VALU-bound on it.  Here ONE form on the whole axis: with t = min(|z| / sqrt 2, 3.95),
//     0.5 erfc(t) = exp2(t r(t) - 1),   r = degree-7 minimax fit of log2(erfc(t)) / t on [0, 3.95] weighted by erfc (tools/fit_gelu.py),
// Phi(z) = that value for z < 0 and 1 - it for z >= 0: 8 fused multiply-adds, one v_exp_f32, a compare / select -- 15 instructions per
// GELU, 20 per (GELU', sharing nothing else).  Absolute error of Phi <= 8.2e-8 and of GELU <= 4.1e-7 over [-8, 8] in fp32 arithmetic
// (torch's own fp32 GELU: 1.2e-6, it forms 1 + erf and loses the left tail; against the exact function this form rounds to a different
// bf16 value than the exact one on 0.002 % of [-3, 8], torch's on 0.03 %).  PTC_FAST_GELU=0 restores libm's erff (A/B builds).
#ifndef PTC_FAST_GELU
#define PTC_FAST_GELU 1
#endif
__device__ __forceinline__ float ptc_gelu_cdf(float z) {
#if PTC_FAST_GELU
  const float t = fminf(fabsf(z) * 0.70710678118654752f, 3.95f);
  float r = -4.535858889e-05f;
  r = fmaf(r, t, 4.455075312e-04f);
  r = fmaf(r, t, -1.489441160e-03f);
  r = fmaf(r, t, -7.746305554e-04f);
  r = fmaf(r, t, 2.825368195e-02f);
  r = fmaf(r, t, -1.484816155e-01f);
  r = fmaf(r, t, -9.184163932e-01f);
  r = fmaf(r, t, -1.627908593e+00f);
  const float h = __builtin_amdgcn_exp2f(fmaf(r, t, -1.0f));      // 0.5 erfc(t)
  return z < 0.f ? h : 1.0f - h;
#else
  return 0.5f * (1.f + erff(z * 0.70710678118654752f));
#endif
}
__device__ __forceinline__ float ptc_gelu(float z) {
#if PTC_FAST_GELU
  return z * ptc_gelu_cdf(z);
#else
  return 0.5f * z * (1.f + erff(z * 0.70710678118654752f));
#endif
}
__device__ __forceinline__ float ptc_gelu_grad(float z) {       // Phi(z) + z phi(z)
#if PTC_FAST_GELU
  return fmaf(z * 0.3989422804014327f, __builtin_amdgcn_exp2f(z * z * -0.72134752044448170f), ptc_gelu_cdf(z));
#else
  return 0.5f * (1.f + erff(z * 0.70710678118654752f)) + z * 0.3989422804014327f * __expf(-0.5f * z * z);
#endif
}

// exactly N waves per SIMD as the register budget of a kernel (launch_bounds' second argument is only a lower bound: the compiler then
// aimed at three waves for gemm3.h / wgrad3.h and spilled their prefetch registers); nothing on the host emulation
#ifdef __HIPCC__
#define PTC_WAVES_PER_EU(lo, hi) __attribute__((amdgpu_waves_per_eu(lo, hi)))
#else
#define PTC_WAVES_PER_EU(lo, hi)
#endif

// ---- library-internal entry points shared by translation units (C++ linkage: not part of include/ptcore.h) ----------------------------
// ptc_sort_keys (scan_sort.hip) that also returns the key words in sorted order; see there.
int ptc_sort_keys_ex(const int64_t* keys, int64_t n, int k, int begin_bit, int end_bit, int64_t* order, int64_t* inverse,
                     int64_t* sorted_keys, void* workspace, size_t workspace_bytes, ptc_stream_t stream);
