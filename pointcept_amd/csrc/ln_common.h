// ln_common.h -- the row helpers of the LayerNorm / residual-joint kernels (norm.hip), shared with the GEMM epilogue that runs a joint
// inside linear2_kernel (fwd2.h, EPI 3): a lane holds LN_VEC = 8 consecutive channels of a row, LPR = C / 8 consecutive lanes hold the row.
#pragma once

#define LN_THREADS 256
#define LN_VEC 8  // channels per lane

template <typename T>
__device__ __forceinline__ void ln_load8(const T* p, float (&v)[LN_VEC]);
template <>
__device__ __forceinline__ void ln_load8<float>(const float* p, float (&v)[LN_VEC]) {
  const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <>
__device__ __forceinline__ void ln_load8<bf16_t>(const bf16_t* p, float (&v)[LN_VEC]) {
  const uint4 u = *reinterpret_cast<const uint4*>(p);
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
template <>
__device__ __forceinline__ void ln_load8<f16_t>(const f16_t* p, float (&v)[LN_VEC]) {
  const uint4 u = *reinterpret_cast<const uint4*>(p);
  const _Float16* h = reinterpret_cast<const _Float16*>(&u);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)h[i];
}
template <typename T>
__device__ __forceinline__ void ln_store8(T* p, const float (&v)[LN_VEC]) {
  T o[LN_VEC];
#pragma unroll
  for (int i = 0; i < LN_VEC; ++i) o[i] = ptc_from_float<T>(v[i]);
  if (sizeof(T) == 4) { reinterpret_cast<uint4*>(p)[0] = reinterpret_cast<uint4*>(o)[0]; reinterpret_cast<uint4*>(p)[1] = reinterpret_cast<uint4*>(o)[1]; }
  else reinterpret_cast<uint4*>(p)[0] = reinterpret_cast<uint4*>(o)[0];
}

// sum over the LPR lanes of a row group (LPR power of two, groups aligned).  The butterfly steps inside a 16-lane row are DPP operands
// of the add itself (quad_perm for partners 1 and 2 lanes away; row_half_mirror / row_mirror for the other quad / the other half
// of the row: after the earlier steps every lane of a quad / half already holds that quad's / half's sum, so the mirrored lane carries the
// same value as the lane 4 / 8 away) instead of __shfl_xor's ds_bpermute round trip with its five address instructions: the same additions
// in the same order, bit for bit (round 4; LN_DPP 0 = the shuffle form, timing A/B: 60 -> 12 instructions per row of the backward joint,
// no difference in the step -- these kernels wait for HBM, profiles/r04_zr_ln_dpp_ab.txt).  Steps of 16 / 32 lanes keep the shuffle.
#ifndef LN_DPP
#define LN_DPP 1
#endif
#if defined(__HIPCC__) && LN_DPP
template <int CTRL>
__device__ __forceinline__ float ln_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
#endif
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#if defined(__HIPCC__) && LN_DPP
  if (LPR >= 2) v += ln_dpp<0xB1>(v);      // quad_perm [1, 0, 3, 2]
  if (LPR >= 4) v += ln_dpp<0x4E>(v);      // quad_perm [2, 3, 0, 1]
  if (LPR >= 8) v += ln_dpp<0x141>(v);     // row_half_mirror
  if (LPR >= 16) v += ln_dpp<0x140>(v);    // row_mirror
#pragma unroll
  for (int d = 16; d < LPR; d <<= 1) v += __shfl_xor(v, d, 64);
#else
#pragma unroll
  for (int d = 1; d < LPR; d <<= 1) v += __shfl_xor(v, d, 64);
#endif
  return v;
}


// LayerNorm of the row a lane group holds, in place; mean / rstd returned for the backward.  Shared by add_norm_fwd_kernel (norm.hip) and
// the GEMM epilogues that run a joint (fwd2_joint.h), which promise the SAME bits: every multiply-add is an explicit fmaf.  Written as
// `q += d * d`, -ffp-contract=fast lets the compiler decide per kernel which terms fuse (SLP-packed v_pk_mul_f32 + plain adds in one
// kernel, a scalar v_fmac_f32 for one of the eight terms in the other: one ulp of variance between the 128-channel instances of the two
// kernels, found on the MI355X in round 4); an fma intrinsic is never split, and a product feeding one is never merged into it.
template <int LPR>
__device__ __forceinline__ void ln_normalize(float (&v)[LN_VEC], float eps, const float (&g)[LN_VEC], const float (&b)[LN_VEC], float& mean,
                                             float& rstd) {
  constexpr float inv_c = 1.f / (LPR * LN_VEC);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_VEC; ++i) s += v[i];
  mean = group_sum<LPR>(s) * inv_c;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_VEC; ++i) { const float d = v[i] - mean; q = fmaf(d, d, q); }
  rstd = rsqrtf(fmaf(group_sum<LPR>(q), inv_c, eps));
#pragma unroll
  for (int i = 0; i < LN_VEC; ++i) v[i] = fmaf((v[i] - mean) * rstd, g[i], b[i]);
}

// the same with the row width as a run-time value (norm.hip, round 5: widths that are no power of two run on the next larger instance with
// the lanes past the row idle): `act` = this lane holds channels of the row (an idle lane's v is zero and stays out of the second moment).
// For inv_c = 1 / (LPR * 8) and act = true everywhere this is ln_normalize statement for statement -- the joints in the GEMM epilogues
// (fwd2_joint.h, compile-time widths) and add_norm_fwd_kernel keep producing the same bits.
template <int LPR>
__device__ __forceinline__ void ln_normalize_rt(float (&v)[LN_VEC], float eps, const float (&g)[LN_VEC], const float (&b)[LN_VEC], float& mean,
                                                float& rstd, float inv_c, bool act) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_VEC; ++i) s += v[i];
  mean = group_sum<LPR>(s) * inv_c;
  float q = 0.f;
  if (act) {
#pragma unroll
    for (int i = 0; i < LN_VEC; ++i) { const float d = v[i] - mean; q = fmaf(d, d, q); }
  }
  rstd = rsqrtf(fmaf(group_sum<LPR>(q), inv_c, eps));
#pragma unroll
  for (int i = 0; i < LN_VEC; ++i) v[i] = fmaf((v[i] - mean) * rstd, g[i], b[i]);
}
