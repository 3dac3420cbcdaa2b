// attention_drop.h -- window attention WITH attention dropout, head_dim 16 (included by attention.hip).
//
// flash_attn_varlen_qkvpacked_func(..., dropout_p = self.attn_drop if self.training else 0) at
// pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py:208-214 (every reference config sets attn_drop = 0.0, so these
// kernels are the rarely-taken branch: plain loops over the same LDS images and fragment layouts as attention.hip, none of its
// software pipelining).  flash-attn's semantics: the softmax is normalised over ALL keys, then every probability is dropped with
// probability p and the survivors scaled by 1 / (1 - p):
//     O = ((M o P) / (1 - p)) V,   lse from the undropped scores,
//     dV = ((M o P) / (1 - p))^T dO,   dP = (M / (1 - p)) o (dO V^T),   dS = P o (dP - delta),   delta_i = sum_j dO_ij O_ij.
// The mask M is a pure function of (seed, sequence, head, query, key) -- a 32-bit integer hash compared against p 2^32 -- so the forward
// and the two backward kernels regenerate it in whatever register layout they hold the tile in (the forward and dQ kernels own a
// query per lane, the dK / dV kernel a key per lane); nothing is stored.  The random stream is NOT flash-attn's Philox stream (that
// library is un-vendored and its stream an implementation detail; the reference's own results under dropout are a function of it):
// parity is stated against the oracle with the SAME mask (oracle/ops.py::attn_dropout_keep), and statistically (keep rate).
// Forward: two accumulators -- [V^T ; 1] (M o P) for the numerator and [V^T ; 1] P whose row 16 is the undropped denominator.
#pragma once

__device__ __forceinline__ uint32_t ad_unit_key(uint32_t seed_lo, uint32_t seed_hi, uint32_t unit) {
  uint32_t h = seed_lo ^ (unit * 0x9E3779B1u);
  h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
  h ^= seed_hi * 0xC2B2AE3Du;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
  return h;
}
// keep decision of element (query q, key k), q, k < 1024, inside the unit with key `uk`: P(keep) = 1 - thresh / 2^32
__device__ __forceinline__ bool ad_keep(uint32_t uk, int q, int k, uint32_t thresh) {
  uint32_t h = (((uint32_t)q << 10) | (uint32_t)k) ^ uk;
  h *= 0x9E3779B1u; h ^= h >> 15; h *= 0x85EBCA77u; h ^= h >> 13; h *= 0xC2B2AE3Du; h ^= h >> 16;
  return h >= thresh;
}

// ------------------------------------------------------------------------------------------------ forward
// LDS: K row-major [lp_max][16] | V^T [17][pitch] (row 16 = 1.0 for keys < L)
template <bool F16>
__global__ void __launch_bounds__(AT_THREADS, 2)
attn_drop_fwd_kernel(const uint16_t* __restrict__ qkv, const int32_t* __restrict__ cu, int H, float scale, int64_t total, int lp_max,
                     int n_units, uint32_t thresh, float rp, uint32_t seed_lo, uint32_t seed_hi, uint16_t* __restrict__ out,
                     float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int unit = at_unit(n_units);
  if (unit >= n_units) return;
  const int seq = unit / H, head = unit % H;
  const int a = cu[seq], L = cu[seq + 1] - a;
  if (L <= 0) return;
  const int Lp = (L + 31) & ~31, n_tiles = Lp >> 5;
  if (Lp > lp_max) {
    at_poison_rows<F16>(out + ((int64_t)a * H + head) * 16, (int64_t)H * 16, L, lse + (int64_t)head * total + a);
    return;
  }
  const int pitch = lp_max + 8;
  unsigned char* Ksm = smem;
  unsigned char* Vt = smem + (size_t)lp_max * 32;
  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  const int64_t rs = (int64_t)3 * H * 16;
  stage_row_major<F16>(qkv + qkv_off(a, 1, H, head), rs, L, Lp, Ksm);
  stage_transposed<F16>(qkv + qkv_off(a, 2, H, head), rs, L, Lp, pitch, Vt);
  for (int key = threadIdx.x; key < Lp; key += AT_THREADS)
    reinterpret_cast<uint16_t*>(Vt + (size_t)16 * pitch * 2)[vt_pos(key)] = key < L ? (uint16_t)0x3F80 : (uint16_t)0;
  __syncthreads();
  const uint32_t uk = ad_unit_key(seed_lo, seed_hi, (uint32_t)unit);
  const int col = lane & 31, h2 = lane >> 5;
  const float c = scale * AT_LOG2E;
  const unsigned char* vbase = Vt + ((size_t)(col <= 16 ? col : (col & 15)) * pitch + 8 * h2) * 2;
  const unsigned char* kbase = Ksm + rm_off(col, h2);
  for (int qt = wave; qt < n_tiles; qt += AT_WAVES) {
    const int q = qt * 32 + col;
    const s16x8 qf = ld_global_frag<F16>(qkv + qkv_off(a + q, 0, H, head) + h2 * 8, q < L);
    s16x8 qhi, qlo;
    split_scaled(qf, c, qhi, qlo);
    f32x16 acc = zero16(), accl = zero16();
    float m = -INFINITY;
    const unsigned char* kp = kbase;
    const unsigned char* vp = vbase;
    for (int kt = 0; kt < n_tiles; ++kt) {
      const s16x8 kf = *reinterpret_cast<const s16x8*>(kp);
      kp += 1024;
      f32x16 s = mfma32(kf, qhi, zero16());
      s = mfma32(kf, qlo, s);
      if (kt == n_tiles - 1 && L < Lp) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kt * 32 + crow(r, h2) >= L) s[r] = -INFINITY;
      }
      float mt = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[r]);
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      const float m_new = fmaxf(m, mt);
      const float alpha = __builtin_amdgcn_exp2f(m - m_new);
      m = m_new;
#pragma unroll
      for (int r = 0; r < 9; ++r) { acc[r] *= alpha; accl[r] *= alpha; }
      uint32_t pk[8], pd[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float e0 = __builtin_amdgcn_exp2f(s[2 * i] - m), e1 = __builtin_amdgcn_exp2f(s[2 * i + 1] - m);
        const bool k0 = ad_keep(uk, q, kt * 32 + crow(2 * i, h2), thresh), k1 = ad_keep(uk, q, kt * 32 + crow(2 * i + 1, h2), thresh);
        pk[i] = pack_bf16x2(e0, e1);
        pd[i] = pack_bf16x2(k0 ? e0 : 0.f, k1 ? e1 : 0.f);
      }
#pragma unroll
      for (int mm = 0; mm < 2; ++mm) {
        const s16x8 vf = *reinterpret_cast<const s16x8*>(vp + mm * 32);
        acc = mfma32(vf, make_frag(pd[4 * mm], pd[4 * mm + 1], pd[4 * mm + 2], pd[4 * mm + 3]), acc);
        accl = mfma32(vf, make_frag(pk[4 * mm], pk[4 * mm + 1], pk[4 * mm + 2], pk[4 * mm + 3]), accl);
      }
      vp += 64;
    }
    const float l = __shfl(accl[8], col, 64);    // undropped denominator (row 16) lives in the h2 = 0 lane of column q
    const float inv = rp / l;
    if (q < L) {
      uint16_t* o = out + ((int64_t)(a + q) * H + head) * 16;
      uint2 w0, w1;
      w0.x = at_out<F16>(pack_bf16x2(acc[0] * inv, acc[1] * inv)); w0.y = at_out<F16>(pack_bf16x2(acc[2] * inv, acc[3] * inv));
      w1.x = at_out<F16>(pack_bf16x2(acc[4] * inv, acc[5] * inv)); w1.y = at_out<F16>(pack_bf16x2(acc[6] * inv, acc[7] * inv));
      *reinterpret_cast<uint2*>(o + 4 * h2) = w0;
      *reinterpret_cast<uint2*>(o + 8 + 4 * h2) = w1;
      if (h2 == 0) lse[(int64_t)head * total + a + q] = m * AT_LN2 + __logf(l);
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward, dQ + delta
// LDS: V row-major [lp_max][16] | K row-major [lp_max][16]
template <bool F16>
__global__ void __launch_bounds__(AT_THREADS, 2)
attn_drop_bwd_dq_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ out, const uint16_t* __restrict__ dout,
                        const float* __restrict__ lse, const int32_t* __restrict__ cu, int H, float scale, int64_t total, int lp_max,
                        int n_units, uint32_t thresh, float rp, uint32_t seed_lo, uint32_t seed_hi, uint16_t* __restrict__ dqkv,
                        float* __restrict__ delta) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int unit = at_unit(n_units);
  if (unit >= n_units) return;
  const int seq = unit / H, head = unit % H;
  const int a = cu[seq], L = cu[seq + 1] - a;
  if (L <= 0) return;
  const int Lp = (L + 31) & ~31, n_tiles = Lp >> 5;
  if (Lp > lp_max) {
    at_poison_rows<F16>(dqkv + qkv_off(a, 0, H, head), (int64_t)3 * H * 16, L, nullptr);
    return;
  }
  unsigned char* Vsm = smem;
  unsigned char* Ksm = smem + (size_t)lp_max * 32;
  const int64_t rs = (int64_t)3 * H * 16;
  stage_row_major<F16>(qkv + qkv_off(a, 2, H, head), rs, L, Lp, Vsm);
  stage_row_major<F16>(qkv + qkv_off(a, 1, H, head), rs, L, Lp, Ksm);
  __syncthreads();
  const uint32_t uk = ad_unit_key(seed_lo, seed_hi, (uint32_t)unit);
  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  const int col = lane & 31, h2 = lane >> 5;
  const float c = scale * AT_LOG2E;
  const TrAddr16 ta = tr_addr16(lane);
  const int rmo = rm_off(col, h2);
  for (int qt = wave; qt < n_tiles; qt += AT_WAVES) {
    const int q = qt * 32 + col;
    const bool qv = q < L;
    const s16x8 qf = ld_global_frag<F16>(qkv + qkv_off(a + q, 0, H, head) + h2 * 8, qv);
    const int64_t orow = ((int64_t)(a + q) * H + head) * 16 + h2 * 8;
    const s16x8 dof = ld_global_frag<F16>(dout + orow, qv);
    const s16x8 of = ld_global_frag<F16>(out + orow, qv);
    float dl = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) dl += bf16_bits_to_float((uint16_t)dof[j]) * bf16_bits_to_float((uint16_t)of[j]);
    dl += __shfl_xor(dl, 32, 64);
    const float l2 = qv ? lse[(int64_t)head * total + a + q] * AT_LOG2E : INFINITY;
    if (qv && h2 == 0) delta[(int64_t)head * total + a + q] = dl;
    s16x8 qhi, qlo;
    split_scaled(qf, c, qhi, qlo);
    const f32x16 negl = splat16(-l2);
    at_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < n_tiles; ++kt) {
      const int o = kt * 1024;
      const s16x8 vf = *reinterpret_cast<const s16x8*>(Vsm + rmo + o);
      const s16x8 kf = *reinterpret_cast<const s16x8*>(Ksm + rmo + o);
      f32x16 s = mfma32(kf, qhi, negl);
      s = mfma32(kf, qlo, s);
      const f32x16 dp = mfma32(vf, dof, zero16());          // raw dP^T = V dO^T
      uint32_t pk[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bool k0 = ad_keep(uk, q, kt * 32 + crow(2 * i, h2), thresh), k1 = ad_keep(uk, q, kt * 32 + crow(2 * i + 1, h2), thresh);
        const float d0 = (k0 ? dp[2 * i] * rp : 0.f) - dl, d1 = (k1 ? dp[2 * i + 1] * rp : 0.f) - dl;
        pk[i] = pack_bf16x2(__builtin_amdgcn_exp2f(s[2 * i]) * d0, __builtin_amdgcn_exp2f(s[2 * i + 1]) * d1);
      }
      s16x8 ds0, ds1;
      at_to_b16(pk, ds0, ds1);
      const s16x8 ktf = ld_tr_pair(Ksm + ta.lo + o, Ksm + ta.hi + o);
      acc0 = mfma16(ktf, ds0, acc0);
      acc1 = mfma16(ktf, ds1, acc1);
    }
    const int g = lane >> 4, n = lane & 15;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int qq = qt * 32 + 16 * t + n;
      if (qq < L) {
        const at_f32x4 v = t ? acc1 : acc0;
        uint2 w;
        w.x = at_out<F16>(pack_bf16x2(v[0] * scale, v[1] * scale));
        w.y = at_out<F16>(pack_bf16x2(v[2] * scale, v[3] * scale));
        *reinterpret_cast<uint2*>(dqkv + qkv_off(a + qq, 0, H, head) + 4 * g) = w;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward, dK / dV
// LDS: Q row-major [lp_max][16] | dO row-major [lp_max][16] | -lse * log2 e fp32 [lp_max] | -delta fp32 [lp_max]
template <bool F16>
__global__ void __launch_bounds__(AT_THREADS, 2)
attn_drop_bwd_dkv_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dout, const float* __restrict__ lse,
                         const float* __restrict__ delta, const int32_t* __restrict__ cu, int H, float scale, int64_t total, int lp_max,
                         int n_units, uint32_t thresh, float rp, uint32_t seed_lo, uint32_t seed_hi, uint16_t* __restrict__ dqkv) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int unit = at_unit(n_units);
  if (unit >= n_units) return;
  const int seq = unit / H, head = unit % H;
  const int a = cu[seq], L = cu[seq + 1] - a;
  if (L <= 0) return;
  const int Lp = (L + 31) & ~31, n_tiles = Lp >> 5;
  if (Lp > lp_max) {
    at_poison_rows<F16>(dqkv + qkv_off(a, 1, H, head), (int64_t)3 * H * 16, L, nullptr);
    at_poison_rows<F16>(dqkv + qkv_off(a, 2, H, head), (int64_t)3 * H * 16, L, nullptr);
    return;
  }
  unsigned char* Qsm = smem;
  unsigned char* Dsm = smem + (size_t)lp_max * 32;
  float* nl = reinterpret_cast<float*>(smem + (size_t)lp_max * 64);
  float* nd = nl + lp_max;
  stage_row_major<F16>(qkv + qkv_off(a, 0, H, head), (int64_t)3 * H * 16, L, Lp, Qsm);
  stage_row_major<F16>(dout + ((int64_t)a * H + head) * 16, (int64_t)H * 16, L, Lp, Dsm);
  for (int q = threadIdx.x; q < Lp; q += AT_THREADS) {
    nl[q] = q < L ? -lse[(int64_t)head * total + a + q] * AT_LOG2E : -AT_PAD_LSE;
    nd[q] = q < L ? -delta[(int64_t)head * total + a + q] : 0.f;
  }
  __syncthreads();
  const uint32_t uk = ad_unit_key(seed_lo, seed_hi, (uint32_t)unit);
  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  const int col = lane & 31, h2 = lane >> 5;
  const float c = scale * AT_LOG2E;
  const TrAddr16 ta = tr_addr16(lane);
  const int rmo = rm_off(col, h2);
  for (int kt = wave; kt < n_tiles; kt += AT_WAVES) {
    const int key = kt * 32 + col;
    const s16x8 kf = ld_global_frag<F16>(qkv + qkv_off(a + key, 1, H, head) + h2 * 8, key < L);
    const s16x8 vf = ld_global_frag<F16>(qkv + qkv_off(a + key, 2, H, head) + h2 * 8, key < L);
    s16x8 khi, klo;
    split_scaled(kf, c, khi, klo);
    at_f32x4 dv0 = {0.f, 0.f, 0.f, 0.f}, dv1 = dv0, dk0 = dv0, dk1 = dv0;
    for (int qt = 0; qt < n_tiles; ++qt) {
      const int o = qt * 1024;
      const s16x8 qf = *reinterpret_cast<const s16x8*>(Qsm + rmo + o);
      const s16x8 dof = *reinterpret_cast<const s16x8*>(Dsm + rmo + o);
      f32x16 s, ndv;                                        // -lse / -delta of the lane's 16 queries (rows crow(r, h2))
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = nl[qt * 32 + crow(r, h2)];
        ndv[r] = nd[qt * 32 + crow(r, h2)];
      }
      s = mfma32(qf, khi, s);                               // S'[q][key] - lse: lane = key, registers = queries
      s = mfma32(qf, klo, s);
      const f32x16 dp = mfma32(dof, vf, zero16());          // raw dP[q][key]
      uint32_t pp[8], ps[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float p0 = __builtin_amdgcn_exp2f(s[2 * i]), p1 = __builtin_amdgcn_exp2f(s[2 * i + 1]);
        const bool k0 = ad_keep(uk, qt * 32 + crow(2 * i, h2), key, thresh), k1 = ad_keep(uk, qt * 32 + crow(2 * i + 1, h2), key, thresh);
        pp[i] = pack_bf16x2(k0 ? p0 * rp : 0.f, k1 ? p1 * rp : 0.f);
        ps[i] = pack_bf16x2(p0 * ((k0 ? dp[2 * i] * rp : 0.f) + ndv[2 * i]), p1 * ((k1 ? dp[2 * i + 1] * rp : 0.f) + ndv[2 * i + 1]));
      }
      s16x8 p0f, p1f, s0f, s1f;
      at_to_b16(pp, p0f, p1f);
      at_to_b16(ps, s0f, s1f);
      const s16x8 dotf = ld_tr_pair(Dsm + ta.lo + o, Dsm + ta.hi + o);
      const s16x8 qtf = ld_tr_pair(Qsm + ta.lo + o, Qsm + ta.hi + o);
      dv0 = mfma16(dotf, p0f, dv0);
      dv1 = mfma16(dotf, p1f, dv1);
      dk0 = mfma16(qtf, s0f, dk0);
      dk1 = mfma16(qtf, s1f, dk1);
    }
    const int g = lane >> 4, n = lane & 15;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int kk = kt * 32 + 16 * t + n;
      if (kk < L) {
        const at_f32x4 vk = t ? dk1 : dk0, vv = t ? dv1 : dv0;
        uint2 wk, wv;
        wk.x = at_out<F16>(pack_bf16x2(vk[0] * scale, vk[1] * scale));
        wk.y = at_out<F16>(pack_bf16x2(vk[2] * scale, vk[3] * scale));
        wv.x = at_out<F16>(pack_bf16x2(vv[0], vv[1]));
        wv.y = at_out<F16>(pack_bf16x2(vv[2], vv[3]));
        *reinterpret_cast<uint2*>(dqkv + qkv_off(a + kk, 1, H, head) + 4 * g) = wk;
        *reinterpret_cast<uint2*>(dqkv + qkv_off(a + kk, 2, H, head) + 4 * g) = wv;
      }
    }
  }
}
