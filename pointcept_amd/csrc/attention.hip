// attention.hip -- PTv3 serialized patch attention: variable-length, non-causal, head_dim 16,
// bf16 in / fp32 accumulate, forward + backward, on gfx950 MFMA.
//
// Replaces flash_attn.flash_attn_varlen_qkvpacked_func (third party, un-vendored) as called at
//   pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py:208-214
// (qkv [T,3,H,16] bf16, cu_seqlens int32, softmax_scale = 16^-0.5, dropout 0).
//
// Shape facts that drive the design (SURVEY section 7 "hard parts"): every PTv3 stage has
// head_dim = 16, windows are <= 1024 keys.  QK^T is ONE k-step of v_mfma_f32_32x32x16_bf16, so
// there is one exp per 64 MFMA flops: the kernel is VALU/transcendental-bound, not MFMA-bound.
// Therefore: MFMA work is spent freely to delete VALU work.
//   * One workgroup = one (sequence, head); all of K (row-major) and V^T for the sequence live in
//     LDS (<= 65 KB), each wave walks 32-query tiles against every 32-key tile.
//   * "Swapped" products S^T = K Q^T: a lane owns ONE query column (q = lane&31) and 16 keys in
//     registers, so row max / exp / scaling are lane-local; the only cross-lane op per tile is one
//     exchange with lane^32.
//   * P V is issued as O^T = [V^T ; 1 ; 0] P^T on the 32x32x16 MFMA: the packed P registers ARE
//     the B operand (no lane shuffles), row 16 of the A operand is all ones so the MFMA also
//     produces the softmax denominator (no VALU row sums), and O^T lands in the same lane as the
//     softmax state (rescale is lane-local).  Half of that MFMA's rows are padding; the matrix
//     pipe has the slack.
//   * The contraction order inside an MFMA is free as long as A and B agree: key slot (h,j) of
//     the P V product is key 16m + 4h + (j&3) + 8(j>>2), i.e. two 8-byte LDS reads of V^T.
// Backward = two kernels with the same skeleton (recompute P from q,k,lse; no atomics, no
// cross-wave reductions, bit-reproducible): dQ is query-stationary, dK/dV key-stationary.
// Roofline: fwd 4 L^2 D flops and L^2 exps per (sequence, head); bwd here 14 L^2 D
// flops (S and dP recomputed in both kernels) and 2 L^2 exps.
#include "ptc_common.h"

typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

#define AT_WAVES 8
#define AT_THREADS (AT_WAVES * 64)
#define AT_MAX_L 1024
#define AT_LOG2E 1.4426950408889634f

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  f32x2_t f = {lo, hi};
  bf16x2_t h = __builtin_convertvector(f, bf16x2_t);  // v_cvt_pk_bf16_f32 (RNE)
  return *reinterpret_cast<uint32_t*>(&h);
}
__device__ __forceinline__ float bf16_bits_to_float(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

__device__ __forceinline__ f32x16 mfma32(s16x8 a, s16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}
// C/D layout of the 32x32 MFMA: column = lane&31, row(reg, lane) = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
__device__ __forceinline__ int crow(int reg, int h2) { return (reg & 3) + 8 * (reg >> 2) + 4 * h2; }

// packed row index of element (t, j, head) in qkv [T,3,H,16]
__device__ __forceinline__ int64_t qkv_off(int64_t t, int j, int H, int head) { return ((t * 3 + j) * H + head) * 16; }

// ---- LDS images -------------------------------------------------------------------------------
// row-major [Lp][16] bf16 (32 B rows); the two 16-byte halves of a row are swapped when bit 3 of
// the row index is set, which makes the 16-lane ds_read_b128 groups conflict-free.
__device__ __forceinline__ int rm_off(int row, int half) { return row * 32 + ((half ^ ((row >> 3) & 1)) << 4); }

// stage rows [0,Lp) of component `comp` (0 q, 1 k, 2 v) of (sequence at a, head) row-major
__device__ __forceinline__ void stage_row_major(const uint16_t* __restrict__ src, int64_t row_stride, int L, int Lp,
                                                unsigned char* lds) {
  for (int row = threadIdx.x; row < Lp; row += AT_THREADS) {
    uint4 v0 = {0, 0, 0, 0}, v1 = {0, 0, 0, 0};
    if (row < L) {
      const uint4* p = reinterpret_cast<const uint4*>(src + (int64_t)row * row_stride);
      v0 = p[0];
      v1 = p[1];
    }
    *reinterpret_cast<uint4*>(lds + rm_off(row, 0)) = v0;
    *reinterpret_cast<uint4*>(lds + rm_off(row, 1)) = v1;
  }
}
// stage transposed [16][pitch] bf16 (pitch = Lp_max + 8 elements): thread handles two rows and
// writes one 32-bit word per channel
__device__ __forceinline__ void stage_transposed(const uint16_t* __restrict__ src, int64_t row_stride, int L, int Lp,
                                                 int pitch, unsigned char* lds) {
  uint32_t* t32 = reinterpret_cast<uint32_t*>(lds);
  for (int p = threadIdx.x; p < Lp / 2; p += AT_THREADS) {
    uint4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
    const int ra = 2 * p, rb = 2 * p + 1;
    if (ra < L) {
      const uint4* q = reinterpret_cast<const uint4*>(src + (int64_t)ra * row_stride);
      a0 = q[0]; a1 = q[1];
    }
    if (rb < L) {
      const uint4* q = reinterpret_cast<const uint4*>(src + (int64_t)rb * row_stride);
      b0 = q[0]; b1 = q[1];
    }
    uint16_t ea[16], eb[16];
    *reinterpret_cast<uint4*>(ea) = a0; *reinterpret_cast<uint4*>(ea + 8) = a1;
    *reinterpret_cast<uint4*>(eb) = b0; *reinterpret_cast<uint4*>(eb + 8) = b1;
#pragma unroll
    for (int d = 0; d < 16; ++d) t32[(d * pitch) / 2 + p] = (uint32_t)ea[d] | ((uint32_t)eb[d] << 16);
  }
}
// A/B operand "[16 channels ; ones/zeros][8 contraction slots]" from a transposed image:
// lane (i = lane&31, h2) -> channel i, slots j -> column base + (j&3) + 8*(j>>2)
__device__ __forceinline__ s16x8 ld_transposed_frag(const unsigned char* lds, int pitch, int i, int col, bool ones_row) {
  s16x8 f = {0, 0, 0, 0, 0, 0, 0, 0};
  if (i < 16) {
    const unsigned char* p = lds + ((int64_t)i * pitch + col) * 2;
    const s16x4 lo = *reinterpret_cast<const s16x4*>(p);
    const s16x4 hi = *reinterpret_cast<const s16x4*>(p + 16);
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  } else if (ones_row && i == 16) {
    const short one = (short)0x3F80;  // bf16 1.0
    f = (s16x8){one, one, one, one, one, one, one, one};
  }
  return f;
}
__device__ __forceinline__ s16x8 ld_global_frag(const uint16_t* p, bool valid) {
  s16x8 f = {0, 0, 0, 0, 0, 0, 0, 0};
  if (valid) f = *reinterpret_cast<const s16x8*>(p);
  return f;
}
__device__ __forceinline__ s16x8 make_frag(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  uint4 u = {a, b, c, d};
  return *reinterpret_cast<s16x8*>(&u);
}

// ================================================================================================
// forward
// ================================================================================================
__global__ void __launch_bounds__(AT_THREADS)
attn_fwd_kernel(const uint16_t* __restrict__ qkv, const int32_t* __restrict__ cu, int H, float scale, int64_t total,
                int lp_max, int n_units, uint16_t* __restrict__ out, float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // XCD-aware order: workgroup b runs on XCD b % 8, so logical unit = (b % 8) * per + b / 8 gives every
  // XCD a contiguous run of (sequence, head) units -- the H heads of a sequence read interleaved
  // 32-byte pieces of the same qkv rows and now share one L2 instead of re-fetching them per XCD.
  const int per_xcd = (n_units + 7) >> 3;
  const int unit = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (unit >= n_units) return;
  const int seq = unit / H, head = unit % H;
  const int a = cu[seq], L = cu[seq + 1] - a;
  if (L <= 0) return;
  const int Lp = (L + 31) & ~31, n_tiles = Lp >> 5;
  const int pitch = lp_max + 8;
  unsigned char* Ksm = smem;                          // [lp_max][16] row-major
  unsigned char* Vt = smem + (size_t)lp_max * 32;     // [16][pitch] transposed
  const int64_t rs = (int64_t)3 * H * 16;
  stage_row_major(qkv + qkv_off(a, 1, H, head), rs, L, Lp, Ksm);
  stage_transposed(qkv + qkv_off(a, 2, H, head), rs, L, Lp, pitch, Vt);
  __syncthreads();

  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  const int col = lane & 31, h2 = lane >> 5;
  const float c = scale * AT_LOG2E;

  for (int qt = wave; qt < n_tiles; qt += AT_WAVES) {
    const int q = qt * 32 + col;
    const s16x8 qf = ld_global_frag(qkv + qkv_off(a + q, 0, H, head) + h2 * 8, q < L);
    float m = -INFINITY;
    f32x16 acc = zero16();
    for (int kt = 0; kt < n_tiles; ++kt) {
      const s16x8 kf = *reinterpret_cast<const s16x8*>(Ksm + rm_off(kt * 32 + col, h2));
      f32x16 s = mfma32(kf, qf, zero16());  // S^T[key][q]: lane = q, regs = keys crow(r,h2)
      if (kt == n_tiles - 1 && L < Lp) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kt * 32 + crow(r, h2) >= L) s[r] = -INFINITY;
      }
      // tile maximum of this lane's 16 keys: 8 three-input maxima (v_max3_f32 costs the same issue
      // slot as v_max_f32, measured in tools/probe_gfx950.hip), then one exchange with lane ^ 32
      float mt = max3(max3(s[0], s[1], s[2]), max3(s[3], s[4], s[5]), max3(s[6], s[7], s[8]));
      mt = max3(mt, max3(s[9], s[10], s[11]), max3(s[12], s[13], s[14]));
      mt = fmaxf(mt, s[15]);
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      // deferred rescale: keep the stale running maximum while the tile maximum exceeds it by less than
      // 2^8 in the exp2 domain for every query of the wave (P <= 256, exact in the fp32 accumulators;
      // softmax is invariant to the reference point).  On trained / random logits almost every tile
      // after the first few skips the 9 accumulator multiplies and one exp.
      if (__builtin_amdgcn_ballot_w64((mt - m) * c > 8.0f) != 0) {
        const float m_new = fmaxf(m, mt);
        const float alpha = __builtin_amdgcn_exp2f((m - m_new) * c);
        m = m_new;
#pragma unroll
        for (int r = 0; r < 9; ++r) acc[r] *= alpha;  // rows 0..15 = O^T, row 16 (reg 8, h2=0) = denominator
      }
      const float mc = m * c;
      uint32_t pk[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        pk[i] = pack_bf16x2(__builtin_amdgcn_exp2f(s[2 * i] * c - mc), __builtin_amdgcn_exp2f(s[2 * i + 1] * c - mc));
#pragma unroll
      for (int mm = 0; mm < 2; ++mm) {
        const s16x8 pf = make_frag(pk[4 * mm], pk[4 * mm + 1], pk[4 * mm + 2], pk[4 * mm + 3]);
        const s16x8 vf = ld_transposed_frag(Vt, pitch, col, kt * 32 + 16 * mm + 4 * h2, true);
        acc = mfma32(vf, pf, acc);
      }
    }
    const float l = __shfl(acc[8], col, 64);  // denominator lives in the h2 = 0 lane of column q
    const float inv = 1.f / l;
    if (q < L) {
      uint16_t* o = out + ((int64_t)(a + q) * H + head) * 16;
      uint2 w0, w1;
      w0.x = pack_bf16x2(acc[0] * inv, acc[1] * inv); w0.y = pack_bf16x2(acc[2] * inv, acc[3] * inv);
      w1.x = pack_bf16x2(acc[4] * inv, acc[5] * inv); w1.y = pack_bf16x2(acc[6] * inv, acc[7] * inv);
      *reinterpret_cast<uint2*>(o + 4 * h2) = w0;       // d = 4*h2 + {0..3}
      *reinterpret_cast<uint2*>(o + 8 + 4 * h2) = w1;   // d = 8 + 4*h2 + {0..3}
      if (h2 == 0) lse[(int64_t)head * total + a + q] = m * scale + __logf(l);
    }
  }
}

// ================================================================================================
// backward, part 1: dQ (query-stationary) + delta = rowsum(dO * O)
// ================================================================================================
__global__ void __launch_bounds__(AT_THREADS)
attn_bwd_dq_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ out, const uint16_t* __restrict__ dout,
                   const float* __restrict__ lse, const int32_t* __restrict__ cu, int H, float scale, int64_t total,
                   int lp_max, int n_units, uint16_t* __restrict__ dqkv, float* __restrict__ delta) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // XCD-aware order: workgroup b runs on XCD b % 8, so logical unit = (b % 8) * per + b / 8 gives every
  // XCD a contiguous run of (sequence, head) units -- the H heads of a sequence read interleaved
  // 32-byte pieces of the same qkv rows and now share one L2 instead of re-fetching them per XCD.
  const int per_xcd = (n_units + 7) >> 3;
  const int unit = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (unit >= n_units) return;
  const int seq = unit / H, head = unit % H;
  const int a = cu[seq], L = cu[seq + 1] - a;
  if (L <= 0) return;
  const int Lp = (L + 31) & ~31, n_tiles = Lp >> 5;
  const int pitch = lp_max + 8;
  unsigned char* Vsm = smem;                          // V row-major
  unsigned char* Kt = smem + (size_t)lp_max * 32;     // K transposed
  const int64_t rs = (int64_t)3 * H * 16;
  stage_row_major(qkv + qkv_off(a, 2, H, head), rs, L, Lp, Vsm);
  stage_transposed(qkv + qkv_off(a, 1, H, head), rs, L, Lp, pitch, Kt);
  __syncthreads();

  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  const int col = lane & 31, h2 = lane >> 5;
  const float c = scale * AT_LOG2E;

  for (int qt = wave; qt < n_tiles; qt += AT_WAVES) {
    const int q = qt * 32 + col;
    const bool qv = q < L;
    const s16x8 qf = ld_global_frag(qkv + qkv_off(a + q, 0, H, head) + h2 * 8, qv);
    const int64_t orow = ((int64_t)(a + q) * H + head) * 16 + h2 * 8;
    const s16x8 dof = ld_global_frag(dout + orow, qv);
    const s16x8 of = ld_global_frag(out + orow, qv);
    float dl = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) dl += bf16_bits_to_float((uint16_t)dof[j]) * bf16_bits_to_float((uint16_t)of[j]);
    dl += __shfl_xor(dl, 32, 64);
    const float l2 = qv ? lse[(int64_t)head * total + a + q] * AT_LOG2E : INFINITY;
    if (qv && h2 == 0) delta[(int64_t)head * total + a + q] = dl;
    f32x16 acc = zero16();
    for (int kt = 0; kt < n_tiles; ++kt) {
      const int key = kt * 32 + col;
      const s16x8 kf = ld_global_frag(qkv + qkv_off(a + key, 1, H, head) + h2 * 8, key < L);
      const f32x16 s = mfma32(kf, qf, zero16());                                        // S^T
      const s16x8 vf = *reinterpret_cast<const s16x8*>(Vsm + rm_off(key, h2));
      const f32x16 dp = mfma32(vf, dof, zero16());                                      // dP^T
      uint32_t pk[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float p0 = __builtin_amdgcn_exp2f(s[2 * i] * c - l2), p1 = __builtin_amdgcn_exp2f(s[2 * i + 1] * c - l2);
        if (kt == n_tiles - 1 && L < Lp) {
          if (kt * 32 + crow(2 * i, h2) >= L) p0 = 0.f;
          if (kt * 32 + crow(2 * i + 1, h2) >= L) p1 = 0.f;
        }
        pk[i] = pack_bf16x2(p0 * (dp[2 * i] - dl), p1 * (dp[2 * i + 1] - dl));          // dS^T
      }
#pragma unroll
      for (int mm = 0; mm < 2; ++mm) {
        const s16x8 dsf = make_frag(pk[4 * mm], pk[4 * mm + 1], pk[4 * mm + 2], pk[4 * mm + 3]);
        const s16x8 ktf = ld_transposed_frag(Kt, pitch, col, kt * 32 + 16 * mm + 4 * h2, false);
        acc = mfma32(ktf, dsf, acc);                                                    // dQ^T[d][q]
      }
    }
    if (qv) {
      uint16_t* o = dqkv + qkv_off(a + q, 0, H, head);
      uint2 w0, w1;
      w0.x = pack_bf16x2(acc[0] * scale, acc[1] * scale); w0.y = pack_bf16x2(acc[2] * scale, acc[3] * scale);
      w1.x = pack_bf16x2(acc[4] * scale, acc[5] * scale); w1.y = pack_bf16x2(acc[6] * scale, acc[7] * scale);
      *reinterpret_cast<uint2*>(o + 4 * h2) = w0;
      *reinterpret_cast<uint2*>(o + 8 + 4 * h2) = w1;
    }
  }
}

// ================================================================================================
// backward, part 2: dK, dV (key-stationary).  Needs delta written by part 1 (same stream).
// ================================================================================================
__global__ void __launch_bounds__(AT_THREADS)
attn_bwd_dkv_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dout, const float* __restrict__ lse,
                    const float* __restrict__ delta, const int32_t* __restrict__ cu, int H, float scale, int64_t total,
                    int lp_max, int n_units, uint16_t* __restrict__ dqkv) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // XCD-aware order: workgroup b runs on XCD b % 8, so logical unit = (b % 8) * per + b / 8 gives every
  // XCD a contiguous run of (sequence, head) units -- the H heads of a sequence read interleaved
  // 32-byte pieces of the same qkv rows and now share one L2 instead of re-fetching them per XCD.
  const int per_xcd = (n_units + 7) >> 3;
  const int unit = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (unit >= n_units) return;
  const int seq = unit / H, head = unit % H;
  const int a = cu[seq], L = cu[seq + 1] - a;
  if (L <= 0) return;
  const int Lp = (L + 31) & ~31, n_tiles = Lp >> 5;
  const int pitch = lp_max + 8;
  unsigned char* Qt = smem;                                        // Q transposed [16][pitch]
  unsigned char* dOt = smem + (size_t)16 * pitch * 2;              // dO transposed
  float* l2s = reinterpret_cast<float*>(smem + (size_t)32 * pitch * 2);  // lse * log2e, +inf beyond L
  float* dls = l2s + lp_max;                                        // delta
  stage_transposed(qkv + qkv_off(a, 0, H, head), (int64_t)3 * H * 16, L, Lp, pitch, Qt);
  stage_transposed(dout + ((int64_t)a * H + head) * 16, (int64_t)H * 16, L, Lp, pitch, dOt);
  for (int q = threadIdx.x; q < Lp; q += AT_THREADS) {
    l2s[q] = q < L ? lse[(int64_t)head * total + a + q] * AT_LOG2E : INFINITY;
    dls[q] = q < L ? delta[(int64_t)head * total + a + q] : 0.f;
  }
  __syncthreads();

  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  const int col = lane & 31, h2 = lane >> 5;
  const float c = scale * AT_LOG2E;

  for (int kt = wave; kt < n_tiles; kt += AT_WAVES) {
    const int key = kt * 32 + col;
    const s16x8 kf = ld_global_frag(qkv + qkv_off(a + key, 1, H, head) + h2 * 8, key < L);
    const s16x8 vf = ld_global_frag(qkv + qkv_off(a + key, 2, H, head) + h2 * 8, key < L);
    f32x16 dv = zero16(), dk = zero16();
    for (int qt = 0; qt < n_tiles; ++qt) {
      const int q = qt * 32 + col;
      const s16x8 qf = ld_global_frag(qkv + qkv_off(a + q, 0, H, head) + h2 * 8, q < L);
      const s16x8 dof = ld_global_frag(dout + ((int64_t)(a + q) * H + head) * 16 + h2 * 8, q < L);
      const f32x16 s = mfma32(qf, kf, zero16());     // S[q][key]: lane = key, regs = queries crow(r,h2)
      const f32x16 dp = mfma32(dof, vf, zero16());   // dP[q][key]
      uint32_t pp[8], ps[8];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // registers 4g..4g+3 <-> queries qt*32 + 8g + 4*h2 + {0..3}: one 16-byte broadcast read each
        const float4 l4 = *reinterpret_cast<const float4*>(l2s + qt * 32 + 8 * g + 4 * h2);
        const float4 d4 = *reinterpret_cast<const float4*>(dls + qt * 32 + 8 * g + 4 * h2);
        const float p0 = __builtin_amdgcn_exp2f(s[4 * g + 0] * c - l4.x);
        const float p1 = __builtin_amdgcn_exp2f(s[4 * g + 1] * c - l4.y);
        const float p2 = __builtin_amdgcn_exp2f(s[4 * g + 2] * c - l4.z);
        const float p3 = __builtin_amdgcn_exp2f(s[4 * g + 3] * c - l4.w);
        pp[2 * g] = pack_bf16x2(p0, p1);
        pp[2 * g + 1] = pack_bf16x2(p2, p3);
        ps[2 * g] = pack_bf16x2(p0 * (dp[4 * g + 0] - d4.x), p1 * (dp[4 * g + 1] - d4.y));
        ps[2 * g + 1] = pack_bf16x2(p2 * (dp[4 * g + 2] - d4.z), p3 * (dp[4 * g + 3] - d4.w));
      }
#pragma unroll
      for (int mm = 0; mm < 2; ++mm) {
        const s16x8 pf = make_frag(pp[4 * mm], pp[4 * mm + 1], pp[4 * mm + 2], pp[4 * mm + 3]);     // P^T[key][q slots]
        const s16x8 dsf = make_frag(ps[4 * mm], ps[4 * mm + 1], ps[4 * mm + 2], ps[4 * mm + 3]);    // dS^T
        const s16x8 dotf = ld_transposed_frag(dOt, pitch, col, qt * 32 + 16 * mm + 4 * h2, false);  // dO[q slots][d]
        const s16x8 qtf = ld_transposed_frag(Qt, pitch, col, qt * 32 + 16 * mm + 4 * h2, false);    // Q[q slots][d]
        dv = mfma32(pf, dotf, dv);   // dV[key][d]
        dk = mfma32(dsf, qtf, dk);   // dK[key][d]
      }
    }
    // D[i = key][j = d]: lane column = d (valid < 16), regs = keys crow(r,h2)
    if (col < 16) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kk = kt * 32 + crow(r, h2);
        if (kk < L) {
          dqkv[qkv_off(a + kk, 1, H, head) + col] = (uint16_t)(pack_bf16x2(dk[r] * scale, 0.f) & 0xffffu);
          dqkv[qkv_off(a + kk, 2, H, head) + col] = (uint16_t)(pack_bf16x2(dv[r], 0.f) & 0xffffu);
        }
      }
    }
  }
}

// ================================================================================================
// host side
// ================================================================================================
// dynamic LDS above 64 KB has to be opted into once per kernel
template <typename K>
static int allow_big_lds(K kernel, size_t bytes) {
  PTC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return PTC_OK;
}

static size_t fwd_lds_bytes(int lp_max) { return (size_t)lp_max * 32 + (size_t)16 * (lp_max + 8) * 2; }
static size_t dkv_lds_bytes(int lp_max) { return (size_t)32 * (lp_max + 8) * 2 + (size_t)2 * lp_max * 4; }

static int check_common(const char* name, const void* qkv, const int32_t* cu, int64_t n_seq, int64_t total, int H,
                        int max_seqlen, int dtype) {
  PTC_REQUIRE(dtype == PTC_BF16, PTC_EUNSUPPORTED, "%s: only bf16 is implemented (the reference casts qkv to bf16, ptv3m1:209)", name);
  PTC_REQUIRE(n_seq >= 0 && total >= 0 && H >= 1, PTC_EINVAL, "%s: bad sizes", name);
  PTC_REQUIRE(max_seqlen >= 1 && max_seqlen <= AT_MAX_L, PTC_EUNSUPPORTED, "%s: max_seqlen=%d not in [1,%d]", name, max_seqlen, AT_MAX_L);
  PTC_REQUIRE(n_seq * H < (1ll << 31), PTC_EUNSUPPORTED, "%s: grid too large", name);
  PTC_REQUIRE(n_seq == 0 || (qkv && cu), PTC_EINVAL, "%s: null buffer", name);
  PTC_REQUIRE((uintptr_t)qkv % 16 == 0, PTC_EINVAL, "%s: qkv must be 16-byte aligned", name);
  return PTC_OK;
}

extern "C" int ptc_attn_varlen_fwd(const void* qkv, const int32_t* cu_seqlens, int64_t n_seq, int64_t total, int H,
                                   int max_seqlen, float softmax_scale, int dtype, void* out, float* lse,
                                   ptc_stream_t stream) {
  int rc = check_common("ptc_attn_varlen_fwd", qkv, cu_seqlens, n_seq, total, H, max_seqlen, dtype);
  if (rc != PTC_OK) return rc;
  if (n_seq == 0 || total == 0) return PTC_OK;
  PTC_REQUIRE(out && lse, PTC_EINVAL, "ptc_attn_varlen_fwd: null buffer");
  const int lp_max = (max_seqlen + 31) & ~31;
  const size_t lds = fwd_lds_bytes(lp_max);
  hipStream_t s = (hipStream_t)stream;
  rc = allow_big_lds(attn_fwd_kernel, lds);
  if (rc != PTC_OK) return rc;
  const int n_units = (int)(n_seq * H);
  hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)(8 * ((n_units + 7) / 8))), dim3(AT_THREADS), lds, s, (const uint16_t*)qkv,
                     cu_seqlens, H, softmax_scale, total, lp_max, n_units, (uint16_t*)out, lse);
  PTC_CHECK_LAUNCH("attn_fwd_kernel");
  return PTC_OK;
}

extern "C" size_t ptc_attn_varlen_bwd_workspace_bytes(int64_t total, int H) {
  return ptc_align_up((size_t)(total > 0 ? total : 1) * (size_t)H * sizeof(float), 256);
}

extern "C" int ptc_attn_varlen_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                                   const int32_t* cu_seqlens, int64_t n_seq, int64_t total, int H, int max_seqlen,
                                   float softmax_scale, int dtype, void* dqkv, void* workspace, size_t workspace_bytes,
                                   ptc_stream_t stream) {
  int rc = check_common("ptc_attn_varlen_bwd", qkv, cu_seqlens, n_seq, total, H, max_seqlen, dtype);
  if (rc != PTC_OK) return rc;
  if (n_seq == 0 || total == 0) return PTC_OK;
  PTC_REQUIRE(out && dout && lse && dqkv && workspace, PTC_EINVAL, "ptc_attn_varlen_bwd: null buffer");
  PTC_REQUIRE(workspace_bytes >= ptc_attn_varlen_bwd_workspace_bytes(total, H), PTC_EWORKSPACE,
              "ptc_attn_varlen_bwd: workspace too small");
  PTC_REQUIRE(((uintptr_t)out % 16 == 0) && ((uintptr_t)dout % 16 == 0) && ((uintptr_t)dqkv % 16 == 0), PTC_EINVAL,
              "ptc_attn_varlen_bwd: buffers must be 16-byte aligned");
  const int lp_max = (max_seqlen + 31) & ~31;
  hipStream_t s = (hipStream_t)stream;
  float* delta = (float*)workspace;
  rc = allow_big_lds(attn_bwd_dq_kernel, fwd_lds_bytes(lp_max));
  if (rc != PTC_OK) return rc;
  rc = allow_big_lds(attn_bwd_dkv_kernel, dkv_lds_bytes(lp_max));
  if (rc != PTC_OK) return rc;
  const int n_units = (int)(n_seq * H);
  const unsigned grid = (unsigned)(8 * ((n_units + 7) / 8));
  hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(grid), dim3(AT_THREADS), fwd_lds_bytes(lp_max), s,
                     (const uint16_t*)qkv, (const uint16_t*)out, (const uint16_t*)dout, lse, cu_seqlens, H,
                     softmax_scale, total, lp_max, n_units, (uint16_t*)dqkv, delta);
  PTC_CHECK_LAUNCH("attn_bwd_dq_kernel");
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(grid), dim3(AT_THREADS), dkv_lds_bytes(lp_max), s,
                     (const uint16_t*)qkv, (const uint16_t*)dout, lse, (const float*)delta, cu_seqlens, H,
                     softmax_scale, total, lp_max, n_units, (uint16_t*)dqkv);
  PTC_CHECK_LAUNCH("attn_bwd_dkv_kernel");
  return PTC_OK;
}
