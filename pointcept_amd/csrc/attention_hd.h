// attention_hd.h -- serialized patch attention for head_dim 17..64 (included by attention.hip).
//
// PT-v3m3 (Utonia) and LitePT pick head_dim = channels / heads = 18 so that the 3-D RoPE can give every axis a third of the
// channels (pointcept/models/point_transformer_v3/point_transformer_v3m3_utonia.py:43-51,183,354; pointcept/models/litept/
// litept_v1.py:236-251; configs/utonia/*: enc_channels (54,...,576) / enc_num_head (3,...,32)).  flash-attn pads such heads
// to the next multiple of 8 internally; here the head is cut into DK SLABS of 16 channels (zero padded beyond head_dim) and every
// slab is one more k-step of the same 32x32x16 products the head_dim-16 kernels use, on the same LDS images (one [Lp][16]
// row-major image per slab, same swizzle, same transposed reads).  What changes against attention.hip:
//   * S' = K (q c)^T - ref : 2 DK MFMAs (hi + lo part of the scaled query per slab)
//   * P V : the A operand carries 32 rows = 32 channels per M-block; the row of ones that yields the softmax denominator is row
//     head_dim, so heads below 32 channels still need ONE block, 32..63 two, 64 three (MB = head_dim / 32 + 1)
//   * transposed operands of the backward products address TWO slabs at once (lanes 0..15 of each half the even slab, lanes
//     16..31 the odd one: ds_read_b64_tr_b16 works per 16-lane group), so a 32-row operand is 32 real channels
//   * rows of qkv / out are head_dim * 2 bytes: 16-byte vector loads when head_dim % 8 == 0, 4-byte loads when it is even
//     (18), 2-byte loads otherwise
//   * no software pipelining by hand and one workgroup per CU (K and V^T of a 1024-key window with 32 padded channels take
//     131 KB): a first correct tier-1 path for the m3 / LitePT files, measured in tools/bench_ops.py attn_hd.
// LDS capacity bounds the window: head_dim <= 32 up to 1024 keys, <= 48 up to 672, <= 64 up to 512 (ptc_attn_varlen_hd_supported).

#define AH_LDS_LIMIT 163840

// ---- element traits: bf16 (the PT-v3 call sites cast their operands to bf16 themselves, ptv3m3:353) or f16 (LitePT hands flash-attn its
// fp16 autocast tensors, pointcept/models/litept/litept_v1.py:259-265: f16 operands, f16 P / dS, fp32 accumulation -- round 4; until then
// f16 operands were re-rounded to bf16, three mantissa bits short of the reference's arithmetic for that call site)
typedef _Float16 ah_f16x8 __attribute__((ext_vector_type(8)));
template <bool F16> struct AE;
template <> struct AE<false> {
  static constexpr uint16_t ONE = 0x3F80, NAN16 = 0x7FC0;
  static constexpr uint32_t NEG1X2 = 0xBF80BF80u;
  static __device__ __forceinline__ float lo(uint32_t w) { return __uint_as_float(w << 16); }
  static __device__ __forceinline__ float hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
  static __device__ __forceinline__ uint32_t pack2(float a, float b) { return pack_bf16x2(a, b); }
  static __device__ __forceinline__ f32x16 mfma(s16x8 a, s16x8 b, f32x16 c) { return mfma32(a, b, c); }
  static __device__ __forceinline__ float pad_lse() { return AT_PAD_LSE; }
};
template <> struct AE<true> {
  static constexpr uint16_t ONE = 0x3C00, NAN16 = 0x7E00;
  static constexpr uint32_t NEG1X2 = 0xBC00BC00u;
  static __device__ __forceinline__ float lo(uint32_t w) { at_h2 h; __builtin_memcpy(&h, &w, 4); return (float)h[0]; }
  static __device__ __forceinline__ float hi(uint32_t w) { at_h2 h; __builtin_memcpy(&h, &w, 4); return (float)h[1]; }
  static __device__ __forceinline__ uint32_t pack2(float a, float b) {
    const at_h2 h = {(_Float16)a, (_Float16)b};      // v_cvt_f16_f32, RNE
    uint32_t r;
    __builtin_memcpy(&r, &h, 4);
    return r;
  }
  static __device__ __forceinline__ f32x16 mfma(s16x8 a, s16x8 b, f32x16 c) {
    ah_f16x8 fa, fb;
    __builtin_memcpy(&fa, &a, 16);
    __builtin_memcpy(&fb, &b, 16);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, c, 0, 0, 0);
  }
  // lse of padding queries: the largest finite f16 keeps hi + lo finite (1e30 would be inf - inf); exp2(s - 60000) = 0 all the same
  static __device__ __forceinline__ float pad_lse() { return 60000.f; }
};
// x * c -> 16-bit hi + lo parts (hi + lo = x c to ~2^-17 (bf16) / 2^-22 (f16) relative)
template <bool F16>
__device__ __forceinline__ void ah_split_scaled(s16x8 x, float c, s16x8& hi, s16x8& lo) {
  uint32_t h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t w = (uint32_t)(uint16_t)x[2 * j] | ((uint32_t)(uint16_t)x[2 * j + 1] << 16);
    const float a = AE<F16>::lo(w) * c, b = AE<F16>::hi(w) * c;
    h[j] = AE<F16>::pack2(a, b);
    l[j] = AE<F16>::pack2(a - AE<F16>::lo(h[j]), b - AE<F16>::hi(h[j]));
  }
  hi = make_frag(h[0], h[1], h[2], h[3]);
  lo = make_frag(l[0], l[1], l[2], l[3]);
}

// 8 channels [ch0, ch0 + 8) of a row of `D` bf16 channels (zeros beyond D / for invalid rows)
__device__ __forceinline__ uint4 ah_ld8(const uint16_t* __restrict__ row, int ch0, int D, bool valid) {
  uint4 v = {0, 0, 0, 0};
  if (!valid || ch0 >= D) return v;
  if ((D & 7) == 0) return *reinterpret_cast<const uint4*>(row + ch0);
  if ((D & 1) == 0) {
    const uint32_t* p = reinterpret_cast<const uint32_t*>(row + ch0);
    v.x = p[0];
    if (ch0 + 2 < D) v.y = p[1];
    if (ch0 + 4 < D) v.z = p[2];
    if (ch0 + 6 < D) v.w = p[3];
    return v;
  }
  uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (ch0 + i < D) w[i >> 1] |= (uint32_t)row[ch0 + i] << (16 * (i & 1));
  v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
  return v;
}
__device__ __forceinline__ s16x8 ah_frag(uint4 u) { return *reinterpret_cast<s16x8*>(&u); }
template <bool F16>
__device__ __forceinline__ float ah_sumsq(uint4 v) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = AE<F16>::lo(w[j]), b = AE<F16>::hi(w[j]);
    ss = fmaf(a, a, fmaf(b, b, ss));
  }
  return ss;
}
template <bool F16>
__device__ __forceinline__ float ah_dot(uint4 x, uint4 y) {
  const uint32_t a[4] = {x.x, x.y, x.z, x.w}, b[4] = {y.x, y.y, y.z, y.w};
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    s = fmaf(AE<F16>::lo(a[j]), AE<F16>::lo(b[j]), s);
    s = fmaf(AE<F16>::hi(a[j]), AE<F16>::hi(b[j]), s);
  }
  return s;
}

// rows [0, Lp) of a [.., D] operand into DK slab images ([lp_max][16] each, rm_off swizzle); returns max |row|^2 of this thread
template <int DK, bool F16>
__device__ __forceinline__ float ah_stage_rows(const uint16_t* __restrict__ src, int64_t row_stride, int D, int L, int Lp,
                                               int slab_bytes, unsigned char* lds) {
  float mx = 0.f;
  for (int row = threadIdx.x; row < Lp; row += AT_THREADS) {
    const uint16_t* p = src + (int64_t)row * row_stride;
    const bool valid = row < L;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < DK; ++j) {
      const uint4 v0 = ah_ld8(p, 16 * j, D, valid), v1 = ah_ld8(p, 16 * j + 8, D, valid);
      *reinterpret_cast<uint4*>(lds + j * slab_bytes + rm_off(row, 0)) = v0;
      *reinterpret_cast<uint4*>(lds + j * slab_bytes + rm_off(row, 1)) = v1;
      ss += ah_sumsq<F16>(v0) + ah_sumsq<F16>(v1);
    }
    mx = fmaxf(mx, ss);
  }
  return mx;
}
// V^T image [D + 1][pitch] with the key permutation of stage_transposed; row D = 1.0 for keys < L (softmax denominator)
template <int DK, bool F16>
__device__ __forceinline__ void ah_stage_vt(const uint16_t* __restrict__ src, int64_t row_stride, int D, int L, int Lp, int pitch,
                                            unsigned char* lds) {
  uint32_t* t32 = reinterpret_cast<uint32_t*>(lds);
  for (int p = threadIdx.x; p < Lp / 2; p += AT_THREADS) {
    const int ra = 2 * p, rb = 2 * p + 1;
    const uint16_t* pa = src + (int64_t)ra * row_stride;
    const uint16_t* pb = src + (int64_t)rb * row_stride;
    const int w = vt_pos(ra) >> 1;
#pragma unroll
    for (int j = 0; j < DK; ++j) {
      uint16_t ea[16], eb[16];
      *reinterpret_cast<uint4*>(ea) = ah_ld8(pa, 16 * j, D, ra < L);
      *reinterpret_cast<uint4*>(ea + 8) = ah_ld8(pa, 16 * j + 8, D, ra < L);
      *reinterpret_cast<uint4*>(eb) = ah_ld8(pb, 16 * j, D, rb < L);
      *reinterpret_cast<uint4*>(eb + 8) = ah_ld8(pb, 16 * j + 8, D, rb < L);
#pragma unroll
      for (int d = 0; d < 16; ++d)
        if (16 * j + d < D) t32[((16 * j + d) * pitch) / 2 + w] = (uint32_t)ea[d] | ((uint32_t)eb[d] << 16);
    }
  }
  for (int key = threadIdx.x; key < Lp; key += AT_THREADS)
    reinterpret_cast<uint16_t*>(lds + (size_t)D * pitch * 2)[vt_pos(key)] = key < L ? AE<F16>::ONE : (uint16_t)0;
}
// NaN rows for units longer than max_seqlen (see at_poison_rows)
template <bool F16>
__device__ __forceinline__ void ah_poison_rows(uint16_t* rows, int64_t row_stride, int D, int L, float* side) {
  for (int q = threadIdx.x; q < L; q += AT_THREADS) {
    for (int d = 0; d < D; ++d) rows[(int64_t)q * row_stride + d] = AE<F16>::NAN16;
    if (side) side[q] = __uint_as_float(0x7FC00000u);
  }
}
// register r of lane half h2 holds row crow(r, h2): the (register, half) that hold row `row` of a 32-row block
__device__ __forceinline__ float ah_pick_row(const f32x16& acc, int row) {
  const int r = (row & 3) + 4 * (row >> 3);
  float v = acc[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) v = (i == r) ? acc[i] : v;
  return v;
}


// ---- 3-D rotary embedding in the attention PROLOGUE / EPILOGUE (round 4; head_dim 18 = every PT-v3m3 / LitePT config) -------------
// The rotation of q and k that the reference applies between its qkv Linear and flash-attn (point_transformer_v3m3_utonia.py:58-101,
// 303-323; libs/pointrope/kernels.cu:19-75 behind litept_v1.py:239-241): per axis a and frequency i the pair
// (x[6 a + i], x[6 a + 3 + i]) <- (u cos f - v sin f, v cos f + u sin f), f = pos[a] inv_freq[i], computed in fp32 and rounded to the
// operand dtype.  It used to be its own pass over the packed [n, 3, H, 18] rows (ptc_rope3d_xyz: one read + one write of q and k, and
// the same again for the gradient); ROPE = true kernels rotate K while they stage it, q when they load it, and turn the gradients
// of the rotated q / k back (the inverse rotation: sign = -1) in their epilogue -- through a wave-private LDS tile, because a row's
// pairs sit in different lanes of the accumulator layout; a lane then owns a whole 36-byte row and stores it in one piece.
#define AH_RD 18                     // head_dim of the fused rotation
#define AH_RW 9                      // 32-bit words per row
struct AhRope { const float* xyz; const float* inv_freq; };   // positions [total, 3] fp32 of the padded, serialized rows; frequencies [3]
// the 9 words of row `row` (global row index t for its position), rotated by sign * angle; invalid rows = zeros
template <bool F16>
__device__ __forceinline__ void ah_rope_row(const uint32_t* __restrict__ src, bool valid, const AhRope& rp, int64_t t, float sign,
                                            uint32_t (&w)[AH_RW]) {
#pragma unroll
  for (int i = 0; i < AH_RW; ++i) w[i] = valid ? src[i] : 0u;
  if (!valid) return;
  float x[AH_RD];
#pragma unroll
  for (int i = 0; i < AH_RW; ++i) { x[2 * i] = AE<F16>::lo(w[i]); x[2 * i + 1] = AE<F16>::hi(w[i]); }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float pos = rp.xyz[t * 3 + a];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float sn, cs;
      sincosf(pos * rp.inv_freq[i], &sn, &cs);
      sn *= sign;
      const float u = x[6 * a + i], v = x[6 * a + 3 + i];
      ptc_rope_pair(u, v, cs, sn, x[6 * a + i], x[6 * a + 3 + i]);
    }
  }
#pragma unroll
  for (int i = 0; i < AH_RW; ++i) w[i] = AE<F16>::pack2(x[2 * i], x[2 * i + 1]);
}
// the 8-channel chunk [ch0, ch0 + 8) of a rotated row (ch0 in {0, 8, 16, 24})
__device__ __forceinline__ uint4 ah_rope_chunk(const uint32_t (&w)[AH_RW], int ch0) {
  uint4 u = {0, 0, 0, 0};
  if (ch0 == 0) u = make_uint4(w[0], w[1], w[2], w[3]);
  else if (ch0 == 8) u = make_uint4(w[4], w[5], w[6], w[7]);
  else if (ch0 == 16) u = make_uint4(w[8], 0, 0, 0);
  return u;
}
// rows [0, Lp) of K (or Q) rotated, into the two slab images; returns this thread's max |row|^2
template <bool F16>
__device__ __forceinline__ float ah_stage_rows_rope(const uint16_t* __restrict__ src, int64_t row_stride, int64_t t0, const AhRope& rp, int L, int Lp,
                                                    int slab_bytes, unsigned char* lds) {
  float mx = 0.f;
  for (int row = threadIdx.x; row < Lp; row += AT_THREADS) {
    uint32_t w[AH_RW];
    ah_rope_row<F16>(reinterpret_cast<const uint32_t*>(src + (int64_t)row * row_stride), row < L, rp, t0 + row, 1.f, w);
    const uint4 c0 = ah_rope_chunk(w, 0), c1 = ah_rope_chunk(w, 8), c2 = ah_rope_chunk(w, 16), z = {0, 0, 0, 0};
    *reinterpret_cast<uint4*>(lds + rm_off(row, 0)) = c0;
    *reinterpret_cast<uint4*>(lds + rm_off(row, 1)) = c1;
    *reinterpret_cast<uint4*>(lds + slab_bytes + rm_off(row, 0)) = c2;
    *reinterpret_cast<uint4*>(lds + slab_bytes + rm_off(row, 1)) = z;
    mx = fmaxf(mx, ah_sumsq<F16>(c0) + ah_sumsq<F16>(c1) + ah_sumsq<F16>(c2));
  }
  return mx;
}
// gradient rows of a 32-row tile, sitting 16-bit rounded in the wave's LDS tile [32][AH_RD]: lane r < 32 turns row r back and stores it
template <bool F16>
__device__ __forceinline__ void ah_unrope_store(const uint16_t* tile, int lane, int row0, int L, const AhRope& rp, int64_t t0, float scale,
                                                uint16_t* __restrict__ dst0, int64_t row_stride) {
  const int row = row0 + (lane & 31);
  if (lane < 32 && row < L) {
    uint32_t w[AH_RW];
    ah_rope_row<F16>(reinterpret_cast<const uint32_t*>(tile + (lane & 31) * AH_RD), true, rp, t0 + row, -1.f, w);
    uint32_t* o = reinterpret_cast<uint32_t*>(dst0 + (int64_t)row * row_stride);
#pragma unroll
    for (int i = 0; i < AH_RW; ++i) o[i] = w[i];
  }
  (void)scale;
}

#ifndef AH_FWD_PIPE
#define AH_FWD_PIPE 1            // 0: the plain in-order key loop of rounds 2-4; 1: software pipeline, the compiler's schedule; 2: pinned with
                                 // sched_group_barrier (timing A/B: `python -m pointcept_amd.build --variant d_AH_FWD_PIPE_2`)
#endif
// ------------------------------------------------------------------------------------------------ forward
// LDS: K slabs DK x [lp_max][16] | V^T [D + 1][pitch] | AT_WAVES floats
template <int DK, int MB, bool F16, bool ROPE = false>
__global__ void __launch_bounds__(AT_THREADS, 2)
attn_hd_fwd_kernel(const uint16_t* __restrict__ qkv, const int32_t* __restrict__ cu, int H, int D, float scale, int64_t total,
                   int lp_max, int n_units, int qs, uint16_t* __restrict__ out, float* __restrict__ lse, AhRope rope = AhRope{nullptr, nullptr}) {
  static_assert(!ROPE || (DK == 2 && MB == 1), "the fused rotation is built for head_dim 18");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lunit = at_unit(n_units * qs);
  if (lunit >= n_units * qs) return;
  const int unit = lunit / qs, part = lunit - unit * qs;
  const int seq = unit / H, head = unit % H;
  const int a = cu[seq], L = cu[seq + 1] - a;
  if (L <= 0) return;
  const int Lp = (L + 31) & ~31, n_tiles = Lp >> 5;
  const int64_t rs = (int64_t)3 * H * D, os = (int64_t)H * D;
  if (Lp > lp_max) {
    if (part == 0) ah_poison_rows<F16>(out + ((int64_t)a * H + head) * D, os, D, L, lse + (int64_t)head * total + a);
    return;
  }
  const int t_per = (n_tiles + qs - 1) / qs, t_lo = part * t_per, t_hi = (t_lo + t_per) < n_tiles ? (t_lo + t_per) : n_tiles;
  if (t_lo >= n_tiles) return;
  const int pitch = lp_max + 8, slab = lp_max * 32;
  unsigned char* Ksm = smem;
  unsigned char* Vt = smem + (size_t)DK * slab;
  float* red = reinterpret_cast<float*>(Vt + (size_t)(D + 1) * pitch * 2);
  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  const uint16_t* qbase = qkv + ((int64_t)a * 3 * H + head) * D;            // q row t: qbase + t * rs; k: + H*D; v: + 2*H*D
  float kn;
  if constexpr (ROPE) kn = ah_stage_rows_rope<F16>(qbase + (int64_t)H * D, rs, a, rope, L, Lp, slab, Ksm);
  else kn = ah_stage_rows<DK, F16>(qbase + (int64_t)H * D, rs, D, L, Lp, slab, Ksm);
  ah_stage_vt<DK, F16>(qbase + (int64_t)2 * H * D, rs, D, L, Lp, pitch, Vt);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) kn = fmaxf(kn, __shfl_xor(kn, o, 64));
  if (lane == 0) red[wave] = kn;
  __syncthreads();
  float kmax2 = red[0];
#pragma unroll
  for (int w = 1; w < AT_WAVES; ++w) kmax2 = fmaxf(kmax2, red[w]);

  const int col = lane & 31, h2 = lane >> 5;
  const float c = scale * AT_LOG2E;
  const unsigned char* vbase[MB];
#pragma unroll
  for (int m = 0; m < MB; ++m) {
    const int row = 32 * m + col;
    vbase[m] = Vt + ((size_t)(row <= D ? row : (col & 15)) * pitch + 8 * h2) * 2;   // rows > D feed outputs nobody reads
  }
  const unsigned char* kbase = Ksm + rm_off(col, h2);
  const int den_row = D & 31;                                                        // block MB - 1 holds the denominator row

  for (int qt = t_lo + wave; qt < t_hi; qt += AT_WAVES) {
    const int q = qt * 32 + col;
    const uint16_t* qrow = qbase + (int64_t)q * rs;
    s16x8 qhi[DK], qlo[DK];
    float qn = 0.f;
    uint32_t qw[ROPE ? AH_RW : 1];
    if constexpr (ROPE) ah_rope_row<F16>(reinterpret_cast<const uint32_t*>(qrow), q < L, rope, (int64_t)a + q, 1.f, qw);
#pragma unroll
    for (int j = 0; j < DK; ++j) {
      uint4 u;
      if constexpr (ROPE) u = ah_rope_chunk(qw, 16 * j + 8 * h2);
      else u = ah_ld8(qrow, 16 * j + 8 * h2, D, q < L);
      qn += ah_sumsq<F16>(u);
      ah_split_scaled<F16>(ah_frag(u), c, qhi[j], qlo[j]);
    }
    qn += __shfl_xor(qn, 32, 64);
    const float bnd = sqrtf(qn * kmax2) * c * 1.0009765625f + 1e-3f;
    f32x16 acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[m] = zero16();
    float ref2;
    // (f16 operands: P = exp2(S' - ref) is rounded to f16, whose exponent range ends at 2^-24 -- a Cauchy-Schwarz reference up to 64 above the
    //  row's true maximum would flush the whole row to zero; f16 always takes the running-maximum loop, P <= 1 with its largest entry = 1)
    if (!F16 && __builtin_amdgcn_ballot_w64(bnd > AT_FIXED_REF_MAX) == 0) {
      ref2 = bnd;
      const f32x16 negb = splat16(-bnd);
      if constexpr (AH_FWD_PIPE != 0 && DK == 2 && MB == 1) {      // head_dim 17 .. 31 (the reference's 18); wider heads: measured slower, below
      // Three-stage software pipeline over the key tiles (round 5; the head_dim-16 forward's principle, attention.hip): the S' products of
      // tile kt + 1 sit in the matrix pipe while the vector pipe exponentiates tile kt, and P V of tile kt - 1 follows -- no MFMA of a
      // trip depends on the trip's own vector work.  A wave is in-order and this kernel runs two waves per SIMD: in the plain loop
      // (S' -> exp -> pack -> P V, each waiting for the one before) the 2 DK + 2 MB MFMAs and the 16 transcendentals of a tile never
      // overlapped.  Same products, same accumulation order per accumulator: bit-identical to the plain loop (AH_FWD_PIPE 0).
      // S' of one tile past the end is computed and never used (its K reads land in the next slab / the V^T image: in-bounds LDS).
      auto s_tile = [&](int kt) {
        f32x16 sv = negb;
#pragma unroll
        for (int j = 0; j < DK; ++j) {
          const s16x8 kf = *reinterpret_cast<const s16x8*>(kbase + j * slab + kt * 1024);
          sv = AE<F16>::mfma(kf, qhi[j], sv);
          sv = AE<F16>::mfma(kf, qlo[j], sv);
        }
        return sv;
      };
      auto expo = [&](const f32x16& sv, uint32_t (&pk)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) pk[i] = AE<F16>::pack2(__builtin_amdgcn_exp2f(sv[2 * i]), __builtin_amdgcn_exp2f(sv[2 * i + 1]));
      };
      auto pv = [&](int kt, const uint32_t (&pk)[8]) {
        const s16x8 p0 = make_frag(pk[0], pk[1], pk[2], pk[3]), p1 = make_frag(pk[4], pk[5], pk[6], pk[7]);
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          acc[m] = AE<F16>::mfma(*reinterpret_cast<const s16x8*>(vbase[m] + kt * 64), p0, acc[m]);
          acc[m] = AE<F16>::mfma(*reinterpret_cast<const s16x8*>(vbase[m] + kt * 64 + 32), p1, acc[m]);
        }
      };
      // pin the interleave of a half trip (6 MFMAs, 16 v_exp_f32, 8 packs; the compiler's own schedule keeps the stages in source order):
      // the four LDS reads first, then behind every MFMA three transcendentals and one or two plain vector instructions
      auto interleave = [&]() {
#if AH_FWD_PIPE >= 2
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#define AH_GRP(T, V) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x400, T, 0); \
                     __builtin_amdgcn_sched_group_barrier(0x002, V, 0);
        AH_GRP(3, 2) AH_GRP(3, 2) AH_GRP(3, 1) AH_GRP(3, 1) AH_GRP(2, 1) AH_GRP(2, 1)
#undef AH_GRP
#endif
      };
      const int kt_last = n_tiles - 1;
      f32x16 s0 = s_tile(0), s1 = s_tile(kt_last > 0 ? 1 : 0);
      uint32_t pa[8], pb[8];
      expo(s0, pa);
      int kt = 1;
      for (; kt + 1 < n_tiles; kt += 2) {               // at the top: s1 = S'(kt), pa = P(kt - 1), P V done for tiles < kt - 1
        s0 = s_tile(kt + 1);
        expo(s1, pb);
        pv(kt - 1, pa);
        interleave();
        s1 = s_tile(kt + 2 <= kt_last ? kt + 2 : kt_last);
        expo(s0, pa);
        pv(kt, pb);
        interleave();
      }
      if (kt < n_tiles) {                               // one tile left in s1, P(kt - 1) in pa
        expo(s1, pb);
        pv(kt - 1, pa);
        pv(kt, pb);
      } else {
        pv(kt - 1, pa);
      }
      } else {
      for (int kt = 0; kt < n_tiles; ++kt) {
        f32x16 s = negb;
#pragma unroll
        for (int j = 0; j < DK; ++j) {
          const s16x8 kf = *reinterpret_cast<const s16x8*>(kbase + j * slab + kt * 1024);
          s = AE<F16>::mfma(kf, qhi[j], s);
          s = AE<F16>::mfma(kf, qlo[j], s);
        }
        uint32_t pk[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) pk[i] = AE<F16>::pack2(__builtin_amdgcn_exp2f(s[2 * i]), __builtin_amdgcn_exp2f(s[2 * i + 1]));
        const s16x8 p0 = make_frag(pk[0], pk[1], pk[2], pk[3]), p1 = make_frag(pk[4], pk[5], pk[6], pk[7]);
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          acc[m] = AE<F16>::mfma(*reinterpret_cast<const s16x8*>(vbase[m] + kt * 64), p0, acc[m]);
          acc[m] = AE<F16>::mfma(*reinterpret_cast<const s16x8*>(vbase[m] + kt * 64 + 32), p1, acc[m]);
        }
      }
      }
    } else {
      float mrun = -INFINITY;
      for (int kt = 0; kt < n_tiles; ++kt) {
        f32x16 s = zero16();
#pragma unroll
        for (int j = 0; j < DK; ++j) {
          const s16x8 kf = *reinterpret_cast<const s16x8*>(kbase + j * slab + kt * 1024);
          s = AE<F16>::mfma(kf, qhi[j], s);
          s = AE<F16>::mfma(kf, qlo[j], s);
        }
        if (kt == n_tiles - 1 && L < Lp) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (kt * 32 + crow(r, h2) >= L) s[r] = -INFINITY;
        }
        float mt = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(mrun, mt);
        const float alpha = __builtin_amdgcn_exp2f(mrun - m_new);
        mrun = m_new;
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[m][r] *= alpha;
        uint32_t pk[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          pk[i] = AE<F16>::pack2(__builtin_amdgcn_exp2f(s[2 * i] - mrun), __builtin_amdgcn_exp2f(s[2 * i + 1] - mrun));
        const s16x8 p0 = make_frag(pk[0], pk[1], pk[2], pk[3]), p1 = make_frag(pk[4], pk[5], pk[6], pk[7]);
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          acc[m] = AE<F16>::mfma(*reinterpret_cast<const s16x8*>(vbase[m] + kt * 64), p0, acc[m]);
          acc[m] = AE<F16>::mfma(*reinterpret_cast<const s16x8*>(vbase[m] + kt * 64 + 32), p1, acc[m]);
        }
      }
      ref2 = mrun;
    }
    const float l = __shfl(ah_pick_row(acc[MB - 1], den_row), col + 32 * ((den_row >> 2) & 1), 64);
    const float inv = 1.f / l;
    if (q < L) {
      uint16_t* o = out + ((int64_t)(a + q) * H + head) * D;
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {                       // registers r, r + 1 = channels ch, ch + 1 (ch even)
          const int ch = 32 * m + crow(r, h2);
          const uint32_t w = AE<F16>::pack2(acc[m][r] * inv, acc[m][r + 1] * inv);
          if (ch + 1 < D && (D & 1) == 0) *reinterpret_cast<uint32_t*>(o + ch) = w;
          else {
            if (ch < D) o[ch] = (uint16_t)(w & 0xffffu);
            if (ch + 1 < D) o[ch + 1] = (uint16_t)(w >> 16);
          }
        }
      if (h2 == 0) lse[(int64_t)head * total + a + q] = ref2 * AT_LN2 + __logf(l);
    }
  }
}

// transposed 32-row operand over a slab PAIR: lanes 0..15 of each half address slab 2m, lanes 16..31 slab 2m + 1
// (slab 2m again when the pair is incomplete: those rows only reach outputs at channels >= 16 DK, never written)
template <int DK>
__device__ __forceinline__ int ah_pair_off(int m, int col, int slab_bytes) {
  const int j = (col >= 16 && 2 * m + 1 < DK) ? 2 * m + 1 : 2 * m;
  return j * slab_bytes;
}
template <bool F16>
__device__ __forceinline__ void ah_store_col(uint16_t* row, int ch, int D, float v) {
  if (ch < D) row[ch] = (uint16_t)(AE<F16>::pack2(v, 0.f) & 0xffffu);
}

// ------------------------------------------------------------------------------------------------ backward: dQ + delta
// LDS: V slabs | K slabs
template <int DK, bool F16, bool ROPE = false>
__global__ void __launch_bounds__(AT_THREADS, 2)
attn_hd_bwd_dq_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ out, const uint16_t* __restrict__ dout,
                      const float* __restrict__ lse, const int32_t* __restrict__ cu, int H, int D, float scale, int64_t total,
                      int lp_max, int n_units, int qs, uint16_t* __restrict__ dqkv, float* __restrict__ delta,
                      AhRope rope = AhRope{nullptr, nullptr}) {
  static_assert(!ROPE || DK == 2, "the fused rotation is built for head_dim 18");
  constexpr int MP = (DK + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lunit = at_unit(n_units * qs);
  if (lunit >= n_units * qs) return;
  const int unit = lunit / qs, part = lunit - unit * qs;
  const int seq = unit / H, head = unit % H;
  const int a = cu[seq], L = cu[seq + 1] - a;
  if (L <= 0) return;
  const int Lp = (L + 31) & ~31, n_tiles = Lp >> 5;
  const int64_t rs = (int64_t)3 * H * D;
  uint16_t* dqbase = dqkv + ((int64_t)a * 3 * H + head) * D;
  if (Lp > lp_max) {
    if (part == 0) ah_poison_rows<F16>(dqbase, rs, D, L, nullptr);
    return;
  }
  const int t_per = (n_tiles + qs - 1) / qs, t_lo = part * t_per, t_hi = (t_lo + t_per) < n_tiles ? (t_lo + t_per) : n_tiles;
  if (t_lo >= n_tiles) return;
  const int slab = lp_max * 32;
  unsigned char* Vsm = smem;
  unsigned char* Ksm = smem + (size_t)DK * slab;
  const uint16_t* qbase = qkv + ((int64_t)a * 3 * H + head) * D;
  ah_stage_rows<DK, F16>(qbase + (int64_t)2 * H * D, rs, D, L, Lp, slab, Vsm);
  if constexpr (ROPE) ah_stage_rows_rope<F16>(qbase + (int64_t)H * D, rs, a, rope, L, Lp, slab, Ksm);
  else ah_stage_rows<DK, F16>(qbase + (int64_t)H * D, rs, D, L, Lp, slab, Ksm);
  uint16_t* rtile = reinterpret_cast<uint16_t*>(smem + (size_t)2 * DK * slab) + (threadIdx.x >> 6) * 32 * AH_RD;   // ROPE: this wave's [32][18] tile
  __syncthreads();

  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  const int col = lane & 31, h2 = lane >> 5;
  const float c = scale * AT_LOG2E;
  const TrAddr ta = tr_addr(lane);
  const int rmo = rm_off(col, h2);
  int poff[MP];
#pragma unroll
  for (int m = 0; m < MP; ++m) poff[m] = ah_pair_off<DK>(m, col, slab);

  for (int qt = t_lo + wave; qt < t_hi; qt += AT_WAVES) {
    const int q = qt * 32 + col;
    const bool qv = q < L;
    const uint16_t* qrow = qbase + (int64_t)q * rs;
    const int64_t orow = ((int64_t)(a + q) * H + head) * D;
    s16x8 qhi[DK], qlo[DK], dof[DK];
    float dl = 0.f;
    uint32_t qw[ROPE ? AH_RW : 1];
    if constexpr (ROPE) ah_rope_row<F16>(reinterpret_cast<const uint32_t*>(qrow), qv, rope, (int64_t)a + q, 1.f, qw);
#pragma unroll
    for (int j = 0; j < DK; ++j) {
      uint4 uq;
      if constexpr (ROPE) uq = ah_rope_chunk(qw, 16 * j + 8 * h2);
      else uq = ah_ld8(qrow, 16 * j + 8 * h2, D, qv);
      const uint4 ud = ah_ld8(dout + orow, 16 * j + 8 * h2, D, qv);
      const uint4 uo = ah_ld8(out + orow, 16 * j + 8 * h2, D, qv);
      dl += ah_dot<F16>(ud, uo);
      dof[j] = ah_frag(ud);
      ah_split_scaled<F16>(ah_frag(uq), c, qhi[j], qlo[j]);
    }
    dl += __shfl_xor(dl, 32, 64);
    const float l2 = qv ? lse[(int64_t)head * total + a + q] * AT_LOG2E : INFINITY;
    if (qv && h2 == 0) delta[(int64_t)head * total + a + q] = dl;
    const f32x16 negl = splat16(-l2), negd = splat16(-dl);
    f32x16 acc[MP];
#pragma unroll
    for (int m = 0; m < MP; ++m) acc[m] = zero16();
    for (int kt = 0; kt < n_tiles; ++kt) {
      f32x16 s = negl, dp = negd;
#pragma unroll
      for (int j = 0; j < DK; ++j) {
        const s16x8 kf = *reinterpret_cast<const s16x8*>(Ksm + j * slab + kt * 1024 + rmo);
        const s16x8 vf = *reinterpret_cast<const s16x8*>(Vsm + j * slab + kt * 1024 + rmo);
        s = AE<F16>::mfma(kf, qhi[j], s);
        s = AE<F16>::mfma(kf, qlo[j], s);
        dp = AE<F16>::mfma(vf, dof[j], dp);
      }
      uint32_t pk[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        pk[i] = AE<F16>::pack2(__builtin_amdgcn_exp2f(s[2 * i]) * dp[2 * i], __builtin_amdgcn_exp2f(s[2 * i + 1]) * dp[2 * i + 1]);
#pragma unroll
      for (int mm = 0; mm < 2; ++mm) {
        const s16x8 dsf = make_frag(pk[4 * mm], pk[4 * mm + 1], pk[4 * mm + 2], pk[4 * mm + 3]);
#pragma unroll
        for (int m = 0; m < MP; ++m) acc[m] = AE<F16>::mfma(ld_tr_frag(Ksm + poff[m], ta, kt * 32 + 16 * mm), dsf, acc[m]);   // dQ^T[d][q]
      }
    }
    if constexpr (ROPE) {
      // gradient of the ROTATED q, rounded as flash-attn returns it, into the wave's tile [query][channel]; then a lane per row turns it back
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ch = crow(r, h2);
        if (ch < AH_RD) rtile[col * AH_RD + ch] = (uint16_t)(AE<F16>::pack2(acc[0][r] * scale, 0.f) & 0xffffu);
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): the tile is written
      __builtin_amdgcn_wave_barrier();
      ah_unrope_store<F16>(rtile, lane, qt * 32, L, rope, a, 1.f, dqbase, rs);
      __builtin_amdgcn_wave_barrier();
    } else if (qv) {
      uint16_t* o = dqbase + (int64_t)q * rs;
#pragma unroll
      for (int m = 0; m < MP; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) ah_store_col<F16>(o, 32 * m + crow(r, h2), D, acc[m][r] * scale);
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
// LDS: Q slabs | dO slabs | aux [lp_max][4] bf16 (lse_hi, lse_lo, delta_hi, delta_lo)
template <int DK, bool F16, bool ROPE = false>
__global__ void __launch_bounds__(AT_THREADS, 2)
attn_hd_bwd_dkv_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dout, const float* __restrict__ lse,
                       const float* __restrict__ delta, const int32_t* __restrict__ cu, int H, int D, float scale, int64_t total,
                       int lp_max, int n_units, int qs, uint16_t* __restrict__ dqkv, AhRope rope = AhRope{nullptr, nullptr}) {
  static_assert(!ROPE || DK == 2, "the fused rotation is built for head_dim 18");
  constexpr int MP = (DK + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lunit = at_unit(n_units * qs);
  if (lunit >= n_units * qs) return;
  const int unit = lunit / qs, part = lunit - unit * qs;
  const int seq = unit / H, head = unit % H;
  const int a = cu[seq], L = cu[seq + 1] - a;
  if (L <= 0) return;
  const int Lp = (L + 31) & ~31, n_tiles = Lp >> 5;
  const int64_t rs = (int64_t)3 * H * D, os = (int64_t)H * D;
  uint16_t* dbase = dqkv + ((int64_t)a * 3 * H + head) * D;
  if (Lp > lp_max) {
    if (part == 0) {
      ah_poison_rows<F16>(dbase + (int64_t)H * D, rs, D, L, nullptr);
      ah_poison_rows<F16>(dbase + (int64_t)2 * H * D, rs, D, L, nullptr);
    }
    return;
  }
  const int t_per = (n_tiles + qs - 1) / qs, t_lo = part * t_per, t_hi = (t_lo + t_per) < n_tiles ? (t_lo + t_per) : n_tiles;
  if (t_lo >= n_tiles) return;
  const int slab = lp_max * 32;
  unsigned char* Qsm = smem;
  unsigned char* dOsm = smem + (size_t)DK * slab;
  uint2* aux = reinterpret_cast<uint2*>(smem + (size_t)2 * DK * slab);
  const uint16_t* qbase = qkv + ((int64_t)a * 3 * H + head) * D;
  if constexpr (ROPE) ah_stage_rows_rope<F16>(qbase, rs, a, rope, L, Lp, slab, Qsm);
  else ah_stage_rows<DK, F16>(qbase, rs, D, L, Lp, slab, Qsm);
  ah_stage_rows<DK, F16>(dout + ((int64_t)a * H + head) * D, os, D, L, Lp, slab, dOsm);
  uint16_t* rtile = reinterpret_cast<uint16_t*>(smem + (size_t)2 * DK * slab + (size_t)lp_max * 8) + (threadIdx.x >> 6) * 32 * AH_RD;   // ROPE: this wave's tile
  for (int q = threadIdx.x; q < Lp; q += AT_THREADS) {
    const float l2 = q < L ? lse[(int64_t)head * total + a + q] * AT_LOG2E : AE<F16>::pad_lse();
    const float dl = q < L ? delta[(int64_t)head * total + a + q] : 0.f;
    const uint32_t hi = AE<F16>::pack2(l2, dl);
    const uint32_t lo = AE<F16>::pack2(l2 - AE<F16>::lo(hi), dl - AE<F16>::hi(hi));
    uint2 w;
    w.x = (hi & 0xffffu) | (lo << 16);
    w.y = (hi >> 16) | (lo & 0xffff0000u);
    aux[q] = w;
  }
  __syncthreads();

  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  const int col = lane & 31, h2 = lane >> 5;
  const float c = scale * AT_LOG2E;
  const TrAddr ta = tr_addr(lane);
  const int rmo = rm_off(col, h2);
  const uint32_t m1 = AE<F16>::NEG1X2;
  const s16x8 bS = make_frag(h2 == 0 ? m1 : 0u, 0u, 0u, 0u);
  const s16x8 bD = make_frag(0u, h2 == 0 ? m1 : 0u, 0u, 0u);
  int poff[MP];
#pragma unroll
  for (int m = 0; m < MP; ++m) poff[m] = ah_pair_off<DK>(m, col, slab);

  for (int kt = t_lo + wave; kt < t_hi; kt += AT_WAVES) {
    const int key = kt * 32 + col;
    const uint16_t* krow = qbase + (int64_t)key * rs + (int64_t)H * D;
    s16x8 khi[DK], klo[DK], vf[DK];
    uint32_t kw[ROPE ? AH_RW : 1];
    if constexpr (ROPE) ah_rope_row<F16>(reinterpret_cast<const uint32_t*>(krow), key < L, rope, (int64_t)a + key, 1.f, kw);
#pragma unroll
    for (int j = 0; j < DK; ++j) {
      if constexpr (ROPE) ah_split_scaled<F16>(ah_frag(ah_rope_chunk(kw, 16 * j + 8 * h2)), c, khi[j], klo[j]);
      else ah_split_scaled<F16>(ah_frag(ah_ld8(krow, 16 * j + 8 * h2, D, key < L)), c, khi[j], klo[j]);
      vf[j] = ah_frag(ah_ld8(krow + (int64_t)H * D, 16 * j + 8 * h2, D, key < L));
    }
    f32x16 dv[MP], dk[MP];
#pragma unroll
    for (int m = 0; m < MP; ++m) { dv[m] = zero16(); dk[m] = zero16(); }
    for (int qt = 0; qt < n_tiles; ++qt) {
      const uint2 ax = aux[qt * 32 + col];
      const s16x8 af = make_frag(ax.x, ax.y, 0u, 0u);
      f32x16 s = AE<F16>::mfma(af, bS, zero16());
      f32x16 dp = AE<F16>::mfma(af, bD, zero16());
#pragma unroll
      for (int j = 0; j < DK; ++j) {
        const s16x8 qf = *reinterpret_cast<const s16x8*>(Qsm + j * slab + qt * 1024 + rmo);
        const s16x8 dof = *reinterpret_cast<const s16x8*>(dOsm + j * slab + qt * 1024 + rmo);
        s = AE<F16>::mfma(qf, khi[j], s);
        s = AE<F16>::mfma(qf, klo[j], s);
        dp = AE<F16>::mfma(dof, vf[j], dp);
      }
      uint32_t pp[8], ps[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float p0 = __builtin_amdgcn_exp2f(s[2 * i]), p1 = __builtin_amdgcn_exp2f(s[2 * i + 1]);
        pp[i] = AE<F16>::pack2(p0, p1);
        ps[i] = AE<F16>::pack2(p0 * dp[2 * i], p1 * dp[2 * i + 1]);
      }
#pragma unroll
      for (int mm = 0; mm < 2; ++mm) {
        const s16x8 pf = make_frag(pp[4 * mm], pp[4 * mm + 1], pp[4 * mm + 2], pp[4 * mm + 3]);
        const s16x8 dsf = make_frag(ps[4 * mm], ps[4 * mm + 1], ps[4 * mm + 2], ps[4 * mm + 3]);
#pragma unroll
        for (int m = 0; m < MP; ++m) {
          dv[m] = AE<F16>::mfma(pf, ld_tr_frag(dOsm + poff[m], ta, qt * 32 + 16 * mm), dv[m]);   // dV[key][d]
          dk[m] = AE<F16>::mfma(dsf, ld_tr_frag(Qsm + poff[m], ta, qt * 32 + 16 * mm), dk[m]);   // dK[key][d]
        }
      }
    }
#pragma unroll
    for (int m = 0; m < MP; ++m) {
      const int ch = 32 * m + col;                                // D[i = key][j = d]: lane column = channel, registers = keys
      if (ch < D && (col < 16 || 2 * m + 1 < DK)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kk = kt * 32 + crow(r, h2);
          if (kk < L) {
            uint16_t* o = dbase + (int64_t)kk * rs;
            if constexpr (!ROPE) o[(int64_t)H * D + ch] = (uint16_t)(AE<F16>::pack2(dk[m][r] * scale, 0.f) & 0xffffu);
            o[(int64_t)2 * H * D + ch] = (uint16_t)(AE<F16>::pack2(dv[m][r], 0.f) & 0xffffu);
          }
          if constexpr (ROPE) rtile[crow(r, h2) * AH_RD + ch] = (uint16_t)(AE<F16>::pack2(dk[m][r] * scale, 0.f) & 0xffffu);   // [key][channel]
        }
      }
    }
    if constexpr (ROPE) {      // the gradient of the rotated k, turned back row by row
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_wave_barrier();
      ah_unrope_store<F16>(rtile, lane, kt * 32, L, rope, a, 1.f, dbase + (int64_t)H * D, rs);
      __builtin_amdgcn_wave_barrier();
    }
  }
}
