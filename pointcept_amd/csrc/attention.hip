// attention.hip -- PTv3 serialized patch attention: variable-length, non-causal, head_dim 16,
// bf16 in / fp32 accumulate, forward + backward, on gfx950 MFMA.
//
// Replaces flash_attn.flash_attn_varlen_qkvpacked_func (third party, un-vendored) as called at
//   pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py:208-214
// (qkv [T,3,H,16] bf16, cu_seqlens int32, softmax_scale = 16^-0.5, dropout 0).
//
// Shape facts that drive the design (SURVEY section 7 "hard parts"): every PTv3 stage has
// head_dim = 16, windows are <= 1024 keys.  QK^T is ONE k-step of v_mfma_f32_32x32x16_bf16, so
// there is one exp per 64 MFMA flops and the kernel is bound by VALU ISSUE SLOTS (r01_b PMC: 27 VALU
// instructions per MFMA in the first version, matrix pipe 26 % busy).  The matrix pipe has slack, the
// vector pipe has none, so every piece of softmax arithmetic that can be phrased as a matrix
// product is moved into MFMA operands:
//   * One workgroup = one (sequence, head); the sequence's operands live in LDS (<= 72 KB, two
//     workgroups per CU), each wave walks 32-row tiles against every 32-row tile of the other side.
//   * "Swapped" products S^T = K Q^T: a lane owns ONE query column (q = lane&31) and 16 keys in
//     registers, so everything per query is lane-local.
//   * The softmax scale is folded into the stationary operand: q*c (c = scale*log2 e) is split
//     into bf16 hi + lo parts (error 2^-17) and S' = K qhi^T + K qlo^T costs a second MFMA instead
//     of 16 multiplies per tile.
//   * The softmax REFERENCE POINT is the MFMA's C operand: S' = K qc^T - ref comes out of the
//     matrix pipe ready for v_exp_f32.  Forward: ref = |q| max_k|k| c, a Cauchy-Schwarz bound on
//     every logit of the row, known before the first key tile -- no running maximum, no rescaling,
//     no cross-lane traffic in the loop.  Softmax is invariant to the reference point; P <= 1 so
//     nothing overflows; the bound is only used while it is <= 64 in the exp2 domain, so the
//     largest P of a row is >= 2^-128 and bf16 / fp32 keep their full relative precision; rows above
//     that limit take the classic online-softmax loop.  Backward: ref = lse (and dP' = V dO^T -
//     delta the same way).
//   * P V is issued as O^T = [V^T ; 1 ; x] P^T: the packed P registers ARE the B operand (no lane
//     shuffles), row 16 of the A operand is all ones so the MFMA also produces the softmax
//     denominator, and O^T lands in the same lane as the softmax state.  Rows 17..31 of that
//     product are never read, so their A lanes may load anything.
//   * Transposed operands of the backward products come from the SAME row-major LDS images through
//     ds_read_b64_tr_b16 (a 16-lane group addresses [4 rows][16 channels], lane j receives rows 0..3
//     of channel j; mapping measured with tools/probe_gfx950.hip): no transposed staging pass.
//   * In the key-stationary backward kernel the per-QUERY constants (lse, delta) vary along the
//     register index, so they enter through one extra MFMA each: A = [lse_hi, lse_lo, d_hi, d_lo]
//     per query (bf16 pairs, staged once per workgroup), B = -1 in the matching contraction slots.
// Per 32x32 tile: forward 4 MFMA + ~26 VALU (was 3 + 80), dQ 5 + ~42, dK/dV 9 + ~50.
// No inline assembly touches MFMA results (the hazard recognizer cannot see into asm blocks: an
// asm v_max3_f32 on fresh accumulators read half-written registers, r01_d).
// Backward = two kernels (recompute P from q,k,lse; no atomics, no cross-wave reductions,
// bit-reproducible): dQ is query-stationary, dK/dV key-stationary.
// Roofline: fwd 4 L^2 D flops and L^2 exps per (sequence, head); bwd 10 L^2 D.
#include "ptc_common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

#define AT_WAVES 8
#define AT_THREADS (AT_WAVES * 64)
#define AT_MAX_L 1024
#define AT_MIN_WAVES 4          // waves per SIMD the register budget must allow (two 8-wave workgroups per CU)
#define AT_LOG2E 1.4426950408889634f
#define AT_LN2 0.6931471805599453f
#ifndef AT_SPLIT_BELOW
#define AT_SPLIT_BELOW 2048      // launches with fewer (sequence, head) parts than this are split further (at_split); 1024 until round 4: the
                                 // enc0 launch (1600 units = 3.1 rounds of 512 slots) gains 3-4 % as 3200 half units, profiles/r04_w_attn_split.txt
#endif
#define AT_FIXED_REF_MAX 64.0f   // largest Cauchy-Schwarz bound (exp2 domain) served by the fixed-reference loop

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  f32x2_t f = {lo, hi};
  bf16x2_t h = __builtin_convertvector(f, bf16x2_t);  // v_cvt_pk_bf16_f32 (RNE)
  return *reinterpret_cast<uint32_t*>(&h);
}
__device__ __forceinline__ float bf16_bits_to_float(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

// F16 I/O (round 4): under the reference's fp16 autocast the call site casts its operands itself -- qkv.to(bfloat16) in front of the
// bf16 kernel, feat.to(qkv.dtype) behind it (ptv3m1:209,215), and autograd the two gradients the other way.  Those four [N', 3 C] /
// [N', C] passes (2.4 ms of the fp16 recipe's step, profiles/r03_v_fp16_recipe_kernel_stats.csv) live in the kernels' load / store
// paths here: F16 = true reads f16 qkv / out / dout and rounds them to bf16 (RNE, what .to(bfloat16) does), and writes
// f16(bf16(result)) -- bit for bit the tensors the reference's casts produce around a bf16 kernel.  The arithmetic stays bf16.
typedef _Float16 at_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t at_f16x2_to_bf16x2(uint32_t w) {
  at_h2 h;
  __builtin_memcpy(&h, &w, 4);
  return pack_bf16x2((float)h[0], (float)h[1]);
}
__device__ __forceinline__ uint32_t at_bf16x2_to_f16x2(uint32_t w) {
  const at_h2 h = {(_Float16)__uint_as_float(w << 16), (_Float16)__uint_as_float(w & 0xffff0000u)};
  uint32_t r;
  __builtin_memcpy(&r, &h, 4);
  return r;
}
template <bool F16> __device__ __forceinline__ uint4 at_in(uint4 v) {
  if constexpr (F16) return make_uint4(at_f16x2_to_bf16x2(v.x), at_f16x2_to_bf16x2(v.y), at_f16x2_to_bf16x2(v.z), at_f16x2_to_bf16x2(v.w));
  return v;
}
template <bool F16> __device__ __forceinline__ uint32_t at_out(uint32_t w) {
  if constexpr (F16) return at_bf16x2_to_f16x2(w);
  return w;
}

__device__ __forceinline__ f32x16 mfma32(s16x8 a, s16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
typedef __attribute__((ext_vector_type(4))) float at_f32x4;
__device__ __forceinline__ at_f32x4 mfma16(s16x8 a, s16x8 b, at_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// a[lanes 16..31] <-> b[lanes 0..15] and a[lanes 48..63] <-> b[lanes 32..47] (v_permlane16_swap: the odd 16-lane rows of the first
// operand trade places with the even rows of the second)
#ifdef __HIPCC__
__device__ __forceinline__ void at_swap16(uint32_t& a, uint32_t& b) {
  const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}
#else
__device__ __forceinline__ void at_swap16(uint32_t& a, uint32_t& b) {
  const int l = emu::lane_id();
  const uint32_t oa = emu::exchange(a, l ^ 16), ob = emu::exchange(b, l ^ 16);
  if (l & 16) a = ob; else b = oa;
}
#endif
__device__ __forceinline__ f32x16 splat16(float v) {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = v;
  return z;
}
__device__ __forceinline__ f32x16 zero16() { return splat16(0.f); }
// C/D layout of the 32x32 MFMA: column = lane&31, row(reg, lane) = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
__device__ __forceinline__ int crow(int reg, int h2) { return (reg & 3) + 8 * (reg >> 2) + 4 * h2; }

// packed row index of element (t, j, head) in qkv [T,3,H,16]
__device__ __forceinline__ int64_t qkv_off(int64_t t, int j, int H, int head) { return ((t * 3 + j) * H + head) * 16; }

__device__ __forceinline__ s16x8 make_frag(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  uint4 u = {a, b, c, d};
  return *reinterpret_cast<s16x8*>(&u);
}
template <bool F16 = false>
__device__ __forceinline__ s16x8 ld_global_frag(const uint16_t* p, bool valid) {
  uint4 u = {0, 0, 0, 0};
  if (valid) u = at_in<F16>(*reinterpret_cast<const uint4*>(p));
  return *reinterpret_cast<s16x8*>(&u);
}
// x * c -> bf16 hi + bf16 lo (hi + lo = x*c to 2^-17 relative)
__device__ __forceinline__ void split_scaled(s16x8 x, float c, s16x8& hi, s16x8& lo) {
  uint32_t h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = bf16_bits_to_float((uint16_t)x[2 * j]) * c, b = bf16_bits_to_float((uint16_t)x[2 * j + 1]) * c;
    h[j] = pack_bf16x2(a, b);
    l[j] = pack_bf16x2(a - __uint_as_float(h[j] << 16), b - __uint_as_float(h[j] & 0xffff0000u));
  }
  hi = make_frag(h[0], h[1], h[2], h[3]);
  lo = make_frag(l[0], l[1], l[2], l[3]);
}

// ---- LDS images -------------------------------------------------------------------------------
// row-major [Lp][16] bf16 (32 B rows); the two 16-byte halves of a row are swapped when bit 3 of
// the row index is set, which makes the 16-lane ds_read_b128 groups conflict-free.
__device__ __forceinline__ int rm_off(int row, int half) { return row * 32 + ((half ^ ((row >> 3) & 1)) << 4); }

// stage rows [0,Lp) (zeros beyond L) row-major; returns this thread's max over its rows of |row|^2
template <bool F16 = false>
__device__ __forceinline__ float stage_row_major(const uint16_t* __restrict__ src, int64_t row_stride, int L, int Lp,
                                                 unsigned char* lds) {
  float mx = 0.f;
  for (int row = threadIdx.x; row < Lp; row += AT_THREADS) {
    uint4 v0 = {0, 0, 0, 0}, v1 = {0, 0, 0, 0};
    if (row < L) {
      const uint4* p = reinterpret_cast<const uint4*>(src + (int64_t)row * row_stride);
      v0 = at_in<F16>(p[0]);
      v1 = at_in<F16>(p[1]);
    }
    *reinterpret_cast<uint4*>(lds + rm_off(row, 0)) = v0;
    *reinterpret_cast<uint4*>(lds + rm_off(row, 1)) = v1;
    const uint32_t w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float a = __uint_as_float(w[j] << 16), b = __uint_as_float(w[j] & 0xffff0000u);
      ss = fmaf(a, a, fmaf(b, b, ss));
    }
    mx = fmaxf(mx, ss);
  }
  return mx;
}
// stage transposed [16][pitch] bf16 (pitch = Lp_max + 8 elements) with the keys of every 16-key block
// PERMUTED into the order in which the P V product consumes them: a lane's 8 contraction slots are keys
// 4*h2 + {0,1,2,3, 8,9,10,11} of the block (the rows a 32x32 MFMA leaves in 8 consecutive registers), so
// key kappa sits at position 8*((kappa>>2)&1) + (kappa&3) + 4*(kappa>>3) and the whole fragment is ONE
// 16-byte read.  (Two 8-byte reads get fused into ds_read2_b64, whose 32-bank addressing made rows i and
// i+8 collide: 1.97 M conflict cycles per launch in profiles/r01_f.)  A thread handles two adjacent rows
// (they stay adjacent under the permutation) and writes one 32-bit word per channel.
__device__ __forceinline__ int vt_pos(int key) {
  const int kk = key & 15;
  return (key & ~15) + 8 * ((kk >> 2) & 1) + (kk & 3) + 4 * (kk >> 3);
}
template <bool F16 = false>
__device__ __forceinline__ void stage_transposed(const uint16_t* __restrict__ src, int64_t row_stride, int L, int Lp,
                                                 int pitch, unsigned char* lds) {
  uint32_t* t32 = reinterpret_cast<uint32_t*>(lds);
  for (int p = threadIdx.x; p < Lp / 2; p += AT_THREADS) {
    uint4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
    const int ra = 2 * p, rb = 2 * p + 1;
    if (ra < L) {
      const uint4* q = reinterpret_cast<const uint4*>(src + (int64_t)ra * row_stride);
      a0 = at_in<F16>(q[0]); a1 = at_in<F16>(q[1]);
    }
    if (rb < L) {
      const uint4* q = reinterpret_cast<const uint4*>(src + (int64_t)rb * row_stride);
      b0 = at_in<F16>(q[0]); b1 = at_in<F16>(q[1]);
    }
    uint16_t ea[16], eb[16];
    *reinterpret_cast<uint4*>(ea) = a0; *reinterpret_cast<uint4*>(ea + 8) = a1;
    *reinterpret_cast<uint4*>(eb) = b0; *reinterpret_cast<uint4*>(eb + 8) = b1;
    const int w = vt_pos(ra) >> 1;
#pragma unroll
    for (int d = 0; d < 16; ++d) t32[(d * pitch) / 2 + w] = (uint32_t)ea[d] | ((uint32_t)eb[d] << 16);
  }
}

// Transposed MFMA operand from a ROW-MAJOR image: lane (i = lane&31, h2 = lane>>5) receives, for channel
// i & 15, the 8 contraction slots rows base + 4*h2 + {0,1,2,3, 8,9,10,11} (the register order in which a
// 32x32 product leaves its rows).  Lanes 16..31 of each half alias lanes 0..15: they feed operand rows
// that only reach output rows / columns >= 16, which no kernel reads.
struct TrAddr { int lo, hi; };   // byte offsets of the two ds_read_b64_tr_b16 for row base 0
__device__ __forceinline__ TrAddr tr_addr(int lane) {
  const int lp = lane & 15, h2 = lane >> 5;
  const int row = 4 * h2 + (lp >> 2), cq = lp & 3;
  TrAddr a;
  a.lo = rm_off(row, cq >> 1) + ((cq & 1) << 3);
  a.hi = rm_off(row + 8, cq >> 1) + ((cq & 1) << 3);
  return a;
}
__device__ __forceinline__ s16x8 ld_tr_frag(const unsigned char* img, TrAddr a, int row_base) {
  const unsigned char* p = img + row_base * 32;   // row_base is a multiple of 16: the swizzle phase is unchanged
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + a.lo));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + a.hi));
  return (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__device__ __forceinline__ s16x8 ld_tr_pair(const unsigned char* plo, const unsigned char* phi) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)plo);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)phi);
  return (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// XCD-aware order: workgroup b runs on XCD b % 8, so logical unit = (b % 8) * per + b / 8 gives every
// XCD a contiguous run of (sequence, head) units -- the H heads of a sequence read interleaved
// 32-byte pieces of the same qkv rows and share one L2 instead of re-fetching them per XCD.
__device__ __forceinline__ int at_unit(int n_units) {
  const int per_xcd = (n_units + 7) >> 3;
  return (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
}
// Small launches (deep stages: 256..1000 (sequence, head) units for 512 workgroup slots) are split
// `qs` ways along the stationary side: part p of a unit owns tiles [p*per, (p+1)*per).  The parts of a
// unit are consecutive workgroups (same XCD); each stages the whole other side again (64 KB, L2-hot).
__host__ __device__ __forceinline__ int at_split(int n_units, int lp_max) {
  int qs = 1;
  while (qs < 4 && (long long)n_units * qs < AT_SPLIT_BELOW && (lp_max >> 5) / (2 * qs) >= AT_WAVES) qs *= 2;
  return qs;
}

// (forcing a finer split at the bench shape -- 6400 half units instead of 3200 units on 512 slots, a shorter last round -- was measured
//  neutral in the step: 50.3 vs 49.8 ms, profiles/r03_a_knob_ab.txt; removed)
static inline int at_split_host(int n_units, int lp_max) { return at_split(n_units, lp_max); }

// ---- forward launch plan (round 5): whole units first, the LAST partial round in finer parts -----------------------------------
// Workgroup b runs on XCD b & 7 and the XCD takes its workgroups in order, two per CU: 3200 equal units are 6.25 rounds of its 64
// slots, i.e. the last quarter round costs a whole one (the "6.25-round tail", ~10 % of the launch).  Splitting EVERY unit only
// moves the problem (6400 halves = 12.5 rounds, each part pays the staging of the other side again: measured neutral, above).  The
// plan keeps the first `whole` units of every XCD's chunk whole and cuts only the ones behind them into `qs` parts (query tiles
// p*per .. (p+1)*per of the unit; every part stages K / V itself, L2-hot): 6 full rounds + one round of quarter units.  Chosen per
// launch by simulating the XCD's in-order dispatch for the candidates (uniform 1 / 2 / 4, tail 2 / 4) with a part costing
// AT_STAGE_FRAC + (1 - AT_STAGE_FRAC) / qs of a unit (0.10: what the uniform split of the 1600-unit launch gained, r04_w_attn_split.txt).
#ifndef AT_STAGE_FRAC
#define AT_STAGE_FRAC 0.10
#endif
#ifndef AT_ALONE
#define AT_ALONE 0.6             // duration of a workgroup that has its CU to itself, relative to one that shares it
#endif
#define AT_PLAN_CACHE 64
struct AtPlan { int whole, qs; };     // per XCD chunk: unit index < whole -> one workgroup; else qs workgroups per unit
__host__ __device__ __forceinline__ int at_plan_blocks_per_xcd(int n_units, AtPlan p) {
  const int per_xcd = (n_units + 7) >> 3, w = p.whole < per_xcd ? p.whole : per_xcd;
  return w + (per_xcd - w) * p.qs;
}
// block -> (unit, part, parts); false: nothing to do
__device__ __forceinline__ bool at_plan_unit(int n_units, int whole, int qs, int& unit, int& part, int& parts) {
  const int per_xcd = (n_units + 7) >> 3, xcd = (int)(blockIdx.x & 7), j = (int)(blockIdx.x >> 3);
  const int w = whole < per_xcd ? whole : per_xcd;
  if (j < w) {
    unit = xcd * per_xcd + j; part = 0; parts = 1;
  } else {
    const int jj = j - w;
    unit = xcd * per_xcd + w + jj / qs; part = jj % qs; parts = qs;
    if (w + jj / qs >= per_xcd) return false;
  }
  return unit < n_units;
}
static double at_plan_makespan(int per_xcd, int slots, AtPlan p) {
  // in-order dispatch of the chunk's workgroups onto the XCD's CUs, two slots per CU: the next workgroup takes the slot that frees first
  // (an idle CU before the second slot of a busy one).  A workgroup that starts ALONE on its CU runs AT_ALONE x as long as one that
  // shares it (the loop is issue-bound: 3200 units on 512 slots cost 459 us, 3584 units 489-528 us, r05_a_attn_tail_probe.txt; the
  // deep-stage launches, 96 .. 384 units, are decided by this term: quarter units on every CU beat half units on three CUs in four).
  if (slots > 512) slots = 512;
  if (slots < 2) slots = 2;
  const int cus = slots / 2;
  double fr[512];
  for (int i = 0; i < 2 * cus; ++i) fr[i] = 0.0;
  const int w = p.whole < per_xcd ? p.whole : per_xcd;
  const long nblk = w + (long)(per_xcd - w) * p.qs;
  const double t_part = AT_STAGE_FRAC + (1.0 - AT_STAGE_FRAC) / p.qs;
  double end = 0.0;
  for (long b = 0; b < nblk; ++b) {
    int best = 0;
    double bt = 1e300, bsib = 1e300;
    for (int i = 0; i < 2 * cus; ++i) {                      // (<= 64 slots per XCD, <= a few hundred workgroups, a handful of plans: cached)
      const double sib = fr[i ^ 1];
      if (fr[i] < bt || (fr[i] == bt && sib < bsib)) { bt = fr[i]; bsib = sib; best = i; }
    }
    const bool alone = fr[best ^ 1] <= bt;
    const double t = bt + (b < w ? 1.0 : t_part) * (alone ? AT_ALONE : 1.0);
    fr[best] = t;
    if (t > end) end = t;
  }
  return end;
}
static AtPlan at_plan_host(int n_units, int lp_max) {
  constexpr int slots_per_xcd = 2 * 256 / 8;                  // gfx950 = MI355X: 256 CUs in 8 XCDs, two 8-wave workgroups per CU (66 KB of LDS each)
  if (const char* e = getenv("PTC_AT_PLAN")) {                // "whole,qs" (A/B and tests); "0" = the round-4 uniform split
    int w = 0, q = 0;
    if (sscanf(e, "%d,%d", &w, &q) == 2 && (q == 1 || q == 2 || q == 4)) return {w, q};
    const int qs = at_split_host(n_units, lp_max);
    return {0, qs};
  }
  // The plan is a function of (workgroups per XCD chunk, query tiles) only.  Real training changes n_units with every batch and stage
  // (a 16-entry cache keyed on n_units that never evicted froze on the first two steps' shapes and re-ran the simulation --
  // 0.2-0.4 ms of host time -- on every later launch): 64 entries, keyed on what the plan depends on, replaced round-robin.
  const int per_xcd = (n_units + 7) >> 3, n_tiles = lp_max >> 5, S = slots_per_xcd;
  static thread_local struct { int per_xcd, n_tiles; AtPlan p; } cache[AT_PLAN_CACHE];     // per thread: launches may come from several host threads
  static thread_local int n_cache = 0, next_victim = 0;
  for (int i = 0; i < n_cache; ++i)
    if (cache[i].per_xcd == per_xcd && cache[i].n_tiles == n_tiles) return cache[i].p;
  AtPlan best = {per_xcd, 1};
  double t_best = at_plan_makespan(per_xcd, S, best);
  auto consider = [&](AtPlan p) {
    if (n_tiles / p.qs < AT_WAVES) return;                   // every wave of a part keeps at least one query tile
    const double t = at_plan_makespan(per_xcd, S, p);
    if (t < t_best * 0.995) { t_best = t; best = p; }
  };
  const int full = (per_xcd / S) * S;
  for (int qs = 2; qs <= 4; qs *= 2) {
    consider({0, qs});                                       // every unit split (the deep stages: fewer units than slots)
    if (full > 0 && full < per_xcd) consider({full, qs});    // only the last partial round
    if (full >= S && full == per_xcd) consider({full - S, qs});   // (exactly full rounds stay whole: nothing to gain)
  }
  if (n_cache < AT_PLAN_CACHE) cache[n_cache++] = {per_xcd, n_tiles, best};
  else { cache[next_victim] = {per_xcd, n_tiles, best}; next_victim = (next_victim + 1) % AT_PLAN_CACHE; }
  return best;
}

// A sequence longer than max_seqlen (cu_seqlens built for a larger patch than the caller's max_seqlen, or a caller
// of the flash_attn API passing inconsistent arguments) would overrun the LDS images, which are sized from
// max_seqlen.  Such units write NaN to every output row they own (16 bf16 per row, optionally the fp32 side vector) and
// return: the error is loud in the loss, memory stays intact.
template <bool F16 = false>
__device__ __forceinline__ void at_poison_rows(uint16_t* rows, int64_t row_stride, int L, float* side) {
  constexpr uint32_t nn = F16 ? 0x7E007E00u : 0x7FC07FC0u;
  const uint4 nan4 = {nn, nn, nn, nn};
  for (int q = threadIdx.x; q < L; q += AT_THREADS) {
    uint4* o = reinterpret_cast<uint4*>(rows + (int64_t)q * row_stride);
    o[0] = nan4;
    o[1] = nan4;
    if (side) side[q] = __uint_as_float(0x7FC00000u);
  }
}

// ================================================================================================
// forward
// ================================================================================================
// LDS: K row-major [lp_max][16] | V^T [17][pitch] (row 16 = 1.0 for keys < L, else 0) | 8 floats (reduction)
// The second-dispatched half of the workgroup's waves runs at s_setprio 1 (MI355X_MICROARCH.md, two waves per SIMD, item 4).  Measured and
// dropped (round 1/2, profiles/r02_g_attn_variants.txt, r02_h_attn_variants.txt): rounding the scaled query to ONE bf16 operand (3 MFMAs per
// tile instead of 4, 2^-9 relative logit error) and packing P by truncation (v_perm_b32 instead of v_cvt_pk_bf16_f32).
template <bool F16>
__global__ void __launch_bounds__(AT_THREADS, AT_MIN_WAVES)
attn_fwd_kernel(const uint16_t* __restrict__ qkv, const int32_t* __restrict__ cu, int H, float scale, int64_t total,
                int lp_max, int n_units, int plan_qs, int plan_whole, uint16_t* __restrict__ out, float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int unit, part, qs;
  if (!at_plan_unit(n_units, plan_whole, plan_qs, unit, part, qs)) return;
  const int seq = unit / H, head = unit % H;
  const int a = cu[seq], L = cu[seq + 1] - a;
  if (L <= 0) return;
  const int Lp = (L + 31) & ~31, n_tiles = Lp >> 5;
  if (Lp > lp_max) {   // sequence longer than the caller's max_seqlen: LDS is sized from max_seqlen -- poison, never overrun
    if (part == 0) at_poison_rows<F16>(out + ((int64_t)a * H + head) * 16, (int64_t)H * 16, L, lse + (int64_t)head * total + a);
    return;
  }
  const int t_per = (n_tiles + qs - 1) / qs, t_lo = part * t_per, t_hi = (t_lo + t_per) < n_tiles ? (t_lo + t_per) : n_tiles;
  if (t_lo >= n_tiles) return;
  const int pitch = lp_max + 8;
  unsigned char* Ksm = smem;                                   // [lp_max][16] row-major
  unsigned char* Vt = smem + (size_t)lp_max * 32;              // [17][pitch] transposed, keys permuted (vt_pos)
  float* red = reinterpret_cast<float*>(Vt + (size_t)17 * pitch * 2);  // [AT_WAVES]
  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  const int64_t rs = (int64_t)3 * H * 16;
  float kn = stage_row_major<F16>(qkv + qkv_off(a, 1, H, head), rs, L, Lp, Ksm);
  stage_transposed<F16>(qkv + qkv_off(a, 2, H, head), rs, L, Lp, pitch, Vt);
  // row 16: the denominator row of the P V product.  Zero beyond L, so padding keys (k = 0, finite P)
  // reach neither the numerator (V^T = 0) nor the denominator: no masking in the key loop.
  for (int key = threadIdx.x; key < Lp; key += AT_THREADS)
    reinterpret_cast<uint16_t*>(Vt + (size_t)16 * pitch * 2)[vt_pos(key)] = key < L ? (uint16_t)0x3F80 : (uint16_t)0;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) kn = fmaxf(kn, __shfl_xor(kn, o, 64));
  if (lane == 0) red[wave] = kn;
  __syncthreads();
  float kmax2 = red[0];
#pragma unroll
  for (int w = 1; w < AT_WAVES; ++w) kmax2 = fmaxf(kmax2, red[w]);

  const int col = lane & 31, h2 = lane >> 5;
  const float c = scale * AT_LOG2E;
  if (wave >= AT_WAVES / 2) __builtin_amdgcn_s_setprio(1);
  // A operand of the P V product: lane col <= 16 reads image row col (16 = denominator row); lanes
  // col > 16 feed output rows nobody reads and alias rows 1..15.  One 16-byte read per 16 keys.
  const unsigned char* vbase = Vt + ((size_t)(col <= 16 ? col : (col & 15)) * pitch + 8 * h2) * 2;
  const int vstride = 64, voff = 32;                           // bytes per 32-key tile / per 16-key block
  const unsigned char* kbase = Ksm + rm_off(col, h2);

#ifndef AT_QPREFETCH
#define AT_QPREFETCH 1           // the next query tile's fragment is requested before this tile's key loop (its latency was exposed at every tile start)
#endif
  int qt = t_lo + wave;
#if AT_QPREFETCH
  s16x8 qf_next = {0, 0, 0, 0, 0, 0, 0, 0};
  if (qt < t_hi) qf_next = ld_global_frag<F16>(qkv + qkv_off(a + qt * 32 + col, 0, H, head) + h2 * 8, qt * 32 + col < L);
#endif
  for (; qt < t_hi; qt += AT_WAVES) {
    const int q = qt * 32 + col;
#if AT_QPREFETCH
    const s16x8 qf = qf_next;
    if (qt + AT_WAVES < t_hi) {
      const int qn_ = (qt + AT_WAVES) * 32 + col;
      qf_next = ld_global_frag<F16>(qkv + qkv_off(a + qn_, 0, H, head) + h2 * 8, qn_ < L);
    }
#else
    const s16x8 qf = ld_global_frag<F16>(qkv + qkv_off(a + q, 0, H, head) + h2 * 8, q < L);
#endif
    float qn = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = bf16_bits_to_float((uint16_t)qf[j]);
      qn = fmaf(x, x, qn);
    }
    qn += __shfl_xor(qn, 32, 64);
    // Cauchy-Schwarz: every logit of this row satisfies s*c <= |q| max|k| c =: bnd.  (1 + 2^-10) and
    // the additive term cover the rounding of the norms and of the MFMA sums.
    const float bnd = sqrtf(qn * kmax2) * c * 1.0009765625f + 1e-3f;
    s16x8 qhi, qlo;
    split_scaled(qf, c, qhi, qlo);
    f32x16 acc = zero16();
    float ref2;                                                // softmax reference point, exp2 domain
    if (__builtin_amdgcn_ballot_w64(bnd > AT_FIXED_REF_MAX) == 0) {
      // ---- fixed reference: P = exp2(S c - bnd) <= ~1
      ref2 = bnd;
      const f32x16 negb = splat16(-bnd);
      const unsigned char* kp = kbase;
      const unsigned char* vp = vbase;
      // software pipeline: the S' product of tile kt+1 is in the matrix pipe while the vector pipe
      // exponentiates tile kt (a wave is in-order: without this its MFMAs and exps never overlap)
      // LDS operands are fetched one stage ahead of the MFMA that consumes them
      auto ldk = [&]() {
        const s16x8 f = *reinterpret_cast<const s16x8*>(kp);
        kp += 1024;
        return f;
      };
      s16x8 kfn = ldk();
      auto s_tile = [&]() {
        const s16x8 kf = kfn;
        f32x16 sv = mfma32(kf, qhi, negb);  // S'^T[key][q] = k.(q c) - bnd: lane = q, regs = keys crow(r,h2)
        sv = mfma32(kf, qlo, sv);
        kfn = ldk();
        return sv;
      };
      // Three-stage software pipeline over key tiles, two tiles per trip with ping-pong registers:
      //   matrix pipe : S'(kt+1) and P V (kt-1)      vector pipe : exp / pack of tile kt
      // so that no MFMA of a trip depends on the trip's own vector work (a wave is in-order: otherwise
      // its MFMAs and exps never overlap), and sched_group_barrier spreads the 24 vector instructions
      // over the 4 MFMA shadows.
      auto expo = [&](const f32x16& sv, uint32_t (&pk)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float e0 = __builtin_amdgcn_exp2f(sv[2 * i]), e1 = __builtin_amdgcn_exp2f(sv[2 * i + 1]);
          pk[i] = pack_bf16x2(e0, e1);
        }
      };
      s16x8 vf0, vf1;
      auto ldv = [&]() {
        vf0 = *reinterpret_cast<const s16x8*>(vp);
        vf1 = *reinterpret_cast<const s16x8*>(vp + voff);
        vp += vstride;
      };
      auto pv = [&](const uint32_t (&pk)[8]) {
        acc = mfma32(vf0, make_frag(pk[0], pk[1], pk[2], pk[3]), acc);
        acc = mfma32(vf1, make_frag(pk[4], pk[5], pk[6], pk[7]), acc);
      };
      auto interleave = [&]() {
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);     // the trip's three LDS reads first
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA, and in its shadow:
          __builtin_amdgcn_sched_group_barrier(0x400, 4, 0);   //   four transcendentals (v_exp_f32)
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   //   two plain VALU (v_cvt_pk_bf16_f32 / v_perm_b32)
        }
      
      };
      // S' of one tile past the end is computed and never used (it reads the V^T image: in-bounds LDS),
      // which keeps the trip body branch-free.
      f32x16 s0 = s_tile(), s1 = s_tile();
      uint32_t pa[8], pb[8];
      expo(s0, pa);
      int kt = 1;
      for (; kt + 1 < n_tiles; kt += 2) {             // at the top: s1 = S'(kt), pa = P(kt-1), P V done for tiles < kt-1
        ldv();
        s0 = s_tile();
        expo(s1, pb);
        pv(pa);
        interleave();
        ldv();
        s1 = s_tile();
        expo(s0, pa);
        pv(pb);
        interleave();
      }
      if (kt < n_tiles) {
        expo(s1, pb);
        ldv();
        pv(pa);
        ldv();
        pv(pb);
      } else {
        ldv();
        pv(pa);
      }
    } else {
      // ---- online softmax (rows whose norm bound is too loose to serve as the reference)
      float m = -INFINITY;                                     // running maximum of S' = s*c
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
        for (int r = 0; r < 9; ++r) acc[r] *= alpha;  // rows 0..15 = O^T, row 16 (reg 8, h2=0) = denominator
        uint32_t pk[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          pk[i] = pack_bf16x2(__builtin_amdgcn_exp2f(s[2 * i] - m), __builtin_amdgcn_exp2f(s[2 * i + 1] - m));
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
          const s16x8 pf = make_frag(pk[4 * mm], pk[4 * mm + 1], pk[4 * mm + 2], pk[4 * mm + 3]);
          const s16x8 vf = *reinterpret_cast<const s16x8*>(vp + mm * voff);
          acc = mfma32(vf, pf, acc);
        }
        vp += vstride;
      }
      ref2 = m;
    }
    const float l = __shfl(acc[8], col, 64);  // denominator lives in the h2 = 0 lane of column q
    const float inv = 1.f / l;
    if (q < L) {
      uint16_t* o = out + ((int64_t)(a + q) * H + head) * 16;
      uint2 w0, w1;
      w0.x = at_out<F16>(pack_bf16x2(acc[0] * inv, acc[1] * inv)); w0.y = at_out<F16>(pack_bf16x2(acc[2] * inv, acc[3] * inv));
      w1.x = at_out<F16>(pack_bf16x2(acc[4] * inv, acc[5] * inv)); w1.y = at_out<F16>(pack_bf16x2(acc[6] * inv, acc[7] * inv));
      *reinterpret_cast<uint2*>(o + 4 * h2) = w0;       // d = 4*h2 + {0..3}
      *reinterpret_cast<uint2*>(o + 8 + 4 * h2) = w1;   // d = 8 + 4*h2 + {0..3}
      if (h2 == 0) lse[(int64_t)head * total + a + q] = ref2 * AT_LN2 + __logf(l);
    }
  }
}

// ================================================================================================
// backward (round 4 form)
// ================================================================================================
// (The split form: launches below AT1_MIN_UNITS (sequence, head) units, which it fills the chip with by splitting units; larger launches
//  take the one-pass kernel of attention_bwd1.h, included below.)
// Two kernels as before (recompute P from q, k, lse; no atomics, no cross-wave reductions, bit-reproducible): dQ query-stationary, dK / dV
// key-stationary.  What changed against rounds 1-3 (14 MFMAs of 32 cycles per (query tile, key tile) pair over the two kernels, and
// v_exp_f32 does not overlap with them: 448 + ~150 cycles of floor per pair and SIMD against 890 measured, profiles/r03_p_attn_pmc.json):
//   * every product whose CONTRACTION runs over a tile side (dQ += dS K, dK += dS^T Q, dV += P^T dO) is a 16x16x32 MFMA: head_dim 16 is
//     exactly its output height, where the 32x32x16 form computed 32 output rows and read 16.  A 32-row side is two MFMAs of 16
//     cycles instead of two of 32.  The operand the softmax produces (P / dS in the 32x32 accumulator layout: lane = column, 16 rows in
//     registers) becomes the 16x16x32 B operand (lane = column mod 16, 8 contraction rows) through FOUR v_permlane16_swap per operand:
//     the packed registers of rows {0-3, 8-11} + 4 h2 and those of rows {16-19, 24-27} + 4 h2 trade their odd / even 16-lane groups, which
//     puts all 32 rows of columns 0-15 into one register quadruple and all 32 rows of columns 16-31 into the other.  The matching A
//     operand (K^T, Q^T, dO^T for the same row order) comes from the row-major LDS images through ds_read_b64_tr_b16 as before;
//   * key-stationary kernel: the per-QUERY constants (lse, delta) vary along the register index of the accumulator, and entered through one
//     extra MFMA each (bf16 hi + lo pairs against -1 slots).  They are now the C OPERAND of the first product, read from fp32 LDS
//     arrays with four broadcast ds_read_b128 each (a lane's 16 rows are four runs of 4): two 32-cycle MFMAs less per tile and the
//     constants exact;
//   * results leave as 8-byte stores (a lane of a 16x16 tile holds 4 consecutive channels of one row) instead of 2-byte ones.
// Per 32x32 tile: dQ 3 x 32 + 2 x 16 = 128 matrix-pipe cycles (was 160), dK / dV 3 x 32 + 4 x 16 = 160 (was 288).

// contraction-row order of the 16x16x32 operands built from a 32x32 accumulator: slot (g = lane >> 4, j) <-> row at_krow(g) + j (j < 4),
// at_krow(g) + 8 + (j - 4) (j >= 4)
__device__ __forceinline__ int at_krow(int g) { return 16 * (g & 1) + 4 * (g >> 1); }
struct TrAddr16 { int lo, hi; };   // byte offsets of the two ds_read_b64_tr_b16 of a 16x16x32 A operand for row base 0
__device__ __forceinline__ TrAddr16 tr_addr16(int lane) {
  const int lp = lane & 15, g = lane >> 4;
  const int row = at_krow(g) + (lp >> 2), cq = lp & 3;
  TrAddr16 a;
  a.lo = rm_off(row, cq >> 1) + ((cq & 1) << 3);
  a.hi = rm_off(row + 8, cq >> 1) + ((cq & 1) << 3);
  return a;
}
// P / dS of a 32x32 tile (8 packed registers: pk[i] = rows crow(2 i), crow(2 i + 1) of column lane & 31) -> the two 16x16x32 B operands
// (columns 0-15 | 16-31)
__device__ __forceinline__ void at_to_b16(uint32_t (&pk)[8], s16x8& lo, s16x8& hi) {
#pragma unroll
  for (int j = 0; j < 4; ++j) at_swap16(pk[j], pk[4 + j]);
  lo = make_frag(pk[0], pk[1], pk[2], pk[3]);
  hi = make_frag(pk[4], pk[5], pk[6], pk[7]);
}

// ---- part 1: dQ (query-stationary) + delta = rowsum(dO * O) ------------------------------------------------------------------
// LDS: V row-major [lp_max][16] | K row-major [lp_max][16]
template <int LP, bool F16>   // LP = lp_max at compile time (1024: image distances become immediate offsets) or 0 = runtime
__global__ void __launch_bounds__(AT_THREADS, AT_MIN_WAVES)
attn_bwd_dq_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ out, const uint16_t* __restrict__ dout,
                   const float* __restrict__ lse, const int32_t* __restrict__ cu, int H, float scale, int64_t total,
                   int lp_max, int n_units, int qs, uint16_t* __restrict__ dqkv, float* __restrict__ delta) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lunit = at_unit(n_units * qs);
  if (lunit >= n_units * qs) return;
  const int unit = lunit / qs, part = lunit - unit * qs;
  const int seq = unit / H, head = unit % H;
  const int a = cu[seq], L = cu[seq + 1] - a;
  if (L <= 0) return;
  const int Lp = (L + 31) & ~31, n_tiles = Lp >> 5;
  if (Lp > lp_max) {   // see attn_fwd_kernel
    if (part == 0) at_poison_rows<F16>(dqkv + qkv_off(a, 0, H, head), (int64_t)3 * H * 16, L, nullptr);
    return;
  }
  const int t_per = (n_tiles + qs - 1) / qs, t_lo = part * t_per, t_hi = (t_lo + t_per) < n_tiles ? (t_lo + t_per) : n_tiles;
  if (t_lo >= n_tiles) return;
  unsigned char* Vsm = smem;
  unsigned char* Ksm = smem + (size_t)lp_max * 32;
  const int64_t rs = (int64_t)3 * H * 16;
  stage_row_major<F16>(qkv + qkv_off(a, 2, H, head), rs, L, Lp, Vsm);
  stage_row_major<F16>(qkv + qkv_off(a, 1, H, head), rs, L, Lp, Ksm);
  __syncthreads();

  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  const int col = lane & 31, h2 = lane >> 5;
  const float c = scale * AT_LOG2E;
  const TrAddr16 ta = tr_addr16(lane);
  const int rmo = rm_off(col, h2);

  for (int qt = t_lo + wave; qt < t_hi; qt += AT_WAVES) {
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
    const f32x16 negl = splat16(-l2), negd = splat16(-dl);
    at_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};    // dQ^T[d = 4 g + e][q = 32 qt + (0 | 16) + lane & 15]
    {
      // three per-lane LDS pointers advanced once per trip, everything else immediate offsets; the V image sits KOFF bytes BEFORE the K image
      const int KOFF = LP ? LP * 32 : lp_max * 32;
      const unsigned char* pv = Vsm + rmo;
      const unsigned char* pl = Ksm + ta.lo;
      const unsigned char* ph = Ksm + ta.hi;
      auto tile = [&](const int o) {
        const s16x8 vf = *reinterpret_cast<const s16x8*>(pv + o);
        const s16x8 kf = *reinterpret_cast<const s16x8*>(pv + o + KOFF);
        f32x16 s = mfma32(kf, qhi, negl);                 // S'^T = k.(q c) - lse  (exp2 domain): lane = query, registers = keys crow(r, h2)
        s = mfma32(kf, qlo, s);
        const f32x16 dp = mfma32(vf, dof, negd);          // dP^T - delta
        // keys >= L have k = 0, so whatever (finite) dS they get multiplies K^T = 0 below
        uint32_t pk[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          pk[i] = pack_bf16x2(__builtin_amdgcn_exp2f(s[2 * i]) * dp[2 * i], __builtin_amdgcn_exp2f(s[2 * i + 1]) * dp[2 * i + 1]);
        s16x8 ds0, ds1;
        at_to_b16(pk, ds0, ds1);                          // dS^T as B operands: queries 0-15 | 16-31 of the tile, all 32 keys each
        const s16x8 ktf = ld_tr_pair(pl + o, ph + o);     // K^T[d][key slots]
        acc0 = mfma16(ktf, ds0, acc0);
        acc1 = mfma16(ktf, ds1, acc1);
      };
      int kt = 0;
      for (; kt + 1 < n_tiles; kt += 2) {
        tile(0);
        tile(1024);
        pv += 2048; pl += 2048; ph += 2048;
      }
      if (kt < n_tiles) tile(0);
    }
    // lane (n = lane & 15, g = lane >> 4) holds channels 4 g .. 4 g + 3 of queries 32 qt + n (acc0) and 32 qt + 16 + n (acc1)
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

// ---- part 2: dK, dV (key-stationary).  Needs delta written by part 1 (same stream). -------------------------------------------------
// LDS: Q row-major [lp_max][16] | dO row-major [lp_max][16] | -lse * log2 e fp32 [lp_max] | -delta fp32 [lp_max]
#define AT_PAD_LSE 1.0e30f   // lse of padding queries: exp2(s - 1e30) = 0
template <int LP, bool F16>   // LP = lp_max at compile time (1024: image distances become immediates) or 0
__global__ void __launch_bounds__(AT_THREADS, AT_MIN_WAVES)
attn_bwd_dkv_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dout, const float* __restrict__ lse,
                    const float* __restrict__ delta, const int32_t* __restrict__ cu, int H, float scale, int64_t total,
                    int lp_max, int n_units, int qs, uint16_t* __restrict__ dqkv) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lunit = at_unit(n_units * qs);
  if (lunit >= n_units * qs) return;
  const int unit = lunit / qs, part = lunit - unit * qs;
  const int seq = unit / H, head = unit % H;
  const int a = cu[seq], L = cu[seq + 1] - a;
  if (L <= 0) return;
  const int Lp = (L + 31) & ~31, n_tiles = Lp >> 5;
  if (Lp > lp_max) {   // see attn_fwd_kernel
    if (part == 0) {
      at_poison_rows<F16>(dqkv + qkv_off(a, 1, H, head), (int64_t)3 * H * 16, L, nullptr);
      at_poison_rows<F16>(dqkv + qkv_off(a, 2, H, head), (int64_t)3 * H * 16, L, nullptr);
    }
    return;
  }
  const int t_per = (n_tiles + qs - 1) / qs, t_lo = part * t_per, t_hi = (t_lo + t_per) < n_tiles ? (t_lo + t_per) : n_tiles;
  if (t_lo >= n_tiles) return;
  unsigned char* Qsm = smem;
  float* nl = reinterpret_cast<float*>(smem + (size_t)lp_max * 64);            // -lse * log2 e per query
  float* nd = nl + lp_max;                                                     // -delta per query
  stage_row_major<F16>(qkv + qkv_off(a, 0, H, head), (int64_t)3 * H * 16, L, Lp, Qsm);
  stage_row_major<F16>(dout + ((int64_t)a * H + head) * 16, (int64_t)H * 16, L, Lp, smem + (size_t)lp_max * 32);
  for (int q = threadIdx.x; q < Lp; q += AT_THREADS) {
    nl[q] = q < L ? -lse[(int64_t)head * total + a + q] * AT_LOG2E : -AT_PAD_LSE;
    nd[q] = q < L ? -delta[(int64_t)head * total + a + q] : 0.f;
  }
  __syncthreads();

  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  const int col = lane & 31, h2 = lane >> 5;
  const float c = scale * AT_LOG2E;
  const TrAddr16 ta = tr_addr16(lane);
  const int rmo = rm_off(col, h2);

  for (int kt = t_lo + wave; kt < t_hi; kt += AT_WAVES) {
    const int key = kt * 32 + col;
    const s16x8 kf = ld_global_frag<F16>(qkv + qkv_off(a + key, 1, H, head) + h2 * 8, key < L);
    const s16x8 vf = ld_global_frag<F16>(qkv + qkv_off(a + key, 2, H, head) + h2 * 8, key < L);
    s16x8 khi, klo;
    split_scaled(kf, c, khi, klo);
    at_f32x4 dv0 = {0.f, 0.f, 0.f, 0.f}, dv1 = dv0, dk0 = dv0, dk1 = dv0;    // dV^T / dK^T [d = 4 g + e][key = 32 kt + (0 | 16) + lane & 15]
    {
      // Per-lane LDS pointers (row-major fragment, the two halves of the transposed fragments, the constants of the lane's first row
      // run), advanced once per trip; everything else is an immediate offset (the dO image sits DOFF bytes after the Q image, the delta
      // array lp_max floats after the lse array)
      const int DOFF = LP ? LP * 32 : lp_max * 32, NOFF = (LP ? LP : lp_max) * 4;
      const unsigned char* pq = Qsm + rmo;
      const unsigned char* pl = Qsm + ta.lo;
      const unsigned char* ph = Qsm + ta.hi;
      const unsigned char* pc = reinterpret_cast<const unsigned char*>(nl) + 16 * h2;       // rows crow(r, h2) = 8 (r / 4) + 4 h2 + r % 4
      auto tile = [&](const int o, const int oc) {          // o = byte offset of the tile in the row-major images, oc in the constant arrays
        const s16x8 qf = *reinterpret_cast<const s16x8*>(pq + o);
        const s16x8 dof = *reinterpret_cast<const s16x8*>(pq + o + DOFF);
        // C operands: -lse / -delta of the lane's 16 queries = four runs of four consecutive rows (the same for every lane of a half: broadcast reads)
        f32x16 s, dp;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const at_f32x4 l4 = *reinterpret_cast<const at_f32x4*>(pc + oc + 32 * r4);
          const at_f32x4 d4 = *reinterpret_cast<const at_f32x4*>(pc + oc + 32 * r4 + NOFF);
#pragma unroll
          for (int e = 0; e < 4; ++e) { s[4 * r4 + e] = l4[e]; dp[4 * r4 + e] = d4[e]; }
        }
        s = mfma32(qf, khi, s);                         // S'[q][key] = q.(k c) - lse (exp2 domain): lane = key, registers = queries crow(r, h2)
        s = mfma32(qf, klo, s);
        dp = mfma32(dof, vf, dp);                       // dP[q][key] - delta[q]
        uint32_t pp[8], ps[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float p0 = __builtin_amdgcn_exp2f(s[2 * i]), p1 = __builtin_amdgcn_exp2f(s[2 * i + 1]);
          pp[i] = pack_bf16x2(p0, p1);
          ps[i] = pack_bf16x2(p0 * dp[2 * i], p1 * dp[2 * i + 1]);
        }
        s16x8 p0f, p1f, s0f, s1f;
        at_to_b16(pp, p0f, p1f);                        // P as B operands: keys 0-15 | 16-31 of the tile, all 32 queries each
        at_to_b16(ps, s0f, s1f);
        const s16x8 dotf = ld_tr_pair(pl + o + DOFF, ph + o + DOFF);      // dO^T[d][query slots]
        const s16x8 qtf = ld_tr_pair(pl + o, ph + o);                     // Q^T[d][query slots]
        dv0 = mfma16(dotf, p0f, dv0);
        dv1 = mfma16(dotf, p1f, dv1);
        dk0 = mfma16(qtf, s0f, dk0);
        dk1 = mfma16(qtf, s1f, dk1);
      };
      int qt = 0;
      for (; qt + 1 < n_tiles; qt += 2) {
        tile(0, 0);
        tile(1024, 128);
        pq += 2048; pl += 2048; ph += 2048; pc += 256;
      }
      if (qt < n_tiles) tile(0, 0);
    }
    // lane (n = lane & 15, g = lane >> 4) holds channels 4 g .. 4 g + 3 of keys 32 kt + n (.0) and 32 kt + 16 + n (.1)
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

#include "attention_bwd1.h"
#include "attention_hd.h"
#include "attention_rpe.h"
#include "attention_drop.h"

// ================================================================================================
// host side
// ================================================================================================
// dynamic LDS above 64 KB has to be opted into once per kernel
template <typename K>
static int allow_big_lds(K kernel, size_t bytes) {
  PTC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return PTC_OK;
}

static size_t fwd_lds_bytes(int lp_max) { return (size_t)lp_max * 32 + (size_t)17 * (lp_max + 8) * 2 + AT_WAVES * 4; }
// (PTC_AT_BWD_PAD_LDS: occupancy probe of tools only -- extra dynamic LDS per workgroup, e.g. 20000 = one workgroup per CU)
static size_t at_bwd_pad() { const char* e = getenv("PTC_AT_BWD_PAD_LDS"); return e ? (size_t)atoi(e) : 0; }
static size_t dq_lds_bytes(int lp_max) { return (size_t)lp_max * 64 + at_bwd_pad(); }
static size_t dkv_lds_bytes(int lp_max) { return (size_t)lp_max * 64 + (size_t)lp_max * 8 + at_bwd_pad(); }

static int check_common(const char* name, const void* qkv, const int32_t* cu, int64_t n_seq, int64_t total, int H,
                        int max_seqlen, int dtype, bool f16_io = false) {
  // f16_io (the head_dim-16 window kernels): PTC_F16 = f16 tensors in and out around the same bf16 arithmetic, i.e. the reference's
  // qkv.to(bfloat16) / feat.to(qkv.dtype) casts folded into the load / store paths (see at_in / at_out)
  PTC_REQUIRE(dtype == PTC_BF16 || (f16_io && dtype == PTC_F16), PTC_EUNSUPPORTED,
              "%s: bf16 arithmetic only (the reference casts qkv to bf16, ptv3m1:209)%s", name, f16_io ? "; tensors bf16 or f16" : "");
  PTC_REQUIRE(n_seq >= 0 && total >= 0 && H >= 1, PTC_EINVAL, "%s: bad sizes", name);
  PTC_REQUIRE(max_seqlen >= 1 && max_seqlen <= AT_MAX_L, PTC_EUNSUPPORTED, "%s: max_seqlen=%d not in [1,%d]", name, max_seqlen, AT_MAX_L);
  PTC_REQUIRE(n_seq * H < (1ll << 29), PTC_EUNSUPPORTED, "%s: grid too large", name);
  PTC_REQUIRE(n_seq == 0 || (qkv && cu), PTC_EINVAL, "%s: null buffer", name);
  PTC_REQUIRE((uintptr_t)qkv % 16 == 0, PTC_EINVAL, "%s: qkv must be 16-byte aligned", name);
  return PTC_OK;
}

extern "C" int ptc_attn_varlen_fwd(const void* qkv, const int32_t* cu_seqlens, int64_t n_seq, int64_t total, int H,
                                   int max_seqlen, float softmax_scale, int dtype, void* out, float* lse,
                                   ptc_stream_t stream) {
  int rc = check_common("ptc_attn_varlen_fwd", qkv, cu_seqlens, n_seq, total, H, max_seqlen, dtype, true);
  if (rc != PTC_OK) return rc;
  if (n_seq == 0 || total == 0) return PTC_OK;
  PTC_REQUIRE(out && lse, PTC_EINVAL, "ptc_attn_varlen_fwd: null buffer");
  const int lp_max = (max_seqlen + 31) & ~31;
  const size_t lds = fwd_lds_bytes(lp_max);
  hipStream_t s = (hipStream_t)stream;
  const int n_units = (int)(n_seq * H);
  const AtPlan plan = at_plan_host(n_units, lp_max);
#define AT_FWD_CASE(F16)                                                                                                                   \
  if ((dtype == PTC_F16) == F16) {                                                                                                          \
    rc = allow_big_lds(attn_fwd_kernel<F16>, lds);                                                                                          \
    if (rc != PTC_OK) return rc;                                                                                                            \
    hipLaunchKernelGGL(attn_fwd_kernel<F16>, dim3((unsigned)(8 * at_plan_blocks_per_xcd(n_units, plan))), dim3(AT_THREADS), lds, s,           \
                       (const uint16_t*)qkv, cu_seqlens, H, softmax_scale, total, lp_max, n_units, plan.qs, plan.whole, (uint16_t*)out, lse); \
    PTC_CHECK_LAUNCH("attn_fwd_kernel");                                                                                                    \
    return PTC_OK;                                                                                                                          \
  }
  AT_FWD_CASE(false) AT_FWD_CASE(true)
#undef AT_FWD_CASE
  return PTC_EINVAL;   // not reached
}

extern "C" size_t ptc_attn_varlen_bwd_workspace_bytes(int64_t total, int H) {
  return ptc_align_up((size_t)(total > 0 ? total : 1) * (size_t)H * sizeof(float), 256);
}

extern "C" int ptc_attn_varlen_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                                   const int32_t* cu_seqlens, int64_t n_seq, int64_t total, int H, int max_seqlen,
                                   float softmax_scale, int dtype, void* dqkv, void* workspace, size_t workspace_bytes,
                                   ptc_stream_t stream) {
  int rc = check_common("ptc_attn_varlen_bwd", qkv, cu_seqlens, n_seq, total, H, max_seqlen, dtype, true);
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
  const int n_units = (int)(n_seq * H);
  // launches that fill the chip with whole units take the one-pass kernel (attention_bwd1.h); the deep stages (a few hundred units)
  // keep the two split kernels below.  PTC_AT_BWD1=0 / 1 forces either form (timing A/B, tests).
  {
    const char* e1 = getenv("PTC_AT_BWD1");          // read per call: tests switch it in-process
    const int forced = e1 ? atoi(e1) : -1;
    if (forced != 0 && lp_max <= AT_MAX_L && (forced == 1 || n_units >= AT1_MIN_UNITS)) {
      const unsigned g1 = (unsigned)(8 * ((n_units + 7) / 8));
#define AT_BWD1_CASE(F16)                                                                                                           \
      if ((dtype == PTC_F16) == F16) {                                                                                              \
        rc = allow_big_lds((attn_bwd1_kernel<F16>), bwd1_lds_bytes(lp_max));                                                        \
        if (rc != PTC_OK) return rc;                                                                                                \
        hipLaunchKernelGGL((attn_bwd1_kernel<F16>), dim3(g1), dim3(AT_THREADS), bwd1_lds_bytes(lp_max), s, (const uint16_t*)qkv,     \
                           (const uint16_t*)out, (const uint16_t*)dout, lse, cu_seqlens, H, softmax_scale, total, lp_max, n_units,  \
                           (uint16_t*)dqkv);                                                                                        \
        PTC_CHECK_LAUNCH("attn_bwd1_kernel");                                                                                       \
        return PTC_OK;                                                                                                              \
      }
      AT_BWD1_CASE(false) AT_BWD1_CASE(true)
#undef AT_BWD1_CASE
    }
  }
  const int qs = at_split_host(n_units, lp_max);
  const unsigned grid = (unsigned)(8 * ((n_units * qs + 7) / 8));
  // (s_setprio on half of the waves, as in the forward: measured neutral to harmful here, profiles/r02_h_attn_variants.txt; a two-stage
  //  software pipeline at one workgroup per CU: slower, profiles/r02_o_slp_ab.txt; round 1's single-pass backward -- 32x32 MFMAs throughout, dQ
  //  accumulated in LDS tiles under a turn counter -- was 2.1-2.3 ms against 1.56, r01_w/x: the one-pass kernel above is a different design)
#define AT_BWD_CASE(LP, F16)                                                                                                        \
  if ((LP == 0 ? lp_max != 1024 : lp_max == LP) && (dtype == PTC_F16) == F16) {                                                                                                             \
    rc = allow_big_lds((attn_bwd_dq_kernel<LP, F16>), dq_lds_bytes(lp_max));                                                           \
    if (rc != PTC_OK) return rc;                                                                                               \
    rc = allow_big_lds((attn_bwd_dkv_kernel<LP, F16>), dkv_lds_bytes(lp_max));                                                         \
    if (rc != PTC_OK) return rc;                                                                                               \
    hipLaunchKernelGGL((attn_bwd_dq_kernel<LP, F16>), dim3(grid), dim3(AT_THREADS), dq_lds_bytes(lp_max), s, (const uint16_t*)qkv,     \
                       (const uint16_t*)out, (const uint16_t*)dout, lse, cu_seqlens, H, softmax_scale, total, lp_max, n_units, \
                       qs, (uint16_t*)dqkv, delta);                                                                            \
    PTC_CHECK_LAUNCH("attn_bwd_dq_kernel");                                                                                    \
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<LP, F16>), dim3(grid), dim3(AT_THREADS), dkv_lds_bytes(lp_max), s, (const uint16_t*)qkv,   \
                       (const uint16_t*)dout, lse, (const float*)delta, cu_seqlens, H, softmax_scale, total, lp_max, n_units,  \
                       qs, (uint16_t*)dqkv);                                                                                   \
    PTC_CHECK_LAUNCH("attn_bwd_dkv_kernel");                                                                                   \
    return PTC_OK;                                                                                                             \
  }
  AT_BWD_CASE(1024, false) AT_BWD_CASE(0, false) AT_BWD_CASE(1024, true) AT_BWD_CASE(0, true)
#undef AT_BWD_CASE
  return PTC_EINVAL;   // not reached
}

// ------------------------------------------------------------------------------------------------
// attention dropout, head_dim 16 (attention_drop.h)
// ------------------------------------------------------------------------------------------------
static int drop_params(const char* name, float p, uint32_t* thresh, float* rp) {
  PTC_REQUIRE(p >= 0.f && p < 1.f, PTC_EINVAL, "%s: dropout_p=%g must lie in [0, 1)", name, (double)p);
  const double t = (double)p * 4294967296.0;
  *thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
  *rp = 1.f / (1.f - p);
  return PTC_OK;
}

extern "C" int ptc_attn_varlen_dropout_fwd(const void* qkv, const int32_t* cu_seqlens, int64_t n_seq, int64_t total, int H, int max_seqlen,
                                           float softmax_scale, int dtype, float dropout_p, uint64_t seed, void* out, float* lse,
                                           ptc_stream_t stream) {
  int rc = check_common("ptc_attn_varlen_dropout_fwd", qkv, cu_seqlens, n_seq, total, H, max_seqlen, dtype, true);
  if (rc != PTC_OK) return rc;
  uint32_t thresh;
  float rp;
  rc = drop_params("ptc_attn_varlen_dropout_fwd", dropout_p, &thresh, &rp);
  if (rc != PTC_OK) return rc;
  if (n_seq == 0 || total == 0) return PTC_OK;
  PTC_REQUIRE(out && lse, PTC_EINVAL, "ptc_attn_varlen_dropout_fwd: null buffer");
  const int lp_max = (max_seqlen + 31) & ~31;
  const size_t lds = fwd_lds_bytes(lp_max);
  const int n_units = (int)(n_seq * H);
  const unsigned grid = (unsigned)(8 * ((n_units + 7) / 8));
#define AD_FWD_CASE(F16)                                                                                                             \
  if ((dtype == PTC_F16) == F16) {                                                                                                   \
    rc = allow_big_lds(attn_drop_fwd_kernel<F16>, lds);                                                                              \
    if (rc != PTC_OK) return rc;                                                                                                     \
    hipLaunchKernelGGL(attn_drop_fwd_kernel<F16>, dim3(grid), dim3(AT_THREADS), lds, (hipStream_t)stream, (const uint16_t*)qkv,      \
                       cu_seqlens, H, softmax_scale, total, lp_max, n_units, thresh, rp, (uint32_t)seed, (uint32_t)(seed >> 32),     \
                       (uint16_t*)out, lse);                                                                                         \
    PTC_CHECK_LAUNCH("attn_drop_fwd_kernel");                                                                                        \
    return PTC_OK;                                                                                                                   \
  }
  AD_FWD_CASE(false) AD_FWD_CASE(true)
#undef AD_FWD_CASE
  return PTC_EINVAL;   // not reached
}

extern "C" int ptc_attn_varlen_dropout_bwd(const void* qkv, const void* out, const void* dout, const float* lse, const int32_t* cu_seqlens,
                                           int64_t n_seq, int64_t total, int H, int max_seqlen, float softmax_scale, int dtype,
                                           float dropout_p, uint64_t seed, void* dqkv, void* workspace, size_t workspace_bytes,
                                           ptc_stream_t stream) {
  int rc = check_common("ptc_attn_varlen_dropout_bwd", qkv, cu_seqlens, n_seq, total, H, max_seqlen, dtype, true);
  if (rc != PTC_OK) return rc;
  uint32_t thresh;
  float rp;
  rc = drop_params("ptc_attn_varlen_dropout_bwd", dropout_p, &thresh, &rp);
  if (rc != PTC_OK) return rc;
  if (n_seq == 0 || total == 0) return PTC_OK;
  PTC_REQUIRE(out && dout && lse && dqkv && workspace, PTC_EINVAL, "ptc_attn_varlen_dropout_bwd: null buffer");
  PTC_REQUIRE(workspace_bytes >= ptc_attn_varlen_bwd_workspace_bytes(total, H), PTC_EWORKSPACE, "ptc_attn_varlen_dropout_bwd: workspace too small");
  PTC_REQUIRE(((uintptr_t)out % 16 == 0) && ((uintptr_t)dout % 16 == 0) && ((uintptr_t)dqkv % 16 == 0), PTC_EINVAL,
              "ptc_attn_varlen_dropout_bwd: buffers must be 16-byte aligned");
  const int lp_max = (max_seqlen + 31) & ~31;
  hipStream_t s = (hipStream_t)stream;
  float* delta = (float*)workspace;
  const int n_units = (int)(n_seq * H);
  const unsigned grid = (unsigned)(8 * ((n_units + 7) / 8));
  const size_t lds_q = (size_t)lp_max * 64, lds_kv = (size_t)lp_max * 64 + (size_t)lp_max * 8;
#define AD_BWD_CASE(F16)                                                                                                             \
  if ((dtype == PTC_F16) == F16) {                                                                                                   \
    rc = allow_big_lds(attn_drop_bwd_dq_kernel<F16>, lds_q);                                                                         \
    if (rc != PTC_OK) return rc;                                                                                                     \
    rc = allow_big_lds(attn_drop_bwd_dkv_kernel<F16>, lds_kv);                                                                       \
    if (rc != PTC_OK) return rc;                                                                                                     \
    hipLaunchKernelGGL(attn_drop_bwd_dq_kernel<F16>, dim3(grid), dim3(AT_THREADS), lds_q, s, (const uint16_t*)qkv, (const uint16_t*)out, \
                       (const uint16_t*)dout, lse, cu_seqlens, H, softmax_scale, total, lp_max, n_units, thresh, rp, (uint32_t)seed, \
                       (uint32_t)(seed >> 32), (uint16_t*)dqkv, delta);                                                              \
    PTC_CHECK_LAUNCH("attn_drop_bwd_dq_kernel");                                                                                     \
    hipLaunchKernelGGL(attn_drop_bwd_dkv_kernel<F16>, dim3(grid), dim3(AT_THREADS), lds_kv, s, (const uint16_t*)qkv,                 \
                       (const uint16_t*)dout, lse, (const float*)delta, cu_seqlens, H, softmax_scale, total, lp_max, n_units, thresh, \
                       rp, (uint32_t)seed, (uint32_t)(seed >> 32), (uint16_t*)dqkv);                                                 \
    PTC_CHECK_LAUNCH("attn_drop_bwd_dkv_kernel");                                                                                    \
    return PTC_OK;                                                                                                                   \
  }
  AD_BWD_CASE(false) AD_BWD_CASE(true)
#undef AD_BWD_CASE
  return PTC_EINVAL;   // not reached
}

// ------------------------------------------------------------------------------------------------
// head_dim 17..64 (attention_hd.h)
// ------------------------------------------------------------------------------------------------
static size_t hd_fwd_lds(int dk, int D, int lp_max) { return (size_t)dk * lp_max * 32 + (size_t)(D + 1) * (lp_max + 8) * 2 + AT_WAVES * 4; }
static size_t hd_dq_lds(int dk, int lp_max) { return (size_t)2 * dk * lp_max * 32; }
static size_t hd_dkv_lds(int dk, int lp_max) { return (size_t)2 * dk * lp_max * 32 + (size_t)lp_max * 8; }

extern "C" int ptc_attn_varlen_hd_supported(int head_dim, int max_seqlen) {
  if (head_dim < 17 || head_dim > 64 || max_seqlen < 1 || max_seqlen > AT_MAX_L) return 0;
  const int dk = (head_dim + 15) / 16, lp_max = (max_seqlen + 31) & ~31;
  return hd_fwd_lds(dk, head_dim, lp_max) <= AH_LDS_LIMIT && hd_dkv_lds(dk, lp_max) <= AH_LDS_LIMIT;
}

static int hd_check(const char* name, const void* qkv, const int32_t* cu, int64_t n_seq, int64_t total, int H, int D,
                    int max_seqlen, int dtype) {
  // PTC_F16 here = f16 OPERANDS (f16 MFMAs, P / dS rounded to f16: what flash-attn does with fp16 tensors, LitePT's call site) -- not the
  // f16 I/O around bf16 arithmetic of the head_dim-16 kernels
  int rc = check_common(name, qkv, cu, n_seq, total, H, max_seqlen, dtype, true);
  if (rc != PTC_OK) return rc;
  PTC_REQUIRE(ptc_attn_varlen_hd_supported(D, max_seqlen), PTC_EUNSUPPORTED,
              "%s: head_dim=%d with max_seqlen=%d is outside the LDS-resident range (17..32: 1024 keys, ..48: 672, ..64: 512)", name, D,
              max_seqlen);
  return PTC_OK;
}

extern "C" int ptc_attn_varlen_hd_fwd(const void* qkv, const int32_t* cu_seqlens, int64_t n_seq, int64_t total, int H, int head_dim,
                                      int max_seqlen, float softmax_scale, int dtype, void* out, float* lse, ptc_stream_t stream) {
  int rc = hd_check("ptc_attn_varlen_hd_fwd", qkv, cu_seqlens, n_seq, total, H, head_dim, max_seqlen, dtype);
  if (rc != PTC_OK) return rc;
  if (n_seq == 0 || total == 0) return PTC_OK;
  PTC_REQUIRE(out && lse, PTC_EINVAL, "ptc_attn_varlen_hd_fwd: null buffer");
  const int lp_max = (max_seqlen + 31) & ~31, dk = (head_dim + 15) / 16, mb = head_dim / 32 + 1;
  const size_t lds = hd_fwd_lds(dk, head_dim, lp_max);
  const int n_units = (int)(n_seq * H);
  const int qs = at_split_host(n_units, lp_max);
  hipStream_t s = (hipStream_t)stream;
#define AH_FWD_CASE(DK, MB, F16)                                                                                                    \
  if (dk == DK && mb == MB && (dtype == PTC_F16) == F16) {                                                                                                  \
    rc = allow_big_lds(attn_hd_fwd_kernel<DK, MB, F16>, lds);                                                                       \
    if (rc != PTC_OK) return rc;                                                                                               \
    hipLaunchKernelGGL((attn_hd_fwd_kernel<DK, MB, F16>), dim3((unsigned)(8 * ((n_units * qs + 7) / 8))), dim3(AT_THREADS), lds, s,  \
                       (const uint16_t*)qkv, cu_seqlens, H, head_dim, softmax_scale, total, lp_max, n_units, qs, (uint16_t*)out, \
                       lse);                                                                                                   \
    PTC_CHECK_LAUNCH("attn_hd_fwd_kernel");                                                                                    \
    return PTC_OK;                                                                                                             \
  }
  AH_FWD_CASE(2, 1, false) AH_FWD_CASE(2, 2, false) AH_FWD_CASE(3, 2, false) AH_FWD_CASE(4, 2, false) AH_FWD_CASE(4, 3, false)
  AH_FWD_CASE(2, 1, true) AH_FWD_CASE(2, 2, true) AH_FWD_CASE(3, 2, true) AH_FWD_CASE(4, 2, true) AH_FWD_CASE(4, 3, true)
#undef AH_FWD_CASE
  ptc_set_error("ptc_attn_varlen_hd_fwd: no instance for head_dim=%d", head_dim);
  return PTC_EUNSUPPORTED;
}

extern "C" int ptc_attn_varlen_hd_bwd(const void* qkv, const void* out, const void* dout, const float* lse, const int32_t* cu_seqlens,
                                      int64_t n_seq, int64_t total, int H, int head_dim, int max_seqlen, float softmax_scale,
                                      int dtype, void* dqkv, void* workspace, size_t workspace_bytes, ptc_stream_t stream) {
  int rc = hd_check("ptc_attn_varlen_hd_bwd", qkv, cu_seqlens, n_seq, total, H, head_dim, max_seqlen, dtype);
  if (rc != PTC_OK) return rc;
  if (n_seq == 0 || total == 0) return PTC_OK;
  PTC_REQUIRE(out && dout && lse && dqkv && workspace, PTC_EINVAL, "ptc_attn_varlen_hd_bwd: null buffer");
  PTC_REQUIRE(workspace_bytes >= ptc_attn_varlen_bwd_workspace_bytes(total, H), PTC_EWORKSPACE,
              "ptc_attn_varlen_hd_bwd: workspace too small");
  PTC_REQUIRE(((uintptr_t)out % 16 == 0) && ((uintptr_t)dout % 16 == 0) && ((uintptr_t)dqkv % 16 == 0), PTC_EINVAL,
              "ptc_attn_varlen_hd_bwd: buffers must be 16-byte aligned");
  const int lp_max = (max_seqlen + 31) & ~31, dk = (head_dim + 15) / 16;
  const int n_units = (int)(n_seq * H);
  const int qs = at_split_host(n_units, lp_max);
  const unsigned grid = (unsigned)(8 * ((n_units * qs + 7) / 8));
  hipStream_t s = (hipStream_t)stream;
  float* delta = (float*)workspace;
#define AH_BWD_CASE(DK, F16)                                                                                                        \
  if (dk == DK && (dtype == PTC_F16) == F16) {                                                                                                              \
    rc = allow_big_lds(attn_hd_bwd_dq_kernel<DK, F16>, hd_dq_lds(DK, lp_max));                                                      \
    if (rc != PTC_OK) return rc;                                                                                               \
    rc = allow_big_lds(attn_hd_bwd_dkv_kernel<DK, F16>, hd_dkv_lds(DK, lp_max));                                                    \
    if (rc != PTC_OK) return rc;                                                                                               \
    hipLaunchKernelGGL((attn_hd_bwd_dq_kernel<DK, F16>), dim3(grid), dim3(AT_THREADS), hd_dq_lds(DK, lp_max), s,                    \
                       (const uint16_t*)qkv, (const uint16_t*)out, (const uint16_t*)dout, lse, cu_seqlens, H, head_dim,        \
                       softmax_scale, total, lp_max, n_units, qs, (uint16_t*)dqkv, delta);                                     \
    PTC_CHECK_LAUNCH("attn_hd_bwd_dq_kernel");                                                                                 \
    hipLaunchKernelGGL((attn_hd_bwd_dkv_kernel<DK, F16>), dim3(grid), dim3(AT_THREADS), hd_dkv_lds(DK, lp_max), s,                  \
                       (const uint16_t*)qkv, (const uint16_t*)dout, lse, (const float*)delta, cu_seqlens, H, head_dim,         \
                       softmax_scale, total, lp_max, n_units, qs, (uint16_t*)dqkv);                                            \
    PTC_CHECK_LAUNCH("attn_hd_bwd_dkv_kernel");                                                                                \
    return PTC_OK;                                                                                                             \
  }
  AH_BWD_CASE(2, false) AH_BWD_CASE(3, false) AH_BWD_CASE(4, false) AH_BWD_CASE(2, true) AH_BWD_CASE(3, true) AH_BWD_CASE(4, true)
#undef AH_BWD_CASE
  ptc_set_error("ptc_attn_varlen_hd_bwd: no instance for head_dim=%d", head_dim);
  return PTC_EUNSUPPORTED;
}

// ---- the same operator with the 3-D rotary embedding of q and k fused into its prologue / epilogue (head_dim 18; attention_hd.h, ROPE) ----------
extern "C" int ptc_attn_varlen_hd_rope_supported(int head_dim, int max_seqlen) {
  if (head_dim != AH_RD) return 0;
  const int lp_max = (max_seqlen + 31) & ~31;
  return ptc_attn_varlen_hd_supported(head_dim, max_seqlen) && hd_dkv_lds(2, lp_max) + (size_t)AT_WAVES * 32 * AH_RD * 2 <= AH_LDS_LIMIT;
}

extern "C" int ptc_attn_varlen_hd_rope_fwd(const void* qkv, const int32_t* cu_seqlens, const float* xyz, const float* inv_freq, int64_t n_seq,
                                           int64_t total, int H, int head_dim, int max_seqlen, float softmax_scale, int dtype, void* out,
                                           float* lse, ptc_stream_t stream) {
  int rc = hd_check("ptc_attn_varlen_hd_rope_fwd", qkv, cu_seqlens, n_seq, total, H, head_dim, max_seqlen, dtype);
  if (rc != PTC_OK) return rc;
  PTC_REQUIRE(ptc_attn_varlen_hd_rope_supported(head_dim, max_seqlen), PTC_EUNSUPPORTED, "ptc_attn_varlen_hd_rope_fwd: head_dim=%d (18 only)", head_dim);
  if (n_seq == 0 || total == 0) return PTC_OK;
  PTC_REQUIRE(out && lse && xyz && inv_freq, PTC_EINVAL, "ptc_attn_varlen_hd_rope_fwd: null buffer");
  const int lp_max = (max_seqlen + 31) & ~31;
  const size_t lds = hd_fwd_lds(2, head_dim, lp_max);
  const int n_units = (int)(n_seq * H);
  const int qs = at_split_host(n_units, lp_max);
  const AhRope rp{xyz, inv_freq};
#define AH_RF_CASE(F16)                                                                                                                \
  if ((dtype == PTC_F16) == F16) {                                                                                                     \
    rc = allow_big_lds(attn_hd_fwd_kernel<2, 1, F16, true>, lds);                                                                      \
    if (rc != PTC_OK) return rc;                                                                                                       \
    hipLaunchKernelGGL((attn_hd_fwd_kernel<2, 1, F16, true>), dim3((unsigned)(8 * ((n_units * qs + 7) / 8))), dim3(AT_THREADS), lds,     \
                       (hipStream_t)stream, (const uint16_t*)qkv, cu_seqlens, H, head_dim, softmax_scale, total, lp_max, n_units, qs,  \
                       (uint16_t*)out, lse, rp);                                                                                       \
    PTC_CHECK_LAUNCH("attn_hd_fwd_kernel(rope)");                                                                                      \
    return PTC_OK;                                                                                                                     \
  }
  AH_RF_CASE(false) AH_RF_CASE(true)
#undef AH_RF_CASE
  return PTC_EINVAL;   // not reached
}

extern "C" int ptc_attn_varlen_hd_rope_bwd(const void* qkv, const void* out, const void* dout, const float* lse, const int32_t* cu_seqlens,
                                           const float* xyz, const float* inv_freq, int64_t n_seq, int64_t total, int H, int head_dim,
                                           int max_seqlen, float softmax_scale, int dtype, void* dqkv, void* workspace, size_t workspace_bytes,
                                           ptc_stream_t stream) {
  int rc = hd_check("ptc_attn_varlen_hd_rope_bwd", qkv, cu_seqlens, n_seq, total, H, head_dim, max_seqlen, dtype);
  if (rc != PTC_OK) return rc;
  PTC_REQUIRE(ptc_attn_varlen_hd_rope_supported(head_dim, max_seqlen), PTC_EUNSUPPORTED, "ptc_attn_varlen_hd_rope_bwd: head_dim=%d (18 only)", head_dim);
  if (n_seq == 0 || total == 0) return PTC_OK;
  PTC_REQUIRE(out && dout && lse && dqkv && workspace && xyz && inv_freq, PTC_EINVAL, "ptc_attn_varlen_hd_rope_bwd: null buffer");
  PTC_REQUIRE(workspace_bytes >= ptc_attn_varlen_bwd_workspace_bytes(total, H), PTC_EWORKSPACE, "ptc_attn_varlen_hd_rope_bwd: workspace too small");
  PTC_REQUIRE(((uintptr_t)out % 16 == 0) && ((uintptr_t)dout % 16 == 0) && ((uintptr_t)dqkv % 16 == 0), PTC_EINVAL,
              "ptc_attn_varlen_hd_rope_bwd: buffers must be 16-byte aligned");
  const int lp_max = (max_seqlen + 31) & ~31;
  const int n_units = (int)(n_seq * H);
  const int qs = at_split_host(n_units, lp_max);
  const unsigned grid = (unsigned)(8 * ((n_units * qs + 7) / 8));
  hipStream_t s = (hipStream_t)stream;
  float* delta = (float*)workspace;
  const size_t tile = (size_t)AT_WAVES * 32 * AH_RD * 2, lds_q = hd_dq_lds(2, lp_max) + tile, lds_kv = hd_dkv_lds(2, lp_max) + tile;
  const AhRope rp{xyz, inv_freq};
#define AH_RB_CASE(F16)                                                                                                                \
  if ((dtype == PTC_F16) == F16) {                                                                                                     \
    rc = allow_big_lds(attn_hd_bwd_dq_kernel<2, F16, true>, lds_q);                                                                    \
    if (rc != PTC_OK) return rc;                                                                                                       \
    rc = allow_big_lds(attn_hd_bwd_dkv_kernel<2, F16, true>, lds_kv);                                                                  \
    if (rc != PTC_OK) return rc;                                                                                                       \
    hipLaunchKernelGGL((attn_hd_bwd_dq_kernel<2, F16, true>), dim3(grid), dim3(AT_THREADS), lds_q, s, (const uint16_t*)qkv,              \
                       (const uint16_t*)out, (const uint16_t*)dout, lse, cu_seqlens, H, head_dim, softmax_scale, total, lp_max,        \
                       n_units, qs, (uint16_t*)dqkv, delta, rp);                                                                       \
    PTC_CHECK_LAUNCH("attn_hd_bwd_dq_kernel(rope)");                                                                                   \
    hipLaunchKernelGGL((attn_hd_bwd_dkv_kernel<2, F16, true>), dim3(grid), dim3(AT_THREADS), lds_kv, s, (const uint16_t*)qkv,            \
                       (const uint16_t*)dout, lse, (const float*)delta, cu_seqlens, H, head_dim, softmax_scale, total, lp_max,         \
                       n_units, qs, (uint16_t*)dqkv, rp);                                                                              \
    PTC_CHECK_LAUNCH("attn_hd_bwd_dkv_kernel(rope)");                                                                                  \
    return PTC_OK;                                                                                                                     \
  }
  AH_RB_CASE(false) AH_RB_CASE(true)
#undef AH_RB_CASE
  return PTC_EINVAL;   // not reached
}

// ------------------------------------------------------------------------------------------------
// relative-position-bias attention, head_dim 16 (attention_rpe.h)
// ------------------------------------------------------------------------------------------------
static int rpe_check(const char* name, const void* qkv, const int32_t* cu, const int32_t* gc, const float* table, int pos_bnd,
                     int64_t n_seq, int64_t total, int H, int max_seqlen, int dtype) {
  int rc = check_common(name, qkv, cu, n_seq, total, H, max_seqlen, dtype, true);
  if (rc != PTC_OK) return rc;
  PTC_REQUIRE(pos_bnd >= 0 && pos_bnd <= 4096, PTC_EINVAL, "%s: pos_bnd=%d out of range", name, pos_bnd);
  PTC_REQUIRE(n_seq == 0 || (gc && table), PTC_EINVAL, "%s: null buffer", name);
  return PTC_OK;
}

extern "C" int ptc_attn_rpe_fwd(const void* qkv, const int32_t* cu_seqlens, const int32_t* grid_coord, const float* rpe_table,
                                int pos_bnd, int64_t n_seq, int64_t total, int H, int max_seqlen, float softmax_scale, int dtype,
                                void* out, float* lse, ptc_stream_t stream) {
  int rc = rpe_check("ptc_attn_rpe_fwd", qkv, cu_seqlens, grid_coord, rpe_table, pos_bnd, n_seq, total, H, max_seqlen, dtype);
  if (rc != PTC_OK) return rc;
  if (n_seq == 0 || total == 0) return PTC_OK;
  PTC_REQUIRE(out && lse, PTC_EINVAL, "ptc_attn_rpe_fwd: null buffer");
  const int lp_max = (max_seqlen + 31) & ~31, R = 2 * pos_bnd + 1;
  const size_t lds = (size_t)lp_max * 32 + (size_t)17 * (lp_max + 8) * 2 + 64 + ar_extra_lds(lp_max, R);
  PTC_REQUIRE(lds <= AH_LDS_LIMIT, PTC_EUNSUPPORTED, "ptc_attn_rpe_fwd: max_seqlen=%d with pos_bnd=%d does not fit LDS", max_seqlen, pos_bnd);
  const int n_units = (int)(n_seq * H);
  // F16: the call site's casts (`qkv` of an fp16-autocast Linear -> bf16 operands, output back to fp16) in the load / store paths, as in
  // the flash-branch kernels: configs/s3dis/semseg-pt-v3m1-1-rpe.py runs under the reference's fp16 AMP
#define AR_FWD(F16)                                                                                                                 \
  {                                                                                                                                 \
    rc = allow_big_lds(attn_rpe_fwd_kernel<F16>, lds);                                                                              \
    if (rc != PTC_OK) return rc;                                                                                                    \
    hipLaunchKernelGGL(attn_rpe_fwd_kernel<F16>, dim3((unsigned)(8 * ((n_units + 7) / 8))), dim3(AT_THREADS), lds, (hipStream_t)stream, \
                       (const uint16_t*)qkv, cu_seqlens, grid_coord, rpe_table, R, pos_bnd, H, softmax_scale, total, lp_max, n_units, \
                       (uint16_t*)out, lse);                                                                                        \
  }
  if (dtype == PTC_F16) AR_FWD(true) else AR_FWD(false)
#undef AR_FWD
  PTC_CHECK_LAUNCH("attn_rpe_fwd_kernel");
  return PTC_OK;
}

extern "C" size_t ptc_attn_rpe_bwd_workspace_bytes(int64_t total, int H, int pos_bnd) {
  return ptc_attn_varlen_bwd_workspace_bytes(total, H) + ptc_align_up((size_t)3 * (2 * (size_t)(pos_bnd > 0 ? pos_bnd : 0) + 1) * H * 8, 256);
}

extern "C" int ptc_attn_rpe_bwd(const void* qkv, const void* out, const void* dout, const float* lse, const int32_t* cu_seqlens,
                                const int32_t* grid_coord, const float* rpe_table, int pos_bnd, int64_t n_seq, int64_t total, int H,
                                int max_seqlen, float softmax_scale, int dtype, void* dqkv, float* d_rpe_table, void* workspace,
                                size_t workspace_bytes, ptc_stream_t stream) {
  int rc = rpe_check("ptc_attn_rpe_bwd", qkv, cu_seqlens, grid_coord, rpe_table, pos_bnd, n_seq, total, H, max_seqlen, dtype);
  if (rc != PTC_OK) return rc;
  const int R = 2 * pos_bnd + 1;
  hipStream_t s = (hipStream_t)stream;
  if (n_seq == 0 || total == 0) {
    if (d_rpe_table) PTC_HIP(hipMemsetAsync(d_rpe_table, 0, (size_t)3 * R * H * sizeof(float), s));
    return PTC_OK;
  }
  PTC_REQUIRE(out && dout && lse && dqkv && d_rpe_table && workspace, PTC_EINVAL, "ptc_attn_rpe_bwd: null buffer");
  PTC_REQUIRE(workspace_bytes >= ptc_attn_rpe_bwd_workspace_bytes(total, H, pos_bnd), PTC_EWORKSPACE, "ptc_attn_rpe_bwd: workspace too small");
  unsigned long long* dt_fix = (unsigned long long*)((char*)workspace + ptc_attn_varlen_bwd_workspace_bytes(total, H));
  PTC_HIP(hipMemsetAsync(dt_fix, 0, (size_t)3 * R * H * 8, s));
  const int lp_max = (max_seqlen + 31) & ~31;
  const size_t lds_q = (size_t)lp_max * 64 + ar_extra_lds(lp_max, R), lds_kv = (size_t)lp_max * 72 + ar_extra_lds(lp_max, R);
  PTC_REQUIRE(lds_kv <= AH_LDS_LIMIT, PTC_EUNSUPPORTED, "ptc_attn_rpe_bwd: max_seqlen=%d with pos_bnd=%d does not fit LDS", max_seqlen, pos_bnd);
  const int n_units = (int)(n_seq * H);
  const unsigned grid = (unsigned)(8 * ((n_units + 7) / 8));
  float* delta = (float*)workspace;
#define AR_BWD(F16)                                                                                                                 \
  {                                                                                                                                 \
    rc = allow_big_lds(attn_rpe_bwd_dq_kernel<F16>, lds_q);                                                                         \
    if (rc != PTC_OK) return rc;                                                                                                    \
    rc = allow_big_lds(attn_rpe_bwd_dkv_kernel<F16>, lds_kv);                                                                       \
    if (rc != PTC_OK) return rc;                                                                                                    \
    hipLaunchKernelGGL(attn_rpe_bwd_dq_kernel<F16>, dim3(grid), dim3(AT_THREADS), lds_q, s, (const uint16_t*)qkv, (const uint16_t*)out, \
                       (const uint16_t*)dout, lse, cu_seqlens, grid_coord, rpe_table, R, pos_bnd, H, softmax_scale, total, lp_max,   \
                       n_units, (uint16_t*)dqkv, delta, dt_fix);                                                                    \
    PTC_CHECK_LAUNCH("attn_rpe_bwd_dq_kernel");                                                                                     \
    hipLaunchKernelGGL(attn_rpe_table_finish_kernel, dim3((unsigned)ptc_cdiv((int64_t)3 * R * H, 256)), dim3(256), 0, s,            \
                       (const unsigned long long*)dt_fix, (int64_t)3 * R * H, d_rpe_table);                                         \
    PTC_CHECK_LAUNCH("attn_rpe_table_finish_kernel");                                                                               \
    hipLaunchKernelGGL(attn_rpe_bwd_dkv_kernel<F16>, dim3(grid), dim3(AT_THREADS), lds_kv, s, (const uint16_t*)qkv,                 \
                       (const uint16_t*)dout, lse, (const float*)delta, cu_seqlens, grid_coord, rpe_table, R, pos_bnd, H,           \
                       softmax_scale, total, lp_max, n_units, (uint16_t*)dqkv);                                                     \
  }
  if (dtype == PTC_F16) AR_BWD(true) else AR_BWD(false)
#undef AR_BWD
  PTC_CHECK_LAUNCH("attn_rpe_bwd_dkv_kernel");
  return PTC_OK;
}
