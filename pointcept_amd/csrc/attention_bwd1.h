// One-pass backward of the window attention, head_dim 16 (round 4; included by attention.hip) -- ptv3m1:208-214 under autograd.
//
// The two-kernel form evaluates S', exp2 and dP for every (query tile, key tile) pair TWICE (once query-stationary for dQ, once
// key-stationary for dK / dV); its time is VALU + matrix cycles of both evaluations (profiles/r04_u_attn_pmc.json: 57 + 73 vector
// instructions and 128 + 160 matrix cycles per pair, no overlap between the two pipes).  Here a workgroup owns one (sequence, head) unit
// whole: wave w keeps key tiles w, w + 8, w + 16, w + 24 stationary (K hi / lo, V, K^T operands and the dK / dV accumulators: 128
// registers of its 256), walks the query tiles once and evaluates every pair ONCE:
//   * S', dP, P, dS as in the key-stationary kernel (lane = key, registers = queries; lse / delta as fp32 C operands from LDS, read
//     once per query tile for the wave's four key tiles);
//   * dV += P^T dO and dK += dS^T Q through the four v_permlane16_swap per operand, as before;
//   * dQ needs the contraction over KEYS, i.e. dS with lane = query: the packed dS tile goes through a wave-private 2 KB LDS slice
//     ([key][query] rows of 64 bytes, 8-byte chunks XOR-swizzled by (key >> 1) & 7: the four ds_write_b64 and the four
//     ds_read_b64_tr_b16 that read it back transposed are both conflict-free) and multiplies the wave's stationary K^T operand in two
//     16x16x32 MFMAs per pair;
//   * the dQ partials of the 8 waves (each the sum over its own key tiles) meet in LDS once per query tile: every wave writes its
//     32 x 16 fp32 partial, ONE barrier, and two waves (rotating with the query tile) add the eight partials in wave order and store
//     the bf16 rows.  No atomics, fixed summation order: bit-reproducible.  The partial buffers are double-buffered, so the barrier
//     of tile qt + 1 is the only other synchronisation the reducers need.
//   * delta = rowsum(dO * O) is computed while dO is staged: no workspace round trip, no second launch.
// Per pair: 3 x 32 + 6 x 16 = 192 matrix cycles (288 before) and one evaluation of the 16 exponentials / products / packs: 51 vector
// instructions (130 before).
// Measured on the MI355X (800 x 4 x 1024 x 16, event-timed, profiles/r04_y_* / r04_z_*): 1105-1108 us against 1335-1386 us for the two
// kernels, and 176.4 -> 182.0 scenes/s on the bench step.  Still far from the instruction floor (~120 ns per pair and SIMD against 346):
// two waves per SIMD (256 registers, 136 KB of LDS) do not cover the dependency chain S' -> exp2 -> pack -> swap -> product, vector
// pipe 54 % busy, matrix pipe 24 % (r04_z_roof_pmc.json).  Timing probes with parts cut out: the per-query-tile barrier + reduction
// costs 8 % of the kernel, the dS round trip + dQ products 8 %.  Measured and dropped: a software pipeline that issues S' of pair j + 1
// and dQ of pair j - 1 ahead of pair j's vector work (1136 us; with a sched_group_barrier sequence 1185 us: the VALU -> MFMA operand
// hazards cost more s_nop than the shadows hide), the form without the deferral below (1186 us), the next query tile's operands requested
// before the barrier of the current one (1150 against 1100 us in one session, profiles/r04_zt_attn_one_pass_prefetch_setprio.txt: two
// more registers spill inside the loop), s_setprio 1 on the second-dispatched half of the waves (no change), a barrier that waits for LDS traffic
// only instead of __syncthreads' vmcnt(0) (1110 vs 1103 us: no change).
// LDS: Q [lp][16] | dO [lp][16] | -lse log2 e [lp] | -delta [lp] | dS slices 8 x 2 x 2 KB | dQ partials 2 x 8 x 2 KB = 72 lp + 64 KB (136 KB at
// 1024): one workgroup of 8 waves per CU, two waves per SIMD.
#pragma once

#define AT1_KT 4                     // key tiles per wave: 8 waves x 4 x 32 = AT_MAX_L keys
#ifndef AT1_MIN_UNITS
#define AT1_MIN_UNITS 176            // (sequence, head) units from which a launch takes this kernel (one workgroup per CU): 192 units 80 us
                                     // against 91 for the split kernels, 128 units 76 against 65 (profiles/r04_zc_attn_small_launches.txt)
#endif
#define AT1_SLICE 2048               // bytes of a wave's dS slice and of one 32 x 16 fp32 dQ partial
#ifndef AT1_PIPE
#define AT1_PIPE 1                   // waves that own four key tiles defer the dQ products of a pair behind the next pair's S' / dP (0: timing A/B)
#endif
static size_t bwd1_lds_bytes(int lp_max) { return (size_t)lp_max * 72 + (size_t)AT_WAVES * AT1_SLICE * 4; }

// byte offset of the 8-byte chunk `c` (queries 4 c .. 4 c + 3) of key row `key` in a dS slice
__device__ __forceinline__ int at1_ds_off(int key, int c) { return key * 64 + ((c ^ ((key >> 1) & 7)) << 3); }

template <bool F16>
__global__ void __launch_bounds__(AT_THREADS, 2)
attn_bwd1_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ out, const uint16_t* __restrict__ dout,
                 const float* __restrict__ lse, const int32_t* __restrict__ cu, int H, float scale, int64_t total, int lp_max, int n_units,
                 uint16_t* __restrict__ dqkv) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int unit = at_unit(n_units);
  if (unit >= n_units) return;
  const int seq = unit / H, head = unit % H;
  const int a = cu[seq], L = cu[seq + 1] - a;
  if (L <= 0) return;
  const int Lp = (L + 31) & ~31, n_tiles = Lp >> 5;
  const int64_t rs = (int64_t)3 * H * 16;
  if (Lp > lp_max) {   // see attn_fwd_kernel
    for (int j = 0; j < 3; ++j) at_poison_rows<F16>(dqkv + qkv_off(a, j, H, head), rs, L, nullptr);
    return;
  }
  unsigned char* Qsm = smem;
  unsigned char* Dsm = smem + (size_t)lp_max * 32;
  float* nl = reinterpret_cast<float*>(smem + (size_t)lp_max * 64);            // -lse * log2 e per query
  float* nd = nl + lp_max;                                                     // -delta per query
  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  unsigned char* slice = smem + (size_t)lp_max * 72 + (size_t)wave * (2 * AT1_SLICE);   // two dS slices per wave
  unsigned char* red = smem + (size_t)lp_max * 72 + (size_t)AT_WAVES * (2 * AT1_SLICE);  // [2][AT_WAVES][32 queries][16] fp32

  stage_row_major<F16>(qkv + qkv_off(a, 0, H, head), rs, L, Lp, Qsm);
  for (int row = threadIdx.x; row < Lp; row += AT_THREADS) {
    uint4 d0 = {0, 0, 0, 0}, d1 = d0, o0 = d0, o1 = d0;
    float l2 = AT_PAD_LSE;
    if (row < L) {
      const int64_t orow = ((int64_t)(a + row) * H + head) * 16;
      const uint4* pd = reinterpret_cast<const uint4*>(dout + orow);
      const uint4* po = reinterpret_cast<const uint4*>(out + orow);
      d0 = at_in<F16>(pd[0]); d1 = at_in<F16>(pd[1]);
      o0 = at_in<F16>(po[0]); o1 = at_in<F16>(po[1]);
      l2 = lse[(int64_t)head * total + a + row] * AT_LOG2E;
    }
    *reinterpret_cast<uint4*>(Dsm + rm_off(row, 0)) = d0;
    *reinterpret_cast<uint4*>(Dsm + rm_off(row, 1)) = d1;
    // delta in the order of the two-kernel form: the two 8-channel halves summed left to right, then added
    const uint32_t wd[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w}, wo[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
    float dl[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      dl[j >> 2] += __uint_as_float(wd[j] << 16) * __uint_as_float(wo[j] << 16);
      dl[j >> 2] += __uint_as_float(wd[j] & 0xffff0000u) * __uint_as_float(wo[j] & 0xffff0000u);
    }
    nl[row] = -l2;
    nd[row] = -(dl[0] + dl[1]);
  }

  const int col = lane & 31, h2 = lane >> 5;
  const int g = lane >> 4, n = lane & 15;
  const float c = scale * AT_LOG2E;
  const TrAddr16 ta = tr_addr16(lane);
  const int rmo = rm_off(col, h2);
  const int n_own = n_tiles > wave ? (n_tiles - wave + AT_WAVES - 1) / AT_WAVES : 0;      // key tiles wave, wave + 8, ... below n_tiles

  // the wave's stationary side
  s16x8 khi[AT1_KT], klo[AT1_KT], vf[AT1_KT], kta[AT1_KT];
  at_f32x4 dk0[AT1_KT], dk1[AT1_KT], dv0[AT1_KT], dv1[AT1_KT];
#pragma unroll
  for (int j = 0; j < AT1_KT; ++j) {
    const at_f32x4 z = {0.f, 0.f, 0.f, 0.f};
    dk0[j] = dk1[j] = dv0[j] = dv1[j] = z;
    const int kt = wave + AT_WAVES * j;
    const int key = kt * 32 + col;
    const bool kv = j < n_own && key < L;
    const s16x8 kf = ld_global_frag<F16>(qkv + qkv_off(a + key, 1, H, head) + h2 * 8, kv);
    vf[j] = ld_global_frag<F16>(qkv + qkv_off(a + key, 2, H, head) + h2 * 8, kv);
    split_scaled(kf, c, khi[j], klo[j]);
    // K^T as the 16x16x32 A operand of dQ^T += K^T dS^T: lane (channel n, slot group g) holds keys 32 kt + 8 g + 0..7 -- the order in
    // which the transposed reads below deliver the keys of the dS slice
    uint32_t w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      uint32_t two = 0;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int k2 = kt * 32 + 8 * g + 2 * e + b;
        uint32_t v = 0;
        if (j < n_own && k2 < L) {
          v = qkv[qkv_off(a + k2, 1, H, head) + n];
          if constexpr (F16) v = at_f16x2_to_bf16x2(v) & 0xffffu;
        }
        two |= v << (16 * b);
      }
      w[e] = two;
    }
    kta[j] = make_frag(w[0], w[1], w[2], w[3]);
  }
  // per-lane addresses of the dS round trip
  int wr[4], rd[2][2];
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) wr[r4] = at1_ds_off(col, 2 * r4 + h2);              // queries 8 r4 + 4 h2 + 0..3 of key `col`
#pragma unroll
  for (int t = 0; t < 2; ++t) {                                                        // query half t: lane reads key row 8 g + (n >> 2) (+ 4), chunk 4 t + (n & 3)
    rd[t][0] = at1_ds_off(8 * g + (n >> 2), 4 * t + (n & 3));
    rd[t][1] = at1_ds_off(8 * g + 4 + (n >> 2), 4 * t + (n & 3));
  }
  __syncthreads();

  // the operands of a query tile: row-major and transposed fragments of Q / dO, -lse / -delta of the lane's 16 queries (four runs of
  // four consecutive rows, broadcast reads) as the C operands of S' / dP
  s16x8 qf, dof, qtf, dotf;
  f32x16 cl, cd;
  auto load_q_tile = [&](int qt) {
    const int o = qt * 1024;
    qf = *reinterpret_cast<const s16x8*>(Qsm + rmo + o);
    dof = *reinterpret_cast<const s16x8*>(Dsm + rmo + o);
    qtf = ld_tr_pair(Qsm + ta.lo + o, Qsm + ta.hi + o);                    // Q^T[d][query slots]
    dotf = ld_tr_pair(Dsm + ta.lo + o, Dsm + ta.hi + o);                   // dO^T[d][query slots]
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const at_f32x4 l4 = *reinterpret_cast<const at_f32x4*>(nl + qt * 32 + 8 * r4 + 4 * h2);
      const at_f32x4 d4 = *reinterpret_cast<const at_f32x4*>(nd + qt * 32 + 8 * r4 + 4 * h2);
#pragma unroll
      for (int e = 0; e < 4; ++e) { cl[4 * r4 + e] = l4[e]; cd[4 * r4 + e] = d4[e]; }
    }
  };
  for (int qt = 0; qt < n_tiles; ++qt) {
    load_q_tile(qt);
    at_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;    // dQ^T[d = 4 g + e][q = 32 qt + (0 | 16) + n], summed over the wave's key tiles
    if (AT1_PIPE && n_own == AT1_KT) {
      // full windows: one straight-line body for the wave's four pairs.  The dS tile of pair j goes to slice j & 1 and its transposed reads
      // + dQ products are issued inside pair j + 1, behind that pair's S' / dP products: the LDS round trip (stores, transposed
      // reads) is covered by matrix work instead of being waited for, and only the last pair's is exposed.
      s16x8 b0 = qf, b1 = qf;
#pragma unroll
      for (int j = 0; j <= AT1_KT; ++j) {
        f32x16 s, dp;
        if (j < AT1_KT) {
          s = mfma32(qf, khi[j], cl);
          s = mfma32(qf, klo[j], s);
          dp = mfma32(dof, vf[j], cd);
        }
        if (j > 0) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const unsigned char* sl = slice + ((j - 1) & 1) * AT1_SLICE;
          b0 = ld_tr_pair(sl + rd[0][0], sl + rd[0][1]);
          b1 = ld_tr_pair(sl + rd[1][0], sl + rd[1][1]);
        }
        if (j < AT1_KT) {
          uint32_t pp[8], ps[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float p0 = __builtin_amdgcn_exp2f(s[2 * i]), p1 = __builtin_amdgcn_exp2f(s[2 * i + 1]);
            pp[i] = pack_bf16x2(p0, p1);
            ps[i] = pack_bf16x2(p0 * dp[2 * i], p1 * dp[2 * i + 1]);
          }
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4)
            *reinterpret_cast<uint2*>(slice + (j & 1) * AT1_SLICE + wr[r4]) = make_uint2(ps[2 * r4], ps[2 * r4 + 1]);
          s16x8 p0f, p1f, s0f, s1f;
          at_to_b16(pp, p0f, p1f);
          at_to_b16(ps, s0f, s1f);
          dv0[j] = mfma16(dotf, p0f, dv0[j]);
          dv1[j] = mfma16(dotf, p1f, dv1[j]);
          dk0[j] = mfma16(qtf, s0f, dk0[j]);
          dk1[j] = mfma16(qtf, s1f, dk1[j]);
        }
        if (j > 0) {
          acc0 = mfma16(kta[j - 1], b0, acc0);
          acc1 = mfma16(kta[j - 1], b1, acc1);
        }
      }
    } else {
#pragma unroll
    for (int j = 0; j < AT1_KT; ++j) {
      if (j < n_own) {
        f32x16 s = mfma32(qf, khi[j], cl);                 // S'[q][key] = q.(k c) - lse (exp2 domain): lane = key, registers = queries crow(r, h2)
        s = mfma32(qf, klo[j], s);
        const f32x16 dp = mfma32(dof, vf[j], cd);          // dP[q][key] - delta[q]
        uint32_t pp[8], ps[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float p0 = __builtin_amdgcn_exp2f(s[2 * i]), p1 = __builtin_amdgcn_exp2f(s[2 * i + 1]);
          pp[i] = pack_bf16x2(p0, p1);
          ps[i] = pack_bf16x2(p0 * dp[2 * i], p1 * dp[2 * i + 1]);
        }
        // dS -> the wave's slice ([key][query]); the previous pair's transposed reads were issued before these stores (LDS is in order
        // per wave)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) *reinterpret_cast<uint2*>(slice + wr[r4]) = make_uint2(ps[2 * r4], ps[2 * r4 + 1]);
        s16x8 p0f, p1f, s0f, s1f;
        at_to_b16(pp, p0f, p1f);                           // P as B operands: keys 0-15 | 16-31 of the tile, all 32 queries each
        at_to_b16(ps, s0f, s1f);
        dv0[j] = mfma16(dotf, p0f, dv0[j]);
        dv1[j] = mfma16(dotf, p1f, dv1[j]);
        dk0[j] = mfma16(qtf, s0f, dk0[j]);
        dk1[j] = mfma16(qtf, s1f, dk1[j]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const s16x8 b0 = ld_tr_pair(slice + rd[0][0], slice + rd[0][1]);   // dS^T[key slots][q = n]
        const s16x8 b1 = ld_tr_pair(slice + rd[1][0], slice + rd[1][1]);   //                  [q = 16 + n]
        acc0 = mfma16(kta[j], b0, acc0);
        acc1 = mfma16(kta[j], b1, acc1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    }
    // the wave's partial: row q = 16 t + n, channels 4 g .. 4 g + 3; the 16-byte piece sits at position g ^ (q & 3) of its row
    unsigned char* rp = red + (size_t)((qt & 1) * AT_WAVES + wave) * AT1_SLICE;
    *reinterpret_cast<at_f32x4*>(rp + n * 64 + ((g ^ (n & 3)) << 4)) = acc0;
    *reinterpret_cast<at_f32x4*>(rp + (16 + n) * 64 + ((g ^ (n & 3)) << 4)) = acc1;
    __syncthreads();
    if ((wave >> 1) == (qt & 3)) {                        // two waves add the eight partials of this query tile, in wave order
      const int t2 = (wave & 1) * 64 + lane, q = t2 >> 2, d4 = t2 & 3;
      const unsigned char* r0 = red + (size_t)((qt & 1) * AT_WAVES) * AT1_SLICE + q * 64 + ((d4 ^ (q & 3)) << 4);
      at_f32x4 sum = *reinterpret_cast<const at_f32x4*>(r0);
#pragma unroll
      for (int w = 1; w < AT_WAVES; ++w) {
        const at_f32x4 v = *reinterpret_cast<const at_f32x4*>(r0 + w * AT1_SLICE);
#pragma unroll
        for (int e = 0; e < 4; ++e) sum[e] += v[e];
      }
      const int qq = qt * 32 + q;
      if (qq < L) {
        uint2 w2;
        w2.x = at_out<F16>(pack_bf16x2(sum[0] * scale, sum[1] * scale));
        w2.y = at_out<F16>(pack_bf16x2(sum[2] * scale, sum[3] * scale));
        *reinterpret_cast<uint2*>(dqkv + qkv_off(a + qq, 0, H, head) + 4 * d4) = w2;
      }
    }
  }
  // lane (n, g) holds channels 4 g .. 4 g + 3 of keys 32 kt + n (.0) and 32 kt + 16 + n (.1)
#pragma unroll
  for (int j = 0; j < AT1_KT; ++j) {
    if (j >= n_own) continue;
    const int kt = wave + AT_WAVES * j;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int kk = kt * 32 + 16 * t + n;
      if (kk < L) {
        const at_f32x4 vk = t ? dk1[j] : dk0[j], vv = t ? dv1[j] : dv0[j];
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
