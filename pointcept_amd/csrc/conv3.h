// conv3.h -- third-generation forward / dgrad kernel of the gather-table convolution for 16-bit
// features with c_in % 32 == 0 and c_out % 32 == 0 (every CPE convolution of PTv3, the gather-fused
// qkv backward, and every 3x3x3 / 2x2x2 convolution of SpUNet beyond the stem).  Included by spconv.hip.
//
// What was wrong with conv2 (profiles/r01_d): it restaged W through LDS once per table row, between two
// workgroup barriers, with an index computation (two integer divisions) per 16-byte vector and with
// nothing else in flight -- for the deep stages (N <= 50k rows, C >= 128: 27 x 256 x 256 weights per
// 64-row workgroup) that staging WAS the kernel: 28 launches of ~0.6 ms for ~20 GFLOP each.
//
// conv3:
//   * the contraction is flattened to v = k * c_in + c (the weight row [kv][c_in] of one output
//     channel is contiguous in the spconv layout) and cut into CHUNKS of 128 v's = four MFMA steps
//     of 32 channels; a chunk of W is 64 output channels x 256 B = 16 KB;
//   * W chunks run through a two-deep LDS pipeline: chunk c+1 is fetched into registers (4 x 16 B per
//     thread, addresses are one multiply-add) while chunk c is multiplied, stored, ONE barrier per chunk;
//   * the LDS image is in MFMA FRAGMENT ORDER ([tile][step][lane][8 channels], 1 KB per fragment, 64 B
//     of padding between fragments so the four steps written by a lane quad hit different banks): the
//     A-operand read is a lane-linear ds_read_b128, conflict-free by construction;
//   * gathered rows go HBM/L2 -> registers directly as B operands (no LDS), through a ring of four
//     step slots: the rows of (chunk c+1, step s) are requested right after (chunk c, step s) is
//     multiplied, i.e. one full chunk (64 MFMAs per wave) ahead; table entries another chunk ahead;
//   * a wave owns RT row tiles of 16 rows x all 64 output channels of the workgroup (RT = 4: 256 rows
//     per workgroup, W fragments reused by 4 row tiles; RT = 2 when the grid would not fill the chip);
//     row tiles with no valid neighbour at a step skip its MFMAs (wave-uniform branch);
//   * workgroups are numbered XCD-first (b % 8 = XCD): consecutive row blocks -- and the c_out tiles
//     of one row block -- share an L2, so neighbouring rows are fetched from HBM once per XCD.
// Output-stationary, no atomics, fixed summation order: bit-reproducible.
//
// (A wave-private LDS bounce with quad-coalesced gathers -- "BNC", round 2 -- measured equal at 64 -> 64 and 20-55 % slower at 96 / 128
//  input channels, profiles/r02_d_conv_stages_ops.txt: the rows still left L2 as half lines.  Removed; conv5.h is the form that coalesces.)
#pragma once

#ifndef C3_NT8
#define C3_NT8 1              // 0: timing A/B only (`python -m pointcept_amd.build --variant d_C3_NT8_0`)
#endif
#ifndef C3_NT8_MIN_WGS
#define C3_NT8_MIN_WGS 256    // fewest 256-row x 128-column workgroups for which the wide tile is used
#endif
#ifndef C3_NT8_SMALL
#define C3_NT8_SMALL 1        // 0: timing A/B only
#endif
#ifndef C3_NT8_SMALL_MIN_WGS
#define C3_NT8_SMALL_MIN_WGS 128
#endif
#ifndef C3_DEEP
#define C3_DEEP 1             // 0: timing A/B only (`python -m pointcept_amd.build --variant d_C3_DEEP_0`)
#endif
#ifndef C3_DEEP_MAX_WGS
#define C3_DEEP_MAX_WGS 1024  // most 128-row x 64-column workgroups for which the two-chunk ring is used: at most one per CU.  (Measured: 2 821 rows x 512 channels
                              // 138 -> 118 us; 12 115 x 256 and 50 360 x 128 -- 380 / 788 workgroups -- unchanged: those are bound by the address path of their
                              // gathers, not by latency, profiles/r04_m_ops_stages.txt)
#endif
#define C3_FRAG 1024
#define C3_FPAD 64
#define C3_BUF(NTILES) (4 * (NTILES) * (C3_FRAG + C3_FPAD))   // one W chunk: NTILES tiles x 4 steps

// DEEP (round 4, the deep stages of the indoor scenes: 2 800 .. 50 000 rows at 128 .. 512 channels -- fewer workgroups than two per CU):
// the gathers run TWO chunks ahead through two ring slots.  A 128-row x 64-column chunk is 32 MFMAs per wave = ~512 matrix-pipe cycles,
// against ~2 us of L2 latency under load: with one or two waves per SIMD the one-chunk ring stalled every chunk (stage 4, 2 821 rows x
// 512 channels: 138 us = 135 TF/s for a problem whose MFMAs take 24 us).
template <typename T, int RT, int KPC, int NTILES, bool GEN, bool IDENT = false, bool DEEP = false>
__global__ void __launch_bounds__(256, 2)
conv3_kernel(const T* __restrict__ in, const T* __restrict__ w, const float* __restrict__ bias,
             const int32_t* __restrict__ nbr, int64_t n_out, int kv, int c_in, int c_out, int n_rowblk, T* __restrict__ out,
             uint32_t in_bytes, const int32_t* __restrict__ skip_hcnt, int skip_max) {
  using M = Mma<T>;
  using frag = typename M::frag;
  const __amdgpu_buffer_rsrc_t in_buf = ptc_buf(in, in_bytes);
  constexpr int NT = NTILES * 16, BM = RT * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int ny = c_out / NT;
  const int nblk = n_rowblk * ny;
  const int per_xcd = (nblk + 7) >> 3;
  const int lb = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (lb >= nblk) return;
  const int rb = lb / ny, n0 = (lb - rb * ny) * NT;
  if (skip_hcnt) {   // follow-up of conv8 (round 6): 128-row blocks whose halo count is in [0, skip_max] were served there
    const int nb128 = (int)((n_out + 127) >> 7);
    bool mine = false;
#pragma unroll
    for (int q = 0; q < BM / 128; ++q) {
      const int b = rb * (BM / 128) + q;
      if (b < nb128) {
        const int c = skip_hcnt[b];
        mine |= !(c >= 0 && c <= skip_max);
      }
    }
    if (!mine) return;
  }
  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int64_t row0 = (int64_t)rb * BM + wave * (RT * 16);
  const int KV = kv * c_in;
  const int nchunks = (KV + 127) >> 7;

  // ---- W staging: thread -> (weight row, step); 4 x 16 B per chunk and pass (NT = 96: two passes of 64 rows)
  constexpr int WP = (NT + 63) / 64;
  const int qd = threadIdx.x & 3;
  const T* wsrc[WP];
  int wdst[WP];
  bool wthread[WP];
#pragma unroll
  for (int ps = 0; ps < WP; ++ps) {
    const int wrow = ps * 64 + (threadIdx.x >> 2);
    wthread[ps] = wrow < NT;
    const int wr = wthread[ps] ? wrow : 0;
    wsrc[ps] = w + (int64_t)(n0 + wr) * KV + qd * 32;
    const int prow = lds_row_of_channel<NTILES>(wr);
    wdst[ps] = ((prow >> 4) * 4 + qd) * (C3_FRAG + C3_FPAD) + (prow & 15) * 16;
  }
  constexpr int WSETS = DEEP ? 2 : 1;   // DEEP: W(c + 1) and W(c + 2) are in flight at the same time (register set = chunk parity)
  uint4 wreg[WSETS][WP][4];
  // SPLIT (NTILES = 8): buffer (c + 1) & 1 is free for the whole of chunk c (its readers passed the barrier that ended chunk c - 1), so
  // pass 0 of W(c + 1) is stored in the MIDDLE of chunk c and pass 1 loaded only then: the staging registers of one pass (16) instead
  // of two (32) are live across the chunk's MFMAs -- the 128-column instance sits at the 256-register cap
  constexpr bool SPLIT = NTILES == 8;
  // (pass range [p0, p1): the two-pass instances -- 96 / 128 output channels -- stage their passes at different points of a chunk,
  //  see SPLIT below, so that only ONE pass of staging registers is live across the MFMAs)
  auto wload_s = [&](auto wset, int c, int p0, int p1) {
    constexpr int WS = decltype(wset)::value;
#pragma unroll
    for (int ps = 0; ps < WP; ++ps) {
      if (ps < p0 || ps >= p1) continue;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        // UNCONDITIONAL load from a clamped address, zeroed by a select: a load under an exec-masked branch
        // makes the compiler's wait-count bookkeeping give up (s_waitcnt vmcnt(0) before every MFMA group: the
        // whole prefetch pipeline of this kernel was serialised, r01_aj ISA)
        const int v0 = c * 128 + qd * 32 + gq * 8;
        const bool ok = wthread[ps] && v0 < KV;
        const int vc = v0 < KV ? v0 : KV - 8;
        uint4 v = *reinterpret_cast<const uint4*>(wsrc[ps] - qd * 32 + vc);
        if (!ok) v = make_uint4(0, 0, 0, 0);
        wreg[WS][ps][gq] = v;
      }
    }
  };
  auto wload = [&](int c, int p0 = 0, int p1 = 4) { wload_s(ptc_int<0>{}, c, p0, p1); };
  auto wstore_s = [&](auto wset, int buf, int p0, int p1) {
    constexpr int WS = decltype(wset)::value;
#pragma unroll
    for (int ps = 0; ps < WP; ++ps) {
      if (ps < p0 || ps >= p1) continue;
      if (wthread[ps]) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) *reinterpret_cast<uint4*>(smem + buf * C3_BUF(NTILES) + wdst[ps] + gq * 256) = wreg[WS][ps][gq];
      }
    }
  };
  auto wstore = [&](int buf, int p0 = 0, int p1 = 4) { wstore_s(ptc_int<0>{}, buf, p0, p1); };

  // ---- gather ring
  // a ring slot holds the MFMA operand itself (lane l: row l & 15, piece l >> 4)
  constexpr int NSLOT = DEEP ? 2 : 1;
  frag ga[NSLOT][4][RT];
  bool anyv[NSLOT][4][RT];           // wave-level "any neighbour at this step"
  // KPC = 16 is the c_in = 8 form (the 6 -> 32 stems, padded to 8): a 32-slot MFMA step spans FOUR table rows, one per lane
  // group, so a lane keeps the entries of ITS table row of each step: ix[s][j] = entry of table row 16 c + 4 s + g
  constexpr int KI = KPC == 16 ? 4 : KPC;
  int32_t idxN[KI][RT], idxNN[KI][RT];
  const int lrow = r;       // tile row whose table entry this lane needs
  const int lpiece = g;     // 16-byte piece of the 64-byte step this lane loads
  auto load_idx = [&](int c, int32_t (&ix)[KI][RT]) {
    // GEN = false: c_in divides 128 or is a multiple of it, a chunk holds exactly KPC whole table rows
    // (or a slice of one); GEN = true (c_in = 96, 160, 192, ...): rows straddle chunks, everything by division
    const int kfirst = (GEN || KPC == 1) ? (c * 128) / c_in : c * KPC;
#pragma unroll
    for (int kk = 0; kk < KI; ++kk) {
      const int k = KPC == 16 ? kfirst + 4 * kk + lpiece : kfirst + kk;
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        const int64_t row = row0 + j * 16 + lrow;
        const bool ok = k < kv && row < n_out;
        // IDENT: the identity table of a dense row-wise GEMM (nn.Linear with a contraction wider than linear2's 256 channels:
        // PTv3's fc2 / proj / qkv of the 128..512-channel stages, K = 512 .. 2048) -- no table in memory
        const int32_t e = IDENT ? (int32_t)(row < n_out ? row : n_out - 1)
                                : nbr[(int64_t)(k < kv ? k : kv - 1) * n_out + (row < n_out ? row : n_out - 1)];   // always in bounds
        ix[kk][j] = ok ? e : -1;
      }
    }
  };
  auto issue = [&](int c, int s, const int32_t (&ix)[KI][RT], auto slot) {
    constexpr int SL = decltype(slot)::value;
    const int v0 = c * 128 + s * 32;                 // flattened contraction index of this step
    int kk, cbase;
    if constexpr (KPC == 16) {
      kk = s;                                        // this lane's table row of the step; its 8 channels are the whole input row
      cbase = 0;
    } else if constexpr (GEN) {
      const int k = v0 / c_in;
      kk = k - (c * 128) / c_in;
      cbase = v0 - k * c_in + lpiece * 8;
    } else {
      kk = (s * KPC) >> 2;
      cbase = v0 % c_in + lpiece * 8;
    }
#pragma unroll
    for (int j = 0; j < RT; ++j) {
      int32_t i;
      if constexpr (GEN) {
        i = ix[0][j];
#pragma unroll
        for (int q = 1; q < KPC; ++q) i = kk == q ? ix[q][j] : i;   // run-time kk: select instead of dynamic indexing
      } else {
        i = ix[kk][j];                                               // kk is a compile-time constant after unrolling
      }
      // one unconditional buffer load: absent neighbours are out-of-range offsets and come back as zeros
      ga[SL][s][j] = ld_frag_buf<T>(in_buf, i >= 0 ? ((uint32_t)i * (uint32_t)c_in + (uint32_t)cbase) * 2u : PTC_BUF_OOB);
      anyv[SL][s][j] = __builtin_amdgcn_ballot_w64(i >= 0) != 0;
    }
  };
  f32x4 acc[RT][NTILES];
  {
    f32x4 breg[NTILES];
    sc_bias_regs<NTILES>(bias, n0, g, breg);
#pragma unroll
    for (int j = 0; j < RT; ++j)
#pragma unroll
      for (int t = 0; t < NTILES; ++t) acc[j][t] = breg[t];
  }

  // ---- prologue: W(0) -> LDS, gathers of chunk 0 (DEEP: and of chunk 1) in flight, W(1) and the next table entries requested
  constexpr int AHEAD = DEEP ? 2 : 1;               // chunks between a gather and its MFMAs
  wload(0);
  load_idx(0, idxN);
  wstore(0);
#pragma unroll
  for (int s = 0; s < 4; ++s) issue(0, s, idxN, ptc_int<0>{});
  load_idx(1, idxN);
  if constexpr (DEEP) {
#pragma unroll
    for (int s = 0; s < 4; ++s) issue(1, s, idxN, ptc_int<1>{});
    load_idx(2, idxN);
  }
  static_assert(!(SPLIT && DEEP), "the two-pass W staging and the two-chunk ring are separate instances");
  if constexpr (DEEP) {           // W(1) -> set 1, W(2) -> set 0 (its first content, W(0), is in LDS already)
    wload_s(ptc_int<1>{}, 1, 0, 4);
    wload_s(ptc_int<0>{}, 2, 0, 4);
  } else if constexpr (SPLIT) wload(1, 0, 1); else wload(1);
  __syncthreads();

  // one chunk: multiply ring slot `slot`, refill it with chunk c + AHEAD, stage W(c + 1)
  auto chunk = [&](int c, auto slot) {
    constexpr int SL = decltype(slot)::value;
    const unsigned char* wb = smem + (c & 1) * C3_BUF(NTILES) + lane * 16;
    load_idx(c + AHEAD + 1, idxNN);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      // (NTILES = 8: the W fragments of a step in two halves of four -- 16 registers less live at the 256-register cap)
      constexpr int WH = NTILES > 6 ? NTILES / 2 : NTILES;
#pragma unroll
      for (int t0 = 0; t0 < NTILES; t0 += WH) {
        frag wf[WH];
#pragma unroll
        for (int t = 0; t < WH; ++t) wf[t] = *reinterpret_cast<const frag*>(wb + ((t0 + t) * 4 + s) * (C3_FRAG + C3_FPAD));
#pragma unroll
        for (int j = 0; j < RT; ++j) {
          if (anyv[SL][s][j]) {
#pragma unroll
            for (int t = 0; t < WH; ++t) acc[j][t0 + t] = M::mma(wf[t], ga[SL][s][j], acc[j][t0 + t]);
          }
        }
      }
      issue(c + AHEAD, s, idxN, slot);
      if constexpr (SPLIT) {
        if (s == 1) {
          wstore((c + 1) & 1, 0, 1);
          wload(c + 1, 1, 2);
        }
      }
    }
    if constexpr (DEEP) {         // W(c + 1) was requested two chunks ago into the set of ITS parity; that set then takes W(c + 3)
      wstore_s(ptc_int<1 - SL>{}, (c + 1) & 1, 0, 4);
      __syncthreads();
      wload_s(ptc_int<1 - SL>{}, c + 3, 0, 4);
    } else {
      if constexpr (SPLIT) wstore((c + 1) & 1, 1, 2); else wstore((c + 1) & 1);
      __syncthreads();
      if constexpr (SPLIT) wload(c + 2, 0, 1); else wload(c + 2);
    }
#pragma unroll
    for (int kk = 0; kk < KI; ++kk)
#pragma unroll
      for (int j = 0; j < RT; ++j) idxN[kk][j] = idxNN[kk][j];
  };
  if constexpr (DEEP) {
    int c = 0;
#pragma unroll 1
    for (; c + 1 < nchunks; c += 2) {
      chunk(c, ptc_int<0>{});
      chunk(c + 1, ptc_int<1>{});
    }
    if (c < nchunks) chunk(c, ptc_int<0>{});
  } else {
#pragma unroll 1
    for (int c = 0; c < nchunks; ++c) chunk(c, ptc_int<0>{});
  }

#pragma unroll
  for (int j = 0; j < RT; j += 2) {
    const int64_t rowA = row0 + j * 16 + r;
    sc_epilogue<T, NTILES>(*reinterpret_cast<f32x4(*)[2][NTILES]>(&acc[j]), nullptr, out, rowA, rowA + 16, n_out, c_out, n0, g);
  }
}

// dense row-wise GEMM out = in W^T + b on the same kernel (identity table): contractions wider than linear2's 256 channels
static inline bool conv3_dense_supported(int dtype, int kv, int c_in, int c_out, const int32_t* nbr) {
  // c_in % 128 != 0 (round 5: the MLP widths of PT-v3m2 / m3 / LitePT, 288 .. 2016): the GEN form, whose last 128-wide chunk is partial
  // (W zero-filled beyond c_in, gathers of table row 1 = out of range = zeros)
  return dtype != PTC_F32 && nbr == nullptr && kv == 1 && c_in > 256 && c_in % 32 == 0 && c_out % 32 == 0;
}

static inline bool conv3_supported(int dtype, int kv, int c_in, int c_out, const int32_t* nbr) {
  if (dtype == PTC_F32 || nbr == nullptr || kv < 2) return false;
  // c_in = 8: the stems (6 input channels padded to 8, k = 5): four table rows per MFMA step instead of one table row per
  // half-empty step in conv2 (685 -> see profiles/r02_am_stem.txt)
  if (c_in == 8) return c_out % 32 == 0;
  if (c_out % 32 != 0 || c_in % 32 != 0) return false;
  // (32 -> 32 / 32 -> 64 ... : conv5 takes every c_in = 32 / 64 shape before this test is reached)
  return !(c_in == 32 && c_out % 64 != 0 && c_out % 96 != 0);
}

// set around a launch_conv3 call by conv8's launcher (conv8.h): the 128-row blocks with a halo count in [0, c3_skip_max] are skipped
static thread_local const int32_t* c3_skip_hcnt = nullptr;
static thread_local int c3_skip_max = 0;

template <typename T, int RT, int KPC, int NTILES, bool GEN, bool IDENT = false, bool DEEP = false>
static int launch_conv3_i(const void* in, int64_t n_in, const void* w, const float* bias, const int32_t* nbr, int64_t n_out, int kv,
                          int c_in, int c_out, void* out, hipStream_t s) {
  const int n_rowblk = (int)ptc_cdiv(n_out, RT * 64);
  const int nblk = n_rowblk * (c_out / (NTILES * 16));
  const size_t lds = 2 * C3_BUF(NTILES);
  auto kern = conv3_kernel<T, RT, KPC, NTILES, GEN, IDENT, DEEP>;
  static size_t allowed = 48 * 1024;   // per instantiation
  if (lds > allowed) {
    PTC_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    allowed = lds;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(8 * ((nblk + 7) / 8))), dim3(256), lds, s, (const T*)in, (const T*)w, bias, nbr, n_out, kv,
                     c_in, c_out, n_rowblk, (T*)out, (uint32_t)((uint64_t)n_in * c_in * sizeof(T)), c3_skip_hcnt, c3_skip_max);
  PTC_CHECK_LAUNCH("conv3_kernel");
  return PTC_OK;
}

template <typename T>
static int launch_conv3(const void* in, int64_t n_in, const void* w, const float* bias, const int32_t* nbr, int64_t n_out, int kv,
                        int c_in, int c_out, void* out, hipStream_t s) {
  // 64 output channels per workgroup (96 when c_out is a multiple of 96 but not of 64 -- SpUNet's decoder --
  // else 32); 256-row workgroups when they still give every CU a workgroup, else 128-row ones
  int nt = c_out % 64 == 0 ? 4 : (c_out % 96 == 0 ? 6 : 2);
  // table rows one 128-wide chunk can touch: 1 (c_in % 128 == 0), 4 (c_in = 32), else 2
  const int kpc = c_in == 8 ? 16 : (c_in % 128 == 0 ? 1 : (c_in == 32 ? 4 : 2));
  // >= one 256-row workgroup per CU (r01_ag: N = 50k, C = 128: 79 vs 88 us).  96-wide tiles: 256 rows x 96 channels of accumulators
  // sit at the 256-register cap -- the c_in = 32 / 8 forms spilled 84-100 bytes per lane and are built with 128 rows only; the
  // c_in = 96 / 192 form keeps 256 rows with 16 bytes of scratch because it is FASTER than the spill-free 128-row one (SpUNet step
  // 28.4 vs 29.6 ms, profiles/r03_q_spunet_nt6_ab.txt); c_in % 128 == 0 does not spill
  const bool big = (nt != 6 || kpc <= 2) && ptc_cdiv(n_out, 256) * (c_out / (nt * 16)) >= 256;
  const bool gen = !(c_in == 8 || c_in == 32 || c_in == 64 || c_in % 128 == 0);   // table rows straddle chunks (kpc == 2)
#if C3_DEEP
  // few workgroups (the deep stages of indoor scenes): the 128-row x 64-column form with its gathers two chunks ahead (see DEEP above)
  if (kpc == 1 && nt == 4 && !big && ptc_cdiv(n_out, 128) * (c_out / 64) <= C3_DEEP_MAX_WGS) {
    if (nbr == nullptr) return launch_conv3_i<T, 2, 1, 4, false, true, true>(in, n_in, w, bias, nbr, n_out, kv, c_in, c_out, out, s);
    return launch_conv3_i<T, 2, 1, 4, false, false, true>(in, n_in, w, bias, nbr, n_out, kv, c_in, c_out, out, s);
  }
#endif
#if C3_NT8
  // 128 output channels per workgroup at c_in % 128 == 0 (round 4): the gathered rows are the B operand of TWICE as many MFMAs.  The
  // kernel is bound by the address path of its gathers at 64 columns (a 1-KB wave gather in B-operand layout costs ~57 address cycles,
  // profiles/r02_e_probe_gather.txt: 8 waves x 4 row tiles per step = 1824 cycles per CU against 1024 matrix-pipe cycles per SIMD),
  // so the same gathers feeding 128 columns halve the time per flop; RT = 4 keeps the LDS reads of W at half the LDS bandwidth
  // (PTC_C3_NT8_MIN_WGS: the CPU emulation tier and tools/conv_kernels.py lower the threshold to reach the instance at test sizes)
  static const long nt8_min = getenv("PTC_C3_NT8_MIN_WGS") ? atol(getenv("PTC_C3_NT8_MIN_WGS")) : C3_NT8_MIN_WGS;
  const bool nt8 = kpc == 1 && c_out % 128 == 0 && ptc_cdiv(n_out, 256) * (c_out / 128) >= nt8_min;
  if (nt8) {
    if (nbr == nullptr) return launch_conv3_i<T, 4, 1, 8, false, true>(in, n_in, w, bias, nbr, n_out, kv, c_in, c_out, out, s);
    return launch_conv3_i<T, 4, 1, 8, false>(in, n_in, w, bias, nbr, n_out, kv, c_in, c_out, out, s);
  }
#if C3_NT8_SMALL
  // the same wide tile on 128-row workgroups for the mid-sized problems (indoor stages 2 / 3: 12 000 .. 50 000 rows): they are bound by the
  // address path too (8 waves x 54 chunks x 8 gathers x ~57 cycles per CU = 85 us at 12 115 rows x 256 channels, measured 99 us), and
  // the 128-column tile halves the gathers per flop
  if (kpc == 1 && c_out % 128 == 0 && ptc_cdiv(n_out, 128) * (c_out / 128) >= C3_NT8_SMALL_MIN_WGS) {
    if (nbr == nullptr) return launch_conv3_i<T, 2, 1, 8, false, true>(in, n_in, w, bias, nbr, n_out, kv, c_in, c_out, out, s);
    return launch_conv3_i<T, 2, 1, 8, false>(in, n_in, w, bias, nbr, n_out, kv, c_in, c_out, out, s);
  }
#endif
#endif
  if (nbr == nullptr && gen) {   // dense GEMM whose contraction is not a multiple of the 128-wide chunk: identity table + general chunking
#define C3_IDG_CASE(N)                                                                                                       \
  if (nt == N)                                                                                                             \
    return (big && N != 6) ? launch_conv3_i<T, 4, 2, N, true, true>(in, n_in, w, bias, nbr, n_out, kv, c_in, c_out, out, s)      \
                           : launch_conv3_i<T, 2, 2, N, true, true>(in, n_in, w, bias, nbr, n_out, kv, c_in, c_out, out, s);
    C3_IDG_CASE(4) C3_IDG_CASE(2) C3_IDG_CASE(6)
#undef C3_IDG_CASE
  }
  if (nbr == nullptr) {   // dense GEMM (kv = 1, c_in % 128 == 0): identity-table instances
#define C3_ID_CASE(N)                                                                                                        \
  if (nt == N)                                                                                                             \
    return big ? launch_conv3_i<T, 4, 1, N, false, true>(in, n_in, w, bias, nbr, n_out, kv, c_in, c_out, out, s)               \
               : launch_conv3_i<T, 2, 1, N, false, true>(in, n_in, w, bias, nbr, n_out, kv, c_in, c_out, out, s);
    C3_ID_CASE(4) C3_ID_CASE(2) C3_ID_CASE(6)
#undef C3_ID_CASE
    ptc_set_error("conv3 (dense): c_in=%d c_out=%d unsupported", c_in, c_out);
    return PTC_EUNSUPPORTED;
  }
#define C3_CASE(K, N, G)                                                                                                   \
  if (kpc == K && nt == N && gen == G)                                                                                     \
    return big ? launch_conv3_i<T, 4, K, N, G>(in, n_in, w, bias, nbr, n_out, kv, c_in, c_out, out, s)                           \
               : launch_conv3_i<T, 2, K, N, G>(in, n_in, w, bias, nbr, n_out, kv, c_in, c_out, out, s);
  C3_CASE(1, 4, false) C3_CASE(2, 4, false) C3_CASE(4, 4, false) C3_CASE(2, 4, true)
  C3_CASE(1, 2, false) C3_CASE(2, 2, false) C3_CASE(2, 2, true) C3_CASE(4, 2, false)
  C3_CASE(16, 2, false) C3_CASE(16, 4, false)
  C3_CASE(1, 6, false) C3_CASE(2, 6, false) C3_CASE(2, 6, true)
#undef C3_CASE
#define C3_CASE6(K, G)                                                                                                     \
  if (kpc == K && nt == 6 && gen == G) return launch_conv3_i<T, 2, K, 6, G>(in, n_in, w, bias, nbr, n_out, kv, c_in, c_out, out, s);
  C3_CASE6(4, false) C3_CASE6(16, false)
#undef C3_CASE6
  ptc_set_error("conv3: c_in=%d c_out=%d unsupported", c_in, c_out);
  return PTC_EUNSUPPORTED;
}
