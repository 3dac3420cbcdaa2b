// wgrad3.h -- wgrad2 (weight gradient of the gather-table convolution, 16-bit features) with COMPACTED gathers, for the
// instances with 32 or 64 input channels in one channel block: (COT, CIT, KG) = (4, 4, 2), (4, 2, 4), (2, 2, 4).
// Included by spconv.hip.
// Candidate, OFF by default (PTC_WGRAD3=1): developed on the host emulation after round 2's GPU time was spent; bit-identical
// to wgrad2 there, not timed yet.
//
// Why (profiles/r02_emu_conv_work_counts.txt): per 32-row step and table row wgrad2 issues CIT = 4 table-entry loads
// and 4 gather instructions (8 rows of 128 B each) whatever the table holds; 56 % of the gathered dwords are "no neighbour"
// lanes and the vector-memory path is paid per instruction.  Here, as in conv6.h:
//   * lane = slot: the step's KG x 32 (table row, row) slots are one (KG = 2) or two (KG = 4) per lane -- one or two entry
//     loads per step instead of KG * CIT;
//   * ballot + prefix count rank the present slots into a wave-private LDS list; gather instruction q takes the pairs of rank
//     8 q .. 8 q + 7 (eight lanes per 128-byte row; 16 pairs of 64-byte rows at 32 channels): W3_Q unconditional
//     instructions cover half of the slots (a third is expected), a second round under a wave-uniform branch takes the rest;
//   * the contraction runs over ROWS, so a row without a neighbour must READ as zero through ds_read_b64_tr_b16 (a lane's
//     fragment holds 8 rows of one channel: no per-row mask at the fragment).  The images start zeroed, and the lanes that
//     wrote a row clear it again after the step's MFMAs;
//   * dout staging, fragments, MFMA order, cross-wave sum and partial layout are wgrad2's: the same operands in the same
//     order, bit-identical partials.
#pragma once

template <typename T, int COT, int CIT, int KG>
__global__ void __launch_bounds__(256, 2)
wgrad3_kernel(const T* __restrict__ in, const T* __restrict__ dout, const int32_t* __restrict__ nbr, int64_t n_out, int kv, int c_out,
              int64_t steps_total, float* __restrict__ partial, int gx, int groups, int nblocks, uint32_t in_bytes, uint32_t dout_bytes) {
  using M = Mma<T>;
  constexpr int C_IN = CIT * 16;
  constexpr int SLOTS = KG * W2_ROWS, NSTEP = SLOTS / 64;   // (table row, row) slots per step; slots per lane
  constexpr int PCS = 2 * CIT, PPI = 64 / PCS;              // 16-byte pieces per row; pairs per gather instruction
  constexpr int W3_Q = SLOTS / (2 * PPI);                   // unconditional gather instructions per round: half of the slots, so that
                                                            // two rounds always suffice (4 in the instances in use)
  constexpr int IMG_BYTES = (COT + KG * CIT) * W2_PLANE;   // dout image + the two `in` images
  constexpr int WAVE_BYTES = IMG_BYTES + SLOTS * 8;        // + [SLOTS] slot | [SLOTS] entry
  const __amdgpu_buffer_rsrc_t in_buf = ptc_buf(in, in_bytes), dout_buf = ptc_buf(dout, dout_bytes);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned char* D = smem + wave * WAVE_BYTES;
  unsigned char* I0 = D + COT * W2_PLANE;
  int32_t* list = reinterpret_cast<int32_t*>(D + IMG_BYTES);
  // workgroup numbering of wgrad2 (the table-row groups of one row range share an XCD)
  const int total = gx * groups * nblocks;
  const int per_xcd = (total + 7) >> 3;
  const int lid = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (lid >= total) return;
  const int bgrp = lid % groups, bz = (lid / groups) % nblocks, bx = lid / (groups * nblocks);
  const int k0 = bgrp * KG;
  const int nk = (kv - k0) < KG ? (kv - k0) : KG;
  const int co0 = bz * COT * 16;
  const int64_t workers = (int64_t)gx * 4, worker = (int64_t)bx * 4 + wave;

  f32x4 acc[KG][COT][CIT];
#pragma unroll
  for (int kk = 0; kk < KG; ++kk)
#pragma unroll
    for (int a = 0; a < COT; ++a)
#pragma unroll
      for (int b = 0; b < CIT; ++b) acc[kk][a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto load_dout = [&](int64_t s, uint4 (&pd)[COT]) {
    const int64_t r0 = s * W2_ROWS;
#pragma unroll
    for (int i = 0; i < COT; ++i) {
      const int v = i * 64 + lane, row = v / (2 * COT), piece = v % (2 * COT);
      const int64_t rr = r0 + row;
      const int ch = co0 + piece * 8;
      const bool ok = rr < n_out && ch < c_out && s < steps_total;
      pd[i] = ptc_buf_load16(dout_buf, ok ? ((uint32_t)rr * (uint32_t)c_out + (uint32_t)ch) * 2u : PTC_BUF_OOB);
    }
  };
  auto store_dout = [&](const uint4 (&pd)[COT]) {
#pragma unroll
    for (int i = 0; i < COT; ++i) {
      const int v = i * 64 + lane, row = v / (2 * COT), piece = v % (2 * COT);
      *reinterpret_cast<uint4*>(D + w2_off(row, piece)) = pd[i];
    }
  };

  // ---- compacted gathers
  const int my_kk = lane >> 5, my_row = lane & 31;                            // the slots this lane owns: lane + 64 st
  const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const int gpiece = lane & (PCS - 1), gpair = lane / PCS;                             // piece / pair-in-instruction of this lane as a gatherer
  auto load_entry = [&](int64_t s, int st) -> int32_t {
    const int kk = st * 2 + my_kk;
    const int64_t rr = s * W2_ROWS + my_row;
    const bool ok = kk < nk && rr < n_out && s < steps_total;
    const int32_t e = nbr[(int64_t)(k0 + (kk < nk ? kk : nk - 1)) * n_out + (rr < n_out ? rr : n_out - 1)];   // always in bounds
    return ok ? e : -1;
  };
  auto rank_step = [&](const int32_t (&e)[NSTEP]) -> int {
    int base = 0;
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      const unsigned long long mask = __builtin_amdgcn_ballot_w64(e[st] >= 0);
      if (e[st] >= 0) {
        const int rk = base + __builtin_popcountll(mask & lt_mask);
        list[rk] = st * 64 + lane;
        list[SLOTS + rk] = e[st];
      }
      base += __builtin_popcountll(mask);
    }
    w2_wave_sync();
    return base;
  };
  uint4 ga[W3_Q];
  int gslot[W3_Q];
  auto issue_round = [&](int rd, int cnt) {
#pragma unroll
    for (int q = 0; q < W3_Q; ++q) {
      const int p = (rd * W3_Q + q) * PPI + gpair;
      const bool ok = p < cnt;
      const int sl = ok ? list[p] : 0;
      const int32_t e = ok ? list[SLOTS + p] : -1;
      gslot[q] = ok ? sl : -1;
      ga[q] = ptc_buf_load16(in_buf, ok ? ((uint32_t)e * (uint32_t)C_IN + (uint32_t)gpiece * 8u) * 2u : PTC_BUF_OOB);
    }
  };
  // slot (kk, row) -> image kk, row-major [plane][32 rows][16] as wgrad2's store_in
  auto slot_ptr = [&](int sl) -> uint4* {
    return reinterpret_cast<uint4*>(I0 + (sl >> 5) * CIT * W2_PLANE + w2_off(sl & 31, gpiece));
  };
  auto write_round = [&](int (&keep)[W3_Q]) {
#pragma unroll
    for (int q = 0; q < W3_Q; ++q) {
      keep[q] = gslot[q];
      if (gslot[q] >= 0) *slot_ptr(gslot[q]) = ga[q];
    }
  };
  auto clear_round = [&](const int (&keep)[W3_Q]) {
#pragma unroll
    for (int q = 0; q < W3_Q; ++q)
      if (keep[q] >= 0) *slot_ptr(keep[q]) = make_uint4(0u, 0u, 0u, 0u);
  };

  // ---- prologue: images zeroed, step `worker` ranked with its first round and its dout rows in flight
  for (int o = lane * 16; o < KG * CIT * W2_PLANE; o += 64 * 16) *reinterpret_cast<uint4*>(I0 + o) = make_uint4(0u, 0u, 0u, 0u);
  uint4 pd[COT];
  load_dout(worker, pd);
  int32_t e_next[NSTEP];
#pragma unroll
  for (int st = 0; st < NSTEP; ++st) e_next[st] = load_entry(worker, st);
  int cnt = rank_step(e_next);                   // (its wave sync also orders the zero fill before the first rows)
  issue_round(0, cnt);
#pragma unroll
  for (int st = 0; st < NSTEP; ++st) e_next[st] = load_entry(worker + workers, st);

  for (int64_t s = worker; s < steps_total; s += workers) {
    // 1. step s lands in the wave's LDS slice
    store_dout(pd);
    int keep1[W3_Q], keep2[W3_Q];
    write_round(keep1);
    const bool two = cnt > PPI * W3_Q;           // wave-uniform, rare
    if (two) {
      issue_round(1, cnt);
      write_round(keep2);
    }
    w2_wave_sync();                              // images complete; the list is free
    // 2. step s + workers: dout rows, ranking, first gather round; entries of the step after it
    load_dout(s + workers, pd);
    cnt = rank_step(e_next);
    issue_round(0, cnt);
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) e_next[st] = load_entry(s + 2 * workers, st);
    // 3. multiply step s
    typename M::frag A[COT];
#pragma unroll
    for (int a = 0; a < COT; ++a) A[a] = w2_frag<T>(D, a, lane);
#pragma unroll
    for (int kk = 0; kk < KG; ++kk) {
      if (kk < nk) {
        typename M::frag B[CIT];
#pragma unroll
        for (int b = 0; b < CIT; ++b) B[b] = w2_frag<T>(I0 + kk * CIT * W2_PLANE, b, lane);
#pragma unroll
        for (int a = 0; a < COT; ++a)
#pragma unroll
          for (int b = 0; b < CIT; ++b) acc[kk][a][b] = M::mma(A[a], B[b], acc[kk][a][b]);
      }
    }
    w2_wave_sync();                              // every read of the slice is done
    // 4. the rows this step wrote read as zero again
    clear_round(keep1);
    if (two) clear_round(keep2);
    w2_wave_sync();                              // cleared before another lane writes the same row for the next step
  }

  // ---- sum the four waves through LDS, write this workgroup's partial (wgrad2's)
  float* red = reinterpret_cast<float*>(smem);
  float* pout = partial + (int64_t)bx * c_out * kv * C_IN;
#pragma unroll
  for (int kk = 0; kk < KG; ++kk) {
#pragma unroll
    for (int a = 0; a < COT; ++a) {
      __syncthreads();
      if (kk < nk) {
#pragma unroll
        for (int b = 0; b < CIT; ++b)
#pragma unroll
          for (int e = 0; e < 4; ++e) red[((wave * CIT + b) * 4 + e) * 64 + lane] = acc[kk][a][b][e];
      }
      __syncthreads();
      if (kk < nk) {
#pragma unroll
        for (int i = 0; i < CIT; ++i) {
          const int q = i * 256 + threadIdx.x;  // (b, e, lane)
          const int ln = q & 63, e = (q >> 6) & 3, b = q >> 8;
          const float v = red[q] + red[CIT * 256 + q] + red[2 * CIT * 256 + q] + red[3 * CIT * 256 + q];
          const int co = co0 + 16 * a + 4 * (ln >> 4) + e, ci = 16 * b + (ln & 15);
          if (co < c_out) pout[((int64_t)co * kv + (k0 + kk)) * C_IN + ci] = v;
        }
      }
    }
  }
}

// PTC_WGRAD3: 1 = the 64-input-channel instances, 2 = the 32-input-channel ones too
static inline bool wgrad3_takes(const W2Plan& p, const int32_t* nbr, int c_in, bool want_bias) {
  const char* e = getenv("PTC_WGRAD3");
  const int v = e ? atoi(e) : 0;
  if (v < 1 || nbr == nullptr || want_bias || p.ci_blocks != 1 || c_in != p.cit * 16) return false;
  if (c_in == 64) return p.cot == 4 && p.kg == 2;   // (2, 4, 4) = 64 -> <= 32 channels would need 8 gathers in flight: spills
  return v >= 2 && c_in == 32 && (p.cot == 4 || p.cot == 2) && p.kg == 4;
}

template <typename T, int COT, int CIT, int KG>
static int launch_wgrad3_inst(const W2Plan& p, const void* in, int64_t n_in, const void* dout, const int32_t* nbr, int64_t n_out, int kv,
                              int c_out, float* partial, hipStream_t s) {
  auto kern = wgrad3_kernel<T, COT, CIT, KG>;
  const size_t lds = (size_t)4 * ((size_t)(COT + KG * CIT) * W2_PLANE + KG * W2_ROWS * 8);
  static bool raised = false;   // per instantiation
  if (!raised && lds > 48 * 1024) {
    PTC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    raised = true;
  }
  const int nblocks = p.co_blocks, total = p.gx * p.groups * nblocks;
  hipLaunchKernelGGL(kern, dim3((unsigned)(8 * ((total + 7) / 8))), dim3(256), lds, s, (const T*)in, (const T*)dout, nbr, n_out, kv, c_out,
                     ptc_cdiv(n_out, W2_ROWS), partial, p.gx, p.groups, nblocks, (uint32_t)((uint64_t)n_in * CIT * 16 * sizeof(T)),
                     (uint32_t)((uint64_t)n_out * c_out * sizeof(T)));
  PTC_CHECK_LAUNCH("wgrad3_kernel");
  return PTC_OK;
}

template <typename T>
static int launch_wgrad3(const W2Plan& p, const void* in, int64_t n_in, const void* dout, const int32_t* nbr, int64_t n_out, int kv,
                         int c_out, float* partial, hipStream_t s) {
  if (p.cot == 4 && p.cit == 4 && p.kg == 2) return launch_wgrad3_inst<T, 4, 4, 2>(p, in, n_in, dout, nbr, n_out, kv, c_out, partial, s);
  if (p.cot == 4 && p.cit == 2 && p.kg == 4) return launch_wgrad3_inst<T, 4, 2, 4>(p, in, n_in, dout, nbr, n_out, kv, c_out, partial, s);
  if (p.cot == 2 && p.cit == 2 && p.kg == 4) return launch_wgrad3_inst<T, 2, 2, 4>(p, in, n_in, dout, nbr, n_out, kv, c_out, partial, s);
  return PTC_EUNSUPPORTED;
}
