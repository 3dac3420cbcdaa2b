// conv6.h -- conv5 (forward / dgrad of the gather-table convolution, 16-bit features, c_in = 32 / 64) with COMPACTED gathers.
// Included by spconv.hip.  Candidate, OFF by default (PTC_CONV6=1): developed on the host emulation after round 2's GPU time was
// spent; bit-identical to conv5 there, not timed yet.
//
// Why (profiles/r02_emu_conv_work_counts.txt): conv5 issues, per wave and 128-wide chunk of the flattened contraction
// (2 table rows x 2 row tiles x 16 rows = 64 (row, tap) slots), 8 table-entry loads and 8 gather loads whatever the table holds; at
// the density of the bench scenes 72 % of the gather lanes are "no neighbour" lanes, and the vector-memory path is paid per
// instruction (~21 cycles per wave-level load on the MI355X whatever its lanes do).  Here:
//   * ONE load fetches the chunk's 64 table entries (lane = slot: tap kk = lane >> 5, tile j = (lane >> 4) & 1, row = lane & 15);
//   * a ballot + prefix count ranks the present slots; their (slot, entry) go to a wave-private LDS list;
//   * gather instruction q takes the pairs of rank 8 q .. 8 q + 7 -- eight lanes per 128-byte row, whole lines as in conv5 -- so a
//     chunk with p present pairs needs ceil(p / 8) instructions: C6_Q = 4 unconditional ones cover 32 pairs (the expected 17 with
//     room), the rest goes through a rarely taken second round;
//   * the rows land in the (tap, tile) tile images of conv5 (same swizzle); rows without a neighbour are never written and are
//     masked to zero when the MFMA operand is read (their presence bit is in the ballot mask, which is wave-uniform).
// Same operands in the same order as conv5: bit-identical results.  LDS: 2 W chunks + 8 KB of images + the list per wave.
// c_in = 32 (NS = 1): 4 table rows per chunk = 128 slots, two per lane (two ballots), 64-byte rows = 16 pairs per gather instruction.
#pragma once

#define C6_Q 4   // unconditional gather instructions per chunk and round (8 pairs each)

template <typename T, int NS, int NTILES>
__global__ void __launch_bounds__(256, 2)
conv6_kernel(const T* __restrict__ in, const T* __restrict__ w, const float* __restrict__ bias, const int32_t* __restrict__ nbr,
             int64_t n_out, int kv, int c_out, int n_rowblk, T* __restrict__ out, uint32_t in_bytes, uint32_t w_bytes) {
  using M = Mma<T>;
  using frag = typename M::frag;
  constexpr int RT = 2, C_IN = NS * 32, TPC = 4 / NS;
  constexpr int SLOTS = TPC * RT * 16, NSTEP = SLOTS / 64;    // (tap, tile, row) slots per chunk; slots per lane
  constexpr int PCS = 4 * NS, PPI = 64 / PCS;                 // 16-byte pieces per row; pairs per gather instruction
  constexpr int NT = NTILES * 16, BM = RT * 64;
  constexpr int WFRAG = C3_FRAG + C3_FPAD;
  constexpr int WBUF = 4 * NTILES * WFRAG;
  constexpr int PITCH = C_IN * 2, IMG = 16 * PITCH;
  constexpr int WAVE_LDS = TPC * RT * IMG + SLOTS * 8;      // tile images + (slot, entry) list
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const __amdgpu_buffer_rsrc_t in_buf = ptc_buf(in, in_bytes), w_buf = ptc_buf(w, w_bytes);
  const int ny = c_out / NT;
  const int nblk = n_rowblk * ny;
  const int per_xcd = (nblk + 7) >> 3;
  const int lb = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (lb >= nblk) return;
  const int rb = lb / ny, n0 = (lb - rb * ny) * NT;
  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int64_t row0 = (int64_t)rb * BM + wave * (RT * 16);
  const int KV = kv * C_IN;
  const int nchunks = (KV + 127) >> 7;

  // ---- W staging (conv5's)
  constexpr int WI = NT / 16;
  const int wpiece = threadIdx.x & 15;
  uint32_t wsrc[WI];
  int wdst[WI];
#pragma unroll
  for (int it = 0; it < WI; ++it) {
    const int wrow = it * 16 + (threadIdx.x >> 4);
    wsrc[it] = (uint32_t)((n0 + wrow) * KV + wpiece * 8) * 2u;
    const int prow = lds_row_of_channel<NTILES>(wrow);
    const int rr = prow & 15;
    wdst[it] = ((prow >> 4) * 4 + (wpiece >> 2)) * WFRAG + rr * 64 + (((wpiece & 3) ^ c5_swz<1>(rr)) << 4);
  }
  uint4 wreg[WI];
  auto wload = [&](int c) {
    const bool ok = c * 128 + wpiece * 8 < KV;
#pragma unroll
    for (int it = 0; it < WI; ++it) wreg[it] = ptc_buf_load16(w_buf, ok ? wsrc[it] + (uint32_t)c * 256u : PTC_BUF_OOB);
  };
  auto wstore = [&](int buf) {
#pragma unroll
    for (int it = 0; it < WI; ++it) *reinterpret_cast<uint4*>(smem + buf * WBUF + wdst[it]) = wreg[it];
  };

  // ---- compacted gathers
  unsigned char* img = smem + 2 * WBUF + wave * WAVE_LDS;
  int32_t* list = reinterpret_cast<int32_t*>(img + TPC * RT * IMG);          // [SLOTS] slot | [SLOTS] entry
  const int my_kk = lane >> 5, my_j = (lane >> 4) & 1;                        // the slots this lane owns: lane + 64 step
  const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const int gpiece = lane & (PCS - 1), gpair = lane / PCS;                    // piece / pair-in-instruction of this lane as a gatherer
  auto load_entry = [&](int c, int step) -> int32_t {
    const int k = c * TPC + step * 2 + my_kk;
    const int64_t row = row0 + my_j * 16 + r;
    const bool ok = k < kv && row < n_out;
    const int32_t e = nbr[(int64_t)(k < kv ? k : kv - 1) * n_out + (row < n_out ? row : n_out - 1)];   // always in bounds
    return ok ? e : -1;
  };
  // ranks the present slots of a chunk and publishes (slot, entry) by rank; fills the presence masks, returns the pair count
  auto rank_chunk = [&](const int32_t (&e)[NSTEP], unsigned long long (&mask)[NSTEP]) -> int {
    int base = 0;
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      mask[st] = __builtin_amdgcn_ballot_w64(e[st] >= 0);
      if (e[st] >= 0) {
        const int rk = base + __builtin_popcountll(mask[st] & lt_mask);
        list[rk] = st * 64 + lane;
        list[SLOTS + rk] = e[st];
      }
      base += __builtin_popcountll(mask[st]);
    }
    w2_wave_sync();
    return base;
  };
  frag ga[C6_Q];
  int gslot[C6_Q];
  // gather instructions of round `rd` (pairs C6_Q PPI rd ...) of a ranked chunk with `cnt` pairs
  auto issue_round = [&](int rd, int cnt) {
#pragma unroll
    for (int q = 0; q < C6_Q; ++q) {
      const int p = (rd * C6_Q + q) * PPI + gpair;
      const bool ok = p < cnt;
      const int s = ok ? list[p] : 0;
      const int32_t e = ok ? list[SLOTS + p] : -1;
      gslot[q] = ok ? s : -1;
      ga[q] = ld_frag_buf<T>(in_buf, ok ? ((uint32_t)e * (uint32_t)C_IN + (uint32_t)gpiece * 8u) * 2u : PTC_BUF_OOB);
    }
  };
  // registers -> tile images: slot s = (kk, j, row) -> image kk * RT + j, row, piece swizzled as conv5's
  auto write_round = [&]() {
#pragma unroll
    for (int q = 0; q < C6_Q; ++q) {
      const int s = gslot[q];
      if (s >= 0) {
        const int rr = s & 15;
        *reinterpret_cast<frag*>(img + (s >> 4) * IMG + rr * PITCH + ((gpiece ^ c5_swz<NS>(rr)) << 4)) = ga[q];
      }
    }
  };
  const int rsw = c5_swz<NS>(r);
  const frag fzero = M::zero();

  f32x4 acc[RT][NTILES];
  {
    f32x4 breg[NTILES];
    sc_bias_regs<NTILES>(bias, n0, g, breg);
#pragma unroll
    for (int j = 0; j < RT; ++j)
#pragma unroll
      for (int t = 0; t < NTILES; ++t) acc[j][t] = breg[t];
  }

  // ---- prologue: W chunk 0 in LDS, chunk 0 ranked and its first round in flight, entries of chunk 1 in flight
  wload(0);
  int32_t e_next[NSTEP];
  unsigned long long mask[NSTEP];
#pragma unroll
  for (int st = 0; st < NSTEP; ++st) e_next[st] = load_entry(0, st);
  wstore(0);
  int cnt = rank_chunk(e_next, mask);
  issue_round(0, cnt);
#pragma unroll
  for (int st = 0; st < NSTEP; ++st) e_next[st] = load_entry(1, st);
  wload(1);
  __syncthreads();

  const int abase = r * 64 + ((g ^ c5_swz<1>(r)) << 4);
#pragma unroll 1
  for (int c = 0; c < nchunks; ++c) {
    const unsigned char* wb = smem + (c & 1) * WBUF + abase;
    // 1. the rows of chunk c go to the images (second round, rare: more than half of the slots present)
    write_round();
    if (cnt > PPI * C6_Q) {         // wave-uniform
      issue_round(1, cnt);
      write_round();
    }
    w2_wave_sync();                 // images complete; the list is free
    // 2. chunk c + 1: rank, first round in flight under the MFMAs of chunk c; entries of chunk c + 2
    unsigned long long mask_c[NSTEP];
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) mask_c[st] = mask[st];
    cnt = rank_chunk(e_next, mask);
    issue_round(0, cnt);
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) e_next[st] = load_entry(c + 2, st);
    // 3. MFMAs of chunk c
#pragma unroll
    for (int kk = 0; kk < TPC; ++kk) {
      bool any[RT], mine[RT];
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        const unsigned cell = (unsigned)(mask_c[kk >> 1] >> ((kk & 1) * 32 + j * 16)) & 0xffffu;
        any[j] = cell != 0;                       // wave-uniform: empty (tile, tap) cells are skipped as in conv5
        mine[j] = (cell >> r) & 1u;               // does row r of the cell have a neighbour
      }
#pragma unroll
      for (int si = 0; si < NS; ++si) {
        const int s = kk * NS + si;
        frag wf[NTILES];
#pragma unroll
        for (int t = 0; t < NTILES; ++t) wf[t] = *reinterpret_cast<const frag*>(wb + (t * 4 + s) * WFRAG);
        const int off = r * PITCH + (((4 * si + g) ^ rsw) << 4);
#pragma unroll
        for (int j = 0; j < RT; ++j) {
          if (any[j]) {
            frag fb = *reinterpret_cast<const frag*>(img + (kk * RT + j) * IMG + off);
            fb = mine[j] ? fb : fzero;
#pragma unroll
            for (int t = 0; t < NTILES; ++t) acc[j][t] = M::mma(wf[t], fb, acc[j][t]);
          }
        }
      }
    }
    w2_wave_sync();                 // every read of the images is done before the next trip rewrites them
    wstore((c + 1) & 1);
    __syncthreads();
    wload(c + 2);
  }

#pragma unroll
  for (int j = 0; j < RT; j += 2) {
    const int64_t rowA = row0 + j * 16 + r;
    sc_epilogue<T, NTILES>(*reinterpret_cast<f32x4(*)[2][NTILES]>(&acc[j]), nullptr, out, rowA, rowA + 16, n_out, c_out, n0, g);
  }
}

template <typename T, int NS, int NTILES>
static int launch_conv6_i(const void* in, int64_t n_in, const void* w, const float* bias, const int32_t* nbr, int64_t n_out, int kv,
                          int c_out, void* out, hipStream_t s) {
  constexpr int C_IN = NS * 32, RT = 2, SLOTS = (4 / NS) * RT * 16;
  const int n_rowblk = (int)ptc_cdiv(n_out, RT * 64);
  const int nblk = n_rowblk * (c_out / (NTILES * 16));
  const size_t lds = (size_t)2 * 4 * NTILES * (C3_FRAG + C3_FPAD) + (size_t)4 * ((4 / NS) * RT * 16 * C_IN * 2 + SLOTS * 8);
  auto kern = conv6_kernel<T, NS, NTILES>;
  static size_t allowed = 48 * 1024;   // per instantiation
  if (lds > allowed) {
    PTC_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    allowed = lds;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(8 * ((nblk + 7) / 8))), dim3(256), lds, s, (const T*)in, (const T*)w, bias, nbr, n_out, kv, c_out,
                     n_rowblk, (T*)out, (uint32_t)((uint64_t)n_in * C_IN * sizeof(T)), (uint32_t)((uint64_t)c_out * kv * C_IN * sizeof(T)));
  PTC_CHECK_LAUNCH("conv6_kernel");
  return PTC_OK;
}

template <typename T, int NS>
static int launch_conv6_n(const void* in, int64_t n_in, const void* w, const float* bias, const int32_t* nbr, int64_t n_out, int kv,
                          int c_out, void* out, hipStream_t s) {
  const int nt = c_out % 64 == 0 ? 4 : (c_out % 96 == 0 ? 6 : 2);
  if (nt == 4) return launch_conv6_i<T, NS, 4>(in, n_in, w, bias, nbr, n_out, kv, c_out, out, s);
  if (nt == 6) return launch_conv6_i<T, NS, 6>(in, n_in, w, bias, nbr, n_out, kv, c_out, out, s);
  return launch_conv6_i<T, NS, 2>(in, n_in, w, bias, nbr, n_out, kv, c_out, out, s);
}

// PTC_CONV6: 1 = c_in 64 only, 2 = c_in 32 and 64
static inline bool conv6_takes(int c_in) {
  const char* e = getenv("PTC_CONV6");
  const int v = e ? atoi(e) : 0;
  return (v >= 1 && c_in == 64) || (v >= 2 && c_in == 32);
}

template <typename T>
static int launch_conv6(const void* in, int64_t n_in, const void* w, const float* bias, const int32_t* nbr, int64_t n_out, int kv,
                        int c_in, int c_out, void* out, hipStream_t s) {
  if (c_in == 32) return launch_conv6_n<T, 1>(in, n_in, w, bias, nbr, n_out, kv, c_out, out, s);
  return launch_conv6_n<T, 2>(in, n_in, w, bias, nbr, n_out, kv, c_out, out, s);
}
