// conv8.h -- block-staged 3^3 submanifold convolution for WIDE rows: c_in a multiple of 32 from 96 up (SpUNet's 96 / 128 / 256-channel levels,
// PT-v3's 128 / 256 / 512-channel stages), forward and input gradient (mirrored weights).  Round 6; included by spconv.hip.
//
// conv3 (the global-gather kernel these shapes ran on) fetches every input row once per table entry that names it AND once per 64-column
// block of the output -- 9.3 x (c_out / 64) times -- in the MFMA operand layout (lane -> row l & 15, piece l >> 4: 57 address cycles per
// 1-KB wave gather, conv5.h).  At 256 -> 256, N = 12115 that is 300 MB of quarter-line requests for 6 MB of rows: 100 us for 19 GFLOP
// (190 TF/s, TA-bound; profiles/r06_a ops table), and 2.4 ms of the PT-v3 step's 30 deep convolutions.  Here, as in conv7 (32 / 64
// channels), a workgroup owns one 128-row block of blocks.hip -- its distinct input rows (the "halo", ~1.7 x 128) and the uint16 table of
// halo slots -- and
//   * stages the halo ONCE per 64-channel chunk of the input, 8 lanes per 128-byte piece of a row (whole lines), into an LDS image
//     [slot][64 channels], row pitch 128 + 16 B; slot HIMG is an all-zero row: "no neighbour" needs no branch;
//   * per tap k with a neighbour anywhere in the block: W[n0 .. n0 + NT)[k][chunk] goes through a two-deep LDS pipeline in MFMA fragment
//     order (conv3's: fetched into registers one tap ahead, ONE barrier per tap and chunk);
//   * FOUR waves = (64-row half of the block) x (half of the workgroup's NT columns): a wave multiplies its four 16-row tiles by its NTW
//     column tiles -- B fragments are ds_read_b128 out of the halo image at the rows' slots, A fragments out of the W image -- skips the
//     taps its 64 rows have no neighbour at (blocks.hip's tile masks) and, inside a tap, every 16-row tile without one (a ballot over
//     the slots the lanes hold: 60 % of the (16-row tile, tap) pairs of a curve-ordered indoor scene are empty);
//   * 80 KB of LDS: TWO workgroups per CU, so that one block's table / halo / weight latencies and its stores run under the other's
//     products.  (First form, profiles/r06_h .. r06_k: one 8-wave workgroup per CU with a 128-channel image -- every block paid its
//     table -> halo list -> rows -> weights chain and its stores alone on the CU, 273 us of 847 at 128 -> 96, N = 819200, with nothing
//     else cut; 32-row x 48-column wave tiles read 5 LDS fragments per 6 MFMAs: 1280 LDS cycles against 768 matrix cycles per tap.)
//   * blocks whose halo does not fit the image (more than HIMG distinct rows, or blocks.hip's own overflow mark) take the same loop with
//     the B fragments gathered from global memory through the neighbour table: correct for any input, slow, rare.
// Output-stationary, fixed summation order (chunk-major, taps ascending, 32-channel steps ascending): bit-reproducible.  The order differs
// from conv3's (tap-major): results agree to fp32 summation order, not bit for bit.
#pragma once

#ifndef C8_HIMG
#define C8_HIMG 288          // halo rows the LDS image holds (indoor scenes: mean 212, p99 283 -- tools/halo_stats.py)
#endif
#define C8_KC 64             // input channels per chunk
#define C8_PITCH (C8_KC * 2 + 16)
#define C8_THREADS 256
#define C8_PASSES ((C8_HIMG + 31) / 32)       // halo rows per thread: 32 rows per workgroup pass (8 lanes per 128-byte row piece)
#define C8_TAB_BYTES (27 * 128 * 2)

static inline int conv8_ntw(int c_out, int64_t n_blocks) {            // column tiles per wave (a workgroup holds 32 NTW columns)
  if (c_out % 128 == 0 && n_blocks * (c_out / 128) >= 384) return 4;  // (fewer workgroups than 1.5 per CU: the 64-column form, twice as many)
  if (c_out % 96 == 0 && c_out % 64 != 0) return 3;
  if (c_out % 64 == 0) return 2;
  return c_out % 96 == 0 ? 3 : 1;
}
static inline size_t conv8_lds(int ntw) { return (size_t)C8_TAB_BYTES + (size_t)(C8_HIMG + 1) * C8_PITCH + (size_t)2 * (2 * ntw) * 2 * 1024; }
static inline bool conv8_supported(int dtype, int kv, int c_in, int c_out, int bm, int hcap, int64_t n_out) {
  return dtype != PTC_F32 && kv == 27 && c_in % 32 == 0 && c_in >= 96 && c_in <= 1024 && c_out % 32 == 0 && bm == 128 && hcap >= 16 && hcap <= 511 &&
         n_out >= 256;
}

static int c8_ablate() { const char* e = getenv("PTC_C8_ABLATE"); return e ? atoi(e) : 0; }     // timing probes only (wrong results): 1 no products,
                                                                                                  // 2 no weight traffic after the first tap, 4 no per-tap barrier, 8 no halo staging, 64 phase timers

template <typename T, int NTW>
__global__ void __launch_bounds__(C8_THREADS, 2)
conv8_kernel(const T* __restrict__ in, const T* __restrict__ w, const float* __restrict__ bias, const int32_t* __restrict__ nbr,
             const uint16_t* __restrict__ tab, const int32_t* __restrict__ hid, const int32_t* __restrict__ hcnt, int64_t n, int c_in, int c_out,
             int hcap, int n_blocks, T* __restrict__ out, uint32_t in_bytes, int abl) {
  using M = Mma<T>;
  using frag = typename M::frag;
  constexpr int NT = 2 * NTW * 16;                      // columns per workgroup: two halves of NTW tiles, each with its own store grouping
  constexpr int WBUF = 2 * NTW * 2 * 1024;              // one W buffer: [2 NTW tiles][2 steps][1 KB fragment]
  constexpr bool PF_H = NTW < 4;                        // the next chunk's halo rows requested under the last tap of the current one (36 registers: the
                                                        // 64-column wave tile has none to spare -- 88 bytes of scratch per lane with it)
  const __amdgpu_buffer_rsrc_t in_buf = ptc_buf(in, in_bytes), w_buf = ptc_buf(w, (uint32_t)c_out * 27u * (uint32_t)c_in * 2u);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* lt = reinterpret_cast<uint16_t*>(smem);                                     // [27][32][4] halo entries (blocks.hip's 128-byte-row variant)
  unsigned char* img = smem + C8_TAB_BYTES;                                             // [HIMG + 1][C8_PITCH]
  unsigned char* wl = img + (size_t)(C8_HIMG + 1) * C8_PITCH;                           // 2 x WBUF
  // XCD-first numbering: the column blocks of one row block are consecutive logical ids and run on one XCD (its halo rows stay in that L2)
  const int ny = c_out / NT, nblk = n_blocks * ny;
  const int per_xcd = (nblk + 7) >> 3;
  const int lb = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (lb >= nblk) return;
  const int b = lb / ny, n0 = (lb - b * ny) * NT;
  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  const int rh = wave & 1, half = wave >> 1;            // 64-row half of the block, column half
  const int r = lane & 15, g = lane >> 4;
  const int64_t row0 = (int64_t)b * 128 + rh * 64;
  // (bit 64 of PTC_C8_ABLATE: cycle totals per phase of every workgroup's wave 0 -> the first 64 bytes of the block's first output row)
  long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
  const bool timing = (abl & 64) != 0;
  auto tick = [&](int i) {
    if (timing) {
      const long long t = clock64();
      tph[i] += t - tlast;
      tlast = t;
    }
  };
  if (timing) tlast = clock64();
  const int cnt = hcnt[b];
  const bool staged = cnt >= 0 && cnt <= C8_HIMG;       // else: the global-gather form of the same loop

  // ---- the block's table (copied as whole 16-byte pieces; an entry is turned into a slot where it is used), tap masks, halo list.
  // Every load of this kernel that feeds a prefetch is a RAW BUFFER LOAD (out-of-range offset = zeros): written as `v = *p; if (!ok) v = 0`
  // the compiler turned each one into a branch around the load with s_waitcnt vmcnt(0) behind it -- nine serial latencies for the halo
  // list and three per tap for the weights (the "skeleton" and "weights" terms of profiles/r06_m_conv8_v2.txt: 229 + 190 of 790 us)
  uint32_t tmask, bmask;
  int32_t hrow[C8_PASSES];                              // this thread's halo rows (row tid >> 3 of every 32-row pass), -1 beyond the list
  if (staged) {
    const uint16_t* tb = tab + (int64_t)b * (28 * 128);
    const __amdgpu_buffer_rsrc_t tab_buf = ptc_buf(tb, C8_TAB_BYTES), hid_buf = ptc_buf(hid + (int64_t)b * hcap, (uint32_t)hcap * 4u);
    uint4 tv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) tv[i] = ptc_buf_load16(tab_buf, (uint32_t)(threadIdx.x + C8_THREADS * i) * 16u);
#pragma unroll
    for (int p = 0; p < C8_PASSES; ++p) hrow[p] = ptc_buf_load4(hid_buf, (uint32_t)(p * 32 + (threadIdx.x >> 3)) * 4u);
    const uint32_t* mw = reinterpret_cast<const uint32_t*>(tb + 27 * 128);
    tmask = mw[2 * rh] | mw[2 * rh + 1];
    bmask = mw[4];
    for (int q = threadIdx.x; q < C8_PITCH / 4; q += C8_THREADS) reinterpret_cast<uint32_t*>(img + (size_t)C8_HIMG * C8_PITCH)[q] = 0u;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (threadIdx.x + C8_THREADS * i < C8_TAB_BYTES / 16) reinterpret_cast<uint4*>(lt)[threadIdx.x + C8_THREADS * i] = tv[i];
#pragma unroll
    for (int p = 0; p < C8_PASSES; ++p)
      if (p * 32 + (int)(threadIdx.x >> 3) >= cnt) hrow[p] = -1;
  } else {
    tmask = bmask = 0x7ffffffu;
#pragma unroll
    for (int p = 0; p < C8_PASSES; ++p) hrow[p] = -1;
  }

  tick(0);                                              // halo count, table + halo list in LDS / registers
  f32x4 acc[4][NTW], breg[NTW];
  const int nw0 = n0 + half * NTW * 16;                 // first column of this wave
  sc_bias_regs<NTW>(bias, nw0, g, breg);
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[j][t] = breg[t];

  // ---- W staging: 8 lanes per 128-byte slice (one tap, one chunk) of a weight row = whole lines per request.  Fragment image [16 rows][4
  // pieces] with the pieces XOR-swizzled by the row (c5_swz<1>): these stores and the ds_read_b128 of the A fragments are both conflict-free.
  constexpr int WI = (NT + 31) / 32;                    // 32 weight rows per pass
  const int wpiece = threadIdx.x & 7;
  uint32_t wsrc[WI];                                    // byte offset of (weight row, tap 0, piece) -- or out of range: zeros
  int wdst[WI];
  bool wact[WI];
#pragma unroll
  for (int it = 0; it < WI; ++it) {
    const int wrow = it * 32 + (threadIdx.x >> 3);
    wact[it] = wrow < NT;
    const int wr = wact[it] ? wrow : 0;
    const int whalf = wr / (NTW * 16), prow = lds_row_of_channel<NTW>(wr - whalf * NTW * 16), rr = prow & 15;
    wsrc[it] = wact[it] ? (uint32_t)((n0 + wr) * 27 * c_in + wpiece * 8) * 2u : PTC_BUF_OOB;
    wdst[it] = ((whalf * NTW + (prow >> 4)) * 2 + (wpiece >> 2)) * 1024 + rr * 64 + (((wpiece & 3) ^ c5_swz<1>(rr)) << 4);
  }
  uint4 wreg[WI];
  auto wload = [&](int k, int c0, int kc) {
    const uint32_t koff = wpiece * 8 < kc ? (uint32_t)(k * c_in + c0) * 2u : PTC_BUF_OOB;     // either part out of range: the sum is forced out of range
#pragma unroll
    for (int it = 0; it < WI; ++it) wreg[it] = ptc_buf_load16(w_buf, (wsrc[it] + koff) | ((wsrc[it] | koff) & PTC_BUF_OOB));
  };
  auto wstore = [&](int buf) {
#pragma unroll
    for (int it = 0; it < WI; ++it)
      if (wact[it]) *reinterpret_cast<uint4*>(wl + (size_t)buf * WBUF + wdst[it]) = wreg[it];
  };
  auto next_tap = [&](int k) {            // first tap >= k with a neighbour somewhere in the block (27: none)
    while (k < 27 && !((bmask >> k) & 1u)) ++k;
    return k;
  };
  // halo rows of a chunk: 8 lanes per row piece of kc * 2 bytes (a full line at 64 channels), every row of the thread in flight together
  const int hpiece = threadIdx.x & 7;
  uint4 hv[C8_PASSES];
  auto hload = [&](int c0, int kc) {
#pragma unroll
    for (int p = 0; p < C8_PASSES; ++p)
      hv[p] = ptc_buf_load16(in_buf, (hrow[p] >= 0 && hpiece * 8 < kc) ? ((uint32_t)hrow[p] * (uint32_t)c_in + (uint32_t)(c0 + hpiece * 8)) * 2u : PTC_BUF_OOB);
  };
  auto hstore = [&]() {
#pragma unroll
    for (int p = 0; p < C8_PASSES; ++p) {
      const int slot = p * 32 + (threadIdx.x >> 3);
      if (hrow[p] >= 0) *reinterpret_cast<uint4*>(img + (size_t)slot * C8_PITCH + hpiece * 16) = hv[p];
    }
  };

  const int nch = (c_in + C8_KC - 1) / C8_KC;
  int buf = 0;
  const int k_first = next_tap(0);
  if (staged && !(abl & 8)) hload(0, c_in < C8_KC ? c_in : C8_KC);
  for (int ch = 0; ch < nch; ++ch) {
    const int c0 = ch * C8_KC;
    const int kc = (c_in - c0) < C8_KC ? (c_in - c0) : C8_KC;
    const int steps = kc >> 5;
    int k = k_first;
    if (!PF_H && ch > 0 && staged && !(abl & 8)) hload(c0, kc);
    if (k < 27) wload(k, c0, kc);         // the first tap's weights travel with the halo rows
    __syncthreads();                      // the image (and the W buffers) of the previous chunk are free
    tick(1);                              // chunk entry: barrier
    if (staged && !(abl & 8)) hstore();
    tick(2);                              // halo rows landed and stored
    bool first = true;
    while (k < 27) {
      if (!(abl & 2) || first) wstore(buf);
      tick(3);                                          // wait for W(k), store
      const int kn = next_tap(k + 1);
      if (!(abl & 4) || first) __syncthreads();         // W(k) complete (and, first tap of a chunk: the halo image)
      tick(4);                                          // per-tap barrier
      first = false;
      // BEHIND the barrier (__syncthreads waits for every outstanding global load: requested in front of it, the next tap's weights were
      // waited for before this tap's products could start -- 190 of 790 us at 128 -> 96, N = 819200, profiles/r06_m_conv8_v2_ablation.txt):
      if (kn < 27 && !(abl & 2)) wload(kn, c0, kc);     // the next tap's weights are in flight under this tap's products
      // last tap of a chunk: the next chunk's halo rows go out under its products (registers only; the image is rewritten behind the barrier above)
      if (PF_H && kn >= 27 && ch + 1 < nch && staged && !(abl & 8)) hload(c0 + C8_KC, (c_in - c0 - C8_KC) < C8_KC ? (c_in - c0 - C8_KC) : C8_KC);
      if (((tmask >> k) & 1u) && !(abl & 1)) {
        const unsigned char* wb = wl + (size_t)buf * WBUF + (size_t)(half * NTW * 2) * 1024 + r * 64 + ((g ^ c5_swz<1>(r)) << 4);
        if (staged) {
          // entries of rows 64 rh + 16 j + r, j = 0..3: 32-row tile 2 rh + (j >> 1), row 16 (j & 1) + r of it -- j and j + 2 are neighbours in the table
          const uint32_t e02 = *reinterpret_cast<const uint32_t*>(lt + (k * 32 + r) * 4 + 2 * rh);
          const uint32_t e13 = *reinterpret_cast<const uint32_t*>(lt + (k * 32 + 16 + r) * 4 + 2 * rh);
          int sl[4] = {(int)((e02 & 0xffffu) >> 7), (int)((e13 & 0xffffu) >> 7), (int)(e02 >> 23), (int)(e13 >> 23)};
          bool act[4];
          const unsigned char* pb[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool has = sl[j] < hcap;                // "no neighbour" (entry = hcap x 128): the zero row
            act[j] = __builtin_amdgcn_ballot_w64(has) != 0;
            pb[j] = img + (size_t)(has ? sl[j] : C8_HIMG) * C8_PITCH + g * 16;
          }
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            if (s < steps) {
              frag fw[NTW];
#pragma unroll
              for (int t = 0; t < NTW; ++t) fw[t] = *reinterpret_cast<const frag*>(wb + (t * 2 + s) * 1024);
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                if (act[j]) {
                  const frag fb = *reinterpret_cast<const frag*>(pb[j] + s * 64);
#pragma unroll
                  for (int t = 0; t < NTW; ++t) acc[j][t] = M::mma(fw[t], fb, acc[j][t]);
                }
              }
            }
          }
        } else {
          int32_t ix[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int64_t rr = row0 + 16 * j + r;
            ix[j] = rr < n ? nbr[(int64_t)k * n + rr] : -1;
          }
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            if (s < steps) {
              const uint32_t col = (uint32_t)(c0 + s * 32 + g * 8);
              frag fw[NTW];
#pragma unroll
              for (int t = 0; t < NTW; ++t) fw[t] = *reinterpret_cast<const frag*>(wb + (t * 2 + s) * 1024);
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const frag fb = ld_frag_buf<T>(in_buf, ix[j] >= 0 ? ((uint32_t)ix[j] * (uint32_t)c_in + col) * 2u : PTC_BUF_OOB);
#pragma unroll
                for (int t = 0; t < NTW; ++t) acc[j][t] = M::mma(fw[t], fb, acc[j][t]);
              }
            }
          }
        }
      }
      tick(5);                                          // products
      buf ^= 1;
      k = kn;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j += 2)
    sc_epilogue<T, NTW>(*reinterpret_cast<f32x4(*)[2][NTW]>(&acc[j]), nullptr, out, row0 + 16 * j + r, row0 + 16 * j + 16 + r, n, c_out, nw0, g);
  tick(6);                                              // epilogue issued
  if (timing && threadIdx.x == 0 && n0 == 0) {
    long long* o = reinterpret_cast<long long*>(out + (int64_t)b * 128 * c_out);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = tph[i];
  }
}

template <typename T, int NTW>
static int launch_conv8_i(const void* in, int64_t n_in, const void* w, const float* bias, const int32_t* nbr, const uint16_t* tab, const int32_t* hid,
                          const int32_t* hcnt, int hcap, int64_t n, int c_in, int c_out, void* out, hipStream_t s) {
  const int n_blocks = (int)ptc_cdiv(n, 128);
  const int nblk = n_blocks * (c_out / (2 * NTW * 16));
  const size_t lds = conv8_lds(NTW);
  auto kern = conv8_kernel<T, NTW>;
  PTC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)(8 * ((nblk + 7) / 8))), dim3(C8_THREADS), lds, s, (const T*)in, (const T*)w, bias, nbr, tab, hid, hcnt, n, c_in,
                     c_out, hcap, n_blocks, (T*)out, (uint32_t)((uint64_t)n_in * c_in * sizeof(T)), c8_ablate());
  PTC_CHECK_LAUNCH("conv8_kernel");
  return PTC_OK;
}

template <typename T>
static int launch_conv8(const void* in, int64_t n_in, const void* w, const float* bias, const int32_t* nbr, const uint16_t* tab, const int32_t* hid,
                        const int32_t* hcnt, int hcap, int64_t n, int c_in, int c_out, void* out, hipStream_t s) {
  switch (conv8_ntw(c_out, ptc_cdiv(n, 128))) {
    case 4: return launch_conv8_i<T, 4>(in, n_in, w, bias, nbr, tab, hid, hcnt, hcap, n, c_in, c_out, out, s);
    case 3: return launch_conv8_i<T, 3>(in, n_in, w, bias, nbr, tab, hid, hcnt, hcap, n, c_in, c_out, out, s);
    case 2: return launch_conv8_i<T, 2>(in, n_in, w, bias, nbr, tab, hid, hcnt, hcap, n, c_in, c_out, out, s);
    default: return launch_conv8_i<T, 1>(in, n_in, w, bias, nbr, tab, hid, hcnt, hcap, n, c_in, c_out, out, s);
  }
}
