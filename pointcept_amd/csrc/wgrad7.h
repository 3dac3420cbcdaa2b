// wgrad7.h -- weight gradient of the SUBMANIFOLD 3^3 gather-table convolution for 16-bit features with c_in = c_out = 32 | 64:
//     dw[co][k][ci] = sum_o dout[o][co] * in[nbr[k][o]][ci]        (contraction over ROWS)
// ACCUMULATOR-STATIONARY IN REGISTERS on the block-local tables of conv7 (blocks.hip): the distinct input rows of a 128-row block (its
// halo) and the block's 128 dout rows are staged ONCE in LDS by the DMA path and every operand of every tap is built from those two
// images.  Included by wgrad7.hip (its own translation unit: conv7.hip is built with MFMA results forced into architectural VGPRs,
// this kernel's accumulators ARE the accumulation-register file).
//
// Why (round 4): wgrad2 (wave-private staging of 32-row steps, 2 taps per workgroup group) runs the dec0 shape
// (64 -> 64, N = 819200) in 369 us = 0.09 of its HBM roof: 14 tap groups re-read dout and re-gather their rows from L2 -- 672 KB
// through the vector-memory path per 128 output rows where the block's operands are 43 KB (profiles/r03_p_conv_pmc_s0.json: TA_BUSY
// 62 %, HBM traffic 1.47 x algorithmic, matrix pipe 22 %).  Here:
//   * a workgroup is PERSISTENT over a strided share of the blocks (the same XCD-aware order as conv7) and keeps its part of the
//     gradient dw[C][27][C] in the ACCUMULATION registers of its four waves as 32x32 fp32 MFMA tiles.  Wave w owns the taps
//     k = w (mod 4) -- seven tap slots -- for all input channels; C = 64: a workgroup owns ONE output-channel half ch (two workgroups of
//     one XCD, dispatched back to back, walk the same blocks: the second finds the halo in the L2): 7 x 2 tiles = 224 of the 256
//     AGPRs, which leaves every architectural VGPR to the pipeline; C = 32: 7 tiles.  (All 27 taps x one 32x32 quadrant per wave = 432
//     accumulator registers was the first build: 16 of the 27 tiles fit the AGPR file, the compiler shuttles the other 11 through it
//     around every MFMA -- 944 v_accvgpr_write -- and spills 125 registers; `-amdgpu-mfma-vgpr-form` crashes its AGPR-rewrite pass at that
//     pressure.)  The gradient leaves the chip ONCE per workgroup (one fp32 partial per workgroup pair / per workgroup, summed by the
//     deterministic reduction of spconv.hip);
//   * the contraction index of a 32x32x16 MFMA is a ROW, so both operands are needed channel-major.  They come out of the row-major
//     LDS images through ds_read_b64_tr_b16, whose 16 lanes address FOUR ROWS INDEPENDENTLY (each lane supplies the address of 8 bytes
//     of "its" row): for the gathered operand the four row addresses are table entries -- the 27-tap gather and the transposition are
//     the same LDS instruction, nothing is materialised.  An MFMA step contracts the 16 rows {32 t + 4 s + q : t = 0..3, q = 0..3} of a
//     block (4 rows of each of its four 32-row tiles): lane group hh = lane >> 5 takes tiles 2 hh and 2 hh + 1, whose table entries are
//     ADJACENT uint16 in conv7's table layout [tap][row in tile][tile] -- one ds_read_b32 per (tap, step) and lane;
//   * tap slot outer (static: the accumulators of a tap are register NAMES), the eight steps of the block inner.  An EMPTY TAP (33 % of
//     the (block, tap) pairs on curve-ordered scenes, the block mask of blocks.hip) is one scalar branch; inside an active tap every
//     gather is unconditional ("no neighbour" entries read the all-zero row) -- 8 KH MFMAs with two transposing reads and two vector-
//     ALU instructions each, the compiler's wait counts exact.  (v1 -- profiles/r04_a_wgrad7_v1_time.txt: 287 us at 64 -> 64 -- skipped at
//     (step, tap) granularity: one MFMA per pair behind ~10 scalar instructions of mask tests and branches, paid by the empty pairs
//     too: 112 slots x ~58 cycles per block where the MFMAs of the active ones are 1800.  At one wave per SIMD an instruction of ANY
//     kind is an issue slot of ~4 cycles; an MFMA hides seven.)  The A fragments of the eight steps are read once per block; table
//     words run one tap ahead, gathered fragments three steps ahead in a ring of six, across tap boundaries when the next tap is
//     active;
//   * block b + 1 (halo rows, table, dout rows) is in flight through global_load_lds while block b is multiplied (conv7's scheme:
//     asm DMA invisible to the compiler's wait-count pass, completion counted by hand).  The halo image keeps conv7's XOR swizzle (the
//     table entries carry it), the dout image holds the workgroup's 32 output channels at 64 bytes per row: the four rows of a transposing
//     read are 256 contiguous bytes = all 64 banks once.  Buffer 1 sits 64 KB after buffer 0: "image base + (entry ^ piece)" is then ONE
//     xor (entries < 64 KB).
// A block whose halo did not fit (hcnt < 0: rows in no spatial order) cannot be served here; the host entry point therefore GATES the
// two kernels on the device-side overflow counter of the table builder: this kernel runs when it is zero, wgrad2 over the whole tensor
// when it is not, each returning at once otherwise -- no host synchronisation, and the reduction reads the partials of whichever ran.
// CHANNEL SLICES (round 4, the 128 .. 512-channel stages and SpUNet's 96-channel decoder; c_in % 32 == 0, c_out % 32 == 0, c_in and c_out
// independent): the C = 64 geometry (the C = 32 one when c_in is not a multiple of 64)
// with a workgroup owning ONE (32 output channels) x (64 | 32 input channels) slice of dw for all 27 taps -- the same 224 accumulation
// registers -- and staging only ITS 128 bytes of every halo row and ITS 64 bytes of every dout row.  (c_out / 32)(c_in / 64) workgroups
// walk one block sequence (same XCD, dispatched back to back: the slices of a row come out of one L2); 256 / slices sequences.  wgrad2 at
// these widths re-streams dout and re-gathers its rows once per (64 x 64 channel tile, tap pair): 16 x 14 times at 256 channels -- 95 TF/s
// on the outdoor stage-3 shape (profiles/r04_h_ops_outdoor.txt).
// Summation order differs from wgrad2 (rows of a block in step order, blocks in the workgroup's stride order): results agree to fp32
// rounding of the accumulation, and are bit-reproducible run to run.
#pragma once

#define W7_BUF1 65536                        // LDS byte offset of buffer 1 (halo image + table); buffer 0 at 0
#ifndef W7_DMA_INLINE
#define W7_DMA_INLINE 1                      // the next block's DMA instructions inside the MFMA stream (0: in front of it; timing A/B)
#endif
#define W7_DMA_EVERY 2                       // ... one per this many (step, tap) pairs

#define W7_MAX_WGS 256                       // one persistent workgroup per CU
// channel slices: input-channel blocks of 64 (c_in % 64 == 0: the 64-channel geometry) or of 32 (c_in % 32 == 0 only -- SpUNet's 96-channel
// decoder: the 32-channel geometry), output-channel blocks of 32
static inline bool wgrad7_sliced(int c_in, int c_out) {
  return c_in % 32 == 0 && c_out % 32 == 0 && c_in <= 1024 && c_out <= 1024 && !(c_in == c_out && (c_in == 64 || c_in == 32));
}
static inline int wgrad7_slice_cin(int c_in) { return c_in % 64 == 0 ? 64 : 32; }
static inline int wgrad7_slices(int c_in, int c_out) {
  return wgrad7_sliced(c_in, c_out) ? (c_in / wgrad7_slice_cin(c_in)) * (c_out / 32) : (c_in == 64 ? 2 : 1);
}
static inline bool wgrad7_supported(int dtype, int kv, int c_in, int c_out, int bm, int hcap, int64_t n_out) {
  if (conv7_supported(dtype, kv, c_in, c_out, bm, hcap, n_out)) return true;
  return dtype != PTC_F32 && kv == 27 && bm == C7_BM && hcap == C7_HCAP && n_out >= 1024 && wgrad7_sliced(c_in, c_out) &&
         wgrad7_slices(c_in, c_out) <= W7_MAX_WGS;
}
// workgroups walking DISTINCT block sequences = fp32 partials per call (C = 64: two workgroups -- the output-channel halves -- per sequence;
// channel slices: (c_out / 32)(c_in / 64) workgroups per sequence, and at least FOUR blocks per sequence -- every sequence writes a full
// fp32 copy of dw, which at 256+ channels outweighs the operands of a short sequence)
static inline int wgrad7_splits(int64_t n_out, int c_in, int c_out) {
  int64_t nb = ptc_cdiv(n_out, C7_BM), cap = W7_MAX_WGS / wgrad7_slices(c_in, c_out);
  if (wgrad7_sliced(c_in, c_out)) {
    if (cap > (nb + 3) / 4) cap = (nb + 3) / 4;
    // ... and no more sequences than keep the partials (one fp32 dw per sequence, written and read once) below four times the operands:
    // 2821 rows at 512 channels are 9 MB of operands against 28 MB per partial -- two sequences ran 136 us where wgrad2 takes 98
    // (profiles/r04_l_ops_stages.txt)
    const int64_t partial = (int64_t)27 * c_in * c_out * 4, operands = n_out * (int64_t)(c_in + c_out) * 2;
    const int64_t by_traffic = 4 * operands / partial;
    if (cap > by_traffic) cap = by_traffic;
  }
  if (cap < 1) cap = 1;
#ifndef __HIPCC__
  if (const char* e = getenv("PTC_EMU_CONV7_WGS")) cap = atoi(e);   // host emulation only: several blocks per workgroup at test sizes
#endif
  if (cap >= 8 && wgrad7_sliced(c_in, c_out)) cap &= ~7;             // whole rounds of the eight XCDs
  return (int)(nb < cap ? nb : cap);
}

int ptc_wgrad7_launch(int dtype, const void* in, const void* dout, const uint16_t* tab, const int32_t* hid, const int32_t* hcnt,
                      const int32_t* gate, int64_t n_out, int c_in, int c_out, float* partial, hipStream_t s);

#ifdef PTC_WGRAD7_IMPL
template <int C> struct W7Geom {
  static constexpr int ROWB = C * 2, PCS = ROWB / 16, RPI = 64 / PCS;
  static constexpr int ROWS_BYTES = (C7_HCAP + 1) * ROWB;                       // + the zero row
  static constexpr int DROWB = 64;                                              // dout image: 32 output channels per row (C = 64: the half `ch`)
  static constexpr int DOUT_BYTES = C7_BM * DROWB;
  static constexpr int DOUT0 = (W7_BUF1 + ROWS_BYTES + C7_TABB + 1023) & ~1023;   // the two dout images follow buffer 1
  static constexpr int LDS = DOUT0 + 2 * DOUT_BYTES;
  static constexpr int NI = (C7_HCAP + RPI - 1) / RPI, NIW = (NI + 3) / 4;      // DMA instructions of a full halo / per wave
  static constexpr int NDW = DOUT_BYTES / 1024 / 4;                             // dout DMA instructions per wave
  static constexpr int NA = 7;                                                  // tap slots per wave: taps 4 a + wave
  static constexpr int KH = C / 32;                                             // 32-channel input halves (accumulators per tap slot)
  static __device__ __forceinline__ int swz(int slot) { return C == 64 ? PTC_SWZ64(slot) : ((slot >> 2) & 3); }
  static_assert(ROWS_BYTES + C7_TABB <= W7_BUF1, "buffer 0 must end before buffer 1");
  static_assert(LDS <= 163840, "LDS budget");
};

template <typename T, int C, bool SL = false>
__global__ void __launch_bounds__(256, 1)
wgrad7_kernel(const T* __restrict__ in, const T* __restrict__ dout, const uint16_t* __restrict__ tab, const int32_t* __restrict__ hid,
              const int32_t* __restrict__ hcnt, const int32_t* __restrict__ gate, int64_t n_out, int n_blocks, float* __restrict__ partial,
              int c_in_full, int c_out_full) {

  using frag = typename Mma<T>::frag;
  using MM = C7Mma<T>;
  using G = W7Geom<C>;
  constexpr int ROWB = G::ROWB, PCS = G::PCS, RPI = G::RPI, NA = G::NA;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (gate != nullptr && *gate != 0) return;              // some block overflowed: wgrad2 serves this call (see the head of the file)
  const int lane = ptc_lane(), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lp = lane & 15, g4 = lane >> 4, hh = g4 >> 1, cb = g4 & 1;   // transposing-read roles: 16-lane group g4 = (k-group hh, channel block cb)
  const int q = lp >> 2, c4 = lp & 3;                                    // ... lane: row q of the read's four, channels 4 c4 .. 4 c4 + 3 of the block
  const int tw = wave;                                                   // this wave's taps: 4 a + tw, a = 0..6 (k = 27 never active)
  // C = 64: workgroup = (block sequence, output-channel half ch); hardware workgroup b runs on XCD b % 8, so the two halves of a sequence
  // are b = x + 8 (2 p) and x + 8 (2 p + 1): same XCD, dispatched back to back.  Sequence `seq` of `step` takes blocks seq', seq' + step, ..
  // with seq' as in conv7 (one round's blocks of an XCD adjacent)
  // SL: n_sl = (c_out / 32)(c_in / 64) slices per sequence; sequence s, slice t <-> hardware workgroup (s % 8) + 8 (t + n_sl (s / 8)) when the
  // sequences come in whole rounds of eight (the generalisation of the C = 64 rule), else s n_sl + t
  const int wgs = (int)gridDim.x;
  const int n_ci = SL ? c_in_full / C : 1, n_sl = SL ? n_ci * (c_out_full / 32) : (C == 64 ? 2 : 1);
  const int step = wgs / n_sl;
  int vb, sl;
  if constexpr (SL) {
    const int bx = (int)blockIdx.x;
    if (step % 8 == 0) { vb = (bx & 7) + 8 * ((bx >> 3) / n_sl); sl = (bx >> 3) % n_sl; }
    else { vb = bx / n_sl; sl = bx - vb * n_sl; }
  } else {
    vb = C == 64 ? (wgs % 16 == 0 ? (((int)blockIdx.x >> 4) << 3 | ((int)blockIdx.x & 7)) : ((int)blockIdx.x >> 1)) : (int)blockIdx.x;
    sl = C == 64 ? (wgs % 16 == 0 ? (((int)blockIdx.x >> 3) & 1) : ((int)blockIdx.x & 1)) : 0;
  }
  const int ch = SL ? sl / n_ci : sl;                         // 32-channel block of dout / dw rows this workgroup owns
  const int cib = SL ? sl - ch * n_ci : 0;                    // SL: C-channel block of the input rows
  const uint32_t in_pitch = SL ? (uint32_t)c_in_full * 2u : (uint32_t)ROWB, in_col = (uint32_t)cib * (uint32_t)ROWB;       // bytes
  const uint32_t do_pitch = SL ? (uint32_t)c_out_full * 2u : (uint32_t)ROWB;
  const int b_begin = (step % 8 == 0) ? (vb & 7) * (step / 8) + (vb >> 3) : vb;
  const int b_end = n_blocks;

  constexpr int KH = G::KH;
  c7_f32x16 acc[NA * KH];                                                // [tap slot a][input-channel half kh]
#pragma unroll
  for (int a = 0; a < NA * KH; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  if (b_begin < b_end) {
    // ---- zero rows of both halo images (never written by the DMA)
    if (threadIdx.x < 2 * PCS) {
      const int bsel = threadIdx.x / PCS, pc = threadIdx.x % PCS;
      *reinterpret_cast<uint4*>(smem + bsel * W7_BUF1 + C7_HCAP * ROWB + pc * 16) = make_uint4(0, 0, 0, 0);
    }
    const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c7_lds_addr(smem));
    const int drow = lane / PCS, dpos = lane % PCS;
    int32_t ids[G::NIW];
    auto load_ids = [&](int blk) {
#pragma unroll
      for (int i = 0; i < G::NIW; ++i) {
        const int slot = (4 * i + wave) * RPI + drow;
        ids[i] = hid[(int64_t)blk * C7_HCAP + (slot < C7_HCAP ? slot : C7_HCAP - 1)];
      }
    };
    // one DMA instruction (1 KB) of a block's operands: pieces 0 .. NIW - 1 = halo rows (this wave's share), then the table (2), then the
    // 32 output channels of the block's dout rows this workgroup owns (NDW; 64 bytes per row: the four rows of a transposing read are
    // 256 contiguous bytes = all 64 banks once; rows past the end repeat the last row -- their table entries are "no neighbour", the
    // products are zero).  cnt = the block's halo count, 0 = there is no such block.
    constexpr int NP = G::NIW + 2 + G::NDW;
    auto dma_piece = [&](auto pc, int blk, int cnt, int bsel, const int32_t (&idv)[G::NIW]) {
      constexpr int P = decltype(pc)::value;
      const uint32_t base = lds0 + (uint32_t)(bsel * W7_BUF1);
      if constexpr (P < G::NIW) {
        const int ii = 4 * P + wave;
        if (ii * RPI < cnt) {                                  // wave-uniform
          const int slot = ii * RPI + drow;
          const int piece = dpos ^ G::swz(slot);
          c7_dma16s(in, (uint32_t)idv[P] * in_pitch + in_col + (uint32_t)(piece * 16), base + (uint32_t)(ii * 1024));
        }
      } else if constexpr (P < G::NIW + 2) {
        const int ii = 4 * (P - G::NIW) + wave;
        if (cnt > 0 && ii < C7_TABB / 1024) {
          const unsigned char* tsrc = reinterpret_cast<const unsigned char*>(tab) + ((int64_t)(C == 64 ? 0 : n_blocks) + blk) * C7_TABB;
          c7_dma16s(tsrc, (uint32_t)(ii * 1024 + lane * 16), base + (uint32_t)(G::ROWS_BYTES + ii * 1024));
        }
      } else {
        if (cnt > 0) {
          const int ii = 4 * (P - G::NIW - 2) + wave;
          const int Pq = ii * 64 + lane, R = Pq >> 2, pos = Pq & 3;
          int64_t row = (int64_t)blk * C7_BM + R;
          row = row < n_out ? row : n_out - 1;
          c7_dma16s(dout, (uint32_t)row * do_pitch + (uint32_t)(ch * 64 + pos * 16), lds0 + (uint32_t)(G::DOUT0 + bsel * G::DOUT_BYTES + ii * 1024));
        }
      }
    };
    auto issue_dma = [&](int blk, int cnt, int bsel, const int32_t (&idv)[G::NIW]) {
      ptc_static_for<NP>([&](auto pc) { dma_piece(pc, blk, cnt, bsel, idv); });
    };
    auto count_of = [&](int blk) -> int { return hcnt[blk < n_blocks ? blk : n_blocks - 1]; };

    int cnt_cur = __builtin_amdgcn_readfirstlane(count_of(b_begin));
    load_ids(b_begin);
    issue_dma(b_begin, cnt_cur, 0, ids);
    int cnt_nxt = count_of(b_begin + step);
    if (b_begin + step < b_end) load_ids(b_begin + step);
    __builtin_amdgcn_s_waitcnt(C7_WAIT_VM0 & C7_WAIT_LGKM0);
    __builtin_amdgcn_s_barrier();
    cnt_nxt = __builtin_amdgcn_readfirstlane(cnt_nxt);

    // lane constants of the two transposing reads of a fragment (read i = 0 | 1: the rows of tile 2 hh + i)
    //   gathered operand: address = image base + (entry ^ pcx), pcx = the 16-byte piece this lane reads of a row (XORed into the
    //   swizzled piece-0 offset the table holds) | the 8-byte half of that piece
    const uint32_t pcx = (uint32_t)((cb * 2 + (c4 >> 1)) << 4) | (uint32_t)((c4 & 1) << 3);     // input-channel half kh: | (kh << 6)
    //   dout operand: row 32 (2 hh + i) + 4 s + q of the 64-byte-per-row image, channel block cb, 8-byte chunk c4
    const uint32_t aoff = (uint32_t)((64 * hh + q) * G::DROWB + cb * 32 + c4 * 8);
    //   table: [tap][row in tile = 4 s + q][tile]: one uint32 = tiles (2 hh, 2 hh + 1); accumulator a = tap TM a + tw
    const uint32_t toff = (uint32_t)(G::ROWS_BYTES + q * 8 + hh * 4 + tw * 256);
    int cur = 0;
#pragma unroll 1
    for (int blk = b_begin; blk < b_end; blk += step) {
      // the next block's operands -> the other buffer.  W7_DMA_INLINE: the DMA instructions ride in the MFMA shadows of this block's
      // stream (one every W7_DMA_EVERY pairs from the first pair on) instead of in front of it; either way every piece has been issued
      // long before the wait at the end of the block
      const int cnt_dma = blk + step < b_end ? cnt_nxt : 0;
      int32_t idd[G::NIW];
#pragma unroll
      for (int i = 0; i < G::NIW; ++i) idd[i] = ids[i];
      if constexpr (!W7_DMA_INLINE) issue_dma(blk + step, cnt_dma, cur ^ 1, idd);
      int cnt_nn = count_of(blk + 2 * step);
      if (blk + 2 * step < b_end) load_ids(blk + 2 * step);

      {   // (no `if (cnt_cur > 0)`: the gate at the kernel's entry guarantees that every block fits, and a conditional around the stream puts
          //  every accumulator through a join -- the register allocator then homes the tiles in VGPRs and moves all 224 registers in and
          //  out of the accumulation file once per block)
        const uint32_t ibase = (uint32_t)(cur * W7_BUF1);
        uint32_t pcb[KH];                                                        // entries < 64 KB: base + (e ^ piece) = e ^ (piece | base)
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) pcb[kh] = pcx | (uint32_t)(kh << 6) | ibase;
        const unsigned char* tb = smem + ibase + toff;                           // this lane's table word of (tap slot 0, step 0)
        const unsigned char* ab = smem + (uint32_t)(G::DOUT0 + cur * G::DOUT_BYTES) + aoff;   // this lane's dout bytes of read 0, step 0
        // ONE straight-line stream over the 8 x 7 (step, tap slot) pairs of the block, L = 7 st + a: KH MFMAs per pair on the tap's own
        // accumulators (consecutive MFMAs write different tiles), the gathered fragments three pairs ahead in a ring of six, the A
        // fragment (dout^T, 32 output channels x 16 rows) and the seven table words of step st + 1 read at the first pair of step st.
        // NO branch: see the head of the file for what the branching versions cost.  "No neighbour" entries read the all-zero row.
        frag af[2];                                                              // A fragment of step st in af[st % 2]
        uint32_t te[2][NA];                                                      // table words of step st in te[st % 2]
        frag bfr[6][KH];
        auto sload = [&](auto sc) {
          constexpr int st = decltype(sc)::value;
          af[st % 2] = ld_tr_pair16<frag>(ab + st * 4 * G::DROWB, ab + st * 4 * G::DROWB + 32 * G::DROWB);
#pragma unroll
          for (int a = 0; a < NA; ++a) te[st % 2][a] = *reinterpret_cast<const uint32_t*>(tb + a * 1024 + st * 32);
        };
        auto gload = [&](auto lc) {
          constexpr int L = decltype(lc)::value, st = L / NA, a = L % NA;
          const uint32_t e = te[st % 2][a];
#pragma unroll
          for (int kh = 0; kh < KH; ++kh) bfr[L % 6][kh] = ld_tr_pair16<frag>(smem + ((e & 0xffffu) ^ pcb[kh]), smem + ((e >> 16) ^ pcb[kh]));
        };
        sload(ptc_int<0>{});
        gload(ptc_int<0>{});
        gload(ptc_int<1>{});
        gload(ptc_int<2>{});
        ptc_static_for<8 * NA>([&](auto lc) {
          constexpr int L = decltype(lc)::value, st = L / NA, a = L % NA;
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (a == 0 && st + 1 < 8) sload(ptc_int<(st + 1 < 8 ? st + 1 : st)>{});
#pragma unroll
          for (int kh = 0; kh < KH; ++kh) acc[a * KH + kh] = MM::mma(af[st % 2], bfr[L % 6][kh], acc[a * KH + kh]);
          if constexpr (L + 3 < 8 * NA) gload(ptc_int<(L + 3 < 8 * NA ? L + 3 : L)>{});
          if constexpr (W7_DMA_INLINE && L % W7_DMA_EVERY == 0 && L / W7_DMA_EVERY < NP)
            dma_piece(ptc_int<(L / W7_DMA_EVERY < NP ? L / W7_DMA_EVERY : 0)>{}, blk + step, cnt_dma, cur ^ 1, idd);
        });
      }
      __builtin_amdgcn_s_waitcnt(C7_WAIT_VM0 & C7_WAIT_LGKM0);   // the next block's rows, table, dout and the ids landed
      __builtin_amdgcn_s_barrier();
      cur ^= 1;
      cnt_cur = cnt_nxt;
      cnt_nxt = __builtin_amdgcn_readfirstlane(cnt_nn);
    }
  }

  // ---- this workgroup's partial: D[i = co][j = ci] of tap k: lane (j = lane & 31, hh) holds co = 8 (r / 4) + 4 hh + r % 4
  const int cif = SL ? c_in_full : C, cof = SL ? c_out_full : C;
  float* pout = partial + (int64_t)vb * ((int64_t)cof * 27 * cif);
  const int jj = lane & 31;
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    const int k = 4 * a + tw;
    if (k < 27) {
#pragma unroll
      for (int kh = 0; kh < KH; ++kh)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = ch * 32 + 8 * (r >> 2) + 4 * hh + (r & 3), ci = cib * C + kh * 32 + jj;
          pout[((int64_t)co * 27 + k) * cif + ci] = acc[a * KH + kh][r];
        }
    }
  }
}

template <typename T, int C, bool SL = false>
static int launch_wgrad7_i(const void* in, const void* dout, const uint16_t* tab, const int32_t* hid, const int32_t* hcnt, const int32_t* gate,
                           int64_t n_out, int c_in, int c_out, float* partial, hipStream_t s) {
  const int n_blocks = (int)ptc_cdiv(n_out, C7_BM);
  const int seqs = wgrad7_splits(n_out, c_in, c_out);
  const int grid = seqs * wgrad7_slices(c_in, c_out);
  auto kern = wgrad7_kernel<T, C, SL>;
  PTC_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W7Geom<C>::LDS));
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), W7Geom<C>::LDS, s, (const T*)in, (const T*)dout, tab, hid, hcnt, gate, n_out, n_blocks,
                     partial, c_in, c_out);
  PTC_CHECK_LAUNCH("wgrad7_kernel");
  return PTC_OK;
}
#endif  // PTC_WGRAD7_IMPL
