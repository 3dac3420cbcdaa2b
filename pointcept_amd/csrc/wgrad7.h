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
//     gradient dw[C][27][C] in the ACCUMULATION registers of its four waves as 32x32 fp32 MFMA tiles.  C = 64: a workgroup owns ONE
//     output-channel half ch (two workgroups of one XCD, dispatched back to back, walk the same blocks: the second finds the halo in
//     the L2), wave (kh, tp) the input-channel half kh of the taps k = tp (mod 2): 14 x 16 = 224 of the 256 AGPRs, which leaves every
//     architectural VGPR to the pipeline.  (All 27 taps x one 32x32 quadrant per wave = 432 accumulator registers was the first version:
//     16 of the 27 tiles fit the AGPR file, the compiler shuttles the other 11 through it around every MFMA -- 944 v_accvgpr_write -- and
//     spills 125 registers; `-amdgpu-mfma-vgpr-form` crashes its AGPR-rewrite pass at that pressure.)  C = 32: wave w owns the taps
//     k = w (mod 4) (7 x 16 registers).  The gradient leaves the chip ONCE per workgroup (one fp32 partial per workgroup pair / per
//     workgroup, summed by the deterministic reduction of spconv.hip);
//   * the contraction index of a 32x32x16 MFMA is a ROW, so both operands are needed channel-major.  They come out of the row-major
//     LDS images through ds_read_b64_tr_b16, whose 16 lanes address FOUR ROWS INDEPENDENTLY (each lane supplies the address of 8 bytes
//     of "its" row): for the gathered operand the four row addresses are table entries -- the 27-tap gather and the transposition are
//     the same LDS instruction, nothing is materialised.  An MFMA step contracts the 16 rows {32 t + 4 s + q : t = 0..3, q = 0..3} of a
//     block (4 rows of each of its four 32-row tiles): lane group hh = lane >> 5 takes tiles 2 hh and 2 hh + 1, whose table entries are
//     ADJACENT uint16 in conv7's table layout [tap][row in tile][tile] -- one ds_read_b32 per (tap, step) and lane;
//   * EMPTY (step, tap) PAIRS ARE SKIPPED by scalar branches on the per-step tap masks blocks.hip leaves in the table's padding row;
//     inside an active pair "no neighbour" entries read the all-zero row;
//   * step outer (a dynamic loop of 8), tap inner (static: the accumulator of a tap is a register NAME): consecutive MFMAs write
//     different accumulators; table entries run 4 taps ahead of their MFMA, the gathered fragment 2 taps ahead, in static rings;
//   * block b + 1 (halo rows, table, dout rows) is in flight through global_load_lds while block b is multiplied (conv7's scheme:
//     asm DMA invisible to the compiler's wait-count pass, completion counted by hand).  The halo image keeps conv7's XOR swizzle (the
//     table entries carry it), the dout image holds the workgroup's 32 output channels at 64 bytes per row: the four rows of a transposing
//     read are 256 contiguous bytes = all 64 banks once.  Buffer 1 sits 64 KB after buffer 0: "image base + (entry ^ piece)" is then ONE xor (entries < 64 KB).
// A block whose halo did not fit (hcnt < 0: rows in no spatial order) cannot be served here; the host entry point therefore GATES the
// two kernels on the device-side overflow counter of the table builder: this kernel runs when it is zero, wgrad2 over the whole tensor
// when it is not, each returning at once otherwise -- no host synchronisation, and the reduction reads the partials of whichever ran.
// Summation order differs from wgrad2 (rows of a block in step order, blocks in the workgroup's stride order): results agree to fp32
// rounding of the accumulation, and are bit-reproducible run to run.
#pragma once

#define W7_BUF1 65536                        // LDS byte offset of buffer 1 (halo image + table); buffer 0 at 0
#define W7_DT 8                              // table entries are read W7_DT slots ahead of their MFMA,
#define W7_DG 4                              // gathered fragments W7_DG slots ahead (rings of W7_DT - W7_DG + 1 words / W7_DG + 2 fragments)

static inline bool wgrad7_supported(int dtype, int kv, int c_in, int c_out, int bm, int hcap, int64_t n_out) {
  return conv7_supported(dtype, kv, c_in, c_out, bm, hcap, n_out);
}
#define W7_MAX_WGS 256                       // one persistent workgroup per CU
// workgroups walking DISTINCT block sequences = fp32 partials per call (C = 64: two workgroups -- the output-channel halves -- per sequence)
static inline int wgrad7_splits(int64_t n_out, int c) {
  int64_t nb = ptc_cdiv(n_out, C7_BM), cap = c == 64 ? W7_MAX_WGS / 2 : W7_MAX_WGS;
#ifndef __HIPCC__
  if (const char* e = getenv("PTC_EMU_CONV7_WGS")) cap = atoi(e);   // host emulation only: several blocks per workgroup at test sizes
#endif
  return (int)(nb < cap ? nb : cap);
}

int ptc_wgrad7_launch(int dtype, const void* in, const void* dout, const uint16_t* tab, const int32_t* hid, const int32_t* hcnt,
                      const int32_t* gate, int64_t n_out, int c, float* partial, hipStream_t s);

#ifdef PTC_WGRAD7_IMPL
// ---- the pipeline's LDS reads as inline assembly (absent from the compiler's wait-count bookkeeping) + hand-counted waits.  `addr` =
//      LDS byte address (the dynamic LDS of this kernel starts at 0: checked at entry).  Host emulation: plain loads, no waits.
#ifdef __HIPCC__
template <int OFF> __device__ __forceinline__ uint32_t w7_lds_u32(uint32_t addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <typename F> __device__ __forceinline__ void w7_tr_pair(F& f, uint32_t a0, uint32_t a1) {
  typedef int w7_i32x2 __attribute__((ext_vector_type(2)));
  w7_i32x2 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0));
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a1));
  const ptc_i32x4 v = {lo[0], lo[1], hi[0], hi[1]};
  __builtin_memcpy(&f, &v, sizeof(f));
}
// wait until at most N LDS operations are outstanding; the operand ties the wait to the first use of the value it guards
template <int N, typename F> __device__ __forceinline__ void w7_wait_frag(F& f) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N)); }
template <int N> __device__ __forceinline__ void w7_wait_word(uint32_t& w) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(w) : "n"(N)); }
__device__ __forceinline__ void w7_wait_all() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#else
template <int OFF> __device__ __forceinline__ uint32_t w7_lds_u32(uint32_t addr) { return *reinterpret_cast<const uint32_t*>(smem + addr + OFF); }
template <typename F> __device__ __forceinline__ void w7_tr_pair(F& f, uint32_t a0, uint32_t a1) { f = ld_tr_pair16<F>(smem + a0, smem + a1); }
template <int N, typename F> __device__ __forceinline__ void w7_wait_frag(F&) {}
template <int N> __device__ __forceinline__ void w7_wait_word(uint32_t&) {}
__device__ __forceinline__ void w7_wait_all() {}
#endif

// table reads certainly issued by slots lo .. hi of the stream (slot j reads the word of pair j + DT while that pair exists; slots start at -DT)
constexpr int w7_certain(int lo, int hi, int tot) {
  int n = 0;
  for (int j = lo; j <= hi; ++j)
    if (j >= -W7_DT && j + W7_DT < tot) ++n;
  return n;
}

template <int C> struct W7Geom {
  static constexpr int ROWB = C * 2, PCS = ROWB / 16, RPI = 64 / PCS;
  static constexpr int ROWS_BYTES = (C7_HCAP + 1) * ROWB;                       // + the zero row
  static constexpr int DROWB = 64;                                              // dout image: 32 output channels per row (C = 64: the half `ch`)
  static constexpr int DOUT_BYTES = C7_BM * DROWB;
  static constexpr int DOUT0 = (W7_BUF1 + ROWS_BYTES + C7_TABB + 1023) & ~1023;   // the two dout images follow buffer 1
  static constexpr int LDS = DOUT0 + 2 * DOUT_BYTES;
  static constexpr int NI = (C7_HCAP + RPI - 1) / RPI, NIW = (NI + 3) / 4;      // DMA instructions of a full halo / per wave
  static constexpr int NDW = DOUT_BYTES / 1024 / 4;                             // dout DMA instructions per wave
  static constexpr int NA = C == 64 ? 14 : 7;                                   // accumulators (taps) per wave: taps TM a + (tp | wave)
  static constexpr int TM = C == 64 ? 2 : 4;
  static constexpr int TSTRIDE = TM * 256;                                      // table bytes between two taps of a wave
  static __device__ __forceinline__ int swz(int slot) { return C == 64 ? ((slot >> 1) & 7) : ((slot >> 2) & 3); }
  static_assert(ROWS_BYTES + C7_TABB <= W7_BUF1, "buffer 0 must end before buffer 1");
  static_assert(LDS <= 163840, "LDS budget");
};

template <typename T, int C>
__global__ void __launch_bounds__(256, 1)
wgrad7_kernel(const T* __restrict__ in, const T* __restrict__ dout, const uint16_t* __restrict__ tab, const int32_t* __restrict__ hid,
              const int32_t* __restrict__ hcnt, const int32_t* __restrict__ gate, int64_t n_out, int n_blocks, float* __restrict__ partial) {
  using frag = typename Mma<T>::frag;
  using MM = C7Mma<T>;
  using G = W7Geom<C>;
  constexpr int ROWB = G::ROWB, PCS = G::PCS, RPI = G::RPI, NA = G::NA;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (gate != nullptr && *gate != 0) return;              // some block overflowed: wgrad2 serves this call (see the head of the file)
  const int lane = ptc_lane(), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lp = lane & 15, g4 = lane >> 4, hh = g4 >> 1, cb = g4 & 1;   // transposing-read roles: 16-lane group g4 = (k-group hh, channel block cb)
  const int q = lp >> 2, c4 = lp & 3;                                    // ... lane: row q of the read's four, channels 4 c4 .. 4 c4 + 3 of the block
  const int kh = C == 64 ? (wave & 1) : 0;                               // input-channel half (B operand)
  const int tw = C == 64 ? (wave >> 1) : wave;                           // this wave's taps: TM a + tw
  // C = 64: workgroup = (block sequence, output-channel half ch); hardware workgroup b runs on XCD b % 8, so the two halves of a sequence
  // are b = x + 8 (2 p) and x + 8 (2 p + 1): same XCD, dispatched back to back.  Sequence `seq` of `step` takes blocks seq', seq' + step, ..
  // with seq' as in conv7 (one round's blocks of an XCD adjacent)
  const int wgs = (int)gridDim.x, vb = C == 64 ? (wgs % 16 == 0 ? (((int)blockIdx.x >> 4) << 3 | ((int)blockIdx.x & 7)) : ((int)blockIdx.x >> 1)) : (int)blockIdx.x;
  const int ch = C == 64 ? (wgs % 16 == 0 ? (((int)blockIdx.x >> 3) & 1) : ((int)blockIdx.x & 1)) : 0;
  const int step = C == 64 ? wgs / 2 : wgs;
  const int b_begin = (step % 8 == 0) ? (vb & 7) * (step / 8) + (vb >> 3) : vb;
  const int b_end = n_blocks;

  c7_f32x16 acc[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  if (b_begin < b_end) {
    // ---- zero rows of both halo images (never written by the DMA)
    if (threadIdx.x < 2 * PCS) {
      const int bsel = threadIdx.x / PCS, pc = threadIdx.x % PCS;
      *reinterpret_cast<uint4*>(smem + bsel * W7_BUF1 + C7_HCAP * ROWB + pc * 16) = make_uint4(0, 0, 0, 0);
    }
    const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c7_lds_addr(smem));
#ifdef __HIPCC__
    if (lds0 != 0) __builtin_trap();                       // the pipeline's assembly reads address LDS from 0 (no static LDS in this kernel)
#endif
    const int drow = lane / PCS, dpos = lane % PCS;
    int32_t ids[G::NIW];
    auto load_ids = [&](int blk) {
#pragma unroll
      for (int i = 0; i < G::NIW; ++i) {
        const int slot = (4 * i + wave) * RPI + drow;
        ids[i] = hid[(int64_t)blk * C7_HCAP + (slot < C7_HCAP ? slot : C7_HCAP - 1)];
      }
    };
    auto issue_dma = [&](int blk, int cnt, int bsel) {
      const uint32_t base = lds0 + (uint32_t)(bsel * W7_BUF1);
#pragma unroll
      for (int i = 0; i < G::NIW; ++i) {
        const int ii = 4 * i + wave;
        if (ii * RPI < cnt) {                                  // wave-uniform
          const int slot = ii * RPI + drow;
          const int piece = dpos ^ G::swz(slot);
          c7_dma16(reinterpret_cast<const unsigned char*>(in) + (int64_t)ids[i] * ROWB + piece * 16, base + (uint32_t)(ii * 1024));
        }
      }
      if (cnt > 0) {
        const unsigned char* tsrc = reinterpret_cast<const unsigned char*>(tab) + ((int64_t)(C == 64 ? 0 : n_blocks) + blk) * C7_TABB;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int ii = 4 * i + wave;
          if (ii < C7_TABB / 1024) c7_dma16(tsrc + ii * 1024 + lane * 16, base + (uint32_t)(G::ROWS_BYTES + ii * 1024));
        }
        // the 32 output channels of the block's dout rows this workgroup owns, 64 bytes per row (the four rows of a transposing read
        // are 256 contiguous bytes: all 64 banks once).  Rows past the end repeat the last row: their table entries are "no neighbour",
        // the products are zero.
#pragma unroll
        for (int i = 0; i < G::NDW; ++i) {
          const int ii = 4 * i + wave;
          const int P = ii * 64 + lane, R = P >> 2, pos = P & 3;
          int64_t row = (int64_t)blk * C7_BM + R;
          row = row < n_out ? row : n_out - 1;
          c7_dma16(reinterpret_cast<const unsigned char*>(dout) + row * ROWB + ch * 64 + pos * 16,
                   lds0 + (uint32_t)(G::DOUT0 + bsel * G::DOUT_BYTES + ii * 1024));
        }
      }
    };
    auto count_of = [&](int blk) -> int { return hcnt[blk < n_blocks ? blk : n_blocks - 1]; };

    int cnt_cur = __builtin_amdgcn_readfirstlane(count_of(b_begin));
    load_ids(b_begin);
    issue_dma(b_begin, cnt_cur, 0);
    int cnt_nxt = count_of(b_begin + step);
    if (b_begin + step < b_end) load_ids(b_begin + step);
    __builtin_amdgcn_s_waitcnt(C7_WAIT_VM0 & C7_WAIT_LGKM0);
    __builtin_amdgcn_s_barrier();
    cnt_nxt = __builtin_amdgcn_readfirstlane(cnt_nxt);

    // lane constants of the two transposing reads of a fragment (read i = 0 | 1: the rows of tile 2 hh + i)
    //   gathered operand: address = image base + (entry ^ pcx), pcx = the 16-byte piece this lane reads of a row (XORed into the
    //   swizzled piece-0 offset the table holds) | the 8-byte half of that piece
    const uint32_t pcx = (uint32_t)(((C == 64 ? kh * 4 : 0) + cb * 2 + (c4 >> 1)) << 4) | (uint32_t)((c4 & 1) << 3);
    //   dout operand: row 32 (2 hh + i) + 4 s + q of the 64-byte-per-row image, channel block cb, 8-byte chunk c4
    const uint32_t aoff = (uint32_t)((64 * hh + q) * G::DROWB + cb * 32 + c4 * 8);
    //   table: [tap][row in tile = 4 s + q][tile]: one uint32 = tiles (2 hh, 2 hh + 1); accumulator a = tap TM a + tw
    const uint32_t toff = (uint32_t)(G::ROWS_BYTES + q * 8 + hh * 4 + tw * 256);
    int cur = 0;
#pragma unroll 1
    for (int blk = b_begin; blk < b_end; blk += step) {
      if (blk + step < b_end) issue_dma(blk + step, cnt_nxt, cur ^ 1);
      int cnt_nn = count_of(blk + 2 * step);
      if (blk + 2 * step < b_end) load_ids(blk + 2 * step);

      if (cnt_cur > 0) {
        const uint32_t ibase = (uint32_t)(cur * W7_BUF1);
        const uint32_t pcb = pcx | ibase;                                        // entries < 64 KB: base + (e ^ pcx) = e ^ (pcx | base)
        const uint32_t tadr = ibase + toff;                                      // this lane's table word of tap slot 0, step 0
        const uint32_t aadr = (uint32_t)(G::DOUT0 + cur * G::DOUT_BYTES) + aoff; // this lane's dout bytes of read 0, step 0
        // the eight per-step tap masks (blocks.hip, table row 27), this wave's taps only: bit TM a of ms[s] = tap TM a + tw at step s
        uint32_t ms[8];
        {
          const unsigned char* mrow = smem + ibase + G::ROWS_BYTES + 27 * 256 + 20;
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)*reinterpret_cast<const uint32_t*>(mrow + 4 * s));
            ms[s] = (m >> tw) & (C == 64 ? 0x5555555u : 0x1111111u);             // (k < 27 follows: the masks have 27 bits)
          }
        }
        // The pipeline: ONE static stream over the 8 x NA (step, tap slot) pairs of the block, L = s NA + a.  Slot L: M(L) = the MFMA,
        // G(L + DG) = the two transposing gathers of a later pair, T(L + DT) = the table word of a still later one (slots -DT .. -1 are
        // the prologue); the A fragment of step s + 1 is read at the first slot of step s.  Everything is a compile-time name: accumulator
        // (a), ring slots (L), mask register (s), LDS offsets (immediates) -- ~110 bytes of code per pair.  The LDS reads of the stream
        // are INLINE ASSEMBLY with hand-counted waits (w7_wait_*): the gathers are conditional, LDS results return in order, and the
        // compiler -- which cannot know which gathers were issued -- guards a use with the count that is safe on EVERY path (the first
        // build waited, in front of every gather, for a read issued three instructions earlier, and in front of every conditional MFMA
        // for everything).  The counts below are the reads CERTAINLY issued behind the one needed: in front of M(L) the table reads of
        // slots L - DG .. L - 1, in front of G(L + DG) those of slots L + DG - DT + 1 .. L - 1.  With every gather in between issued, a
        // wait for "at most c outstanding" ends with the (c - 3)-th newest read of slot L - 2 (a slot issues two gathers and one table
        // word): c = DG = 4 and c = DT - DG - 1 = 3 keep every wait on reads at least two MFMAs old.  A fragment buffer is rewritten two MFMAs after the MFMA that read it (a reload whose destination an
        // MFMA in flight still reads is held until it retires, conv7.h).
        constexpr int TOT = 8 * NA, RB = W7_DG + 2, RT = W7_DT - W7_DG + 1;
        uint32_t te[RT];
        frag bf[RB], af[2];
        w7_tr_pair(af[0], aadr, aadr + 32 * G::DROWB);
        ptc_static_for<W7_DT + TOT>([&](auto sc) {
          constexpr int L = decltype(sc)::value - W7_DT;
          if constexpr (L >= 0) {
            constexpr int s = L / NA, a = L % NA;
            if constexpr (a == 0 && s + 1 < 8) w7_tr_pair(af[(s + 1) % 2], aadr + (s + 1) * 4 * G::DROWB, aadr + (s + 1) * 4 * G::DROWB + 32 * G::DROWB);
            if ((ms[s] >> (G::TM * a)) & 1u) {
              w7_wait_frag<w7_certain(L - W7_DG, L - 1, TOT)>(bf[L % RB]);
              acc[a] = MM::mma(af[s % 2], bf[L % RB], acc[a]);
            }
          }
          if constexpr (L + W7_DG >= 0 && L + W7_DG < TOT) {
            constexpr int Lg = L + W7_DG, s = Lg / NA, a = Lg % NA;
            if ((ms[s] >> (G::TM * a)) & 1u) {
              uint32_t& e = te[Lg % RT];
              w7_wait_word<w7_certain(Lg - W7_DT + 1, L - 1, TOT)>(e);
              w7_tr_pair(bf[Lg % RB], (e & 0xffffu) ^ pcb, (e >> 16) ^ pcb);
            }
          }
          if constexpr (L + W7_DT < TOT) {
            constexpr int Lt = L + W7_DT, s = Lt / NA, a = Lt % NA;
            te[Lt % RT] = w7_lds_u32<s * 32 + a * G::TSTRIDE>(tadr);
          }
        });
        w7_wait_all();                                                           // (table words of skipped pairs)
      }
      __builtin_amdgcn_s_waitcnt(C7_WAIT_VM0 & C7_WAIT_LGKM0);   // the next block's rows, table, dout and the ids landed
      __builtin_amdgcn_s_barrier();
      cur ^= 1;
      cnt_cur = cnt_nxt;
      cnt_nxt = __builtin_amdgcn_readfirstlane(cnt_nn);
    }
  }

  // ---- this workgroup's partial: D[i = co][j = ci] of tap k: lane (j = lane & 31, hh) holds co = 8 (r / 4) + 4 hh + r % 4
  float* pout = partial + (int64_t)vb * ((int64_t)C * 27 * C);
  const int jj = lane & 31;
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    const int k = G::TM * a + tw;
    if (k < 27) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = ch * 32 + 8 * (r >> 2) + 4 * hh + (r & 3), ci = kh * 32 + jj;
        pout[((int64_t)co * 27 + k) * C + ci] = acc[a][r];
      }
    }
  }
}

template <typename T, int C>
static int launch_wgrad7_i(const void* in, const void* dout, const uint16_t* tab, const int32_t* hid, const int32_t* hcnt, const int32_t* gate,
                           int64_t n_out, float* partial, hipStream_t s) {
  const int n_blocks = (int)ptc_cdiv(n_out, C7_BM);
  const int seqs = wgrad7_splits(n_out, C);
  const int grid = C == 64 ? 2 * seqs : seqs;
  auto kern = wgrad7_kernel<T, C>;
  PTC_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W7Geom<C>::LDS));
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), W7Geom<C>::LDS, s, (const T*)in, (const T*)dout, tab, hid, hcnt, gate, n_out, n_blocks,
                     partial);
  PTC_CHECK_LAUNCH("wgrad7_kernel");
  return PTC_OK;
}
#endif  // PTC_WGRAD7_IMPL
