// conv7.h -- forward / dgrad kernel of the SUBMANIFOLD 3^3 gather-table convolution for 16-bit features with c_in = c_out = 32 | 64
// (PTv3's stage-0 / stage-1 positional-encoding convolutions and SpUNet's level-0 / level-1 blocks: the N ~ 8e5-row, HBM-bound ones).
// WEIGHT-STATIONARY IN REGISTERS, input rows staged once per 128-row block in LDS by the DMA path.  Included by spconv.hip.
//
// Why (round 3, profiles/r03_a_conv6_*): conv5 (whole-line global gathers, W streamed through LDS per 128-row workgroup) runs the
// dec0 shape (64 -> 64, N = 819200) in 236 us = 0.16 of its HBM roof with the vector-memory address path 71 % busy, and COMPACTING
// its gathers (conv6: half the vector-memory instructions) made it SLOWER (298 us): the kernel is bound by dependent-load latency
// (table entry -> gather -> LDS bounce -> MFMA at 2-3 waves per SIMD) and by re-streaming W -- 221 KB per 128 rows, more bytes
// through the L1 / LDS-store path than the gathered rows themselves.  conv7 removes both:
//   * a workgroup is PERSISTENT over a share of the 128-row blocks and keeps the whole weight tensor in the registers of
//     its four waves (one wave per SIMD, 512-register budget) as 32x32x16 MFMA A-fragments (32 output x 16 input channels), 54 per
//     wave.  C = 64: wave w holds input-channel half kh = w & 1 of output-channel half ch = w >> 1 (27 taps x 2 k-steps);
//     C = 32: every wave holds all of W (27 x 2) and the waves split the block's row tiles.  W is read from L2 ONCE per workgroup;
//   * the distinct input rows a block names (its "halo": 1.66 x 128 rows on curve-ordered scenes, blocks.hip) and the block's
//     local table arrive in LDS through global_load_lds (16 B per lane, no VGPR staging, no ds_write), double-buffered: block
//     b + 1 is in flight while block b is multiplied; rows are XOR-swizzled on the SOURCE side (the DMA image is lane-linear) so
//     that the 27-tap gather -- a per-lane ds_read_b128 in MFMA B layout -- is conflict-free for neighbouring rows; the table
//     holds ready-made LDS byte offsets, so a gather costs two vector-ALU instructions (field extract, xor-add);
//   * tap outer (static: the fragment index must be a compile-time register name), the block's 32-row tiles inner; EMPTY TAPS ARE
//     SKIPPED by a scalar branch on the tap mask blocks.hip leaves in the table (33 % of the (block, tap) pairs, 46 % of the (tile,
//     tap) pairs on indoor scenes); inside an active tap "no neighbour" entries read an all-zero row.  Per (tap, tile, k-step): one
//     LDS gather + one 32-cycle MFMA per wave, the gathers one tap ahead in a rolling register ring.  (The first version used
//     16x16x32 MFMAs: at one wave per SIMD a 16-cycle MFMA hides ~1 other instruction and the loop ran at 36 cycles per MFMA --
//     tools/probe_mfma_agpr.hip, profiles/r03_f_probe_mfma_agpr.txt; a 32-cycle MFMA hides ~5);
//   * workgroups take blocks STRIDED over the scene (not a contiguous range): block cost follows the local geometry and
//     contiguous ranges left the waves idle for a quarter of the kernel;
//   * C = 64: the two input-channel halves of a row tile are summed through a 32 KB LDS scratch (each wave finishes half of the
//     tiles: symmetric work), bias added in fp32, rows stored as bf16 / f16 in 16-byte pieces (channel groups traded across the
//     two 32-lane halves by v_permlane32_swap).
// Where the cycles of a block go at 64 -> 64 (wave 0, profiles/r03_o_conv7_phases.txt): tap loop 5630 of 8900 (its MFMAs alone: 4500),
// DMA issue 1020, accumulators -> scratch 430, add + store 1290, first gathers 350, barriers 190.  One wave per SIMD: nothing overlaps
// the phases outside the tap loop -- that, not the LDS (a conflict-free gather pattern changes nothing) or the MFMA count, is what
// separates the kernel from its roof.
// A block whose halo does not fit (rows in no spatial order: hcnt < 0) is skipped here and served by conv5, launched behind this
// kernel over exactly those blocks (its 128-row workgroups coincide with the blocks).  Summation order differs from conv5 (k-halves
// summed last, bias last): results agree to fp32 rounding of the accumulation, not bit for bit.
#pragma once

// timing ablations (tools only: `python -m pointcept_amd.build --variant d_C7_ABLATE_<bits>`, results are then WRONG): 1 no halo DMA
// after the first block, 2 no tap loop, 4 no epilogue at all, 8 no LDS exchange of the k-halves, 16 no global stores, 32 gathers read consecutive slots (no bank conflicts),
// 64 per-phase cycle counters instead of results (tools/conv7_time.py --phases)
#ifndef C7_ABLATE
#define C7_ABLATE 0
#endif
#ifndef C7_DMA_SADDR
#define C7_DMA_SADDR 0                      // 1: DMA source = SGPR base + 32-bit lane offset (wgrad7's form) instead of 64-bit lane pointers: measured
                                            // SLOWER here, 148.0 vs 141.4 us at 64 -> 64 (profiles/r04_c_conv7_time.txt)
#endif
#ifndef C7_DMA_IN_TAPS
#define C7_DMA_IN_TAPS 0                    // 1 (round 5, measured and left off): the DMA pieces of the NEXT block's halo and the id loads of the block
                                            // after it issued between the statically unrolled taps of THIS block's loop (piece q in front of tap q)
                                            // instead of in front of the loop, where nothing overlaps them (1020 of a block's 8900 cycles at
                                            // 64 -> 64).  SLOWER: 160.1 vs 145.0 us at 64 -> 64, 68.8 vs 58.0 us at 32 -> 32 (N = 819200,
                                            // profiles/r05_i_conv7_dma_in_taps.txt) -- each piece is an M0 save / set / restore around the DMA plus
                                            // its address arithmetic in front of a tap's first MFMA, and the wave has nothing else to issue
#endif
#ifndef C7_LAG
#define C7_LAG 2                            // MFMAs between a fragment's use and its reload (C = 64; 3 and 4 measured: see conv7_time)
#endif
#define C7_BM 128                           // rows per block
#define C7_NT 8                             // 16-row tiles per block
#define C7_HCAP 416                         // halo capacity (rows); max observed on curve-ordered indoor scenes: 352
#define C7_TABB (28 * 16 * C7_NT * 2)       // bytes of one block's local table: [28 taps (27 + pad)][16 rows-in-tile][8 tiles] u16, each
                                            // entry = LDS byte offset of piece 0 of the neighbour row inside the row image (blocks.hip)

static inline bool conv7_supported(int dtype, int kv, int c_in, int c_out, int bm, int hcap, int64_t n_out) {
  return dtype != PTC_F32 && kv == 27 && c_in == c_out && (c_in == 32 || c_in == 64) && bm == C7_BM && hcap == C7_HCAP && n_out >= 4096;
}


// the entry point of conv7.hip (its own translation unit: built with MFMA accumulators in architectural VGPRs, see build.py)
int ptc_conv7_launch(int dtype, const void* in, int64_t n_in, const void* w, const float* bias, const uint16_t* tab, const int32_t* hid,
                     const int32_t* hcnt, int64_t n_out, int c, void* out, hipStream_t s);

#ifdef PTC_CONV7_IMPL
// s_waitcnt immediates (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14])
#define C7_WAIT_VM0 0x0F70
#define C7_WAIT_LGKM0 0xC07F

// 16 bytes per lane global -> LDS (destination = wave-uniform LDS byte address `lds_dst` + 16 * lane).  Inline assembly on purpose: the
// compiler's wait-count pass treats an LDS-DMA it can see as a pending LDS write that any later ds_read may alias and drains the
// vector-memory counter in front of the first gather -- the DMA of block b + 1 would never overlap block b's MFMAs.  An asm DMA is
// absent from that bookkeeping (cdna_hip_programming.md, "what hipcc does not do"): its completion is counted by hand -- vmcnt(0),
// then a workgroup barrier, then the reads.  M0 (the DMA's LDS base) is saved and restored inside the statement.
#ifdef __HIPCC__
__device__ __forceinline__ void c7_dma16(const void* g, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}
// the same in the SGPR-base + 32-bit-VGPR-offset form (round 4): a piece costs one 32-bit multiply-add of address arithmetic instead of
// a 64-bit add chain, and nothing per-lane and 64 bits wide is loop-invariant (with the 64-bit form the compiler hoists one lane
// pointer per piece out of the block loop -- 26 registers, spilled in wgrad7's first builds).  Tensors < 2 GiB (the host checks).
__device__ __forceinline__ void c7_dma16s(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ uint32_t c7_lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
// keep a weight fragment in the accumulation-register half of the unified file (MFMA A operands may be AGPRs): the 216 weight
// registers then leave the 256 architectural VGPRs to the accumulators and the gather ring
#define C7_PIN_AGPR(x) asm volatile("" : "+a"(x))
#else
__device__ __forceinline__ void c7_dma16(const void* g, uint32_t lds_dst) { emu_global_load_lds(g, smem + lds_dst, 16); }
__device__ __forceinline__ void c7_dma16s(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  emu_global_load_lds(reinterpret_cast<const unsigned char*>(sbase) + voff, smem + lds_dst, 16);
}
__device__ __forceinline__ uint32_t c7_lds_addr(const void* p) { return (uint32_t)((const unsigned char*)p - smem); }
#define C7_PIN_AGPR(x) ((void)0)
#endif

// a[lanes 32..63] <-> b[lanes 0..31] (v_permlane32_swap): afterwards the low half holds (own a, the high half's a) and the high half
// (the low half's b, own b)
#ifdef __HIPCC__
__device__ __forceinline__ void c7_swap32(uint32_t& a, uint32_t& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}
#else
__device__ __forceinline__ void c7_swap32(uint32_t& a, uint32_t& b) {
  const int l = emu::lane_id();
  const uint32_t oa = emu::exchange(a, l ^ 32), ob = emu::exchange(b, l ^ 32);
  if (l < 32) b = oa; else a = ob;
}
#endif

template <int C> struct C7Geom {
  static constexpr int ROWB = C * 2;                  // bytes per feature row
  static constexpr int PCS = ROWB / 16;               // 16-byte pieces per row
  static constexpr int RPI = 64 / PCS;                // rows per DMA instruction (1 KB)
  static constexpr int ROWS_BYTES = (C7_HCAP + 1) * ROWB;   // + the zero row
  static constexpr int BUF = ROWS_BYTES + C7_TABB;    // one halo buffer
  static constexpr int SCRATCH = C == 64 ? 32768 : 0;
  static constexpr int LDS = 2 * BUF + SCRATCH;
  static constexpr int NI = (C7_HCAP + RPI - 1) / RPI;        // DMA instructions of a full halo
  static constexpr int NIW = (NI + 3) / 4;                    // ... per wave
  // swizzle: piece p of the row in slot s sits at position p ^ swz(s); 16 consecutive slots x one piece index = 16 distinct bank quads
  static __device__ __forceinline__ int swz(int slot) { return C == 64 ? PTC_SWZ64(slot) : ((slot >> 2) & 3); }
};

typedef __attribute__((ext_vector_type(16))) float c7_f32x16;
template <typename T> struct C7Mma;
template <> struct C7Mma<bf16_t> {
  static __device__ __forceinline__ c7_f32x16 mma(s16x8 a, s16x8 b, c7_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct C7Mma<f16_t> {
  static __device__ __forceinline__ c7_f32x16 mma(h16x8 a, h16x8 b, c7_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

template <typename T, int C>
__global__ void __launch_bounds__(256, 1)
conv7_kernel(const T* __restrict__ in, const T* __restrict__ w, const float* __restrict__ bias, const uint16_t* __restrict__ tab,
             const int32_t* __restrict__ hid, const int32_t* __restrict__ hcnt, int64_t n_out, int n_blocks, T* __restrict__ out) {
  using frag = typename Mma<T>::frag;
  using MM = C7Mma<T>;
  using G = C7Geom<C>;
  constexpr int ROWB = G::ROWB, PCS = G::PCS, RPI = G::RPI;
  constexpr int TW = C == 64 ? 4 : 1;                  // 32-row tiles a wave multiplies per block
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = ptc_lane(), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int j = lane & 31, h = lane >> 5;              // MFMA 32x32x16: A row (output channel) / B column (row of the tile), k-group
  const int kh = C == 64 ? (wave & 1) : 0;             // input-channel half
  const int ch = C == 64 ? (wave >> 1) : 0;            // output-channel half (C = 64); C = 32: all 32 output channels in every wave
  // Which blocks: workgroup w takes blocks p(w), p(w) + S, p(w) + 2 S, ... (S = the grid).  STRIDED, not a contiguous range: the cost
  // of a block follows the local geometry (active taps, halo size) and contiguous ranges left the waves alive for only 74 % of
  // the kernel's duration (rocprofv3 SQ_WAVE_CYCLES vs GRBM_GUI_ACTIVE, profiles/r03_o_conv7_pmc_sq.json); a strided workgroup samples
  // the whole scene.  p keeps one round's blocks of an XCD adjacent (hardware workgroup w runs on XCD w % 8, each XCD has its own
  // L2): XCD x takes blocks x S/8 .. (x + 1) S/8 - 1 of every round, so the halo rows neighbouring blocks share meet in one L2.
  const int step = (int)gridDim.x;
  const int b_begin = (step % 8 == 0) ? ((int)blockIdx.x & 7) * (step / 8) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  const int b_end = n_blocks;
  if (b_begin >= b_end) return;

  // ---- the weights: 27 taps x 2 k-steps of 16 = 54 A-fragments (32 output channels x 16 input channels) per wave, straight from
  //      global memory (L2) into registers
  frag wf[27][2];
#pragma unroll
  for (int k = 0; k < 27; ++k)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wf[k][ks] = ld_frag<T>(w + ((int64_t)(ch * 32 + j) * 27 + k) * C + kh * 32 + ks * 16 + h * 8);
  // the accumulators of an MFMA live in the accumulation half of the register file (64 of its 256 registers here), which leaves it
  // 192 registers = 48 of the 54 weight fragments; the last three taps stay in architectural VGPRs beside the gather ring
#pragma unroll
  for (int k = 0; k < (C == 64 ? 24 : 27); ++k)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) C7_PIN_AGPR(wf[k][ks]);
  // D[i][jj] of the 32x32 MFMA: lane (jj = row, h) holds output channels i = 8 (r / 4) + 4 h + r % 4, r = 0..15
  float bsv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bsv[r] = bias ? bias[ch * 32 + 8 * (r >> 2) + 4 * h + (r & 3)] : 0.f;

  // ---- zero rows of both buffers (never written by the DMA)
  if (threadIdx.x < 2 * PCS) {
    const int bsel = threadIdx.x / PCS, pc = threadIdx.x % PCS;
    *reinterpret_cast<uint4*>(smem + bsel * G::BUF + C7_HCAP * ROWB + pc * 16) = make_uint4(0, 0, 0, 0);
  }

  // ---- DMA of one block's halo rows + table into buffer `bsel`; ids = this wave's share of the halo list (instruction q of the
  //      wave = instruction 4 q + wave of the block: rows (4 q + wave) * RPI + lane / PCS)
  const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c7_lds_addr(smem));
  const int drow = lane / PCS, dpos = lane % PCS;
  int32_t ids[G::NIW];
  auto load_ids = [&](int blk) {
#pragma unroll
    for (int q = 0; q < G::NIW; ++q) {
      const int slot = (4 * q + wave) * RPI + drow;
      ids[q] = hid[(int64_t)blk * C7_HCAP + (slot < C7_HCAP ? slot : C7_HCAP - 1)];
    }
  };
  auto issue_dma = [&](int blk, int cnt, int bsel) {
    const uint32_t base = lds0 + (uint32_t)(bsel * G::BUF);
#pragma unroll
    for (int q = 0; q < G::NIW; ++q) {
      const int i = 4 * q + wave;
      if (i * RPI < cnt) {                                   // wave-uniform
        const int slot = i * RPI + drow;
        const int piece = dpos ^ G::swz(slot);
        if constexpr (C7_DMA_SADDR) c7_dma16s(in, (uint32_t)ids[q] * (uint32_t)ROWB + (uint32_t)(piece * 16), base + (uint32_t)(i * 1024));
        else c7_dma16(reinterpret_cast<const unsigned char*>(in) + (int64_t)ids[q] * ROWB + piece * 16, base + (uint32_t)(i * 1024));
      }
    }
    if (cnt > 0) {
      const unsigned char* tsrc = reinterpret_cast<const unsigned char*>(tab) + ((int64_t)(C == 64 ? 0 : n_blocks) + blk) * C7_TABB;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int i = 4 * q + wave;
        if (i < C7_TABB / 1024) {
          if constexpr (C7_DMA_SADDR) c7_dma16s(tsrc, (uint32_t)(i * 1024 + lane * 16), base + (uint32_t)(G::ROWS_BYTES + i * 1024));
          else c7_dma16(tsrc + i * 1024 + lane * 16, base + (uint32_t)(G::ROWS_BYTES + i * 1024));
        }
      }
    }
  };
  // piece q of a block's halo (this wave's DMA instruction q) and of its id list, for the in-loop form (C7_DMA_IN_TAPS)
  auto issue_dma_piece = [&](auto qc, int cnt, int bsel) {
    constexpr int q = decltype(qc)::value;
    const int i = 4 * q + wave;
    if (i * RPI < cnt) {                                     // wave-uniform
      const int slot = i * RPI + drow;
      const int piece = dpos ^ G::swz(slot);
      const uint32_t base = lds0 + (uint32_t)(bsel * G::BUF);
      if constexpr (C7_DMA_SADDR) c7_dma16s(in, (uint32_t)ids[q] * (uint32_t)ROWB + (uint32_t)(piece * 16), base + (uint32_t)(i * 1024));
      else c7_dma16(reinterpret_cast<const unsigned char*>(in) + (int64_t)ids[q] * ROWB + piece * 16, base + (uint32_t)(i * 1024));
    }
  };
  auto load_id_piece = [&](auto qc, int blk) {
    constexpr int q = decltype(qc)::value;
    const int slot = (4 * q + wave) * RPI + drow;
    ids[q] = hid[(int64_t)blk * C7_HCAP + (slot < C7_HCAP ? slot : C7_HCAP - 1)];
  };
  auto issue_table_dma = [&](int blk, int cnt, int bsel) {
    if (cnt > 0) {
      const uint32_t base = lds0 + (uint32_t)(bsel * G::BUF);
      const unsigned char* tsrc = reinterpret_cast<const unsigned char*>(tab) + ((int64_t)(C == 64 ? 0 : n_blocks) + blk) * C7_TABB;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int i = 4 * q + wave;
        if (i < C7_TABB / 1024) {
          if constexpr (C7_DMA_SADDR) c7_dma16s(tsrc, (uint32_t)(i * 1024 + lane * 16), base + (uint32_t)(G::ROWS_BYTES + i * 1024));
          else c7_dma16(tsrc + i * 1024 + lane * 16, base + (uint32_t)(G::ROWS_BYTES + i * 1024));
        }
      }
    }
  };
  // (Tried: plain 16-byte loads into 60 staging registers at the top of a block, ds_write_b128 behind its tap loop.  The loads issue
  //  faster -- 940 vs 1450 cycles per block -- but the stores into LDS cost 780, 270 more than they save: profiles/r03_o_conv7_phases.txt.
  //  Also tried: the halo ids as two lane-linear vectors per wave + a cross-lane read per piece instead of 13 small loads: the
  //  dependent chain load -> ds_bpermute -> address -> DMA issues SLOWER, 1350 vs 1020 cycles per block.)
  // (a count is LOADED early and USED late: the compiler waits for an ordinary load at the first use of its result, and that wait
  //  would also drain the DMAs issued before it)
  auto count_of = [&](int blk) -> int { return hcnt[blk < n_blocks ? blk : n_blocks - 1]; };

  // prologue: block b_begin -> buffer 0
  int cnt_cur = __builtin_amdgcn_readfirstlane(count_of(b_begin));
  load_ids(b_begin);
  issue_dma(b_begin, cnt_cur, 0);
  int cnt_nxt = count_of(b_begin + step);
  if (b_begin + step < b_end) load_ids(b_begin + step);
  __builtin_amdgcn_s_waitcnt(C7_WAIT_VM0 & C7_WAIT_LGKM0);
  __builtin_amdgcn_s_barrier();
  cnt_nxt = __builtin_amdgcn_readfirstlane(cnt_nxt);

  // the 16-byte piece of a gathered row this lane feeds to k-step ks: input channels kh * 32 + ks * 16 + h * 8 ..
  const uint32_t pxor0 = (uint32_t)(kh * 4 + h) << 4, pxor1 = (uint32_t)(kh * 4 + 2 + h) << 4;
  // (bit 64 of C7_ABLATE: per-phase cycle totals of every workgroup's wave 0 -> the first 64 bytes of output row blockIdx.x)
  long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
  auto tick = [&](int i) {
    if constexpr ((C7_ABLATE & 64) != 0) {
      const long long t = clock64();
      tph[i] += t - tlast;
      tlast = t;
    }
  };
  if constexpr ((C7_ABLATE & 64) != 0) tlast = clock64();
  int cur = 0;
#pragma unroll 1
  for (int blk = b_begin; blk < b_end; blk += step) {
    // this workgroup's next block -> the other buffer (its ids arrived before the barrier that ended the previous iteration)
    const bool dma_next = blk + step < b_end && !(C7_ABLATE & 1);
    const bool ids_nn = blk + 2 * step < b_end;
    // in-loop form: only the block's table goes out here; halo piece q is issued in front of tap q of the loop below (a piece's id
    // register is free as soon as its DMA has issued: the id of the block after next is loaded into it at once).  A block whose own
    // halo overflowed has no tap loop: everything is issued here, as before.
    const bool in_taps = C7_DMA_IN_TAPS != 0 && cnt_cur > 0 && !(C7_ABLATE & 2);
    if (dma_next) {
      if (in_taps) issue_table_dma(blk + step, cnt_nxt, cur ^ 1);
      else issue_dma(blk + step, cnt_nxt, cur ^ 1);
    }
    int cnt_nn = count_of(blk + 2 * step);
    if (ids_nn && !in_taps) load_ids(blk + 2 * step);

    tick(0);                                            // DMA issue + id loads
    const unsigned char* rowsL = smem + cur * G::BUF;
    // table [28 taps][32 rows of a tile][4 tiles] u16; a wave's LOCAL tile tl is the block's tile (tl + 2 kh) & 3 (C = 64: local tiles
    // 0, 1 are the ones it finishes in the epilogue, 2, 3 the ones it hands to its partner -- every accumulator index is then a
    // compile-time constant); C = 32: wave w multiplies tile w
    const unsigned char* tabL = rowsL + G::ROWS_BYTES + j * 8 + (C == 64 ? 0 : wave * 2);
    const int64_t row0 = (int64_t)blk * C7_BM;
    c7_f32x16 acc[C == 64 ? 4 : 2];                      // C = 32: one accumulator per k-step (no dependent MFMA pair)
#pragma unroll
    for (int t = 0; t < (C == 64 ? 4 : 2); ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (cnt_cur > 0 && !(C7_ABLATE & 2)) {   // (a block whose halo did not fit is left to the global-gather kernel launched behind this one)
      // (table addresses: the lane part is fixed per block, the tap part is a scalar -- one shift-add per read)
      uint32_t tabA = (uint32_t)(tabL - smem) + (C == 64 ? kh * 4 : 0), tabB = (uint32_t)(tabL - smem) + (1 - kh) * 4;
#ifdef __HIPCC__
      asm volatile("" : "+v"(tabA), "+v"(tabB));           // keep both as registers (not re-derived per use)
#endif
      auto entries = [&](int k, uint32_t (&te)[2]) {
        if constexpr (C == 64) {
          te[0] = *reinterpret_cast<const uint32_t*>(smem + (tabA + (uint32_t)(k * 256)));
          te[1] = *reinterpret_cast<const uint32_t*>(smem + (tabB + (uint32_t)(k * 256)));
        } else {
          te[0] = *reinterpret_cast<const uint16_t*>(smem + (tabA + (uint32_t)(k * 256)));
        }
      };
      auto gather1 = [&](const uint32_t (&te)[2], int t, int ks) -> frag {
        uint32_t off = C == 64 ? ((t & 1) ? (te[t >> 1] >> 16) : (te[t >> 1] & 0xffffu)) : te[0];   // piece 0 of the row (zero row: "none")
        if constexpr ((C7_ABLATE & 32) != 0)   // every lane its own consecutive slot: the conflict-free pattern the swizzle is built for
          off = (off & 0) + (uint32_t)((j + 32 * t) * ROWB + G::swz(j + 32 * t) * 16);
        return *reinterpret_cast<const frag*>(rowsL + (off ^ (ks ? pxor1 : pxor0)));
      };
      // EMPTY TAPS ARE SKIPPED: bit k of the block's (C = 64) / this wave's tile's (C = 32) tap mask (blocks.hip, table row 27) says
      // whether any row has a neighbour at tap k -- 67 % of the (block, tap) and 54 % of the (32-row tile, tap) pairs on curve-ordered
      // indoor scenes (tools/halo_stats.py).  The tap loop stays unrolled (the weight fragment of a tap is a register NAME), each tap
      // one basic block behind a scalar branch; the gather ring runs over the ACTIVE taps only (its table address is data, not a
      // name) and is a ROLLING single buffer: the B fragment an MFMA has just consumed is at once reloaded with the same (tile,
      // k-step) of the next active tap and is needed again 2 TW MFMAs later (C = 64: 256 cycles >> the LDS latency).  (Two named
      // buffers would need both parities of every step as code, and the compiler then copies the accumulators at every join.)
      const uint32_t mword = *reinterpret_cast<const uint32_t*>(rowsL + G::ROWS_BYTES + 27 * 256 + (C == 64 ? 16 : wave * 4));
      const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)mword);
      uint32_t rest = m;                                   // active taps whose table entries have not been read yet
      auto pop_tap = [&]() -> int {                        // (past the last active tap: tap 26 again -- a few wasted reads, no branch)
        int k;
#ifdef __HIPCC__
        // scalar unit, spelled out: left to itself the compiler runs this chain on the vector ALU, in the MFMAs' issue slots
        uint32_t nr;
        asm("s_ff1_i32_b32 %0, %2\n\ts_cmp_eq_u32 %2, 0\n\ts_cselect_b32 %0, 26, %0\n\ts_add_u32 %1, %2, -1\n\ts_and_b32 %1, %1, %2"
            : "=&s"(k), "=&s"(nr) : "s"(rest) : "scc");
        rest = nr;
#else
        k = rest ? __builtin_ctz(rest) : 26;
        rest &= rest - 1u;
#endif
        return k;
      };
      // The reload of a fragment trails its MFMA by LAG MFMAs (C = 64: 2): a gather whose destination is still an operand of an
      // MFMA in flight is held at issue until that MFMA retires, and behind it, in order, every later MFMA -- reloading "at once"
      // ran the loop at one MFMA per 60 cycles, its full latency (profiles/r03_o_conv7_ablate.txt).  So the first LAG fragments of
      // a step are still loaded from the CURRENT tap's entries (they are multiplied at the end of this step), the others from
      // the next tap's.
      // (C = 32, one tile = two MFMAs per tap: its loop runs 2460 cycles per block for 930 cycles of MFMAs.  A three-tap-deep ring rotated
      //  by register copies -- gathers two taps ahead -- was measured and is SLOWER, 2620: the copies write registers an MFMA in flight
      //  still reads and wait for it like the reloads do; profiles/r03_o_conv7_phases.txt section 9.)
      constexpr int NF = 2 * TW, LAG = C == 64 ? C7_LAG : 0;
      uint32_t teC[2], teN[2], teNN[2];
      frag b[NF];                                          // fragment i = (k-step i / TW, tile i % TW)
      entries(pop_tap(), teC);
#pragma unroll
      for (int i = 0; i < NF - LAG; ++i) b[i] = gather1(teC, i % TW, i / TW);
      entries(pop_tap(), teN);
      if constexpr (LAG == 0) { teC[0] = teN[0]; teC[1] = teN[1]; }
      tick(1);                                          // accumulator reset, mask, first gathers
      ptc_static_for<27>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (C7_DMA_IN_TAPS != 0 && k < G::NIW) {  // halo piece k of the next block, then the id of the block after next into its register
          if (dma_next) issue_dma_piece(ptc_int<k>{}, cnt_nxt, cur ^ 1);
          if (ids_nn) load_id_piece(ptc_int<k>{}, blk + 2 * step);
        }
        if (m & (1u << k)) {                               // wave-uniform
          __builtin_amdgcn_sched_barrier(0);               // the pinned interleave below starts here
          entries(pop_tap(), teNN);                        // the active tap after the next one
          // k-step outer, tile inner: two MFMAs on one accumulator are TW MFMAs apart
#pragma unroll
          for (int i = 0; i < NF; ++i) {
            c7_f32x16& a = acc[C == 64 ? i % TW : i / TW];
            a = MM::mma(wf[k][i / TW], b[i], a);
            const int g = (i + NF - LAG) % NF;             // the fragment consumed LAG MFMAs ago
            b[g] = gather1(i < LAG ? teC : teN, g % TW, g / TW);
          }
          teC[0] = teN[0];
          teN[0] = teNN[0];
          if constexpr (C == 64) { teC[1] = teN[1]; teN[1] = teNN[1]; }
          if constexpr (LAG == 0) { teC[0] = teN[0]; teC[1] = teN[1]; }
          // pin the interleave (one wave per SIMD: nothing else hides the LDS latency, and a 32-cycle MFMA hides ~5 other issues):
          // the table read rides in front, then per MFMA: the MFMA, the address computation and one fragment reload
          __builtin_amdgcn_sched_group_barrier(0x002, C == 64 ? 2 : 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, C == 64 ? 2 : 1, 0);
#pragma unroll
          for (int q = 0; q < NF; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                 // <= 2 VALU: field extract, xor-add
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                 // 1 LDS gather
          }
        }
      });
      tick(2);                                          // the tap loop
    }

    // ---- epilogue: lane (j, h) holds, of row j of a tile, output channels ch * 32 + 8 q + 4 h + 0..3, q = 0..3
    // 16-byte stores: lane (j, 0) takes channels 8 q .. 8 q + 7 of q = 0 and 2, lane (j, 1) those of q = 1 and 3 -- the halves of a
    // row's 8-channel groups trade places across the two 32-lane halves (4 swaps per tile), then 2 stores per tile, each covering
    // 32 contiguous bytes of every row (the first version: 4 stores of 8 scattered bytes per lane)
    auto store_tile = [&](int tile, const c7_f32x16& a) {
      const int64_t row = row0 + tile * 32 + j;
      uint32_t pk[4][2];                                   // [q][0 | 1]: channels 8 q + 4 h + (0, 1 | 2, 3) as a 16-bit pair
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          T o[2] = {ptc_from_float<T>(a[4 * q + 2 * e] + bsv[4 * q + 2 * e]), ptc_from_float<T>(a[4 * q + 2 * e + 1] + bsv[4 * q + 2 * e + 1])};
          __builtin_memcpy(&pk[q][e], o, 4);
        }
#pragma unroll
      for (int qq = 0; qq < 2; ++qq)
#pragma unroll
        for (int e = 0; e < 2; ++e) c7_swap32(pk[2 * qq][e], pk[2 * qq + 1][e]);
      // now (pk[2 qq][*], pk[2 qq + 1][*]) = channels 8 (2 qq + h) .. + 7 of row j
      if (row < n_out && cnt_cur > 0) {
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const uint4 v = make_uint4(pk[2 * qq][0], pk[2 * qq][1], pk[2 * qq + 1][0], pk[2 * qq + 1][1]);
          if constexpr ((C7_ABLATE & 16) != 0) asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));      // (the value stays live: no dead MFMAs)
          else *reinterpret_cast<uint4*>(out + row * C + ch * 32 + 8 * (2 * qq + h)) = v;
        }
      }
    };
    if constexpr ((C7_ABLATE & 4) != 0) {
#pragma unroll
      for (int t = 0; t < (C == 64 ? 4 : 2); ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" :: "v"(acc[t][r]));
      __builtin_amdgcn_s_waitcnt(C7_WAIT_VM0 & C7_WAIT_LGKM0);
      __builtin_amdgcn_s_barrier();
    } else if constexpr (C == 64) {
      // the two input-channel halves of a tile meet in LDS: wave kh finishes the block's tiles 2 kh, 2 kh + 1 (its local tiles 0, 1)
      // and hands the others (local 2, 3 = the partner's local 0, 1) to its partner
      unsigned char* scr = smem + 2 * G::BUF;
      const int partner = wave ^ 1;
#pragma unroll
      for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = {acc[2 + tl][4 * q], acc[2 + tl][4 * q + 1], acc[2 + tl][4 * q + 2], acc[2 + tl][4 * q + 3]};
          if constexpr ((C7_ABLATE & 8) != 0) asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
          else *reinterpret_cast<f32x4*>(scr + ((partner * 2 + tl) * 4 + q) * 1024 + lane * 16) = v;
        }
      tick(3);                                          // accumulators -> scratch
      __builtin_amdgcn_s_waitcnt(C7_WAIT_VM0 & C7_WAIT_LGKM0);   // scratch written; next block's rows + table and the ids landed
      tick(4);                                          // wait: DMA of the next block, scratch writes
      __builtin_amdgcn_s_barrier();
      tick(5);                                          // barrier
#pragma unroll
      for (int tl = 0; tl < 2; ++tl) {
        c7_f32x16 a = acc[tl];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 o = {0.f, 0.f, 0.f, 0.f};
          if (!(C7_ABLATE & 8)) o = *reinterpret_cast<const f32x4*>(scr + ((wave * 2 + tl) * 4 + q) * 1024 + lane * 16);
#pragma unroll
          for (int e = 0; e < 4; ++e) a[4 * q + e] += o[e];
        }
        store_tile(2 * kh + tl, a);
      }
      // the partner must not reach its next scratch write before this wave has read: a second barrier (the epilogues are symmetric)
      tick(6);                                          // partner's half added, rows stored
      __builtin_amdgcn_s_waitcnt(C7_WAIT_LGKM0);
      __builtin_amdgcn_s_barrier();
      tick(7);                                          // second barrier
    } else {
      // wait for the DMA of the next block BEFORE the stores of this one are issued: they retire during the next block's MFMAs
      tick(3);
      __builtin_amdgcn_s_waitcnt(C7_WAIT_VM0 & C7_WAIT_LGKM0);
      tick(4);                                          // wait: DMA of the next block
      __builtin_amdgcn_s_barrier();
      tick(5);                                          // barrier
      c7_f32x16 a = acc[0];
#pragma unroll
      for (int r = 0; r < 16; ++r) a[r] += acc[1][r];
      store_tile(wave, a);
      tick(6);                                          // add + store
    }
    if (!(C7_ABLATE & 1)) cur ^= 1;
    cnt_cur = cnt_nxt;
    cnt_nxt = __builtin_amdgcn_readfirstlane(cnt_nn);
  }
  if constexpr ((C7_ABLATE & 64) != 0) {
    __builtin_amdgcn_s_waitcnt(C7_WAIT_VM0 & C7_WAIT_LGKM0);
    __builtin_amdgcn_s_barrier();
    if (threadIdx.x == 0) {
      long long* o = reinterpret_cast<long long*>(out + (int64_t)blockIdx.x * C);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = tph[i];
    }
  }
}

template <typename T, int C>
static int launch_conv7_i(const void* in, int64_t n_in, const void* w, const float* bias, const uint16_t* tab,
                          const int32_t* hid, const int32_t* hcnt, int64_t n_out, void* out, hipStream_t s) {
  const int n_blocks = (int)ptc_cdiv(n_out, C7_BM);
  int max_wgs = 256;                          // one persistent workgroup per CU
#ifndef __HIPCC__
  if (const char* e = getenv("PTC_EMU_CONV7_WGS")) max_wgs = atoi(e);   // host emulation only: few workgroups = many blocks each at test sizes
#endif
  int grid = n_blocks < max_wgs ? n_blocks : max_wgs;
  auto kern = conv7_kernel<T, C>;
  // every launch, like the other large-LDS kernels: the attribute is per DEVICE (a process-wide flag left the second GPU of a process
  // without it) and the call is a host-side table write
  PTC_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C7Geom<C>::LDS));
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), C7Geom<C>::LDS, s, (const T*)in, (const T*)w, bias, tab, hid, hcnt, n_out,
                     n_blocks, (T*)out);
  PTC_CHECK_LAUNCH("conv7_kernel");
  return PTC_OK;
}
#endif  // PTC_CONV7_IMPL
