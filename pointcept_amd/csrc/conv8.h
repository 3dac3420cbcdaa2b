// conv8.h -- block-staged 3^3 submanifold convolution for WIDE rows: c_in a multiple of 32 from 96 up (SpUNet's 96 / 128 / 256-channel levels,
// PT-v3's 128 / 256 / 512-channel stages), forward and input gradient (mirrored weights).  Round 6; included by spconv.hip.
//
// conv3 (the global-gather kernel these shapes ran on) fetches every input row once per table entry that names it AND once per 64-column
// block of the output -- 9.3 x (c_out / 64) times -- in the MFMA operand layout (lane -> row l & 15, piece l >> 4: 57 address cycles per
// 1-KB wave gather, conv5.h).  At 256 -> 256, N = 12115 that is 300 MB of quarter-line requests for 6 MB of rows: 100 us for 19 GFLOP
// (190 TF/s, TA-bound; profiles/r06_a ops table), and 2.4 ms of the PT-v3 step's 30 deep convolutions.  Here, as in conv7 (32 / 64
// channels), a workgroup owns one 128-row block of blocks.hip -- its distinct input rows (the "halo", ~1.7 x 128) and the uint16 table of
// halo slots -- and
//   * stages the halo ONCE per 64-channel chunk of the input by LDS-DMA (global_load_lds, 8 lanes per 128-byte piece of a row: whole
//     lines, no staging registers) into an image [slot][64 channels] of 128-byte rows whose 16-byte pieces are XOR-swizzled by the slot
//     (conv7's layout: the table entries of blocks.hip are byte offsets into it); slot HIMG is an all-zero row: "no neighbour" needs no
//     branch;
//   * WEIGHTS NEVER TOUCH LDS (third form): a ~3 us pre-pass (conv8_wfrag_kernel) rewrites W into MFMA FRAGMENT ORDER -- [32-column group]
//     [tap][32-channel step][tile][lane][8] -- so that a wave fetches the A operands of (tap, step) as 1-KB lane-linear loads straight
//     into registers, one tap ahead of its products.  The waves of a workgroup then share nothing but the halo image: NO workgroup
//     barrier inside the tap loop, every wave skips exactly the taps ITS rows have no neighbour at, and the workgroup's LDS is table +
//     image = 48 KB: THREE workgroups per CU, whose table / halo-list / row latencies and stores run under each other's products;
//   * a wave owns 32 output columns (two MFMA tiles: its lanes end with 8 consecutive channels, one 16-byte store per row) x all 128 rows
//     (c_out a multiple of 128 / 96: four / three waves work), 64 rows (64-column workgroups) or 32 rows (32-column ones); inside a tap
//     it multiplies only the 16-row tiles that have a neighbour (a ballot over the slots its lanes hold: 60 % of the (16-row tile, tap)
//     pairs of a curve-ordered indoor scene are empty);
//   * blocks whose halo does not fit the image (more than HIMG distinct rows, or blocks.hip's own overflow mark) return at once and are
//     served by a follow-up launch of conv3 whose workgroups skip the row blocks done here (its `skip_hcnt` argument).  (Until r06_v this
//     was a second instance of this kernel with the B fragments gathered from global memory: a serial latency chain per block -- 150 us
//     for the 1 % of the blocks of a curve-ordered level, and 6.8 ms of a SpUNet step whose coarse levels were numbered lexicographically
//     and overflowed everywhere, profiles/r06_v_conv8_spunet_prof.txt.)
// History (profiles/r06_h .. r06_n): (1) one 8-wave workgroup per CU, 128-channel image, W through a two-deep LDS ring with a barrier
// per tap: 820 us at 128 -> 96, N = 819200 against conv3's 585 -- every block paid its latency chain alone on the CU; (2) 64-channel
// image, two 4-wave workgroups per CU: 790 us, 682 with every prefetch a raw buffer load (as plain loads behind a select the compiler
// serialised them: s_waitcnt vmcnt(0) after each); phase timers: 55 % products at ~20 % matrix-pipe occupancy, 23 % waiting at the
// per-tap barrier for the row half with more neighbours, 12 % waiting for W.
// Measured and dropped (profiles/r06_ab_*, r06_ac_*): three workgroups per CU (168 registers + 72 bytes of scratch: 1-5 %); 64- / 32-column
// workgroups at c_out = 128 / 96 (3-20 % slower: more weight traffic per row); weights three taps ahead (no change); the B fragments of tap
// k + 1 read tile by tile behind the products of the same tile of tap k (a software pipeline over the taps: 2-7 % SLOWER -- the entries of
// the next tap then sit behind those reads in the LDS queue and the wait for them drains it).
// Output-stationary, fixed summation order (chunk-major, taps ascending, 32-channel steps ascending): bit-reproducible.  The order differs
// from conv3's (tap-major): results agree to fp32 summation order, not bit for bit.
#pragma once
#include <mutex>

#ifndef C8_HIMG
#define C8_HIMG 352          // halo rows the LDS image of the 64- / 32-row forms holds (three workgroups per CU; the 128-row form: C8_HIMG_WIDE, below).
                             // At 288 the ~1 % of blocks beyond it cost 150 us of a 650-us launch (profiles/r06_t_conv8_himg.txt); 352 = 52.5 KB of LDS
#endif
#define C8_KC 64             // input channels per chunk
#define C8_PITCH (C8_KC * 2) // image rows are 128 bytes, their 16-byte pieces XOR-swizzled by the slot (PTC_SWZ64: blocks.hip's table entries carry it)
#define C8_THREADS 256
#ifndef C8_HIMG_WIDE
#define C8_HIMG_WIDE 416     // the four-tiles-per-wave form (RT = 4) is held at two workgroups per CU by its registers: its image can be 62 KB.  At 352 ONE
#endif                       // block of SpUNet's level 1 (354 rows, 8 x 100000 voxels) sent nine launches per step into a 52-us follow-up (r06_ah / r06_ai)
__host__ __device__ constexpr int c8_himg(int rt) { return rt == 4 ? C8_HIMG_WIDE : C8_HIMG; }
__host__ __device__ constexpr int c8_passes(int rt) { return (c8_himg(rt) + 31) / 32; }   // DMA instructions per wave and chunk: 8 rows x 128 bytes each, 32 rows per workgroup pass
#define C8_TAB_BYTES (28 * 128 * 2)           // 27 table rows + the row of masks: seven whole 1-KB DMA pieces
#ifndef C8_READ_ALL
#define C8_READ_ALL 0        // 1: the B fragments of four row tiles in flight in front of their MFMAs, the zero row for tiles without a neighbour (timing A/B)
#endif
#ifndef C8_WDEPTH
#define C8_WDEPTH 2          // weight register sets: the fragments of tap k + C8_WDEPTH - 1 are requested before tap k's products (2 | 3)
#endif
#ifndef C8_WGS_PER_CU
#define C8_WGS_PER_CU 2      // (the B fragments of all four tiles of a tap in flight: 64 registers)
#endif

// LDS-DMA, conv7.h's helpers (that header is its own translation unit): 16 bytes per lane global -> LDS at the wave-uniform byte address
// `lds_dst` + 16 * lane, SGPR base + 32-bit VGPR offset.  Inline assembly on purpose: the compiler's wait-count pass never sees the DMA, its
// completion is counted by hand (s_waitcnt vmcnt(0), then a workgroup barrier, then the reads).
#define C8_WAIT_VM0 0x0F70
#define C8_WAIT_LGKM0 0xC07F
#ifdef __HIPCC__
__device__ __forceinline__ void c8_dma16s(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ uint32_t c8_lds_addr(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p; }
#else
__device__ __forceinline__ void c8_dma16s(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  emu_global_load_lds(reinterpret_cast<const unsigned char*>(sbase) + voff, smem + lds_dst, 16);
}
__device__ __forceinline__ uint32_t c8_lds_addr(const void* p) { return (uint32_t)((const unsigned char*)p - smem); }
#endif

static inline size_t conv8_lds(int rt) { return (size_t)C8_TAB_BYTES + (size_t)(c8_himg(rt) + 1) * C8_PITCH + (size_t)c8_passes(rt) * 32 * 4; }   // table | image | halo list
static inline bool conv8_supported(int dtype, int kv, int c_in, int c_out, int bm, int hcap, int64_t n_out) {
  return dtype != PTC_F32 && kv == 27 && c_in % 32 == 0 && c_in >= 96 && c_in <= 1024 && c_out % 32 == 0 && c_out <= 1024 && bm == 128 && hcap >= 16 &&
         hcap <= 511 && n_out >= 256;
}
// columns per workgroup: 128 | 96 (every wave all 128 rows of the block), 64 (two waves per 64-row half), 32 (a wave per 32 rows) -- the
// narrower forms where the wide one would leave CUs without a workgroup (the deep stages: 12115 rows x 256 channels = 190 wide workgroups)
static inline int conv8_nt(int c_out, int64_t n_blocks) {
  static const long forced = getenv("PTC_C8_NT") ? atol(getenv("PTC_C8_NT")) : 0;           // sweeps
  if (forced > 0 && c_out % forced == 0) return (int)forced;
  static const long min_wgs = getenv("PTC_C8_MIN_WGS") ? atol(getenv("PTC_C8_MIN_WGS")) : 512;
  const int wide = c_out % 128 == 0 ? 128 : (c_out % 96 == 0 ? 96 : (c_out % 64 == 0 ? 64 : 32));
  if (wide <= 32 || n_blocks * (c_out / wide) >= min_wgs) return wide;
  if (c_out % 64 == 0 && (wide == 64 || n_blocks * (c_out / 64) >= min_wgs)) return 64;
  if (c_out % 64 == 0 && wide != 64) return n_blocks * (c_out / 64) * 2 >= min_wgs ? 64 : 32;
  return wide;
}

static int c8_ablate() { const char* e = getenv("PTC_C8_ABLATE"); return e ? atoi(e) : 0; }     // timing probes only (wrong results): 1 no products,
                                                                                                  // 8 no halo staging, 16 the first launch alone, 64 phase timers

// ---- W [c_out][27][c_in] -> fragment order [c_out / 32][27][c_in / 16][64 lanes][8] (32x32x16 MFMA A operands): lane (i, h) of the
// fragment of (column group cg, tap k, 16-channel step ks) holds W[32 cg + 16 ((i >> 2) & 1) + 4 (i >> 3) + (i & 3)][k][16 ks + 8 h ..]:
// after the MFMAs (D[i][j]: lane j + 32 h, element e: i = 8 (e >> 2) + 4 h + (e & 3)) lane (row j, h) holds channels 32 cg + 16 h + e
template <typename T>
__global__ void __launch_bounds__(256)
conv8_wfrag_kernel(const T* __restrict__ w, int c_out, int c_in, T* __restrict__ wf) {
  const int KS = c_in >> 4;
  const int64_t total = (int64_t)(c_out >> 5) * 27 * KS * 64;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
    const int lane = (int)(q & 63);
    int64_t rest = q >> 6;
    const int ks = (int)(rest % KS); rest /= KS;
    const int k = (int)(rest % 27);
    const int cg = (int)(rest / 27);
    const int i = lane & 31, h = lane >> 5;
    const int ch = 32 * cg + 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3);
    reinterpret_cast<uint4*>(wf)[q] = *reinterpret_cast<const uint4*>(w + ((int64_t)ch * 27 + k) * c_in + 16 * ks + 8 * h);
  }
}

// per-process scratch for the fragment-ordered weights, one buffer per stream that ever called (grown on demand; hipFree waits for the
// device, so a kernel still reading the old buffer is never cut off).  The library allocates nothing else itself.
static void* c8_wf_scratch(size_t bytes, hipStream_t s) {
  struct Slot { hipStream_t s; void* p; size_t n; bool used; };
  static Slot slots[8] = {};
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  Slot* sl = nullptr;
  for (auto& x : slots) if (x.used && x.s == s) { sl = &x; break; }
  if (!sl) for (auto& x : slots) if (!x.used) { sl = &x; sl->used = true; sl->s = s; sl->p = nullptr; sl->n = 0; break; }
  if (!sl) { sl = &slots[0]; if (sl->p) (void)hipFree(sl->p); sl->p = nullptr; sl->n = 0; sl->s = s; }
  if (sl->n < bytes) {
    if (sl->p) (void)hipFree(sl->p);
    sl->p = nullptr; sl->n = 0;
    const size_t want = bytes < ((size_t)4 << 20) ? ((size_t)4 << 20) : ptc_align_up(bytes, (size_t)1 << 20);
    if (hipMalloc(&sl->p, want) != hipSuccess) { sl->p = nullptr; return nullptr; }
    sl->n = want;
  }
  return sl->p;
}

typedef __attribute__((ext_vector_type(16))) float c8_f32x16;
template <typename T> __device__ __forceinline__ c8_f32x16 c8_mma32(typename Mma<T>::frag a, typename Mma<T>::frag b, c8_f32x16 c);
template <> __device__ __forceinline__ c8_f32x16 c8_mma32<bf16_t>(s16x8 a, s16x8 b, c8_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
template <> __device__ __forceinline__ c8_f32x16 c8_mma32<f16_t>(h16x8 a, h16x8 b, c8_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
// one workgroup per (block, column block); blocks whose halo does not fit return at once (conv3 follows up)
template <typename T, int RT>
__global__ void __launch_bounds__(C8_THREADS, C8_WGS_PER_CU)
conv8_kernel(const T* __restrict__ in, const T* __restrict__ wf, const float* __restrict__ bias,
             const uint16_t* __restrict__ tab, const int32_t* __restrict__ hid, const int32_t* __restrict__ hcnt, int64_t n, int c_in, int c_out,
             int hcap, int n_blocks, int cgw, T* __restrict__ out, int abl) {
  using frag = typename Mma<T>::frag;
  constexpr int HIMG = c8_himg(RT), PASSES = c8_passes(RT);
  const int NT = 32 * cgw;                              // columns per workgroup
  const __amdgpu_buffer_rsrc_t wf_buf = ptc_buf(wf, (uint32_t)c_out * 27u * (uint32_t)c_in * 2u);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* lt = reinterpret_cast<uint16_t*>(smem);                                     // [27][32][4] halo entries (blocks.hip's 128-byte-row variant)
  unsigned char* img = smem + C8_TAB_BYTES;                                             // [HIMG + 1][C8_PITCH]
  // XCD-first numbering: the column blocks of one row block are consecutive logical ids and run on one XCD (its halo rows stay in that L2)
  const int ny = c_out / NT, nblk = n_blocks * ny;
  const int per_xcd = (nblk + 7) >> 3;
  const int lane = ptc_lane(), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // (uniform: DMA destinations are scalar operands)
  const int j = lane & 31, h = lane >> 5;               // MFMA 32x32x16: B column (row of the tile) / A row, k-group
  // this wave: column group cg of the workgroup, 32-row tiles t0 .. t0 + RT - 1 of the block
  const int cg = RT == 4 ? wave : (RT == 2 ? (wave & 1) : 0);
  const int t0 = RT == 4 ? 0 : (RT == 2 ? 2 * (wave >> 1) : wave);
  const bool works = cg < cgw;                          // (96-column workgroups: the fourth wave only helps to stage)
  const int KS = c_in >> 4;
  const int lb = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (lb >= nblk) return;
  const int b = lb / ny, n0 = (lb - b * ny) * NT;
  const int64_t row0 = (int64_t)b * 128 + 32 * t0;
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
  if (!(cnt >= 0 && cnt <= HIMG)) return;            // conv3's block (wave-uniform: before any barrier)

  // ---- the block's table (LDS-DMA), tap masks of this wave's 32-row tiles, halo list.  Every load of this kernel that feeds a prefetch is
  // a RAW BUFFER LOAD (out-of-range offset = zeros): written as `v = *p; if (!ok) v = 0` the compiler turned each one into a branch around
  // the load with s_waitcnt vmcnt(0) behind it
  uint32_t tmk[RT], tm = 0u;                            // taps at which tile t / any tile of this wave has a neighbour
  const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c8_lds_addr(smem));
  int32_t* hl = reinterpret_cast<int32_t*>(img + (size_t)(HIMG + 1) * C8_PITCH);   // [32 PASSES] halo rows by slot (0 beyond the list: never named by the table)
  {
    const uint16_t* tb = tab + (int64_t)b * (28 * 128);
    const __amdgpu_buffer_rsrc_t hid_buf = ptc_buf(hid + (int64_t)b * hcap, (uint32_t)hcap * 4u);
    int32_t h0 = ptc_buf_load4(hid_buf, threadIdx.x * 4u), h1 = ptc_buf_load4(hid_buf, (threadIdx.x + C8_THREADS) * 4u);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (4 * i + wave < C8_TAB_BYTES / 1024) c8_dma16s(tb, (uint32_t)((4 * i + wave) * 1024 + lane * 16), lds0 + (uint32_t)((4 * i + wave) * 1024));
    const uint32_t* mw = reinterpret_cast<const uint32_t*>(tb + 27 * 128);
#pragma unroll
    for (int t = 0; t < RT; ++t) { tmk[t] = mw[t0 + t]; tm |= tmk[t]; }
    for (int q = threadIdx.x; q < C8_PITCH / 4; q += C8_THREADS) reinterpret_cast<uint32_t*>(img + (size_t)HIMG * C8_PITCH)[q] = 0u;
    hl[threadIdx.x] = (int)threadIdx.x < cnt ? h0 : 0;
    if (threadIdx.x + C8_THREADS < PASSES * 32) hl[threadIdx.x + C8_THREADS] = (int)threadIdx.x + C8_THREADS < cnt ? h1 : 0;
    __syncthreads();                                    // the halo list is in LDS (the table's DMA is waited for with the first chunk's rows)
  }
  if (!works || (abl & 1)) tm = 0u;
  tick(0);                                              // halo count, table DMA issued, halo list in LDS

  // accumulators: lane (row j of tile t, h) holds channels nw0 + 16 h + e, e = 0..15; they start at the bias
  c8_f32x16 acc[RT];
  const int nw0 = n0 + 32 * cg;                         // first column of this wave
  {
    c8_f32x16 bv;
#pragma unroll
    for (int e = 0; e < 16; ++e) bv[e] = 0.f;
    if (bias && works) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(bias + nw0 + 16 * h + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[4 * q + e] = v[e];
      }
    }
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = bv;
  }

  // ---- weights: the fragments of (tap, 16-channel step) of this wave's column group, lane-linear 1-KB loads, two register sets
  // One load = the wave's lane-constant vector offset (16 bytes per lane; out of range for a wave without a column group: zeros, no
  // traffic) + a SCALAR offset that names (column group, tap, step).  No vector instruction computes an address inside the tap loop:
  // with a per-tap vector offset the compiler built it in the registers of the weight set it was about to fetch and put s_waitcnt
  // vmcnt(0) in front -- the fetch for tap k + 1 went out only after tap k's weights had landed (prefetch distance: the products of one
  // tap, ~200 cycles against ~700 of L2 latency).  Past the last tap the fetch re-reads tap 26 (in range, never used).
  const uint32_t wlane = works ? (uint32_t)lane * 16u : PTC_BUF_OOB;
  const uint32_t wgrp = (uint32_t)((n0 >> 5) + (works ? cg : 0)) * 27u * (uint32_t)KS;       // (wave-uniform: scalar registers)
  frag wq[C8_WDEPTH][4];
  auto wfetch = [&](auto bi, auto nk, int k, int ks0) {
    constexpr int B = decltype(bi)::value, NK = decltype(nk)::value;
    const uint32_t so = (wgrp + (uint32_t)((k < 27 ? k : 26) * KS + ks0)) * 1024u;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      const ptc_i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wf_buf, (int)wlane, (int)(so + (uint32_t)ks * 1024u), 0);
      __builtin_memcpy(&wq[B][ks], &v, sizeof(frag));
    }
  };
  auto next_tap = [&](int k) {            // first tap >= k at which this wave has work (27: none).  (Written as a loop it compiled to a
    const uint32_t rest = k < 27 ? tm >> k : 0u;        //  ladder of 27 compare-and-branch pairs per call.)
    return rest ? k + __builtin_ctz(rest) : 27;
  };
  // halo rows of a chunk -> the image: DMA instruction q of this wave carries slots 32 q + 8 wave .. + 7, lane (row l >> 3, position l & 7)
  // fetches the piece that belongs at its position (source-side swizzle); a 32-channel tail chunk has four pieces per row: positions whose
  // piece does not exist fetch an existing one (never read)
  auto hdma = [&](int c0, int kc) {
    const int np = kc >> 3;                             // pieces per row: 8 | 4
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int slot0 = p * 32 + 8 * wave;
      if (slot0 < cnt) {                                // wave-uniform
        const int slot = slot0 + (lane >> 3);
        const int piece = ((lane & 7) ^ PTC_SWZ64(slot)) & (np - 1);
        c8_dma16s(in, ((uint32_t)hl[slot] * (uint32_t)c_in + (uint32_t)(c0 + piece * 8)) * 2u, lds0 + (uint32_t)(C8_TAB_BYTES + slot0 * C8_PITCH));
      }
    }
  };

  // one tap of this wave: the table entries of its rows (byte offsets of piece 0 of the neighbour's image row; "no neighbour": the zero row)
  // were requested a tap ago; per 32-row tile with a neighbour anywhere (blocks.hip's tile masks: 46 % of the (tile, tap) pairs of a
  // curve-ordered indoor scene are empty) the B fragments of the chunk's 16-channel steps, ALL tiles' reads in flight before the first
  // MFMA.  (Counters of the form that read and multiplied tile by tile, profiles/r06_t_conv8_pmc.txt: 53 % of the wave cycles in s_waitcnt,
  // 31 % issuing, matrix pipe 28 % busy, 47 % of the LDS cycles bank conflicts of the gathers.)
  using entries = typename std::conditional<RT == 4, uint2, uint32_t>::type;
  auto load_entries = [&](int k) -> entries {
    const int kc_ = k < 27 ? k : 26;
    if constexpr (RT == 4) return *reinterpret_cast<const uint2*>(lt + (kc_ * 32 + j) * 4);          // [tap][row in tile][tile]
    else if constexpr (RT == 2) return *reinterpret_cast<const uint32_t*>(lt + (kc_ * 32 + j) * 4 + t0);
    else return (uint32_t)lt[(kc_ * 32 + j) * 4 + t0];
  };
  entries en_cur;
  auto tap = [&](auto bi, auto nk, int k, int kn) {
    constexpr int B = decltype(bi)::value, NK = decltype(nk)::value;
    uint32_t en[RT];
    if constexpr (RT == 4) { en[0] = en_cur.x & 0xffffu; en[1] = en_cur.x >> 16; en[2] = en_cur.y & 0xffffu; en[3] = en_cur.y >> 16; }
    else if constexpr (RT == 2) { en[0] = en_cur & 0xffffu; en[1] = en_cur >> 16; }
    else en[0] = en_cur;
    en_cur = load_entries(kn);                          // the next tap's entries: in flight under this tap's products
    uint32_t po[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
      po[t] = (uint32_t)C8_TAB_BYTES + ((en[t] < (uint32_t)hcap * 128u ? en[t] : (uint32_t)(HIMG * C8_PITCH)) ^ (uint32_t)(h << 4));
    // piece 2 ks + h of the row sits at position (2 ks + h) ^ swizzle; a full chunk has four 16-channel steps, the 32-channel tail two
    frag fb[RT][NK];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      if ((tmk[t] >> k) & 1u) {                         // wave-uniform
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) fb[t][ks] = *reinterpret_cast<const frag*>(smem + (po[t] ^ (uint32_t)(ks << 5)));
      }
    }
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      if ((tmk[t] >> k) & 1u) {
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) acc[t] = c8_mma32<T>(wq[B][ks], fb[t][ks], acc[t]);
      }
    }
  };

  // One chunk of the input channels: NK = 4 (64 channels) | 2 (the 32-channel tail of c_in = 96, 160, ...).  The two forms are SEPARATE
  // loops, not a run-time choice inside one tap body: with `if (nks == 4) ... else ...` per tap the two bodies kept the accumulators in
  // different registers and every (tile, tap) pair ended with 8 v_mov_b64 that read the MFMA's result the cycle after it issued -- the
  // wave stalled for the full latency of its four products before the next tile's could issue (9.5 vector instructions per MFMA,
  // matrix pipe 28 % busy: the r06_s counters).
  const int k_first = next_tap(0);
  auto chunk = [&](auto nk, int ch) {
    constexpr int NK = decltype(nk)::value;
    const int c0 = ch * C8_KC, ks0 = c0 >> 4;
    int k = k_first;
    if (ch > 0) __syncthreads();                        // every wave is done with the image of the previous chunk
    tick(1);                                            // chunk entry barrier
    if (!(abl & 8)) hdma(c0, 16 * NK);
    wfetch(ptc_int<0>{}, nk, k, ks0);                   // the first tap's weights travel with the halo rows (k = 27: out-of-range offsets, no traffic)
    __builtin_amdgcn_s_waitcnt(C8_WAIT_VM0 & C8_WAIT_LGKM0);   // this wave's DMAs (rows; first chunk: table) landed -- the compiler does not count them
    __syncthreads();
    tick(2);                                            // halo image complete
    en_cur = load_entries(k);
    // the tap loop, C8_WDEPTH taps per trip (the weight register sets rotate): no workgroup synchronisation
    // (the fetches are UNCONDITIONAL -- past the last tap their offsets are out of range: under a branch the compiler's wait-count
    //  bookkeeping gave up at the join and every MFMA waited for vmcnt(0), i.e. for the NEXT tap's weights as well)
#if C8_WDEPTH == 2
    while (k < 27) {
      const int k1 = next_tap(k + 1);
      wfetch(ptc_int<1>{}, nk, k1, ks0);
      tap(ptc_int<0>{}, nk, k, k1);
      if (k1 >= 27) break;
      const int k2 = next_tap(k1 + 1);
      wfetch(ptc_int<0>{}, nk, k2, ks0);
      tap(ptc_int<1>{}, nk, k1, k2);
      k = k2;
    }
#else
    {
      int ka = k, kb = next_tap(ka + 1);
      wfetch(ptc_int<1>{}, nk, kb, ks0);
      while (ka < 27) {
        const int kc2 = next_tap(kb + 1);
        wfetch(ptc_int<2>{}, nk, kc2, ks0);
        tap(ptc_int<0>{}, nk, ka, kb);
        if (kb >= 27) break;
        const int kd = next_tap(kc2 + 1);
        wfetch(ptc_int<0>{}, nk, kd, ks0);
        tap(ptc_int<1>{}, nk, kb, kc2);
        if (kc2 >= 27) break;
        const int ke = next_tap(kd + 1);
        wfetch(ptc_int<1>{}, nk, ke, ks0);
        tap(ptc_int<2>{}, nk, kc2, kd);
        ka = kd; kb = ke;
      }
    }
#endif
    tick(3);                                            // this wave's taps of the chunk
  };
  for (int ch = 0; ch < (c_in >> 6); ++ch) chunk(ptc_int<4>{}, ch);
  if (c_in & 32) chunk(ptc_int<2>{}, c_in >> 6);
  if (works) {
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const int64_t row = row0 + 32 * t + j;
      if (row < n) {
        uint32_t pk[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) pk[e] = sc_pack2<T>(acc[t][2 * e], acc[t][2 * e + 1]);
        uint4* dst = reinterpret_cast<uint4*>(out + row * c_out + nw0 + 16 * h);
        dst[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        dst[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
      }
    }
  }
  tick(4);                                              // epilogue issued
  if (timing) __builtin_amdgcn_s_waitcnt(C8_WAIT_VM0);     // (the epilogue's stores of this row first: different types, the compiler may reorder)
  if (timing && threadIdx.x == 0 && n0 == 0) {
    uint4* o = reinterpret_cast<uint4*>(out + (int64_t)b * 128 * c_out);
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = make_uint4((uint32_t)tph[2 * i], (uint32_t)(tph[2 * i] >> 32), (uint32_t)tph[2 * i + 1], (uint32_t)(tph[2 * i + 1] >> 32));
  }
}

template <typename T, int RT>
static int launch_conv8_i(const void* in, int64_t n_in, const void* wf, const float* bias, const uint16_t* tab, const int32_t* hid, const int32_t* hcnt,
                          int hcap, int64_t n, int c_in, int c_out, int nt, void* out, hipStream_t s) {
  const int n_blocks = (int)ptc_cdiv(n, 128);
  const int nblk = n_blocks * (c_out / nt);
  const size_t lds = conv8_lds(RT);
  auto kern = conv8_kernel<T, RT>;
  static bool attr = false;                   // per instantiation
  if (!attr) { PTC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
  hipLaunchKernelGGL(kern, dim3((unsigned)(8 * ((nblk + 7) / 8))), dim3(C8_THREADS), lds, s, (const T*)in, (const T*)wf, bias, tab, hid, hcnt, n, c_in, c_out,
                     hcap, n_blocks, nt / 32, (T*)out, c8_ablate());
  PTC_CHECK_LAUNCH("conv8_kernel");
  return PTC_OK;
}

template <typename T>
static int launch_conv8(const void* in, int64_t n_in, const void* w, const float* bias, const int32_t* nbr, const uint16_t* tab, const int32_t* hid,
                        const int32_t* hcnt, int hcap, int64_t n, int c_in, int c_out, void* out, hipStream_t s) {
  const size_t wbytes = (size_t)c_out * 27 * c_in * sizeof(T);
  void* wf = c8_wf_scratch(wbytes, s);
  PTC_REQUIRE(wf != nullptr, PTC_EWORKSPACE, "ptc_spconv_fwd_blk: no device memory for %zu bytes of fragment-ordered weights", wbytes);
  const int64_t pieces = (int64_t)wbytes / 16;
  hipLaunchKernelGGL((conv8_wfrag_kernel<T>), dim3((unsigned)(pieces / 256 < 2048 ? ptc_cdiv(pieces, 256) : 2048)), dim3(256), 0, s, (const T*)w, c_out, c_in, (T*)wf);
  PTC_CHECK_LAUNCH("conv8_wfrag_kernel");
  const int nt = conv8_nt(c_out, ptc_cdiv(n, 128));
  int rc;
  if (nt >= 96) rc = launch_conv8_i<T, 4>(in, n_in, wf, bias, tab, hid, hcnt, hcap, n, c_in, c_out, nt, out, s);
  else if (nt == 64) rc = launch_conv8_i<T, 2>(in, n_in, wf, bias, tab, hid, hcnt, hcap, n, c_in, c_out, nt, out, s);
  else rc = launch_conv8_i<T, 1>(in, n_in, wf, bias, tab, hid, hcnt, hcap, n, c_in, c_out, nt, out, s);
  if (rc != PTC_OK || (c8_ablate() & 16)) return rc;      // (bit 16, timing probe: the first launch alone)
  // the blocks whose halo did not fit: conv3 over exactly those (their counts are on the device)
  c3_skip_hcnt = hcnt;
  c3_skip_max = c8_himg(nt >= 96 ? 4 : 1);
  rc = launch_conv3<T>(in, n_in, w, bias, nbr, n, 27, c_in, c_out, out, s);
  c3_skip_hcnt = nullptr;
  return rc;
}
