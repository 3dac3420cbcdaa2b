// gemm3.h -- the row-wise GEMM of the DEEP stages: out[m][n] = sum_{k < kv} sum_c in[nbr[k][m]][c] * w[n][k][c] + bias[n] for 16-bit features,
// c_in a multiple of 64 and >= 128, c_out a multiple of 128 (nn.Linear of the Blocks at 128 / 256 / 512 channels, the gather-fused qkv / proj
// GEMMs and their input gradients -- kv = 2 for the qkv gradient, whose rows appear up to twice in the padded sequence -- and the two MLP
// GEMMs with their GELU epilogues; ptv3m1:173-248).  Included by spconv.hip behind fwd2.h (shares its epilogues and weight-row permutation).
//
// Why (round 6, profiles/r06_ap_step_sequence.txt): linear2_kernel is built for N = 819200 rows of 32 / 64 channels -- W of a column block
// stationary in LDS, the input row fragments of a 128-row tile in registers, HBM-bound.  At 20 000 rows x 256 channels the same kernel holds
// 4 x 8 row fragments + 64 accumulators (one wave per SIMD), stages 64 KB of W for fewer than two row tiles per workgroup and ran the
// qkv GEMM of a stage-3 Block (7.9 GF, 41 MB) in 38 us: 0.2 PF/s, 8 % of the matrix peak and 13 % of HBM -- bound by neither.  The Blocks of
// stages 2-4 (18 of the 30) spend ~290 us per Block and direction in such GEMMs.  This kernel is the textbook form for that regime:
//   * 128 x 128 output tile per workgroup of 4 waves (2 x 2, 64 x 64 per wave: 4 x 4 tiles of v_mfma_f32_16x16x32, 64 accumulators);
//   * the contraction in chunks of 64 channels: A (128 gathered rows) and W (128 channels) chunks of 16 KB each go global -> registers ->
//     LDS, double buffered: the loads of chunk i + 1 are in flight while chunk i is multiplied, ONE barrier per chunk;
//   * LDS rows of 128 bytes, 16-byte pieces XOR-swizzled by (row >> 1) & 7: the ds_read_b128 fragment reads (16 rows x 4 pieces per
//     instruction, lane groups of MI355X_MICROARCH.md) and the ds_write_b128 staging writes are conflict-free without padding: 64 KB per
//     workgroup, two workgroups per CU (two waves per SIMD at <= 256 registers);
//   * absent rows (table entry -1, rows beyond n_out) are raw-buffer loads past the end: zeros, no branch;
//   * workgroup -> tile mapping is XCD-aware: the column tiles of one row tile run on ONE XCD back to back, so the gathered rows are read
//     from HBM once and served from that XCD's L2 to the others.
// Arithmetic intensity of a workgroup is 64 flop per byte staged -- with two workgroups per CU the 64 B/clk of the CU's L2 port and the matrix
// pipe balance, so the ceiling of this form is ~half the MFMA peak; the launches it replaces ran at 8-17 %.
// Dispatch is by channel widths only, never by the row count: a row's result must not depend on the batch it is part of
// (tests/test_gpu_fullsize.py: batch = sum of its scenes, bit for bit).
#pragma once
#include "ptc_common.h"

// host side (gemm3.hip): epilogue 0 plain (bias), 1 = out: h, aux_out: GELU(h) (h rounded to the feature dtype first), 2 = out: acc * GELU'(aux_in)
int ptc_gemm3_launch(int dtype, const void* in, int64_t n_in, const void* w, const float* bias, const int32_t* nbr, int64_t n_out, int kv, int c_in,
                     int c_out, void* out, hipStream_t s, int epi, const void* aux_in, void* aux_out);
bool ptc_gemm3_supported(int dtype, int kv, int c_in, int c_out);

#ifdef PTC_GEMM3_IMPL
#include "mma.h"

#define G3_BN 128
#define G3_BK 64
#ifndef G3_STAGGER
#define G3_STAGGER 0     // > 0 (library variant d_G3_STAGGER_n, timing A/B): the second half of an XCD's workgroups -- the second workgroup of every CU --
#endif                   // sleeps n x 64 cycles before its first chunk, so that the two workgroups of a CU do not run their phases in lockstep

// weight row permutation of a 64-column wave block (the 4-tile group of spconv.hip's TileGroups): MFMA tile tt, A-row 4 gq + e holds channel
// 16 gq + 4 tt + e, so that lane (row r, group g) ends up with the 16 CONSECUTIVE channels 16 g .. 16 g + 15 of its row
__device__ __forceinline__ int g3_lds_row_of_channel(int n) { return 16 * ((n & 15) >> 2) + 4 * (n >> 4) + (n & 3); }
template <typename T> __device__ __forceinline__ uint32_t g3_pack2(float lo, float hi);
template <> __device__ __forceinline__ uint32_t g3_pack2<bf16_t>(float lo, float hi) { return ptc_pack_bf16x2(lo, hi); }
template <> __device__ __forceinline__ uint32_t g3_pack2<f16_t>(float lo, float hi) {
  const _Float16 a = (_Float16)lo, b = (_Float16)hi;
  return (uint32_t)(*reinterpret_cast<const uint16_t*>(&a)) | ((uint32_t)(*reinterpret_cast<const uint16_t*>(&b)) << 16);
}
template <typename T> __device__ __forceinline__ void g3_store16(T* dst, const float (&v)[16]) {
  uint32_t pk[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) pk[i] = g3_pack2<T>(v[2 * i], v[2 * i + 1]);
  reinterpret_cast<uint4*>(dst)[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
  reinterpret_cast<uint4*>(dst)[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
}

// byte offset of 16-byte piece p (0..7) of LDS row `row` (128-byte rows)
__device__ __forceinline__ int g3_off(int row, int p) { return row * (G3_BK * 2) + ((p ^ ((row >> 1) & 7)) << 4); }

// RS = 16-row sub-tiles per wave along the rows: 4 -> 128-row tiles, 2 -> 64-row tiles (launches with too few 128-row tiles to fill the
// chip; the same products in the same order per output row: the tile height never changes a result).
// PERSISTENT workgroups (second form, round 6): the first form ran one workgroup per tile -- 4 chunks of work behind a cold prologue (table
// entry -> gathered rows -> LDS -> barrier) and in front of the stores, ~25 % of a workgroup's life in MFMAs (qkv of a stage-3 Block: 16.4 us
// = 0.48 PF/s).  Now a workgroup walks ITS tiles as one flat chunk sequence: the loads of the next tile's first chunk, its table entries
// (one segment ahead) and its bias go out while the current tile is still being multiplied and stored.
template <typename T, int EPI, int RS>
__global__ void __launch_bounds__(256) PTC_WAVES_PER_EU(2, 2)
gemm3_kernel(const T* __restrict__ in, const T* __restrict__ w, const float* __restrict__ bias, const int32_t* __restrict__ nbr, int64_t n_out,
             int kv, int c_in, int c_out, T* __restrict__ out, const T* __restrict__ aux_in, T* __restrict__ aux_out, uint32_t in_bytes,
             int n_col_tiles, int n_tiles) {
  using M = Mma<T>;
  constexpr int BM = RS * 32;
  constexpr int A_BYTES = BM * G3_BK * 2, STAGE_BYTES = A_BYTES + G3_BN * G3_BK * 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const __amdgpu_buffer_rsrc_t in_buf = ptc_buf(in, in_bytes);
  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int wm = wave & 1, wn = wave >> 1;

  // tiles of this workgroup: XCD x (workgroup b runs on XCD b % 8) owns the contiguous tile range [T x / 8, T (x + 1) / 8) -- tile id =
  // row tile * n_col_tiles + column tile, so the column tiles of a row tile meet in ONE L2 -- and its workgroups take them round-robin
  const int PX = (int)gridDim.x >> 3, px = (int)blockIdx.x >> 3, xcd = (int)blockIdx.x & 7;
  const int tx0 = (int)((int64_t)n_tiles * xcd / 8), tx1 = (int)((int64_t)n_tiles * (xcd + 1) / 8);
  if (tx0 + px >= tx1) return;
  if (G3_STAGGER > 0 && px >= PX / 2) __builtin_amdgcn_s_sleep(G3_STAGGER);
  const int n_my = (tx1 - tx0 - px + PX - 1) / PX;
  const int cpk = c_in / G3_BK, cpt = kv * cpk, total = n_my * cpt;

  // staging roles: thread t moves piece kp = t & 7 of A rows (t >> 3) + 32 i, i < RS, and of W rows (t >> 3) + 32 i, i < 4
  const int kp = threadIdx.x & 7, rr = threadIdx.x >> 3;
  int lds_a[RS], lds_w[4];
#pragma unroll
  for (int i = 0; i < RS; ++i) lds_a[i] = g3_off(rr + 32 * i, kp);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = rr + 32 * i;
    lds_w[i] = A_BYTES + g3_off((row & 64) + g3_lds_row_of_channel(row & 63), kp);   // the epilogue's channel permutation, per 64-column wave block
  }

  // ---- load cursor: the chunk that is fetched next = (tile lj of mine, table row lk, chunk lc of the row) -------------------------------
  int lj = 0, lk = 0, lc = 0;
  int64_t l_row0 = 0;
  int l_n0 = 0;
  uint32_t a_cur[RS];            // byte offsets of this thread's A rows in the cursor's segment (tile, table row); PTC_BUF_OOB: absent
  int32_t idx_next[RS];          // table entries of the segment after it (requested one segment ahead)
  const T* w_cur[4];
  f32x4 bias_r[4];               // bias of the cursor's tile, in the accumulator layout (requested one tile ahead)
  // TWO register sets in flight: the chunk that goes to LDS at the end of this iteration and the one after it (prefetch distance 2 -- one
  // chunk of MFMAs, ~0.4 us, did not cover an L2 round trip under load and every chunk waited on its loads).  Native vectors: the HIP uint4
  // struct kept such arrays in scratch (80 bytes per lane, every prefetch behind a full wait)
  ptc_i32x4 ra0[RS], rw0[4], ra1[RS], rw1[4];
  auto set_tile = [&]() __attribute__((always_inline)) {
    const int t = tx0 + px + lj * PX;
    const int row_tile = t / n_col_tiles;
    l_row0 = (int64_t)row_tile * BM;
    l_n0 = (t - row_tile * n_col_tiles) * G3_BN;
  };
  auto request_idx = [&](int j, int k) __attribute__((always_inline)) {       // entries of segment (j, k) -> idx_next
    const int t = tx0 + px + j * PX;
    const int64_t row0 = (int64_t)(t / n_col_tiles) * BM;
#pragma unroll
    for (int i = 0; i < RS; ++i) {
      const int64_t row = row0 + rr + 32 * i;
      int32_t v = -1;
      if (row < n_out) v = nbr ? nbr[(int64_t)k * n_out + row] : (int32_t)row;
      idx_next[i] = v;
    }
  };
  auto enter_segment = [&]() __attribute__((always_inline)) {                 // idx_next -> a_cur, W pointers of (tile, table row)
#pragma unroll
    for (int i = 0; i < RS; ++i)
      a_cur[i] = idx_next[i] >= 0 ? ((uint32_t)idx_next[i] * (uint32_t)c_in + (uint32_t)(kp * 8)) * 2u : PTC_BUF_OOB;
#pragma unroll
    for (int i = 0; i < 4; ++i) w_cur[i] = w + ((int64_t)(l_n0 + rr + 32 * i) * kv + lk) * c_in + kp * 8;
  };
  auto issue_chunk = [&](ptc_i32x4 (&ra)[RS], ptc_i32x4 (&rw)[4]) __attribute__((always_inline)) {
    const int c0 = lc * G3_BK;
#pragma unroll
    for (int i = 0; i < RS; ++i)
      ra[i] = __builtin_amdgcn_raw_buffer_load_b128(in_buf, (int)(a_cur[i] == PTC_BUF_OOB ? PTC_BUF_OOB : a_cur[i] + (uint32_t)c0 * 2u), 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) rw[i] = *reinterpret_cast<const ptc_i32x4*>(w_cur[i] + c0);
  };
  auto request_bias = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      bias_r[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (bias) bias_r[t] = *reinterpret_cast<const f32x4*>(bias + l_n0 + wn * 64 + 16 * g + 4 * t);
    }
  };
  auto request_following = [&]() __attribute__((always_inline)) {             // the segment after the cursor's, if any
    if (lk + 1 < kv) request_idx(lj, lk + 1);
    else if (lj + 1 < n_my) request_idx(lj + 1, 0);
  };
  auto store_chunk = [&](int st, const ptc_i32x4 (&ra)[RS], const ptc_i32x4 (&rw)[4]) __attribute__((always_inline)) {
    unsigned char* base = smem + st * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < RS; ++i) *reinterpret_cast<ptc_i32x4*>(base + lds_a[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<ptc_i32x4*>(base + lds_w[i]) = rw[i];
  };
  auto advance = [&]() __attribute__((always_inline)) {
    if (++lc == cpk) {
      lc = 0;
      if (++lk == kv) { lk = 0; ++lj; }
    }
  };

  // the load cursor's step: fetch its chunk into (ra, rw) and move on
  int64_t p_row0 = 0;            // coordinates of the tile the load cursor entered last (the compute cursor takes them over when it gets there)
  int p_n0 = 0;
  // `live` = there is such a chunk.  The chunk loads are issued UNCONDITIONALLY (past the end: the last chunk again, never stored) and every
  // conditional load (table entries, bias) goes out BEFORE them: the compiler's s_waitcnt in front of the LDS stores of the other register
  // set counts the loads issued since on the path with the fewest -- with the chunk loads under a branch that count was zero and each
  // store waited for the loads just issued (the first build of this form: prefetch distance 1 again)
  auto fetch = [&](ptc_i32x4 (&ra)[RS], ptc_i32x4 (&rw)[4], bool live) __attribute__((always_inline)) {
    const bool seg_start = live && lc == 0, tile_start = seg_start && lk == 0;
    if (tile_start) { set_tile(); p_row0 = l_row0; p_n0 = l_n0; }
    if (seg_start) enter_segment();
    if (tile_start) request_bias();          // (consumed two iterations later, after this tile's predecessor has taken its own)
    if (seg_start) request_following();
    issue_chunk(ra, rw);
    if (live) advance();
  };

  // prologue: chunk 0 cold, chunk 1 behind it (total >= 2: a tile has at least two chunks)
  request_idx(0, 0);
  fetch(ra0, rw0, true);
  int64_t c_row0 = p_row0;       // ---- compute cursor: the tile being multiplied, its chunk counter
  int c_n0 = p_n0;
  int crem = 0;
  fetch(ra1, rw1, true);
  store_chunk(0, ra0, rw0);
  __syncthreads();

  f32x4 acc[RS][4];
  auto half = [&](int it, ptc_i32x4 (&raL)[RS], ptc_i32x4 (&rwL)[4], const ptc_i32x4 (&raS)[RS], const ptc_i32x4 (&rwS)[4]) __attribute__((always_inline)) {
    if (crem == 0) {               // before the load cursor may replace bias_r with the next tile's
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int s = 0; s < RS; ++s) acc[s][t] = bias_r[t];
    }
    fetch(raL, rwL, it + 2 < total);
    const unsigned char* As = smem + (it & 1) * STAGE_BYTES;
    const unsigned char* Ws = As + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < G3_BK / 32; ++ks) {
      typename M::frag fw[4], fa[RS];
#pragma unroll
      for (int t = 0; t < 4; ++t) fw[t] = *reinterpret_cast<const typename M::frag*>(Ws + g3_off(wn * 64 + t * 16 + r, ks * 4 + g));
#pragma unroll
      for (int s = 0; s < RS; ++s) fa[s] = *reinterpret_cast<const typename M::frag*>(As + g3_off(wm * (RS * 16) + s * 16 + r, ks * 4 + g));
#pragma unroll
      for (int s = 0; s < RS; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[s][t] = M::mma(fw[t], fa[s], acc[s][t]);
    }
    if (crem == cpt - 1) {
      // epilogue: lane (r, g) holds channels n0 + 64 wn + 16 g .. + 15 of row (row0 + RS 16 wm + 16 s + r): two 16-byte stores per sub-tile
#pragma unroll
      for (int s = 0; s < RS; ++s) {
        const int64_t row = c_row0 + wm * (RS * 16) + s * 16 + r;
        if (row >= n_out) continue;
        const int64_t off = row * c_out + c_n0 + wn * 64 + 16 * g;
        float v[16];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[4 * t + e] = acc[s][t][e];
        if constexpr (EPI == 2) {
          T hv[16];
          *reinterpret_cast<uint4*>(hv) = reinterpret_cast<const uint4*>(aux_in + off)[0];
          *reinterpret_cast<uint4*>(hv + 8) = reinterpret_cast<const uint4*>(aux_in + off)[1];
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] *= ptc_gelu_grad(ptc_to_float(hv[i]));
        }
        g3_store16<T>(out + off, v);
        if constexpr (EPI == 1) {
          float u[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) u[i] = ptc_gelu(ptc_to_float(ptc_from_float<T>(v[i])));   // the activation sees h rounded to the feature dtype (as fwd2.h)
          g3_store16<T>(aux_out + off, u);
        }
      }
      crem = 0;
      c_row0 = p_row0;        // (the load cursor is inside the next tile by now and not yet in the one after: cpt >= 2)
      c_n0 = p_n0;
    } else {
      ++crem;
    }
    if (it + 1 < total) store_chunk((it + 1) & 1, raS, rwS);
    __syncthreads();
  };
  // both halves unconditional inside the loop (a skipped second half on the back edge made the compiler assume the worst about the loads in
  // flight and drain them at the top of every first half); an odd last chunk runs behind it
  int it = 0;
#pragma unroll 1
  for (; it + 1 < total; it += 2) {
    half(it, ra0, rw0, ra1, rw1);
    half(it + 1, ra1, rw1, ra0, rw0);
  }
  if (it < total) half(it, ra0, rw0, ra1, rw1);
}

static inline bool gemm3_enabled() {
  static const bool on = [] { const char* e = getenv("PTC_GEMM3"); return !(e && e[0] == '0'); }();   // PTC_GEMM3=0: the former kernels (timing A/B)
  return on;
}
bool ptc_gemm3_supported(int dtype, int kv, int c_in, int c_out) {
  return gemm3_enabled() && dtype != PTC_F32 && kv >= 1 && kv <= 4 && c_in >= 128 && c_in % G3_BK == 0 && c_out % G3_BN == 0;
}

template <typename T>
static int launch_gemm3(const void* in, int64_t n_in, const void* w, const float* bias, const int32_t* nbr, int64_t n_out, int kv, int c_in, int c_out,
                        void* out, hipStream_t s, int epi, const void* aux_in, void* aux_out) {
  const int n_col = c_out / G3_BN;
  // 64-row tiles when 128-row tiles would leave workgroup slots idle (two workgroups per CU, 256 CUs)
  const bool small = ptc_cdiv(n_out, 128) * n_col < 512;
  const int bm = small ? 64 : 128;
  const int64_t n_tiles = ptc_cdiv(n_out, bm) * n_col;
  if (n_tiles > 0x7fffffffll) { ptc_set_error("gemm3: %lld tiles", (long long)n_tiles); return PTC_EUNSUPPORTED; }
  int64_t wgs = (n_tiles + 7) & ~(int64_t)7;
  if (wgs > 512) wgs = 512;
  const dim3 grid((unsigned)wgs);
  const uint32_t in_bytes = (uint32_t)((uint64_t)n_in * c_in * sizeof(T));
#define G3_LAUNCH_RS(EE, RR)                                                                                                           \
  {                                                                                                                                    \
    auto kern = gemm3_kernel<T, EE, RR>;                                                                                               \
    const int lds = 2 * (RR * 32 + G3_BN) * G3_BK * 2;                                                                                 \
    PTC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));                \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, (const T*)in, (const T*)w, bias, nbr, n_out, kv, c_in, c_out, (T*)out,           \
                       (const T*)aux_in, (T*)aux_out, in_bytes, n_col, (int)n_tiles);                                                  \
  }
#define G3_LAUNCH(EE) { if (small) G3_LAUNCH_RS(EE, 2) else G3_LAUNCH_RS(EE, 4) }
  if (epi == 1) G3_LAUNCH(1) else if (epi == 2) G3_LAUNCH(2) else G3_LAUNCH(0)
#undef G3_LAUNCH
#undef G3_LAUNCH_RS
  PTC_CHECK_LAUNCH("gemm3_kernel");
  return PTC_OK;
}

int ptc_gemm3_launch(int dtype, const void* in, int64_t n_in, const void* w, const float* bias, const int32_t* nbr, int64_t n_out, int kv, int c_in,
                     int c_out, void* out, hipStream_t s, int epi, const void* aux_in, void* aux_out) {
  if (dtype == PTC_BF16) return launch_gemm3<bf16_t>(in, n_in, w, bias, nbr, n_out, kv, c_in, c_out, out, s, epi, aux_in, aux_out);
  return launch_gemm3<f16_t>(in, n_in, w, bias, nbr, n_out, kv, c_in, c_out, out, s, epi, aux_in, aux_out);
}
#endif  // PTC_GEMM3_IMPL
