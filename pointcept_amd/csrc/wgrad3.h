// wgrad3.h -- weight gradient of the Linear layers of the DEEP stages: dw[co][ci] = sum_o dout[o][co] * in[nbr[o]][ci] (+ the bias gradient
// sum_o dout[o][co]) for 16-bit features, c_in and c_out multiples of 128 (the 128 / 256 / 512-channel Blocks of PT-v3m1: qkv, proj, fc1, fc2
// and the Linear of the positional encoding; ptv3m1:173-248,278-284).
//
// Why (round 6, profiles/r06_at_step_sequence.txt): wgrad2 makes every WAVE an independent worker with a 64 x 64 output tile that streams
// 32-row steps through a wave-private LDS slice -- right at N = 819200 x 32 / 64 channels, where the whole weight is ONE such tile.  At
// 256 -> 1024 the weight is 64 tiles: every dout row is fetched 4 times and every input row 16 times, ~900 MB of L2 -> CU traffic for the
// five Linears of a stage-3 Block (20 000 rows) whose operands are 70 MB: 81 us for 34 GF (0.42 PF/s), plus 41 us to add the ~50 partial
// sums per weight.  Here a WORKGROUP owns a 128 x 128 tile of dw (4 waves, 2 x 2, 64 x 64 each) over a contiguous range of 64-row chunks:
//   * both operands of a chunk go global -> registers -> LDS ONCE per workgroup (row-major [16-channel plane][32 rows][16] images, the
//     layout of wgrad2.h) and are read back transposed by ds_read_b64_tr_b16: half the traffic per flop of the 64 x 64 form;
//   * two register sets in flight (prefetch distance 2), double-buffered LDS, one barrier per chunk -- the pipeline of gemm3.h;
//   * the bias gradient rides in the matrix pipe: the waves of a workgroup's first input-column half multiply the dout fragments with a
//     fragment of ones (only in workgroups of the first input tile);
//   * split-K by contiguous chunk ranges, ~192 workgroups per weight: 3-32 partial sums per weight instead of ~50; the partials go to the
//     deterministic reduction of spconv.hip (no atomics, bit-reproducible);
//   * several weights in ONE launch (the five of a Block), workgroup -> (weight, split, tile) with the tiles of a split on one XCD.
#pragma once
#include "ptc_common.h"

#define W3_MAX 8
// workgroups per weight the split-K plan aims at (library variants d_W3_TARGET_n).  96 | 192 | 384: wgrad3 903 | 903 | 912 us per step, the
// reduction of its partials 406 | 499 | 540 us (profiles/r06_bd_wgrad3_split_target.txt): the kernel does not care, the reduction does
#ifndef W3_TARGET
#define W3_TARGET 96
#endif
struct W3Problem {
  const void* in; const void* dout; const int32_t* nbr; int64_t n_out; int c_in, c_out;
  float* partial; float* bias_partial;       // [splits][c_out][c_in], [splits][c_out] (bias_partial may be null)
  int splits, cps, tiles, tiles_i;           // cps = 64-row chunks per split
  uint32_t in_bytes, dout_bytes;
};
struct W3Group { int n; int start[W3_MAX + 1]; W3Problem p[W3_MAX]; };

bool ptc_wgrad3_supported(int dtype, int kv, int c_in, int c_out, int64_t n_out);
// number of partial sums (splits) of a weight and the chunks each of them covers
void ptc_wgrad3_plan(int64_t n_out, int c_in, int c_out, int* splits, int* cps);
int ptc_wgrad3_launch(int dtype, const W3Group& g, hipStream_t s);

#ifdef PTC_WGRAD3_IMPL
#include "wgrad2.h"          // w2_off / w2_frag / W2_PLANE: the LDS image format and its transposing fragment read

#define W3_IMG (8 * W2_PLANE)                 // one 32-row x 128-channel image
#define W3_STAGE (4 * W3_IMG)                 // dout rows 0..31 | dout rows 32..63 | in rows 0..31 | in rows 32..63
#define W3_LDS_BYTES (2 * W3_STAGE)

template <typename T>
__global__ void __launch_bounds__(256) PTC_WAVES_PER_EU(2, 2)
wgrad3_kernel(W3Group g) {
  using M = Mma<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  const int r = lane & 15, gq = lane >> 4;
  const int wo = wave & 1, wi = wave >> 1;

  // XCD-aware logical id (workgroup b runs on XCD b % 8): consecutive logical ids -- the tiles of one split -- share an L2
  const int nwg = (int)gridDim.x, id = (int)blockIdx.x;
  const int q8 = nwg >> 3, rem = nwg & 7, xcd = id & 7;
  const int logical = xcd * q8 + (xcd < rem ? xcd : rem) + (id >> 3);
  int pj = 0;
#pragma unroll
  for (int q = 1; q < W3_MAX; ++q)
    if (q < g.n && logical >= g.start[q]) pj = q;
  const W3Problem& P = g.p[pj];
  const int local = logical - g.start[pj];
  const int split = local / P.tiles, tile = local - split * P.tiles;
  const int to = tile / P.tiles_i, ti = tile - to * P.tiles_i;
  const int c_in = P.c_in, c_out = P.c_out;
  const int64_t n_out = P.n_out;
  const int32_t* __restrict__ nbr = P.nbr;
  const __amdgpu_buffer_rsrc_t in_buf = ptc_buf(P.in, P.in_bytes), dout_buf = ptc_buf(P.dout, P.dout_bytes);
  const int64_t n_chunks = (n_out + 63) >> 6;
  const int64_t ch0 = (int64_t)split * P.cps;
  int64_t ch1 = ch0 + P.cps;
  if (ch1 > n_chunks) ch1 = n_chunks;
  const int total = (int)(ch1 - ch0);              // >= 1 by the plan
  const bool do_bias = P.bias_partial != nullptr && ti == 0 && wi == 0;

  // staging roles: thread t moves the 16-byte piece t & 15 (8 channels) of rows (t >> 4) + 16 i, i < 4, of both operands
  const int piece = threadIdx.x & 15, rr = threadIdx.x >> 4;
  int lds_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = rr + 16 * i;
    lds_off[i] = (row >> 5) * W3_IMG + w2_off(row & 31, piece);
  }
  const uint32_t d_col = (uint32_t)(to * 128 + piece * 8) * 2u, x_col = (uint32_t)(ti * 128 + piece * 8) * 2u;

  ptc_i32x4 rd0[4], rx0[4], rd1[4], rx1[4];
  int32_t idx_next[4];          // table entries of the chunk AFTER the one being fetched (requested one chunk ahead)
  auto request_idx = [&](int64_t ch) __attribute__((always_inline)) {       // raw entries (clamped address, unconditional): nothing here may
#pragma unroll                                                              // touch the loaded value, or the wave waits for it on the spot
    for (int i = 0; i < 4; ++i) {
      int64_t row = (ch << 6) + rr + 16 * i;
      if (row >= n_out) row = n_out - 1;
      idx_next[i] = nbr ? nbr[row] : (int32_t)row;
    }
  };
  auto fetch = [&](int64_t ch, ptc_i32x4 (&rd)[4], ptc_i32x4 (&rx)[4]) __attribute__((always_inline)) {
    uint32_t xo[4], dofs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t row = (ch << 6) + rr + 16 * i;
      const bool ok = row < n_out;
      xo[i] = (ok && idx_next[i] >= 0) ? (uint32_t)idx_next[i] * (uint32_t)c_in * 2u + x_col : PTC_BUF_OOB;
      dofs[i] = ok ? (uint32_t)row * (uint32_t)c_out * 2u + d_col : PTC_BUF_OOB;
    }
    request_idx(ch + 1);                           // before the chunk loads (gemm3.h: conditional / early loads go first)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      rd[i] = __builtin_amdgcn_raw_buffer_load_b128(dout_buf, (int)dofs[i], 0, 0);
      rx[i] = __builtin_amdgcn_raw_buffer_load_b128(in_buf, (int)xo[i], 0, 0);
    }
  };
  auto store_chunk = [&](int st, const ptc_i32x4 (&rd)[4], const ptc_i32x4 (&rx)[4]) __attribute__((always_inline)) {
    unsigned char* base = smem + st * W3_STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<ptc_i32x4*>(base + lds_off[i]) = rd[i];
      *reinterpret_cast<ptc_i32x4*>(base + 2 * W3_IMG + lds_off[i]) = rx[i];
    }
  };

  f32x4 acc[4][4], accb[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    accb[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[t][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  typename M::frag ones;
  {
    const short one = sizeof(T) == 2 && std::is_same<T, bf16_t>::value ? (short)0x3f80 : (short)0x3c00;
    s16x8 o = {one, one, one, one, one, one, one, one};
    __builtin_memcpy(&ones, &o, sizeof(ones));
  }

  // prologue: chunk 0 cold, chunk 1 behind it (past the end: the loads are clamped / out of range and never stored)
  request_idx(ch0);
  fetch(ch0, rd0, rx0);
  fetch(ch0 + 1, rd1, rx1);
  store_chunk(0, rd0, rx0);
  __syncthreads();

  auto half = [&](int it, ptc_i32x4 (&rdL)[4], ptc_i32x4 (&rxL)[4], const ptc_i32x4 (&rdS)[4], const ptc_i32x4 (&rxS)[4]) __attribute__((always_inline)) {
    fetch(ch0 + it + 2, rdL, rxL);                 // unconditional (rows past the end read as zeros)
    const unsigned char* st = smem + (it & 1) * W3_STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      typename M::frag fa[4], fb[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) fa[t] = w2_frag<T>(st + ks * W3_IMG, wo * 4 + t, lane);
#pragma unroll
      for (int u = 0; u < 4; ++u) fb[u] = w2_frag<T>(st + 2 * W3_IMG + ks * W3_IMG, wi * 4 + u, lane);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t][u] = M::mma(fa[t], fb[u], acc[t][u]);
      if (do_bias) {
#pragma unroll
        for (int t = 0; t < 4; ++t) accb[t] = M::mma(fa[t], ones, accb[t]);
      }
    }
    if (it + 1 < total) store_chunk((it + 1) & 1, rdS, rxS);
    __syncthreads();
  };
  // both halves unconditional inside the loop (a skipped second half on the back edge made the compiler assume the worst about the loads in
  // flight and drain them at the top of every first half); an odd last chunk runs behind it
  int it = 0;
#pragma unroll 1
  for (; it + 1 < total; it += 2) {
    half(it, rd0, rx0, rd1, rx1);
    half(it + 1, rd1, rx1, rd0, rx0);
  }
  if (it < total) half(it, rd0, rx0, rd1, rx1);

  // epilogue: acc[t][u][e] = dw[co = 128 to + 64 wo + 16 t + 4 gq + e][ci = 128 ti + 64 wi + 16 u + r]
  float* __restrict__ part = P.partial + (int64_t)split * c_out * c_in;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int co = to * 128 + wo * 64 + t * 16 + 4 * gq + e;
      float* dst = part + (int64_t)co * c_in + ti * 128 + wi * 64 + r;
#pragma unroll
      for (int u = 0; u < 4; ++u) dst[16 * u] = acc[t][u][e];
    }
  if (do_bias && r == 0) {
    float* bp = P.bias_partial + (int64_t)split * c_out + to * 128 + wo * 64 + 4 * gq;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) bp[16 * t + e] = accb[t][e];
  }
}

static inline bool wgrad3_enabled() {
  static const bool on = [] { const char* e = getenv("PTC_WGRAD3"); return !(e && e[0] == '0'); }();   // PTC_WGRAD3=0: wgrad2 (timing A/B)
  return on;
}
bool ptc_wgrad3_supported(int dtype, int kv, int c_in, int c_out, int64_t n_out) {
  return wgrad3_enabled() && dtype != PTC_F32 && kv == 1 && c_in >= 128 && c_in % 128 == 0 && c_out % 128 == 0 && n_out >= 1;
}
void ptc_wgrad3_plan(int64_t n_out, int c_in, int c_out, int* splits, int* cps) {
  const int64_t n_chunks = (n_out + 63) >> 6;
  const int tiles = (c_in / 128) * (c_out / 128);
  int64_t s = (W3_TARGET + tiles - 1) / tiles;     // ~W3_TARGET workgroups per weight
  const int64_t min_s = (n_chunks + 127) / 128;    // ... and no workgroup walks more than 128 chunks (8192 rows): at 200 000+ rows (the outdoor
  if (s < min_s) s = min_s;                        // configuration's deep stages) the target alone left 96 workgroups on 512 slots
  if (s > 64) s = 64;
  const int64_t max_s = (n_chunks + 3) / 4;        // at least 4 chunks per split
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  const int64_t c = (n_chunks + s - 1) / s;
  *cps = (int)c;
  *splits = (int)((n_chunks + c - 1) / c);         // no empty split
}

int ptc_wgrad3_launch(int dtype, const W3Group& g, hipStream_t s) {
  const int grid = g.start[g.n];
  if (grid <= 0) return PTC_OK;
  if (dtype == PTC_BF16) {
    auto kern = wgrad3_kernel<bf16_t>;
    PTC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, W3_LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), W3_LDS_BYTES, s, g);
  } else {
    auto kern = wgrad3_kernel<f16_t>;
    PTC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, W3_LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), W3_LDS_BYTES, s, g);
  }
  PTC_CHECK_LAUNCH("wgrad3_kernel");
  return PTC_OK;
}
#endif  // PTC_WGRAD3_IMPL
