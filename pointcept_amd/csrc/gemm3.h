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

#define G3_BM 128
#define G3_BN 128
#define G3_BK 64
#define G3_STAGE_ELEMS ((G3_BM + G3_BN) * G3_BK)          // 16-bit elements per pipeline stage (32 KB)
#define G3_LDS_BYTES (2 * G3_STAGE_ELEMS * 2)

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

template <typename T, int EPI>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
gemm3_kernel(const T* __restrict__ in, const T* __restrict__ w, const float* __restrict__ bias, const int32_t* __restrict__ nbr, int64_t n_out,
             int kv, int c_in, int c_out, T* __restrict__ out, const T* __restrict__ aux_in, T* __restrict__ aux_out, uint32_t in_bytes,
             int n_col_tiles) {
  using M = Mma<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const __amdgpu_buffer_rsrc_t in_buf = ptc_buf(in, in_bytes);
  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int wm = wave & 1, wn = wave >> 1;

  // XCD-aware tile id: workgroup b runs on XCD b % 8; XCD x owns the logical ids [x q + min(x, rem), ...) -- consecutive logical ids are the
  // column tiles of one row tile
  const int nwg = (int)gridDim.x, id = (int)blockIdx.x;
  const int q = nwg >> 3, rem = nwg & 7, xcd = id & 7;
  const int logical = xcd * q + (xcd < rem ? xcd : rem) + (id >> 3);
  const int row_tile = logical / n_col_tiles, col_tile = logical - row_tile * n_col_tiles;
  const int64_t row0 = (int64_t)row_tile * G3_BM;
  const int n0 = col_tile * G3_BN;

  // staging roles: thread t moves piece kp = t & 7 of rows (t >> 3) + 32 i, i < 4, of both operands
  const int kp = threadIdx.x & 7, rr = threadIdx.x >> 3;
  const int chunks_per_k = c_in / G3_BK, n_chunks = kv * chunks_per_k;
  uint32_t a_off[4];            // byte offset of this thread's A rows at channel 0 of the current table row (PTC_BUF_OOB: absent)
  const T* w_ptr[4];
  int lds_a[4], lds_w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = rr + 32 * i;
    lds_a[i] = g3_off(row, kp);
    const int wrow = (row & 64) + g3_lds_row_of_channel(row & 63);       // the epilogue's channel permutation, per 64-column wave block
    lds_w[i] = G3_BM * G3_BK * 2 + g3_off(wrow, kp);
    w_ptr[i] = w + (int64_t)(n0 + row) * kv * c_in + kp * 8;
  }
  auto load_idx = [&](int k) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t row = row0 + rr + 32 * i;
      int32_t j = -1;
      if (row < n_out) j = nbr ? nbr[(int64_t)k * n_out + row] : (int32_t)row;
      a_off[i] = j >= 0 ? ((uint32_t)j * (uint32_t)c_in + (uint32_t)(kp * 8)) * 2u : PTC_BUF_OOB;
    }
  };
  ptc_i32x4 ra[4], rw[4];          // native vectors: the HIP uint4 struct kept these arrays in scratch (80 bytes per lane, every prefetch behind a full wait)
  auto load_chunk = [&](int ch) __attribute__((always_inline)) {
    const int k = ch / chunks_per_k, c0 = (ch - k * chunks_per_k) * G3_BK;
    if (c0 == 0 && ch > 0) load_idx(k);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = __builtin_amdgcn_raw_buffer_load_b128(in_buf, (int)(a_off[i] == PTC_BUF_OOB ? PTC_BUF_OOB : a_off[i] + (uint32_t)c0 * 2u), 0, 0);
      rw[i] = *reinterpret_cast<const ptc_i32x4*>(w_ptr[i] + (int64_t)k * c_in + c0);
    }
  };
  auto store_chunk = [&](int st) __attribute__((always_inline)) {
    unsigned char* base = smem + st * (G3_STAGE_ELEMS * 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<ptc_i32x4*>(base + lds_a[i]) = ra[i];
      *reinterpret_cast<ptc_i32x4*>(base + lds_w[i]) = rw[i];
    }
  };

  // accumulators start at the bias of the channel they are stored to (sc_bias_regs mapping of a 4-tile group)
  f32x4 acc[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    f32x4 b = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (bias) b = *reinterpret_cast<const f32x4*>(bias + n0 + wn * 64 + 16 * g + 4 * t);
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[s][t] = b;
  }

  load_idx(0);
  load_chunk(0);
  store_chunk(0);
  __syncthreads();
#pragma unroll 1
  for (int ch = 0; ch < n_chunks; ++ch) {
    const bool more = ch + 1 < n_chunks;
    if (more) load_chunk(ch + 1);
    const unsigned char* As = smem + (ch & 1) * (G3_STAGE_ELEMS * 2);
    const unsigned char* Ws = As + G3_BM * G3_BK * 2;
#pragma unroll
    for (int ks = 0; ks < G3_BK / 32; ++ks) {
      typename M::frag fw[4], fa[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) fw[t] = *reinterpret_cast<const typename M::frag*>(Ws + g3_off(wn * 64 + t * 16 + r, ks * 4 + g));
#pragma unroll
      for (int s = 0; s < 4; ++s) fa[s] = *reinterpret_cast<const typename M::frag*>(As + g3_off(wm * 64 + s * 16 + r, ks * 4 + g));
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[s][t] = M::mma(fw[t], fa[s], acc[s][t]);
    }
    if (more) store_chunk((ch + 1) & 1);
    __syncthreads();
  }

  // epilogue: lane (r, g) holds channels n0 + 64 wn + 16 g .. + 15 of row (row0 + 64 wm + 16 s + r): two 16-byte stores per sub-tile
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int64_t row = row0 + wm * 64 + s * 16 + r;
    if (row >= n_out) continue;
    const int64_t off = row * c_out + n0 + wn * 64 + 16 * g;
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
  const int64_t n_row = ptc_cdiv(n_out, G3_BM);
  if (n_row * n_col > 0x7fffffffll) { ptc_set_error("gemm3: %lld x %d tiles", (long long)n_row, n_col); return PTC_EUNSUPPORTED; }
  const dim3 grid((unsigned)(n_row * n_col));
  const uint32_t in_bytes = (uint32_t)((uint64_t)n_in * c_in * sizeof(T));
#define G3_LAUNCH(EE)                                                                                                                  \
  {                                                                                                                                    \
    auto kern = gemm3_kernel<T, EE>;                                                                                                   \
    PTC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS_BYTES));       \
    hipLaunchKernelGGL(kern, grid, dim3(256), G3_LDS_BYTES, s, (const T*)in, (const T*)w, bias, nbr, n_out, kv, c_in, c_out, (T*)out,  \
                       (const T*)aux_in, (T*)aux_out, in_bytes, n_col);                                                                \
  }
  if (epi == 1) G3_LAUNCH(1) else if (epi == 2) G3_LAUNCH(2) else G3_LAUNCH(0)
#undef G3_LAUNCH
  PTC_CHECK_LAUNCH("gemm3_kernel");
  return PTC_OK;
}

int ptc_gemm3_launch(int dtype, const void* in, int64_t n_in, const void* w, const float* bias, const int32_t* nbr, int64_t n_out, int kv, int c_in,
                     int c_out, void* out, hipStream_t s, int epi, const void* aux_in, void* aux_out) {
  if (dtype == PTC_BF16) return launch_gemm3<bf16_t>(in, n_in, w, bias, nbr, n_out, kv, c_in, c_out, out, s, epi, aux_in, aux_out);
  return launch_gemm3<f16_t>(in, n_in, w, bias, nbr, n_out, kv, c_in, c_out, out, s, epi, aux_in, aux_out);
}
#endif  // PTC_GEMM3_IMPL
