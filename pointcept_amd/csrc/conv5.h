// conv5.h -- forward / dgrad kernel of the gather-table convolution for 16-bit features with c_in in {32, 64, 128}:
// conv3's chunked pipeline with every global access turned into WHOLE-CACHE-LINE, lane-contiguous requests.
// Included by spconv.hip.
//
// Measured on MI355X (tools/probe_gather.hip, profiles/r02_e_probe_gather.txt): what a 1-KB wave gather of 16 rows
// costs on the vector-memory path depends on how lanes map to addresses, not on the bytes moved:
//     lane -> (row l & 15, 16-B piece l >> 4)   [MFMA B layout]   57 cycles (L1 or L2 resident), 131-153 beyond L2
//     lane -> (row l >> 2, piece l & 3)         [quad = 64 B]     15 (L1) / 33 (L2) / 131-153 (beyond: half lines)
//     lane -> (row l >> 3, piece l & 7)         [8 lanes = 128 B] 19 / 20 / 69-79
// conv3 gathers in the MFMA layout, fetches every 128-byte row as two half-line requests in different instructions, and
// stages W with four lanes 64 bytes apart: ~60 address cycles per non-empty gather and per W load; at the dec0 shape
// that is 236 of the kernel's 263 us (TA_BUSY 72 %, matrix pipe 12 %, profiles/r02_a_conv_pmc_s0.json).  A wave-private
// LDS bounce with quad-coalesced gathers alone (conv3 BNC) bought nothing: the rows still left L2 as half lines.
//
// conv5:
//   * per (row tile, table row) the c_in/32 = NS load instructions each fetch 16/NS COMPLETE rows: lane l ->
//     (row l / (4 NS), piece l % (4 NS)), 4 NS adjacent lanes = one contiguous row = whole 128-byte lines;
//   * the registers go to a wave-private LDS tile image [16 rows][c_in] (pieces XOR-swizzled by f(row) so that the
//     ds_write_b128 in load layout and the ds_read_b128 in MFMA layout are both conflict-free for the lane groups of
//     MI355X_MICROARCH.md's LDS table) one table row before its MFMAs; the freed registers are reloaded at once with
//     the same table row of the next chunk (gather distance: one chunk, as conv3);
//   * W chunks are fetched with 16 lanes per 256-byte weight-row slice (two full lines) and stored in the same
//     swizzled tile format, fragment stride 1088 B (the eight stores of a lane group cover two fragments);
//   * everything else is conv3: flattened contraction, 128-wide W chunks double-buffered with one barrier per chunk,
//     table entries two chunks ahead, empty (tile, table row) cells skipped, XCD-first block order.
// Same operands in the same order as conv3: bit-identical results.
#pragma once

// XOR swizzle of the 16-byte piece index inside a tile-image row, by pieces per row (4 NS)
template <int NS> __device__ __forceinline__ int c5_swz(int row);
template <> __device__ __forceinline__ int c5_swz<1>(int row) { return (0x1230 >> ((row >> 2) * 4)) & 3; }   // {0,3,2,1}[row >> 2]
template <> __device__ __forceinline__ int c5_swz<2>(int row) { return (row >> 1) & 7; }
template <> __device__ __forceinline__ int c5_swz<4>(int row) { return row & 15; }

template <typename T, int RT, int NS, int NTILES>
__global__ void __launch_bounds__(256, 2)
conv5_kernel(const T* __restrict__ in, const T* __restrict__ w, const float* __restrict__ bias, const int32_t* __restrict__ nbr,
             int64_t n_out, int kv, int c_out, int n_rowblk, T* __restrict__ out, uint32_t in_bytes, uint32_t w_bytes,
             const int32_t* __restrict__ only_where_negative) {
  using M = Mma<T>;
  using frag = typename M::frag;
  constexpr int C_IN = NS * 32, TPC = 4 / NS;             // table rows per 128-wide chunk
  constexpr int NT = NTILES * 16, BM = RT * 64;
  constexpr int WFRAG = C3_FRAG + C3_FPAD;                  // 1088: fragment stride of the W image
  constexpr int WBUF = 4 * NTILES * WFRAG;                  // one W chunk
  constexpr int PITCH = C_IN * 2;                           // tile-image row pitch (no padding: swizzled)
  constexpr int IMG = 16 * PITCH;                           // one tile image
  constexpr int LPR = 4 * NS, RPI = 16 / NS;                // lanes per row, rows per load instruction
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const __amdgpu_buffer_rsrc_t in_buf = ptc_buf(in, in_bytes), w_buf = ptc_buf(w, w_bytes);
  const int ny = c_out / NT;
  const int nblk = n_rowblk * ny;
  const int per_xcd = (nblk + 7) >> 3;
  const int lb = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (lb >= nblk) return;
  const int rb = lb / ny, n0 = (lb - rb * ny) * NT;
  // second launch behind conv7 (RT = 2: the 128-row workgroups are conv7's blocks): only the blocks it left (count < 0)
  if (only_where_negative != nullptr && only_where_negative[rb] >= 0) return;
  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int64_t row0 = (int64_t)rb * BM + wave * (RT * 16);
  const int KV = kv * C_IN;
  const int nchunks = (KV + 127) >> 7;

  // ---- W staging: instruction `it` -> weight rows it*16 .. it*16+15, 16 lanes per 256-byte slice
  constexpr int WI = NT / 16;
  const int wpiece = threadIdx.x & 15;
  uint32_t wsrc[WI];
  int wdst[WI];
#pragma unroll
  for (int it = 0; it < WI; ++it) {
    const int wrow = it * 16 + (threadIdx.x >> 4);
    wsrc[it] = (uint32_t)((n0 + wrow) * KV + wpiece * 8) * 2u;             // bytes; + c * 256 per chunk
    const int prow = lds_row_of_channel<NTILES>(wrow);
    const int rr = prow & 15;
    wdst[it] = ((prow >> 4) * 4 + (wpiece >> 2)) * WFRAG + rr * 64 + (((wpiece & 3) ^ c5_swz<1>(rr)) << 4);
  }
  uint4 wreg[WI];
  auto wload = [&](int c) {
    const bool ok = c * 128 + wpiece * 8 < KV;                                // KV is a multiple of 8: whole pieces
#pragma unroll
    for (int it = 0; it < WI; ++it) wreg[it] = ptc_buf_load16(w_buf, ok ? wsrc[it] + (uint32_t)c * 256u : PTC_BUF_OOB);
  };
  auto wstore = [&](int buf) {
#pragma unroll
    for (int it = 0; it < WI; ++it) *reinterpret_cast<uint4*>(smem + buf * WBUF + wdst[it]) = wreg[it];
  };

  // ---- gather ring: slot kk * NS + i = load instruction i of table row kk of the chunk, all RT row tiles
  frag ga[4][RT];
  bool anyv[TPC][RT];
  int32_t idxN[4][RT], idxNN[4][RT];
  const int lrow = lane / LPR, lpiece = lane % LPR;          // tile row (within the instruction's RPI rows) and piece of this lane
  auto load_idx = [&](int c, int32_t (&ix)[4][RT]) {
#pragma unroll
    for (int kk = 0; kk < TPC; ++kk) {
      const int k = c * TPC + kk;
#pragma unroll
      for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int j = 0; j < RT; ++j) {
          const int64_t row = row0 + j * 16 + i * RPI + lrow;
          const bool ok = k < kv && row < n_out;
          const int32_t e = nbr[(int64_t)(k < kv ? k : kv - 1) * n_out + (row < n_out ? row : n_out - 1)];   // always in bounds
          ix[kk * NS + i][j] = ok ? e : -1;
        }
    }
  };
  auto issue_tap = [&](int kk, const int32_t (&ix)[4][RT]) {
#pragma unroll
    for (int j = 0; j < RT; ++j) {
      bool any = false;
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int32_t e = ix[kk * NS + i][j];
        ga[kk * NS + i][j] = ld_frag_buf<T>(in_buf, e >= 0 ? ((uint32_t)e * (uint32_t)C_IN + (uint32_t)lpiece * 8u) * 2u : PTC_BUF_OOB);
        any = any || (__builtin_amdgcn_ballot_w64(e >= 0) != 0);
      }
      anyv[kk][j] = any;
    }
  };
  unsigned char* img = smem + 2 * WBUF + wave * (RT * IMG);
  int woff[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int rr = i * RPI + lrow;
    woff[i] = rr * PITCH + ((lpiece ^ c5_swz<NS>(rr)) << 4);
  }
  const int rsw = c5_swz<NS>(r);
  // table row kk of the ring -> the wave's tile images
  auto bounce_write = [&](int kk, bool (&fa)[RT]) {
#pragma unroll
    for (int j = 0; j < RT; ++j) {
      fa[j] = anyv[kk][j];
      if (fa[j]) {
#pragma unroll
        for (int i = 0; i < NS; ++i) *reinterpret_cast<frag*>(img + j * IMG + woff[i]) = ga[kk * NS + i][j];
      }
    }
    w2_wave_sync();
  };
  // MFMA operands of step si of the table row currently in the images
  auto read_frags = [&](int si, const bool (&fa)[RT], frag (&fb)[RT]) {
    const int off = r * PITCH + (((4 * si + g) ^ rsw) << 4);
#pragma unroll
    for (int j = 0; j < RT; ++j)
      if (fa[j]) fb[j] = *reinterpret_cast<const frag*>(img + j * IMG + off);
  };

  f32x4 acc[RT][NTILES];
  {
    f32x4 breg[NTILES];
    sc_bias_regs<NTILES>(bias, n0, g, breg);
#pragma unroll
    for (int j = 0; j < RT; ++j)
#pragma unroll
      for (int t = 0; t < NTILES; ++t) acc[j][t] = breg[t];
  }

  // ---- prologue
  wload(0);
  load_idx(0, idxN);
  wstore(0);
#pragma unroll
  for (int kk = 0; kk < TPC; ++kk) issue_tap(kk, idxN);
  load_idx(1, idxN);
  wload(1);
  __syncthreads();

  frag fB[RT], fN[RT];
  bool aB[RT], aN[RT];
  bounce_write(0, aB);
  read_frags(0, aB, fB);
  issue_tap(0, idxN);                       // table row 0 of chunk 1

  const int abase = r * 64 + ((g ^ c5_swz<1>(r)) << 4);
#pragma unroll 1
  for (int c = 0; c < nchunks; ++c) {
    const unsigned char* wb = smem + (c & 1) * WBUF + abase;
    load_idx(c + 2, idxNN);
#pragma unroll
    for (int kk = 0; kk < TPC; ++kk) {
#pragma unroll
      for (int si = 0; si < NS; ++si) {
        const int s = kk * NS + si;
        frag wf[NTILES];
#pragma unroll
        for (int t = 0; t < NTILES; ++t) wf[t] = *reinterpret_cast<const frag*>(wb + (t * 4 + s) * WFRAG);
        if (si == NS - 1) {
          // the next table row goes through the images now (all reads of the current one are issued); its
          // registers are refilled with the same table row one chunk further
          const int nk = (kk + 1) % TPC;            // compile-time after unrolling
          bounce_write(nk, aN);
          read_frags(0, aN, fN);
          if (kk + 1 < TPC) issue_tap(nk, idxN); else issue_tap(nk, idxNN);
        } else {
#pragma unroll
          for (int j = 0; j < RT; ++j) aN[j] = aB[j];
          read_frags(si + 1, aB, fN);
        }
#pragma unroll
        for (int j = 0; j < RT; ++j) {
          if (aB[j]) {
#pragma unroll
            for (int t = 0; t < NTILES; ++t) acc[j][t] = M::mma(wf[t], fB[j], acc[j][t]);
          }
        }
#pragma unroll
        for (int j = 0; j < RT; ++j) {
          fB[j] = fN[j];
          aB[j] = aN[j];
        }
      }
    }
    wstore((c + 1) & 1);
    __syncthreads();
    wload(c + 2);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < RT; ++j) idxN[q][j] = idxNN[q][j];
  }

#pragma unroll
  for (int j = 0; j < RT; j += 2) {
    const int64_t rowA = row0 + j * 16 + r;
    sc_epilogue<T, NTILES>(*reinterpret_cast<f32x4(*)[2][NTILES]>(&acc[j]), nullptr, out, rowA, rowA + 16, n_out, c_out, n0, g);
  }
}

static inline bool conv5_supported(int dtype, int kv, int c_in, int c_out, const int32_t* nbr, int64_t n_in) {
  if (dtype == PTC_F32 || nbr == nullptr || kv < 2) return false;
  // (c_in = 128 on this form measured slower than conv3's direct gathers: 946 vs 562 us at 128 -> 96, N = 819200 -- 4-KB tile images, two
  //  waves per SIMD; not instantiated)
  if (!(c_in == 32 || c_in == 64) || c_out % 32 != 0) return false;
  return (uint64_t)n_in * (uint64_t)c_in * 2 <= PTC_BUF_MAX_BYTES && (uint64_t)c_out * kv * c_in * 2 <= PTC_BUF_MAX_BYTES;
}

template <typename T, int RT, int NS, int NTILES>
static int launch_conv5_i(const void* in, int64_t n_in, const void* w, const float* bias, const int32_t* nbr, int64_t n_out, int kv,
                          int c_out, void* out, hipStream_t s, const int32_t* only_where_negative) {
  constexpr int C_IN = NS * 32;
  const int n_rowblk = (int)ptc_cdiv(n_out, RT * 64);
  const int nblk = n_rowblk * (c_out / (NTILES * 16));
  const size_t lds = (size_t)2 * 4 * NTILES * (C3_FRAG + C3_FPAD) + (size_t)4 * RT * 16 * C_IN * 2;
  auto kern = conv5_kernel<T, RT, NS, NTILES>;
  static size_t allowed = 48 * 1024;   // per instantiation
  if (lds > allowed) {
    PTC_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    allowed = lds;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(8 * ((nblk + 7) / 8))), dim3(256), lds, s, (const T*)in, (const T*)w, bias, nbr, n_out, kv, c_out,
                     n_rowblk, (T*)out, (uint32_t)((uint64_t)n_in * C_IN * sizeof(T)), (uint32_t)((uint64_t)c_out * kv * C_IN * sizeof(T)),
                     only_where_negative);
  PTC_CHECK_LAUNCH("conv5_kernel");
  return PTC_OK;
}

template <typename T>
static int launch_conv5(const void* in, int64_t n_in, const void* w, const float* bias, const int32_t* nbr, int64_t n_out, int kv,
                        int c_in, int c_out, void* out, hipStream_t s, const int32_t* only_where_negative = nullptr) {
  const int nt = c_out % 64 == 0 ? 4 : (c_out % 96 == 0 ? 6 : 2);
  // 128-row workgroups (RT = 2: three waves per SIMD) measured faster than 256-row ones at every shape
  // (64 -> 64, N = 819200: 230 vs 244 us; 32 -> 32: 94 vs 118 us, profiles/r02_f_conv_stages_ops.txt)
  const int ns = c_in / 32;
#define C5_CASE(S, N)                                                                                             \
  if (ns == S && nt == N)                                                                                           \
    return launch_conv5_i<T, 2, S, N>(in, n_in, w, bias, nbr, n_out, kv, c_out, out, s, only_where_negative);
  C5_CASE(1, 2) C5_CASE(1, 4) C5_CASE(1, 6) C5_CASE(2, 2) C5_CASE(2, 4) C5_CASE(2, 6)
#undef C5_CASE
  ptc_set_error("conv5: c_in=%d c_out=%d unsupported", c_in, c_out);
  return PTC_EUNSUPPORTED;
}
