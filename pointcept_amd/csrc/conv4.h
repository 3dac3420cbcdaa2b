// conv4.h -- fourth-generation forward / dgrad kernel of the SUBMANIFOLD gather-table convolution for 16-bit
// features, c_in in {32, 64, 96, 128}: the input rows a block of output rows needs are staged ONCE in LDS
// (block-local rulebook of blocks.hip) and the MFMA B operands are gathered from LDS instead of L1/L2.
// Included by spconv.hip.
//
// Why: conv3 is bound by the vector-memory ADDRESS path, not by HBM, L2 or the matrix pipe.  rocprofv3 PMC at the
// dec0 shape of PT-v3m1 (SubM 3^3, 64 -> 64, N = 819200, profiles/r02_a_conv_pmc_s0.json): HBM traffic 308 MB =
// 1.03 x algorithmic, L2 hit rate 87 %, matrix pipe 12 % busy, TA_BUSY 72 % of the kernel: 5.6 M wave-level 1-KB
// gathers x 16 address cycles.  Every input row is gathered once per table entry that names it (9.3 x per voxel).
// With rows in curve order the DISTINCT rows behind a block of 256 outputs are ~1.5 x 256, so:
//   * prologue: the block's halo list (ascending global rows, blocks.hip) is read, each row is fetched from
//     global memory once (16-byte pieces, fully coalesced) into an LDS image [slot][c_in] (+16 B pitch padding);
//     slot `hmax` is a zero row: "no neighbour" entries point there, so the gather is an unconditional ds_read_b128;
//   * main loop = conv3's: contraction flattened to v = k * c_in + c, 128-wide chunks of W double-buffered through
//     LDS in MFMA fragment order (one barrier per chunk), B fragments of (chunk c+1, step s) requested right after
//     (chunk c, step s) is multiplied, local table entries (int16) two chunks ahead, row tiles without any
//     neighbour at a step skipped; 8 waves x RT row tiles of 16 rows, all sharing one W image and one halo image;
//   * blocks whose halo does not fit (hcnt > hmax: rows in no spatial order) run a plain, un-pipelined loop over
//     the global table -- correct for any input, fast for the ordered ones.
// Same summation order as conv3 (chunk, step, tile): results are bit-identical to it.
#pragma once

#define C4_THREADS 512
#define C4_WAVES 8

template <typename T, int RT, int KPC, int NTILES, bool GEN>
__global__ void __launch_bounds__(C4_THREADS, 2)
conv4_kernel(const T* __restrict__ in, const T* __restrict__ w, const float* __restrict__ bias, const int32_t* __restrict__ nbr,
             const int16_t* __restrict__ lnbr, const int32_t* __restrict__ halo, const int32_t* __restrict__ hcnt, int hmax,
             int64_t n_out, int kv, int c_in, int c_out, int n_rowblk, T* __restrict__ out, uint32_t in_bytes) {
  using M = Mma<T>;
  using frag = typename M::frag;
  constexpr int NT = NTILES * 16, BM = C4_WAVES * RT * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int ny = c_out / NT;
  const int nblk = n_rowblk * ny;
  const int per_xcd = (nblk + 7) >> 3;
  const int lb = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (lb >= nblk) return;
  const int rb = lb / ny, n0 = (lb - rb * ny) * NT;
  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int64_t row0 = (int64_t)rb * BM + wave * (RT * 16);
  const int KV = kv * c_in;
  const int nchunks = (KV + 127) >> 7;
  const int pitch = c_in * 2 + 16;
  unsigned char* wbuf = smem;                                          // 2 x C3_BUF(NTILES)
  int* ids = reinterpret_cast<int*>(smem + 2 * C3_BUF(NTILES));         // [hmax]
  unsigned char* rows = smem + 2 * C3_BUF(NTILES) + hmax * 4;           // [hmax + 1][pitch]

  // ---- W staging (as conv3; 512 threads: one pass covers 128 weight rows) ----
  const int qd = threadIdx.x & 3;
  const int wrow = threadIdx.x >> 2;
  const bool wthread = wrow < NT;
  const T* wsrc = w + (int64_t)(n0 + (wthread ? wrow : 0)) * KV;
  int wdst;
  {
    const int prow = lds_row_of_channel<NTILES>(wthread ? wrow : 0);
    wdst = ((prow >> 4) * 4 + qd) * (C3_FRAG + C3_FPAD) + (prow & 15) * 16;
  }
  uint4 wreg[4];
  auto wload = [&](int c) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int v0 = c * 128 + qd * 32 + gq * 8;
      const bool ok = wthread && v0 < KV;
      const int vc = v0 < KV ? v0 : KV - 8;
      uint4 v = *reinterpret_cast<const uint4*>(wsrc + vc);   // unconditional, clamped (see conv3.h)
      if (!ok) v = make_uint4(0, 0, 0, 0);
      wreg[gq] = v;
    }
  };
  auto wstore = [&](int buf) {
    if (wthread) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) *reinterpret_cast<uint4*>(wbuf + buf * C3_BUF(NTILES) + wdst + gq * 256) = wreg[gq];
    }
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

  const int hc = hcnt[rb];
  if (hc <= hmax) {
    // =========================== LDS path ===========================
    for (int i = threadIdx.x; i < hc; i += C4_THREADS) ids[i] = halo[(int64_t)rb * hmax + i];
    wload(0);
    __syncthreads();
    {
      const int pr = c_in >> 3;                     // 16-byte pieces per row
      const int total = hc * pr;
      for (int p0 = threadIdx.x; p0 < total; p0 += 4 * C4_THREADS) {
        uint4 v[4];
        int off[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int p = p0 + u * C4_THREADS;
          const int pc = p < total ? p : total - 1;
          const int row = pc / pr, q = pc - row * pr;
          v[u] = *reinterpret_cast<const uint4*>(in + (int64_t)ids[row] * c_in + q * 8);
          off[u] = p < total ? row * pitch + q * 16 : -1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (off[u] >= 0) *reinterpret_cast<uint4*>(rows + off[u]) = v[u];
      }
      for (int q = threadIdx.x; q < (pitch >> 4); q += C4_THREADS) *reinterpret_cast<uint4*>(rows + hmax * pitch + q * 16) = make_uint4(0, 0, 0, 0);
    }
    wstore(0);

    frag ga[4][RT];
    bool anyv[4][RT];
    int idxN[KPC][RT], idxNN[KPC][RT];
    auto load_idx = [&](int c, int (&ix)[KPC][RT]) {
      const int kfirst = (GEN || KPC == 1) ? (c * 128) / c_in : c * KPC;
#pragma unroll
      for (int kk = 0; kk < KPC; ++kk) {
        const int k = kfirst + kk;
#pragma unroll
        for (int j = 0; j < RT; ++j) {
          const int64_t row = row0 + j * 16 + r;
          const bool ok = k < kv && row < n_out;
          const int e = lnbr[(int64_t)(k < kv ? k : kv - 1) * n_out + (row < n_out ? row : n_out - 1)];   // always in bounds
          ix[kk][j] = ok ? e : -1;
        }
      }
    };
    auto issue = [&](int c, int s, const int (&ix)[KPC][RT]) {
      const int v0 = c * 128 + s * 32;
      int kk, cbase;
      if constexpr (GEN) {
        const int k = v0 / c_in;
        kk = k - (c * 128) / c_in;
        cbase = v0 - k * c_in + g * 8;
      } else {
        kk = (s * KPC) >> 2;
        cbase = v0 % c_in + g * 8;
      }
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        int i;
        if constexpr (GEN) {
          i = ix[0][j];
#pragma unroll
          for (int q = 1; q < KPC; ++q) i = kk == q ? ix[q][j] : i;
        } else {
          i = ix[kk][j];
        }
        const int slot = i >= 0 ? i : hmax;                      // zero row
        ga[s][j] = *reinterpret_cast<const frag*>(rows + slot * pitch + cbase * 2);
        anyv[s][j] = __builtin_amdgcn_ballot_w64(i >= 0) != 0;
      }
    };

    load_idx(0, idxN);
    __syncthreads();                       // halo rows, zero row and W(0) are in LDS
#pragma unroll
    for (int s = 0; s < 4; ++s) issue(0, s, idxN);
    load_idx(1, idxN);
    wload(1);

#pragma unroll 1
    for (int c = 0; c < nchunks; ++c) {
      const unsigned char* wb = wbuf + (c & 1) * C3_BUF(NTILES) + lane * 16;
      load_idx(c + 2, idxNN);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        frag wf[NTILES];
#pragma unroll
        for (int t = 0; t < NTILES; ++t) wf[t] = *reinterpret_cast<const frag*>(wb + (t * 4 + s) * (C3_FRAG + C3_FPAD));
#pragma unroll
        for (int j = 0; j < RT; ++j) {
          if (anyv[s][j]) {
#pragma unroll
            for (int t = 0; t < NTILES; ++t) acc[j][t] = M::mma(wf[t], ga[s][j], acc[j][t]);
          }
        }
        issue(c + 1, s, idxN);
      }
      wstore((c + 1) & 1);
      __syncthreads();
      wload(c + 2);
#pragma unroll
      for (int kk = 0; kk < KPC; ++kk)
#pragma unroll
        for (int j = 0; j < RT; ++j) idxN[kk][j] = idxNN[kk][j];
    }
  } else {
    // =========================== fallback: global table, no pipelining ===========================
    const __amdgpu_buffer_rsrc_t in_buf = ptc_buf(in, in_bytes);
#pragma unroll 1
    for (int c = 0; c < nchunks; ++c) {
      __syncthreads();
      wload(c);
      wstore(0);
      __syncthreads();
      const unsigned char* wb = wbuf + lane * 16;
#pragma unroll 1
      for (int s = 0; s < 4; ++s) {
        const int v0 = c * 128 + s * 32;
        if (v0 >= KV) break;
        const int k = v0 / c_in, cb = v0 - k * c_in + g * 8;
        frag wf[NTILES];
#pragma unroll
        for (int t = 0; t < NTILES; ++t) wf[t] = *reinterpret_cast<const frag*>(wb + (t * 4 + s) * (C3_FRAG + C3_FPAD));
#pragma unroll
        for (int j = 0; j < RT; ++j) {
          const int64_t row = row0 + j * 16 + r;
          const int32_t e = row < n_out ? nbr[(int64_t)k * n_out + row] : -1;
          const frag b = ld_frag_buf<T>(in_buf, e >= 0 ? ((uint32_t)e * (uint32_t)c_in + (uint32_t)cb) * 2u : PTC_BUF_OOB);
          if (__builtin_amdgcn_ballot_w64(e >= 0) != 0) {
#pragma unroll
            for (int t = 0; t < NTILES; ++t) acc[j][t] = M::mma(wf[t], b, acc[j][t]);
          }
        }
      }
    }
  }

  if constexpr (RT >= 2) {
#pragma unroll
    for (int j = 0; j < RT; j += 2) {
      const int64_t rowA = row0 + j * 16 + r;
      sc_epilogue<T, NTILES>(*reinterpret_cast<f32x4(*)[2][NTILES]>(&acc[j]), nullptr, out, rowA, rowA + 16, n_out, c_out, n0, g);
    }
  } else {
    f32x4 two[2][NTILES];
#pragma unroll
    for (int t = 0; t < NTILES; ++t) { two[0][t] = acc[0][t]; two[1][t] = acc[0][t]; }
    const int64_t rowA = row0 + r;
    sc_epilogue<T, NTILES>(two, nullptr, out, rowA, n_out, n_out, c_out, n0, g);   // second row >= n_out: skipped
  }
}

// LDS bytes of one workgroup
static inline size_t conv4_lds_bytes(int ntiles, int hmax, int c_in) {
  return (size_t)2 * C3_BUF(ntiles) + (size_t)hmax * 4 + (size_t)(hmax + 1) * (c_in * 2 + 16);
}

static inline bool conv4_supported(int dtype, int kv, int c_in, int c_out, int bm, int hmax) {
  if (dtype == PTC_F32 || kv < 2) return false;
  if (c_in % 32 != 0 || c_out % 32 != 0 || c_in > 128) return false;
  if (bm != 128 && bm != 256) return false;
  const int nt = c_out % 64 == 0 ? 4 : (c_out % 96 == 0 ? 6 : 2);
  return conv4_lds_bytes(nt, hmax, c_in) <= 160 * 1024 && (hmax * 4) % 16 == 0;
}

template <typename T, int RT, int KPC, int NTILES, bool GEN>
static int launch_conv4_i(const void* in, int64_t n_in, const void* w, const float* bias, const int32_t* nbr, const int16_t* lnbr,
                          const int32_t* halo, const int32_t* hcnt, int hmax, int64_t n_out, int kv, int c_in, int c_out, void* out,
                          hipStream_t s) {
  const int n_rowblk = (int)ptc_cdiv(n_out, C4_WAVES * RT * 16);
  const int nblk = n_rowblk * (c_out / (NTILES * 16));
  const size_t lds = conv4_lds_bytes(NTILES, hmax, c_in);
  auto kern = conv4_kernel<T, RT, KPC, NTILES, GEN>;
  static size_t allowed = 0;   // per instantiation
  if (lds > allowed) {
    PTC_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    allowed = lds;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(8 * ((nblk + 7) / 8))), dim3(C4_THREADS), lds, s, (const T*)in, (const T*)w, bias, nbr, lnbr,
                     halo, hcnt, hmax, n_out, kv, c_in, c_out, n_rowblk, (T*)out, (uint32_t)((uint64_t)n_in * c_in * sizeof(T)));
  PTC_CHECK_LAUNCH("conv4_kernel");
  return PTC_OK;
}

template <typename T>
static int launch_conv4(const void* in, int64_t n_in, const void* w, const float* bias, const int32_t* nbr, const int16_t* lnbr,
                        const int32_t* halo, const int32_t* hcnt, int bm, int hmax, int64_t n_out, int kv, int c_in, int c_out, void* out,
                        hipStream_t s) {
  const int nt = c_out % 64 == 0 ? 4 : (c_out % 96 == 0 ? 6 : 2);
  const int kpc = c_in % 128 == 0 ? 1 : (c_in == 32 ? 4 : 2);
  const bool gen = !(c_in == 32 || c_in == 64 || c_in % 128 == 0);
  const int rt = bm / (C4_WAVES * 16);
#define C4_CASE(R, K, N, G)                                                                                                \
  if (rt == R && kpc == K && nt == N && gen == G)                                                                           \
    return launch_conv4_i<T, R, K, N, G>(in, n_in, w, bias, nbr, lnbr, halo, hcnt, hmax, n_out, kv, c_in, c_out, out, s);
  // bm = 256 (c_in <= 64): 32 / 64 input channels
  C4_CASE(2, 4, 2, false) C4_CASE(2, 4, 4, false) C4_CASE(2, 4, 6, false)
  C4_CASE(2, 2, 2, false) C4_CASE(2, 2, 4, false) C4_CASE(2, 2, 6, false)
  // bm = 128 (c_in = 96, 128)
  C4_CASE(1, 2, 2, true) C4_CASE(1, 2, 4, true) C4_CASE(1, 2, 6, true)
  C4_CASE(1, 1, 2, false) C4_CASE(1, 1, 4, false) C4_CASE(1, 1, 6, false)
#undef C4_CASE
  ptc_set_error("conv4: c_in=%d c_out=%d bm=%d unsupported", c_in, c_out, bm);
  return PTC_EUNSUPPORTED;
}
