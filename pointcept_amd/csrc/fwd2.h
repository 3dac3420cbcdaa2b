// fwd2.h -- second-generation forward kernels of the gather-table convolution for 16-bit features
// and c_in <= 256 (the large-N stages of PTv3 / SpUNet).  Included by spconv.hip after the v1
// kernel, whose weight-row permutation (lds_row_of_channel) and epilogue (sc_epilogue) they share.
//
// v1 walked the table one row k at a time: two workgroup barriers, a restaged W_k and an unhidden
// dependent load chain (table entry -> gathered row) per k, ~10 % of the HBM roofline at C = 32.
//   conv2_kernel   (kv > 1): W is staged for a GROUP of table rows at once (<= 40 KB), the table
//                  entries of the group sit in a wave-private LDS block, and every wave streams its
//                  (k, 32-channel step) units through a PD-deep register ring: the gathers of unit
//                  u + PD are in flight while unit u is multiplied.  Table rows with no valid entry
//                  in a wave's 32 output rows cost that wave nothing.  Two barriers per GROUP.
//   linear2_kernel (kv == 1: nn.Linear, 1x1x1 convs, the gather-fused qkv / proj GEMMs): persistent
//                  workgroups keep W in LDS for their whole life and walk 128-row tiles; the rows of
//                  tile i+1 (and the table entries of tile i+2) are loaded before tile i is multiplied.
#pragma once
#ifndef F2_FULL_PATH
#define F2_FULL_PATH 1       // 0: timing A/B only (`python -m pointcept_amd.build --variant d_F2_FULL_PATH_0`): the guarded W reads for every width
#endif

#define F2_ROWS 128
#define F2_MAX_W_BYTES (40 * 1024)

template <typename T, int NTILES, int PD, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, WAVES == 4 ? 3 : 2)  // both: <= 170 registers per lane
conv2_kernel(const T* __restrict__ in, const T* __restrict__ w, const float* __restrict__ bias,
             const int32_t* __restrict__ nbr, int64_t n_out, int kv, int c_in, int c_out, int kg, T* __restrict__ out,
             uint32_t in_bytes) {
  using M = Mma<T>;
  const __amdgpu_buffer_rsrc_t in_buf = ptc_buf(in, in_bytes);
  constexpr int NT = NTILES * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // contraction chunks of <= 128 channels: a "virtual table row" v = k * nch + chunk reads W[.][k][chunk]
  // and the same table entry nbr[k]; c_in <= 128 is the single-chunk case
  const int cc_len = c_in < 128 ? c_in : 128;          // c_in % 128 == 0 when c_in > 128 (host-checked)
  const int nch = c_in / cc_len;
  const int kvv = kv * nch;
  const int pitch = cc_len + 8;                        // elements; +16 B staggers the fragment reads
  T* wl = reinterpret_cast<T*>(smem);                  // [kg][NT][pitch]
  const int w_bytes = (kg * NT * pitch * 2 + 15) & ~15;
  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  int32_t* il = reinterpret_cast<int32_t*>(smem + w_bytes) + wave * kg * 32;  // this wave's [kg][32] entries
  const int r = lane & 15, g = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * (WAVES * 32) + wave * 32;
  const int n0 = blockIdx.y * NT;
  const int64_t rowA = row0 + r, rowB = row0 + 16 + r;
  const int S = (cc_len + 31) >> 5;                    // 32-channel steps per virtual table row
  const int vpr = cc_len >> 3;                         // 16-byte vectors per staged weight row

  f32x4 acc[2][NTILES], breg[NTILES];
  sc_bias_regs<NTILES>(bias, n0, g, breg);
#pragma unroll
  for (int t = 0; t < NTILES; ++t) { acc[0][t] = breg[t]; acc[1][t] = breg[t]; }

#pragma unroll 1
  for (int k0 = 0; k0 < kvv; k0 += kg) {
    const int gk = (kvv - k0) < kg ? (kvv - k0) : kg;
    if (k0 > 0) __syncthreads();  // every wave is done with the previous group's W slices
    for (int q = lane; q < gk * 32; q += 64) {
      const int64_t row = row0 + (q & 31);
      il[q] = row < n_out ? nbr[(int64_t)((k0 + (q >> 5)) / nch) * n_out + row] : -1;
    }
    for (int q = threadIdx.x; q < gk * NT * vpr; q += WAVES * 64) {
      const int kk = q / (NT * vpr), rem = q - kk * NT * vpr;
      const int n = rem / vpr, cc = rem - n * vpr;
      const int v = k0 + kk, k = v / nch, ch = v - k * nch;
      *reinterpret_cast<uint4*>(wl + (kk * NT + lds_row_of_channel<NTILES>(n)) * pitch + cc * 8) =
          *reinterpret_cast<const uint4*>(w + ((int64_t)(n0 + n) * kv + k) * c_in + ch * cc_len + cc * 8);
    }
    __syncthreads();

    const int U = gk * S;
    typename M::frag ra[PD], rb[PD];
    int32_t xa[PD], xb[PD];
    auto issue = [&](int j, int u) {
      const int kk = u / S, s = u - kk * S;
      const int col = s * 32 + g * 8;
      const int base_c = ((k0 + kk) % nch) * cc_len;
      const int32_t ia = il[kk * 32 + r], ib = il[kk * 32 + 16 + r];
      // unconditional buffer loads (absent rows = out-of-range offsets = zeros): loads under exec-masked branches
      // force s_waitcnt vmcnt(0) at every use and serialise the ring against the MFMAs
      const uint32_t cb = (uint32_t)(base_c + col);
      typename M::frag fa = ld_frag_buf<T>(in_buf, (col < cc_len && ia >= 0) ? ((uint32_t)ia * (uint32_t)c_in + cb) * 2u : PTC_BUF_OOB);
      typename M::frag fb = ld_frag_buf<T>(in_buf, (col < cc_len && ib >= 0) ? ((uint32_t)ib * (uint32_t)c_in + cb) * 2u : PTC_BUF_OOB);
      ra[j] = fa; rb[j] = fb; xa[j] = ia; xb[j] = ib;
    };
#pragma unroll
    for (int j = 0; j < PD; ++j)
      if (j < U) issue(j, j);
#pragma unroll 1
    for (int base = 0; base < U; base += PD) {
#pragma unroll
      for (int j = 0; j < PD; ++j) {
        const int u = base + j;
        if (u < U) {
          const typename M::frag fa = ra[j], fb = rb[j];
          const bool any = __builtin_amdgcn_ballot_w64((xa[j] >= 0) | (xb[j] >= 0)) != 0;
          if (u + PD < U) issue(j, u + PD);
          if (any) {
            const int kk = u / S, s = u - kk * S;
            const int col = s * 32 + g * 8;
            const T* wrow = wl + (kk * NT + r) * pitch + col;
#pragma unroll
            for (int t = 0; t < NTILES; ++t) {
              typename M::frag fw = M::zero();
              if (col < cc_len) fw = ld_frag<T>(wrow + t * 16 * pitch);
              acc[0][t] = M::mma(fw, fa, acc[0][t]);
              acc[1][t] = M::mma(fw, fb, acc[1][t]);
            }
          }
        }
      }
    }
  }
  sc_epilogue<T, NTILES>(acc, nullptr, out, rowA, rowB, n_out, c_out, n0, g);
}

// ------------------------------------------------------------------------------------------------
// Epilogues of the MLP (ptv3m1:225-248: fc1 -> GELU -> fc2), fused into the GEMMs that produce the values:
//   EPI 1 (fc1 forward)    : out = h (pre-activation, kept for the backward), aux_out = GELU(h)
//   EPI 2 (fc2 input grad) : out = acc * GELU'(aux_in)   (aux_in = h: the gradient leaves the kernel already
//                            multiplied through the activation; no dA tensor is ever written)
// Same lane -> channel mapping as sc_epilogue; GELU is nn.GELU()'s erf form, evaluated in fp32 (ptc_gelu).
__device__ __forceinline__ float f2_gelu(float z) { return ptc_gelu(z); }             // ptc_common.h: branch-free since round 6
__device__ __forceinline__ float f2_gelu_grad(float z) { return ptc_gelu_grad(z); }

template <typename T, int NTILES, int EPI>
__device__ __forceinline__ void f2_epilogue_ex(f32x4 (&acc)[2][NTILES], T* __restrict__ out, const T* __restrict__ aux_in,
                                               T* __restrict__ aux_out, int64_t rowA, int64_t rowB, int64_t n_out, int c_out,
                                               int n0, int g) {
  static_assert(sizeof(T) == 2, "16-bit features only");
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int64_t row = s ? rowB : rowA;
    if (row >= n_out) continue;
#pragma unroll
    for (int t = 0; t < NTILES; ++t) {
      const int gs = TileGroups<NTILES>::gstart(t), G = TileGroups<NTILES>::gsize(t);
      if (t != gs) continue;
      const int ch0 = n0 + 16 * gs + 4 * G * g;
      const int64_t off = row * c_out + ch0;
      float v[16], u[16];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * tt + e] = acc[s][(gs + tt) < NTILES ? (gs + tt) : t][e];
      const int nv = 4 * G;                                   // valid values: 16, 8 or 4 consecutive channels
      if (EPI == 2) {
        T hv[16];
        if (G == 4) { *reinterpret_cast<uint4*>(hv) = reinterpret_cast<const uint4*>(aux_in + off)[0]; *reinterpret_cast<uint4*>(hv + 8) = reinterpret_cast<const uint4*>(aux_in + off)[1]; }
        else if (G == 2) { *reinterpret_cast<uint4*>(hv) = reinterpret_cast<const uint4*>(aux_in + off)[0]; }
        else { *reinterpret_cast<uint2*>(hv) = reinterpret_cast<const uint2*>(aux_in + off)[0]; }
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (i < nv) v[i] *= f2_gelu_grad(ptc_to_float(hv[i]));
      }
      if (EPI == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (i < nv) {
            // the activation sees the value the reference's GELU sees: h rounded to the feature dtype
            const float hr = ptc_to_float(ptc_from_float<T>(v[i]));
            u[i] = f2_gelu(hr);
          }
      }
      auto store = [&](T* dst, const float* val) {
        uint32_t pk[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) pk[i] = sc_pack2<T>(val[2 * i], val[2 * i + 1]);
        if (G == 4) {
          reinterpret_cast<uint4*>(dst)[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
          reinterpret_cast<uint4*>(dst)[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
        } else if (G == 2) {
          reinterpret_cast<uint4*>(dst)[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        } else {
          reinterpret_cast<uint2*>(dst)[0] = make_uint2(pk[0], pk[1]);
        }
      };
      store(out + off, v);
      if (EPI == 1) store(aux_out + off, u);
    }
  }
}

// Row-contiguous stores through a wave-private LDS slice (plain epilogue, c_in <= 64 only).  In the MFMA accumulator layout the four lanes that
// hold one output row are 16 lanes apart; the texture-address unit charges such a store like a scattered one (tools/probe_gather.hip:
// 57 cycles per KB against 15-19 when adjacent lanes share a line) and rocprofv3 shows linear2 with TA_BUSY 58-66 % at N = 819200, the
// epilogue stores being 60-75 % of its vector-memory instructions (profiles/r02_aj_linear_pmc_*.json).  Each 16-row half of a wave's
// tile is written to LDS as [row][channel] (pitch + 16 B: the 16 rows of a ds_write_b128 pass land on distinct banks) and read back 16 B
// per lane in row-major order, so that every global store instruction writes whole, consecutive 128-byte lines: 32->256 118 -> 99 us,
// 64->128 70 -> 61, 64->256 121 -> 107 (profiles/r02_ak_linear_probe.txt).  A COMPILE-TIME variant (LDSS): carrying it as a run-time
// branch inside every instance cost the GELU epilogues and the 128 / 256-channel instances 10-60 % (registers; kernel stats of r02_at
// against r02_ag), and the LDS forms of the two GELU epilogues themselves measured slower -- they keep the direct epilogue.
#define F2_OUT_ROWS 16
template <typename T, int NTILES>
__device__ __forceinline__ void f2_store_rows_lds(f32x4 (&acc)[2][NTILES], unsigned char* slice, T* __restrict__ out, int64_t row0,
                                                  int64_t n_out, int c_out, int n0, int r, int g, int lane) {
  static_assert(sizeof(T) == 2, "16-bit features only");
  constexpr int NT = NTILES * 16, RB = NT * 2, P = RB + 16, PIECES = RB / 16;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int t = 0; t < NTILES; ++t) {
      const int gs = TileGroups<NTILES>::gstart(t), G = TileGroups<NTILES>::gsize(t);
      if (t != gs) continue;
      uint32_t pk[8];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const f32x4 v = acc[s][(gs + tt) < NTILES ? (gs + tt) : t];
        pk[2 * tt] = sc_pack2<T>(v[0], v[1]);
        pk[2 * tt + 1] = sc_pack2<T>(v[2], v[3]);
      }
      unsigned char* dst = slice + r * P + (16 * gs + 4 * G * g) * 2;
      if (G == 4) {
        reinterpret_cast<uint4*>(dst)[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        reinterpret_cast<uint4*>(dst)[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
      } else if (G == 2) {
        reinterpret_cast<uint4*>(dst)[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
      } else {
        reinterpret_cast<uint2*>(dst)[0] = make_uint2(pk[0], pk[1]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int it = 0; it < (F2_OUT_ROWS * PIECES + 63) / 64; ++it) {
      const int q = it * 64 + lane, row = q / PIECES, piece = q - row * PIECES;
      const int64_t grow = row0 + s * 16 + row;
      if (row < F2_OUT_ROWS && grow < n_out) {
        const uint4 v = *reinterpret_cast<const uint4*>(slice + row * P + piece * 16);
        *reinterpret_cast<uint4*>(out + grow * c_out + n0 + piece * 8) = v;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();      // the slice is rewritten by the next half / tile
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}
__host__ __device__ static inline size_t f2_out_slice_bytes(int nt) { return (size_t)F2_OUT_ROWS * (nt * 2 + 16); }

template <typename T, int NTILES, int S, int EPI = 0, bool LDSS = false>
__global__ void __launch_bounds__(256)
linear2_kernel(const T* __restrict__ in, const T* __restrict__ w, const float* __restrict__ bias,
               const int32_t* __restrict__ nbr, int64_t n_out, int c_in, int c_out, int nh, T* __restrict__ out,
               const T* __restrict__ aux_in, T* __restrict__ aux_out, uint32_t in_bytes) {
  // A workgroup owns `nh` column blocks of NT channels (W of all of them in LDS) and walks them per row tile with the
  // row fragments held in registers: the input is read ONCE.  (One column block per workgroup and gridDim.y = 2 read
  // every input row from two workgroups and ran at 0.31-0.45 of the HBM roof for c_out = 192 / 256 against 0.56-0.77
  // for c_out <= 128, N = 819200: profiles/r02_v_linear_probe.txt.)
  using M = Mma<T>;
  const __amdgpu_buffer_rsrc_t in_buf = ptc_buf(in, in_bytes);
  constexpr int NT = NTILES * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int pitch = c_in + 8;
  T* wl = reinterpret_cast<T*>(smem);  // [nh][NT][pitch]
  float* bl = reinterpret_cast<float*>(smem + (((size_t)nh * NT * pitch * 2 + 15) & ~(size_t)15));   // [nh][NT] bias
  [[maybe_unused]] unsigned char* oslice = reinterpret_cast<unsigned char*>(bl + nh * NT) + (threadIdx.x >> 6) * f2_out_slice_bytes(NT);   // LDSS only
  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.y * NT * nh;
  const int vpr = c_in >> 3;
  for (int q = threadIdx.x; q < nh * NT * vpr; q += 256) {
    const int n = q / vpr, cc = q - n * vpr;
    const int h = n / NT, nl = n - h * NT;
    *reinterpret_cast<uint4*>(wl + (h * NT + lds_row_of_channel<NTILES>(nl)) * pitch + cc * 8) =
        *reinterpret_cast<const uint4*>(w + (int64_t)(n0 + n) * c_in + cc * 8);
  }
  for (int q = threadIdx.x; q < nh * NT; q += 256) bl[q] = bias ? bias[n0 + q] : 0.f;
  __syncthreads();

  const int64_t tiles = (n_out + F2_ROWS - 1) / F2_ROWS;
  auto load_idx = [&](int64_t tile, int32_t& ia, int32_t& ib) {
    const int64_t rowA = tile * F2_ROWS + wave * 32 + r, rowB = rowA + 16;
    // unconditional loads (clamped), validity applied by select: keeps the compiler's vmcnt bookkeeping exact
    const bool okA = tile < tiles && rowA < n_out, okB = tile < tiles && rowB < n_out;
    const int64_t ca = okA ? rowA : 0, cb = okB ? rowB : 0;
    const int32_t ja = nbr ? nbr[ca] : (int32_t)ca, jb = nbr ? nbr[cb] : (int32_t)cb;
    ia = okA ? ja : -1;
    ib = okB ? jb : -1;
  };
  auto load_rows = [&](int32_t ia, int32_t ib, typename M::frag (&fa)[S], typename M::frag (&fb)[S]) {
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int col = s * 32 + g * 8;
      fa[s] = ld_frag_buf<T>(in_buf, (col < c_in && ia >= 0) ? ((uint32_t)ia * (uint32_t)c_in + (uint32_t)col) * 2u : PTC_BUF_OOB);
      fb[s] = ld_frag_buf<T>(in_buf, (col < c_in && ib >= 0) ? ((uint32_t)ib * (uint32_t)c_in + (uint32_t)col) * 2u : PTC_BUF_OOB);
    }
  };

  int64_t tile = blockIdx.x;
  int32_t ia, ib, na, nb;
  typename M::frag ca[S], cb[S], pa[S], pb[S];
  load_idx(tile, ia, ib);
  load_rows(ia, ib, ca, cb);
  load_idx(tile + gridDim.x, na, nb);
#pragma unroll 1
  for (; tile < tiles; tile += gridDim.x) {
    load_rows(na, nb, pa, pb);                       // next tile's rows (entries fetched one round ago)
    load_idx(tile + 2 * (int64_t)gridDim.x, na, nb);  // entries of the tile after next
    const int64_t rowA = tile * F2_ROWS + wave * 32 + r;
#pragma unroll 1
    for (int h = 0; h < nh; ++h) {
      f32x4 acc[2][NTILES];
      // accumulators start at the bias of the channel they are stored to (sc_bias_regs mapping), read from LDS: a
      // global load here would queue behind the rows prefetched for the next tile
#pragma unroll
      for (int t = 0; t < NTILES; ++t) {
        const int gs = TileGroups<NTILES>::gstart(t), G = TileGroups<NTILES>::gsize(t);
        acc[0][t] = *reinterpret_cast<const f32x4*>(bl + h * NT + 16 * gs + 4 * G * g + 4 * (t - gs));
        acc[1][t] = acc[0][t];
      }
      // FULL (c_in a multiple of 32: every width of PT-v3m1) reads the W fragments unconditionally.  With the per-lane guard `col < c_in`
      // every ds_read_b128 sat under its own exec mask with s_waitcnt lgkmcnt(0) behind it: one LDS round trip per two MFMAs, nothing
      // in flight (251 v_mov of zero-initialisation beside 128 MFMAs in the 128-column instance; round 6)
      auto products = [&](auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
        for (int s = 0; s < S; ++s) {
          const int col = s * 32 + g * 8;
          const T* wrow = wl + (h * NT + r) * pitch + col;
#pragma unroll
          for (int t = 0; t < NTILES; ++t) {
            typename M::frag fw;
            if constexpr (FULL) fw = ld_frag<T>(wrow + t * 16 * pitch);
            else {
              fw = M::zero();
              if (col < c_in) fw = ld_frag<T>(wrow + t * 16 * pitch);
            }
            acc[0][t] = M::mma(fw, ca[s], acc[0][t]);
            acc[1][t] = M::mma(fw, cb[s], acc[1][t]);
          }
        }
      };
      if (F2_FULL_PATH && (c_in & 31) == 0) products(std::true_type{}); else products(std::false_type{});
      if constexpr (LDSS) f2_store_rows_lds<T, NTILES>(acc, oslice, out, tile * F2_ROWS + wave * 32, n_out, c_out, n0 + h * NT, r, g, lane);
      else if constexpr (EPI == 0) sc_epilogue<T, NTILES>(acc, nullptr, out, rowA, rowA + 16, n_out, c_out, n0 + h * NT, g);
      else f2_epilogue_ex<T, NTILES, EPI>(acc, out, aux_in, aux_out, rowA, rowA + 16, n_out, c_out, n0 + h * NT, g);
    }
#pragma unroll
    for (int s = 0; s < S; ++s) { ca[s] = pa[s]; cb[s] = pb[s]; }
  }
}

// ---- host side ---------------------------------------------------------------------------------
static inline bool fwd2_supported(int dtype, int kv, int c_in) {
  if (dtype == PTC_F32) return false;
  return kv == 1 ? c_in <= 256 : (c_in <= 128 || c_in % 128 == 0);
}

static inline int conv2_kg(int kv, int c_in, int nt, int waves) {  // kv = number of VIRTUAL table rows, c_in = chunk length
  const int per_k = nt * (c_in + 8) * 2 + waves * 32 * 4;
  int kgmax = F2_MAX_W_BYTES / per_k;
  if (kgmax < 1) kgmax = 1;
  const int groups = (kv + kgmax - 1) / kgmax;
  return (kv + groups - 1) / groups;
}

template <typename T, int NTILES, int WAVES>
static int launch_conv2_w(const void* in, int64_t n_in, const void* w, const float* bias, const int32_t* nbr, int64_t n_out, int kv,
                          int c_in, int c_out, void* out, hipStream_t s) {
  constexpr int NT = NTILES * 16;
  const int cc_len = c_in < 128 ? c_in : 128;
  const int kg = conv2_kg(kv * (c_in / cc_len), cc_len, NT, WAVES);
  const size_t lds = (((size_t)kg * NT * (cc_len + 8) * 2 + 15) & ~(size_t)15) + (size_t)WAVES * kg * 32 * 4;
  auto kern = conv2_kernel<T, NTILES, 4, WAVES>;
  if (lds > 48 * 1024)
    PTC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  dim3 grid((unsigned)ptc_cdiv(n_out, WAVES * 32), (unsigned)(c_out / NT));
  hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), lds, s, (const T*)in, (const T*)w, bias, nbr, n_out, kv, c_in, c_out, kg,
                     (T*)out, (uint32_t)((uint64_t)n_in * c_in * sizeof(T)));
  PTC_CHECK_LAUNCH("conv2_kernel");
  return PTC_OK;
}

template <typename T, int NTILES>
static int launch_fwd2(const void* in, int64_t n_in, const void* w, const float* bias, const int32_t* nbr, int64_t n_out, int kv,
                       int c_in, int c_out, void* out, hipStream_t s, int epi = 0, const void* aux_in = nullptr, void* aux_out = nullptr) {
  constexpr int NT = NTILES * 16;
  if (kv == 1) {
    // column blocks per workgroup: as many as keep W within 64 KB of LDS (two workgroups per CU), see linear2_kernel
    const int nblk = c_out / NT;
    bool lds_store = epi == 0 && c_in <= 64;          // see f2_store_rows_lds
    const size_t slices = lds_store ? 4 * f2_out_slice_bytes(NT) : 0;
    int nh = 1;
    for (int cand = 4; cand >= 2; --cand)
      if (nblk % cand == 0 && (size_t)cand * NT * (c_in + 8) * 2 + (size_t)cand * NT * 4 + 16 + slices <= 64 * 1024) { nh = cand; break; }
    const size_t lds = (((size_t)nh * NT * (c_in + 8) * 2 + 15) & ~(size_t)15) + (size_t)nh * NT * 4 + slices;
    const int64_t tiles = ptc_cdiv(n_out, F2_ROWS);
#ifndef F2_PER_CU
#define F2_PER_CU 4
#endif
    int64_t per_cu = lds > 40 * 1024 ? (F2_PER_CU > 2 ? 2 : F2_PER_CU) : F2_PER_CU;
    if (lds_store) { per_cu = (160 * 1024) / (int64_t)lds; per_cu = per_cu > F2_PER_CU ? F2_PER_CU : (per_cu < 1 ? 1 : per_cu); }
    int64_t gx = 256 * per_cu / (nblk / nh);
    if (gx > tiles) gx = tiles;
    if (gx < 1) gx = 1;
    dim3 grid((unsigned)gx, (unsigned)(nblk / nh));
    const int S = (c_in + 31) / 32;
#define L2_LAUNCH(SS, EE) L2_LAUNCH_X(SS, EE, false)
#define L2_LAUNCH_X(SS, EE, LL)                                                                                         \
  {                                                                                                                     \
    auto kern = linear2_kernel<T, NTILES, SS, EE, LL>;                                                                      \
    if (lds > 48 * 1024)                                                                                                \
      PTC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, (const T*)in, (const T*)w, bias, nbr, n_out, c_in, c_out, nh, (T*)out, \
                       (const T*)aux_in, (T*)aux_out, (uint32_t)((uint64_t)n_in * c_in * sizeof(T)));                   \
  }
#define L2_CASE(SS)                                                                                                     \
  case SS: {                                                                                                            \
    if constexpr (NTILES == 8) {       /* the GELU epilogues exist for 128-wide output tiles: the MLP hidden width 4 C, C % 32 == 0 */ \
      if (epi == 1) L2_LAUNCH(SS, 1) else if (epi == 2) L2_LAUNCH(SS, 2) else L2_LAUNCH(SS, 0)                            \
    } else L2_LAUNCH(SS, 0)                                                                                             \
  } break;
    if (lds_store && S == 1) L2_LAUNCH_X(1, 0, true)
    else if (lds_store && S == 2) L2_LAUNCH_X(2, 0, true)
    else switch (S) {
      L2_CASE(1) L2_CASE(2) L2_CASE(3) L2_CASE(4) L2_CASE(5) L2_CASE(6) L2_CASE(7) L2_CASE(8)
      default: ptc_set_error("linear2: c_in=%d unsupported", c_in); return PTC_EUNSUPPORTED;
    }
#undef L2_CASE
#undef L2_LAUNCH
#undef L2_LAUNCH_X
    PTC_CHECK_LAUNCH("linear2_kernel");
    return PTC_OK;
  }
  // Deep stages (few rows, wide channels) leave a 128-row grid under-filled (N = 11400, C = 256: 180
  // workgroups on 256 CUs, one latency-bound chain per CU): there 64-row workgroups double the number
  // of independent chains.  (8-wave / 256-row workgroups measured slower everywhere: r01 session s7.)
  const int64_t wg128 = ptc_cdiv(n_out, 128) * (c_out / NT);
  if (wg128 < 512) return launch_conv2_w<T, NTILES, 2>(in, n_in, w, bias, nbr, n_out, kv, c_in, c_out, out, s);
  return launch_conv2_w<T, NTILES, 4>(in, n_in, w, bias, nbr, n_out, kv, c_in, c_out, out, s);
}

template <typename T>
static int dispatch_fwd2(const void* in, int64_t n_in, const void* w, const float* bias, const int32_t* nbr, int64_t n_out, int kv,
                         int c_in, int c_out, void* out, hipStream_t s, int epi = 0, const void* aux_in = nullptr, void* aux_out = nullptr) {
  // Linear layers of the deep stages (c_in >= 128, fewer 128-row x 128-column workgroups than CUs): 64-column workgroups -- twice as many,
  // each staging half as much of W before its ONE row tile (round 6; PTC_L2_NARROW=0: the 128-column form, timing A/B)
  static const bool narrow_on = [] { const char* e = getenv("PTC_L2_NARROW"); return !(e && e[0] == '0'); }();
  const bool narrow = narrow_on && kv == 1 && epi == 0 && c_in >= 128 && ptc_cdiv(n_out, F2_ROWS) * (c_out / 128) < 256;
  if (c_out % 128 == 0 && !narrow) return launch_fwd2<T, 8>(in, n_in, w, bias, nbr, n_out, kv, c_in, c_out, out, s, epi, aux_in, aux_out);
  if (c_out % 96 == 0 && !narrow) return launch_fwd2<T, 6>(in, n_in, w, bias, nbr, n_out, kv, c_in, c_out, out, s, epi, aux_in, aux_out);
  if (c_out % 64 == 0) return launch_fwd2<T, 4>(in, n_in, w, bias, nbr, n_out, kv, c_in, c_out, out, s, epi, aux_in, aux_out);
  if (c_out % 48 == 0) return launch_fwd2<T, 3>(in, n_in, w, bias, nbr, n_out, kv, c_in, c_out, out, s, epi, aux_in, aux_out);
  if (c_out % 32 == 0) return launch_fwd2<T, 2>(in, n_in, w, bias, nbr, n_out, kv, c_in, c_out, out, s, epi, aux_in, aux_out);
  return launch_fwd2<T, 1>(in, n_in, w, bias, nbr, n_out, kv, c_in, c_out, out, s, epi, aux_in, aux_out);
}
