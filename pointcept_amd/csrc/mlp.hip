// mlp.hip -- PT-v3m1's MLP (fc1 -> GELU -> fc2, ptv3m1:225-248) and the residual joint behind it (ptv3m1:334-337) as ONE kernel per
// direction, for the large-N stages (C = 32 | 64, hidden = 4 C = 128 | 256).  Round 6 (VERDICT r5 next 3 (i) + (iii)).
//
// As separate GEMMs the hidden tensor crosses HBM six times per Block and step: forward h and GELU(h) out of fc1 (2 x N x 4C x 2 B),
// GELU(h) into fc2; backward h into fc2's input gradient, dh out of it, dh into fc1's input gradient, GELU(h) and dh into the two weight
// gradients -- 2.5 GB of a stage-0 Block's backward and 1.3 GB of its forward at N = 819200, C = 64, for 0.1 GB of operands that matter
// (profiles/r06_a_step_traffic.txt: the Linear + weight-gradient families move 51 of the step's 93 GB).  Here the hidden tile never
// leaves the CU:
//   forward  (mlp_fwd_kernel, 4 waves x 32 rows, W1 and W2 resident in LDS): per 64-channel chunk of the hidden width the fc1 accumulators
//            go through bias, the rounding to the feature dtype and GELU IN REGISTERS and become the B operand of fc2's MFMAs directly --
//            fc1's weight rows are permuted in LDS so that a lane ends up with 8 consecutive hidden channels of its row, which is exactly
//            the fragment the second product wants; same operands, same contraction order as linear2_kernel (EPI 1) followed by
//            linear2_joint_kernel: the output is BIT-IDENTICAL to the two-launch form.  Nothing of the hidden width is written: the backward
//            recomputes it.
//   backward (mlp_bwd_kernel, 8 waves x 16 rows per 128-row tile, W1 and W2^T resident in LDS): per chunk h = y W1^T + b1 is recomputed (same
//            bits as the forward), dA = dm W2, dh = dA * GELU'(h), dy += dh W1 -- W1^T fragments are read TRANSPOSED out of the W1 image
//            (ds_read_b64_tr_b16), no second copy --; GELU(h) and dh of the tile's 128 rows meet in two LDS images [128][64] and are read
//            back transposed as the operands of the weight gradients dW2 += dm^T GELU(h), dW1 += dh^T y, whose 2 x 4C x C accumulators are
//            spread over the workgroup's 8 waves (64 registers per lane at C = 64) for the workgroup's whole life; db1 / db2 are column sums
//            of the same images.  Per workgroup one partial of every parameter gradient; the library's deterministic reduction adds them.
// HBM bytes per row: forward C x (2 + 4 + 4 + 2), backward 3 x 2 C.  What bounds the kernels is the GELU arithmetic (ptc_gelu: 15 / 20 vector
// instructions per hidden value) beside ~1 MFMA per 13 hidden values; see DESIGN 4.2.
#include "ptc_common.h"
#include "spconv_internal.h"
#include "mma.h"
#include "ln_common.h"

#define MLP_ROWS 128
#define MLP_OUT_ROWS 16
#ifndef MLP_ABLATE
#define MLP_ABLATE 0         // timing probes (python -m pointcept_amd.build --variant d_MLP_ABLATE_1): 1 = no GELU arithmetic in the forward
#endif

template <typename T> __device__ __forceinline__ uint32_t mlp_pack2(float lo, float hi);
template <> __device__ __forceinline__ uint32_t mlp_pack2<bf16_t>(float lo, float hi) { return ptc_pack_bf16x2(lo, hi); }
template <> __device__ __forceinline__ uint32_t mlp_pack2<f16_t>(float lo, float hi) {
  const _Float16 a = (_Float16)lo, b = (_Float16)hi;
  return (uint32_t)(*reinterpret_cast<const uint16_t*>(&a)) | ((uint32_t)(*reinterpret_cast<const uint16_t*>(&b)) << 16);
}
template <typename T> __device__ __forceinline__ float mlp_round(float v) { return ptc_to_float(ptc_from_float<T>(v)); }

__device__ __forceinline__ void mlp_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// LDS row of hidden channel n in the W1 / W2^T images: inside a 64-channel chunk, MFMA tile t = 2 ks + p (ks = 32-channel step of the second
// product, p = half of an 8-channel run), A-row i = 4 gq + e  <->  channel 64 c + 32 ks + 8 gq + 4 p + e.  After the fc1 MFMAs lane (row, g)
// then holds channels 32 ks + 8 g + {0..3} (tile 2 ks) and + {4..7} (tile 2 ks + 1): the 8 contraction values of the natural fragment.
__host__ __device__ __forceinline__ int mlp_hidden_row(int n) {
  const int c = n >> 6, l = n & 63, ks = l >> 5, gq = (l & 31) >> 3, p = (l >> 2) & 1, e = l & 3;
  return 64 * c + 16 * (2 * ks + p) + 4 * gq + e;
}
// LDS row of output channel n of a C-wide product whose G = C / 16 tiles form one store group: tile tt, A-row 4 gq + e <-> channel 4 G gq + 4 tt + e
// (lane (row, g) ends up with the 4 G consecutive channels 4 G g ..: spconv.hip's TileGroups for one group)
template <int G> __host__ __device__ __forceinline__ int mlp_out_row(int n) {
  const int gq = n / (4 * G), rem = n - gq * 4 * G;
  return 16 * (rem >> 2) + 4 * gq + (rem & 3);
}

static size_t mlp_slice_bytes(int c) { return (size_t)MLP_OUT_ROWS * (c * 2 + 16); }
static size_t mlp_fwd_lds(int c, int waves) {
  const int hid = 4 * c;
  return (size_t)hid * (c + 8) * 2 + (size_t)c * (hid + 8) * 2 + (size_t)hid * 4 + (size_t)c * 4 + (size_t)waves * mlp_slice_bytes(c);
}

struct MlpOut {
  const float* a;          // residual stream [n, C] fp32, or NULL: plain output
  const float* row_scale;  // DropPath row factors [n] or NULL
  float* z;                // [n, C] fp32 = a + row_scale * m       (a != NULL)
  void* y;                 // [n, C] feature dtype: cast(z) (may be NULL), or m itself when a == NULL
};

// ------------------------------------------------------------------------------------------------------------------------------- forward
// WAVES x 32 rows per workgroup step.  16 waves (one workgroup per CU, four waves per SIMD, <= 128 registers) instead of 2 x 4: the chain
// fc1 MFMAs -> GELU -> fc2 MFMAs -> LDS round trip -> residual loads is one long dependency per wave, and two waves per SIMD left it exposed
// (819200 x 64: 200 us with 2 x 4 waves per CU, 295 us with 4: profiles/r06_d_mlp_sweeps.txt).
template <typename T, int C, int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
mlp_fwd_kernel(const T* __restrict__ x, const T* __restrict__ w1, const float* __restrict__ b1, const T* __restrict__ w2,
               const float* __restrict__ b2, int64_t n, uint32_t x_bytes, MlpOut J) {
  using M = Mma<T>;
  static_assert(sizeof(T) == 2, "16-bit features only");
  constexpr int HID = 4 * C, NCH = HID / 64, S = C / 32, G = C / 16, P1 = C + 8, P2 = HID + 8, LPR = C / LN_VEC, RB = C * 2, P = RB + 16;
  const __amdgpu_buffer_rsrc_t x_buf = ptc_buf(x, x_bytes);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* wl1 = reinterpret_cast<T*>(smem);                                    // [HID (mlp_hidden_row)][P1]
  T* wl2 = wl1 + HID * P1;                                                // [C (mlp_out_row)][P2]
  float* bl1 = reinterpret_cast<float*>(wl2 + C * P2);                    // [HID] natural order
  float* bl2 = bl1 + HID;                                                 // [C] natural order
  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  unsigned char* slice = reinterpret_cast<unsigned char*>(bl2 + C) + wave * (MLP_OUT_ROWS * P);
  const int r = lane & 15, g = lane >> 4;
  constexpr int TROWS = WAVES * 32;
  for (int q = threadIdx.x; q < HID * (C / 8); q += WAVES * 64) {
    const int nn = q / (C / 8), cc = q - nn * (C / 8);
    *reinterpret_cast<uint4*>(wl1 + mlp_hidden_row(nn) * P1 + cc * 8) = *reinterpret_cast<const uint4*>(w1 + (int64_t)nn * C + cc * 8);
  }
  for (int q = threadIdx.x; q < C * (HID / 8); q += WAVES * 64) {
    const int nn = q / (HID / 8), cc = q - nn * (HID / 8);
    *reinterpret_cast<uint4*>(wl2 + mlp_out_row<G>(nn) * P2 + cc * 8) = *reinterpret_cast<const uint4*>(w2 + (int64_t)nn * HID + cc * 8);
  }
  for (int q = threadIdx.x; q < HID; q += WAVES * 64) bl1[q] = b1 ? b1[q] : 0.f;
  for (int q = threadIdx.x; q < C; q += WAVES * 64) bl2[q] = b2 ? b2[q] : 0.f;
  __syncthreads();

  const int64_t tiles = (n + TROWS - 1) / TROWS;
  auto load_rows = [&](int64_t tile, typename M::frag (&fa)[S], typename M::frag (&fb)[S]) {
    const int64_t rowA = tile * TROWS + wave * 32 + r, rowB = rowA + 16;
    const bool okA = tile < tiles && rowA < n, okB = tile < tiles && rowB < n;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const uint32_t col = (uint32_t)(s * 32 + g * 8);
      fa[s] = ld_frag_buf<T>(x_buf, okA ? ((uint32_t)rowA * (uint32_t)C + col) * 2u : PTC_BUF_OOB);
      fb[s] = ld_frag_buf<T>(x_buf, okB ? ((uint32_t)rowB * (uint32_t)C + col) * 2u : PTC_BUF_OOB);
    }
  };
  const int slot = lane % LPR;
  int64_t tile = blockIdx.x;
  typename M::frag ca[S], cb[S], pa[S], pb[S];
  load_rows(tile, ca, cb);
#pragma unroll 1
  for (; tile < tiles; tile += gridDim.x) {
    load_rows(tile + gridDim.x, pa, pb);                    // the next tile's rows are in flight under this tile's products
    f32x4 oacc[2][G];
#pragma unroll
    for (int tt = 0; tt < G; ++tt) {
      oacc[0][tt] = *reinterpret_cast<const f32x4*>(bl2 + 4 * G * g + 4 * tt);
      oacc[1][tt] = oacc[0][tt];
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      // fc1 for 64 hidden channels: tile t = 2 ks + p of the chunk
      f32x4 hacc[2][4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        hacc[0][t] = *reinterpret_cast<const f32x4*>(bl1 + 64 * c + 32 * (t >> 1) + 8 * g + 4 * (t & 1));
        hacc[1][t] = hacc[0][t];
      }
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const T* wrow = wl1 + (64 * c + r) * P1 + s * 32 + g * 8;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const typename M::frag fw = ld_frag<T>(wrow + t * 16 * P1);
          hacc[0][t] = M::mma(fw, ca[s], hacc[0][t]);
          hacc[1][t] = M::mma(fw, cb[s], hacc[1][t]);
        }
      }
      // GELU of the value the reference's activation sees (h rounded to the feature dtype), rounded again as the operand of fc2
      typename M::frag act[2][2];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          uint32_t pk[4];
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const f32x4 v = hacc[hf][2 * ks + p];
#if MLP_ABLATE == 1        // timing probe only: no GELU arithmetic
            pk[2 * p] = mlp_pack2<T>(v[0], v[1]);
            pk[2 * p + 1] = mlp_pack2<T>(v[2], v[3]);
#else
            pk[2 * p] = mlp_pack2<T>(ptc_gelu(mlp_round<T>(v[0])), ptc_gelu(mlp_round<T>(v[1])));
            pk[2 * p + 1] = mlp_pack2<T>(ptc_gelu(mlp_round<T>(v[2])), ptc_gelu(mlp_round<T>(v[3])));
#endif
          }
          const uint4 u = make_uint4(pk[0], pk[1], pk[2], pk[3]);
          __builtin_memcpy(&act[hf][ks], &u, 16);
        }
      // fc2: contraction steps 2 c, 2 c + 1 of linear2's s = 0 .. HID / 32 - 1, in that order
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const T* wrow = wl2 + r * P2 + 64 * c + 32 * ks + g * 8;
#pragma unroll
        for (int tt = 0; tt < G; ++tt) {
          const typename M::frag fw = ld_frag<T>(wrow + tt * 16 * P2);
          oacc[0][tt] = M::mma(fw, act[0][ks], oacc[0][tt]);
          oacc[1][tt] = M::mma(fw, act[1][ks], oacc[1][tt]);
        }
      }
    }
    // ---- epilogue: linear2_joint_kernel's, statement for statement (the 16-row halves through the wave's LDS slice, rows read back by
    // C / 8 consecutive lanes): z = a + row_scale * m, y = cast(z); without a residual the rows of m leave as they are
    const int64_t row0 = tile * TROWS + wave * 32;
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) {
      uint32_t pk[2 * G];
#pragma unroll
      for (int tt = 0; tt < G; ++tt) {
        const f32x4 v = oacc[sx][tt];
        pk[2 * tt] = mlp_pack2<T>(v[0], v[1]);
        pk[2 * tt + 1] = mlp_pack2<T>(v[2], v[3]);
      }
      unsigned char* dst = slice + r * P + (4 * G * g) * 2;
      reinterpret_cast<uint4*>(dst)[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
      if constexpr (G == 4) reinterpret_cast<uint4*>(dst)[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
      mlp_wave_sync();
#pragma unroll
      for (int it = 0; it < (MLP_OUT_ROWS * LPR + 63) / 64; ++it) {
        const int q = it * 64 + lane, row = q / LPR;
        const int64_t grow = row0 + sx * 16 + row;
        const bool ok = row < MLP_OUT_ROWS && grow < n;
        float v[LN_VEC], rr[LN_VEC];
        ln_load8<T>(reinterpret_cast<const T*>(slice + (row < MLP_OUT_ROWS ? row : 0) * P) + slot * LN_VEC, v);
        if (J.a) {
          if (ok) ln_load8<float>(J.a + grow * C + slot * LN_VEC, rr);
          else {
#pragma unroll
            for (int i = 0; i < LN_VEC; ++i) rr[i] = 0.f;
          }
          const float sc = (J.row_scale && ok) ? J.row_scale[grow] : 1.f;
#pragma unroll
          for (int i = 0; i < LN_VEC; ++i) rr[i] = fmaf(sc, v[i], rr[i]);
          if (ok) ln_store8<float>(J.z + grow * C + slot * LN_VEC, rr);
          if (ok && J.y) ln_store8<T>(reinterpret_cast<T*>(J.y) + grow * C + slot * LN_VEC, rr);
        } else if (ok) {
          ln_store8<T>(reinterpret_cast<T*>(J.y) + grow * C + slot * LN_VEC, v);          // v holds T-representable values: exact
        }
      }
      mlp_wave_sync();
    }
#pragma unroll
    for (int s = 0; s < S; ++s) { ca[s] = pa[s]; cb[s] = pb[s]; }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------ backward
#define MLPB_THREADS 512
#define MLPB_WAVES 8
#ifndef MLPB_PF_CHUNK
#define MLPB_PF_CHUNK 0
#endif
#ifndef MLPB_SCHED_FENCE
#define MLPB_SCHED_FENCE 0
#endif
#ifndef MLPB_HOLD_T
#define MLPB_HOLD_T 0        // 1: the transposed dm / y fragments of a tile stay in registers (32 more: spills at C = 64); 0: re-read per chunk
#endif
typedef short mlp_s16x4 __attribute__((ext_vector_type(4)));

// One MFMA fragment read TRANSPOSED out of a row-major 16-bit image: lane (j = lane & 15, g = lane >> 4) receives column col0 + j (or, with
// `cstride` = 4 G, the permuted column 4 G (j >> 2) + (j & 3) + col0) of rows row0 + 4 g + {0..3} and row0 + 16 + 4 g + {0..3}: the 8
// contraction values of K-group g of a 32-row step.  Each lane supplies the address of one 8-byte piece (4 columns) of its 16-lane group's
// [4 rows][16 columns] block.
template <typename T>
__device__ __forceinline__ typename Mma<T>::frag mlp_tr_frag(const unsigned char* img, int pitch_bytes, int row0, int col0, int cstride, int lane) {
  const int lp = lane & 15, g = lane >> 4;
  const unsigned char* p = img + (row0 + 4 * g + (lp >> 2)) * pitch_bytes + (col0 + cstride * (lp & 3)) * 2;
  const mlp_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) mlp_s16x4*)(p));
  const mlp_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) mlp_s16x4*)(p + 16 * pitch_bytes));
  const s16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  typename Mma<T>::frag out;
  __builtin_memcpy(&out, &f, sizeof(out));
  return out;
}

template <int C> struct MlpBwdLds {
  static constexpr int HID = 4 * C;
  static constexpr int PW = C + 8;            // W1 / W2^T image pitch (elements): conflict-free ds_read_b128 fragments
  static constexpr int PD = C + 16;           // dm / y image pitch: + 32 B -- 8 consecutive rows x 32 B cover the 64 banks once (transposed reads)
  static constexpr int PA = 64 + 16;          // GELU(h) / dh chunk images [128][64]
  static constexpr size_t w_bytes = (size_t)HID * PW * 2;
  static constexpr size_t d_bytes = (size_t)MLP_ROWS * PD * 2;
  static constexpr size_t a_bytes = (size_t)MLP_ROWS * PA * 2;
  static constexpr size_t total = 2 * w_bytes + 2 * d_bytes + 2 * a_bytes + (size_t)HID * 4;
};

// partial sums of one workgroup: [dW1 HID x C][dW2 C x HID][db1 HID][db2 C] live in four arrays [gridDim.x][count] (the layout
// ptc_wgrad_reduce_jobs sums)
struct MlpPartials { float* w1; float* w2; float* b1; float* b2; };

template <typename T, int C>
__global__ void __launch_bounds__(MLPB_THREADS, 2)
mlp_bwd_kernel(const T* __restrict__ dm, const T* __restrict__ x, const T* __restrict__ w1, const float* __restrict__ b1,
               const T* __restrict__ w2t, int64_t n, uint32_t row_bytes, T* __restrict__ dx, MlpPartials part) {
  using M = Mma<T>;
  using L = MlpBwdLds<C>;
  static_assert(sizeof(T) == 2, "16-bit features only");
  constexpr int HID = 4 * C, NCH = HID / 64, S = C / 32, G = C / 16, KPW = G / 2, PW = L::PW, PD = L::PD, PA = L::PA, VPR = C / 8;
  constexpr int PIECES = MLP_ROWS * VPR / MLPB_THREADS;             // 16-byte pieces of a 128-row tile per thread and tensor: 2 (C = 64) | 1
  const __amdgpu_buffer_rsrc_t dm_buf = ptc_buf(dm, row_bytes), x_buf = ptc_buf(x, row_bytes);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* wl1 = reinterpret_cast<T*>(smem);                                   // [HID (mlp_hidden_row)][PW]       W1
  T* wl2 = wl1 + HID * PW;                                               // [HID (mlp_hidden_row)][PW]       W2^T
  unsigned char* DM = smem + 2 * L::w_bytes;                             // [128][PD]
  unsigned char* Y = DM + L::d_bytes;
  unsigned char* ACT = Y + L::d_bytes;                                   // [128][PA]
  unsigned char* DH = ACT + L::a_bytes;
  float* bl1 = reinterpret_cast<float*>(DH + L::a_bytes);                // [HID]
  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  for (int q = threadIdx.x; q < HID * VPR; q += MLPB_THREADS) {
    const int nn = q / VPR, cc = q - nn * VPR;
    const int row = mlp_hidden_row(nn);
    *reinterpret_cast<uint4*>(wl1 + row * PW + cc * 8) = *reinterpret_cast<const uint4*>(w1 + (int64_t)nn * C + cc * 8);
    *reinterpret_cast<uint4*>(wl2 + row * PW + cc * 8) = *reinterpret_cast<const uint4*>(w2t + (int64_t)nn * C + cc * 8);
  }
  for (int q = threadIdx.x; q < HID; q += MLPB_THREADS) bl1[q] = b1 ? b1[q] : 0.f;

  // this wave's share of the weight gradients: for every chunk c, the dW2 tiles (output tile q, hidden tiles kh KPW ..) and the dW1 tiles
  // (the same hidden tiles, input tile q)
  const int wq = wave % G, kh = wave / G;
  f32x4 accW2[NCH][KPW], accW1[NCH][KPW];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int j = 0; j < KPW; ++j) { accW2[c][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; accW1[c][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  float accb1[NCH], accb2 = 0.f;            // column sums of dh (thread: hidden channel t & 63 of each chunk, rows 16 (t >> 6) ..) and of dm
#pragma unroll
  for (int c = 0; c < NCH; ++c) accb1[c] = 0.f;

  const int64_t tiles = (n + MLP_ROWS - 1) / MLP_ROWS;
  auto load_tile = [&](int64_t tile, uint4 (&pd)[PIECES], uint4 (&px)[PIECES]) {
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int q = i * MLPB_THREADS + threadIdx.x, row = q / VPR, pc = q - row * VPR;
      const int64_t grow = tile * MLP_ROWS + row;
      const uint32_t off = (tile < tiles && grow < n) ? ((uint32_t)grow * (uint32_t)C + (uint32_t)pc * 8u) * 2u : PTC_BUF_OOB;
      pd[i] = ptc_buf_load16(dm_buf, off);
      px[i] = ptc_buf_load16(x_buf, off);
    }
  };
  uint4 pd[PIECES], px[PIECES];
  int64_t tile = blockIdx.x;
  load_tile(tile, pd, px);
#pragma unroll 1
  for (; tile < tiles; tile += gridDim.x) {
    // (every wave has passed the barriers of the previous tile's chunk loop since it last read DM / Y: the images are free)
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int q = i * MLPB_THREADS + threadIdx.x, row = q / VPR, pc = q - row * VPR;
      *reinterpret_cast<uint4*>(DM + row * PD * 2 + pc * 16) = pd[i];
      *reinterpret_cast<uint4*>(Y + row * PD * 2 + pc * 16) = px[i];
    }
    __syncthreads();
    // operands of this wave for the whole tile
    typename M::frag yb[S], db[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      yb[s] = *reinterpret_cast<const typename M::frag*>(Y + (16 * wave + r) * PD * 2 + (32 * s + 8 * g) * 2);
      db[s] = *reinterpret_cast<const typename M::frag*>(DM + (16 * wave + r) * PD * 2 + (32 * s + 8 * g) * 2);
    }
#if MLPB_HOLD_T
    typename M::frag dmT[4], yT[4];           // the transposed dm / y operands of the weight gradients, held for the whole tile (32 registers)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      dmT[kk] = mlp_tr_frag<T>(DM, PD * 2, 32 * kk, 16 * wq, 4, lane);
      yT[kk] = mlp_tr_frag<T>(Y, PD * 2, 32 * kk, 16 * wq, 4, lane);
    }
#endif
    {  // db2: column sums of dm (rounded values, as the split kernels' bias gradient sums them)
      const int ch = threadIdx.x % C, rg = threadIdx.x / C;
      constexpr int RPG = MLP_ROWS / (MLPB_THREADS / C);
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < RPG; ++i) sum += ptc_to_float(*reinterpret_cast<const T*>(DM + (rg * RPG + i) * PD * 2 + ch * 2));
      accb2 += sum;
    }
    f32x4 yacc[G];
#pragma unroll
    for (int tt = 0; tt < G; ++tt) yacc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      // the next tile's rows go out in the second half of this one (their 16 registers would not fit beside the first chunks' operands)
      if (c == MLPB_PF_CHUNK) load_tile(tile + gridDim.x, pd, px);
#if MLPB_SCHED_FENCE
      __builtin_amdgcn_sched_barrier(0);
#endif
      // h (recomputed: the forward's bits) and dA = dm W2 for this wave's 16 rows x 64 hidden channels
      f32x4 hacc[4], dacc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        hacc[t] = *reinterpret_cast<const f32x4*>(bl1 + 64 * c + 32 * (t >> 1) + 8 * g + 4 * (t & 1));
        dacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const T* w1row = wl1 + (64 * c + r) * PW + s * 32 + g * 8;
        const T* w2row = wl2 + (64 * c + r) * PW + s * 32 + g * 8;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          hacc[t] = M::mma(ld_frag<T>(w1row + t * 16 * PW), yb[s], hacc[t]);
          dacc[t] = M::mma(ld_frag<T>(w2row + t * 16 * PW), db[s], dacc[t]);
        }
      }
      typename M::frag aF[2], dF[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        uint32_t pa_[4], pd_[4];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const f32x4 hv = hacc[2 * ks + p], dv = dacc[2 * ks + p];
          float a4[4], d4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float hr = mlp_round<T>(hv[e]);
            a4[e] = ptc_gelu(hr);
            d4[e] = dv[e] * ptc_gelu_grad(hr);
          }
          pa_[2 * p] = mlp_pack2<T>(a4[0], a4[1]); pa_[2 * p + 1] = mlp_pack2<T>(a4[2], a4[3]);
          pd_[2 * p] = mlp_pack2<T>(d4[0], d4[1]); pd_[2 * p + 1] = mlp_pack2<T>(d4[2], d4[3]);
        }
        const uint4 ua = make_uint4(pa_[0], pa_[1], pa_[2], pa_[3]), ud = make_uint4(pd_[0], pd_[1], pd_[2], pd_[3]);
        __builtin_memcpy(&aF[ks], &ua, 16);
        __builtin_memcpy(&dF[ks], &ud, 16);
      }
      // dy += dh W1: A = W1^T[output channel][hidden], read transposed out of the W1 image (rows = hidden channels in mlp_hidden_row order:
      // K-group g of step ks sits in LDS rows 64 c + 32 ks + 4 g .. (first four values) and + 16 (last four))
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int tt = 0; tt < G; ++tt)
          yacc[tt] = M::mma(mlp_tr_frag<T>(reinterpret_cast<const unsigned char*>(wl1), PW * 2, 64 * c + 32 * ks, 4 * tt, 4 * G, lane), dF[ks], yacc[tt]);
      __syncthreads();                    // every wave is done with the previous chunk's images
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        *reinterpret_cast<typename M::frag*>(ACT + (16 * wave + r) * PA * 2 + (32 * ks + 8 * g) * 2) = aF[ks];
        *reinterpret_cast<typename M::frag*>(DH + (16 * wave + r) * PA * 2 + (32 * ks + 8 * g) * 2) = dF[ks];
      }
      __syncthreads();
      {  // db1
        const int ch = threadIdx.x & 63, rg = threadIdx.x >> 6;
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) sum += ptc_to_float(*reinterpret_cast<const T*>(DH + (rg * 16 + i) * PA * 2 + ch * 2));
        accb1[c] += sum;
      }
      // weight gradients of the chunk: contraction over the tile's 128 rows in four 32-row steps
#pragma unroll
      for (int j = 0; j < KPW; ++j) {
        const int kt = kh * KPW + j;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#if MLPB_HOLD_T
          const typename M::frag fdm = dmT[kk], fy = yT[kk];
#else
          const typename M::frag fdm = mlp_tr_frag<T>(DM, PD * 2, 32 * kk, 16 * wq, 4, lane), fy = mlp_tr_frag<T>(Y, PD * 2, 32 * kk, 16 * wq, 4, lane);
#endif
          accW2[c][j] = M::mma(fdm, mlp_tr_frag<T>(ACT, PA * 2, 32 * kk, 16 * kt, 4, lane), accW2[c][j]);
          accW1[c][j] = M::mma(mlp_tr_frag<T>(DH, PA * 2, 32 * kk, 16 * kt, 4, lane), fy, accW1[c][j]);
        }
      }
    }
    // dy rows: lane (row r, g) holds the 4 G consecutive channels 4 G g ..
    {
      const int64_t grow = tile * MLP_ROWS + 16 * wave + r;
      if (grow < n) {
        uint32_t pk[2 * G];
#pragma unroll
        for (int tt = 0; tt < G; ++tt) {
          pk[2 * tt] = mlp_pack2<T>(yacc[tt][0], yacc[tt][1]);
          pk[2 * tt + 1] = mlp_pack2<T>(yacc[tt][2], yacc[tt][3]);
        }
        uint4* dst = reinterpret_cast<uint4*>(dx + grow * C + 4 * G * g);
        dst[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        if constexpr (G == 4) dst[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
      }
    }
  }

  // ---- this workgroup's partials: every weight-gradient tile is owned by exactly one wave ---------------------------------------------
  float* pw1 = part.w1 + (int64_t)blockIdx.x * HID * C;
  float* pw2 = part.w2 + (int64_t)blockIdx.x * HID * C;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int j = 0; j < KPW; ++j) {
      const int k0 = 64 * c + 16 * (kh * KPW + j);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pw2[(int64_t)(16 * wq + 4 * g + e) * HID + k0 + r] = accW2[c][j][e];          // D[i = output channel][j = hidden channel]
        pw1[(int64_t)(k0 + 4 * g + e) * C + 16 * wq + r] = accW1[c][j][e];            // D[i = hidden channel][j = input channel]
      }
    }
  __syncthreads();
  float* red = reinterpret_cast<float*>(ACT);                 // [8][64] per chunk, then [512 / C][C]
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    red[(threadIdx.x >> 6) * 64 + (threadIdx.x & 63)] = accb1[c];
    __syncthreads();
    if (threadIdx.x < 64) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < MLPB_WAVES; ++i) s += red[i * 64 + threadIdx.x];
      part.b1[(int64_t)blockIdx.x * HID + 64 * c + threadIdx.x] = s;
    }
    __syncthreads();
  }
  red[threadIdx.x] = accb2;                                   // [rg][ch], ch = t % C
  __syncthreads();
  if (threadIdx.x < C) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MLPB_THREADS / C; ++i) s += red[i * C + threadIdx.x];
    part.b2[(int64_t)blockIdx.x * C + threadIdx.x] = s;
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------------------------
extern "C" int ptc_mlp_supported(int c, int dtype) { return (c == 32 || c == 64) && (dtype == PTC_BF16 || dtype == PTC_F16) ? 1 : 0; }

static int mlp_grid_bwd(int64_t n) {
  const int64_t tiles = ptc_cdiv(n, MLP_ROWS);
  int g = 256;                                               // one 8-wave workgroup per CU (LDS), MI355X: 256 CUs
  if (const char* e = getenv("PTC_MLP_BWD_WGS")) { const int v = atoi(e); if (v > 0) g = v; }     // (host emulation / sweeps)
  return (int)(tiles < g ? (tiles < 1 ? 1 : tiles) : g);
}

extern "C" int ptc_mlp_fwd(const void* x, int64_t n, int c, int dtype, const void* w1, const float* b1, const void* w2, const float* b2,
                           const float* a, const float* row_scale, float* z, void* y, ptc_stream_t stream) {
  PTC_REQUIRE(n >= 0 && ptc_mlp_supported(c, dtype), PTC_EUNSUPPORTED, "ptc_mlp_fwd: c=%d dtype=%d (C = 32 | 64, 16-bit features)", c, dtype);
  PTC_REQUIRE((uint64_t)n * (uint64_t)c * 2 <= PTC_BUF_MAX_BYTES, PTC_EUNSUPPORTED, "ptc_mlp_fwd: input of 2 GiB or more");
  if (n == 0) return PTC_OK;
  PTC_REQUIRE(x && w1 && w2 && (a ? z != nullptr : y != nullptr), PTC_EINVAL, "ptc_mlp_fwd: null buffer");
  PTC_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)w1 % 16 == 0) && ((uintptr_t)w2 % 16 == 0) && ((uintptr_t)a % 16 == 0) && ((uintptr_t)z % 16 == 0) &&
              ((uintptr_t)y % 16 == 0), PTC_EINVAL, "ptc_mlp_fwd: buffers must be 16-byte aligned");
  int waves = 4;            // (16-wave workgroups -- four waves per SIMD -- measured equal to 2 x 4 at C = 64, profiles/r06_d_mlp_sweeps.txt: kept as an instance)
  if (const char* e = getenv("PTC_MLP_FWD_WAVES")) { const int v = atoi(e); if (v == 4 || v == 16) waves = v; }
  const size_t lds = mlp_fwd_lds(c, waves);
  const int64_t tiles = ptc_cdiv(n, (int64_t)waves * 32);
  int64_t per_cu = (160 * 1024) / (int64_t)lds;          // workgroups a CU holds by LDS: 1 (16 waves) | 2 (C = 64) | 4+ (C = 32: 81 -> 72 us at N = 819200)
  per_cu = per_cu > 4 ? 4 : (per_cu < 1 ? 1 : per_cu);
  int64_t gx = 256 * per_cu;
  if (const char* e = getenv("PTC_MLP_FWD_WGS")) { const int v = atoi(e); if (v > 0) gx = v; }
  if (gx > tiles) gx = tiles;
  const MlpOut J{a, row_scale, z, y};
#define MLP_FWD(T, CC, WW)                                                                                                      \
  {                                                                                                                             \
    auto kern = mlp_fwd_kernel<T, CC, WW>;                                                                                      \
    PTC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));    \
    hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(WW * 64), lds, (hipStream_t)stream, (const T*)x, (const T*)w1, b1, (const T*)w2, b2, n, \
                       (uint32_t)((uint64_t)n * c * 2), J);                                                                     \
  }
#define MLP_FWD_W(T, CC) { if (waves == 16) MLP_FWD(T, CC, 16) else MLP_FWD(T, CC, 4) }
  if (dtype == PTC_BF16) { if (c == 64) MLP_FWD_W(bf16_t, 64) else MLP_FWD_W(bf16_t, 32) }
  else { if (c == 64) MLP_FWD_W(f16_t, 64) else MLP_FWD_W(f16_t, 32) }
#undef MLP_FWD_W
#undef MLP_FWD
  PTC_CHECK_LAUNCH("mlp_fwd_kernel");
  return PTC_OK;
}

// workspace: partials [grid][HID C] x 2 | [grid][HID] | [grid][C]
static size_t mlp_part_bytes(int64_t n, int c, size_t (&off)[4]) {
  const size_t g = (size_t)mlp_grid_bwd(n), hid = 4 * (size_t)c;
  off[0] = 0;
  off[1] = off[0] + ptc_align_up(g * hid * c * 4, 256);
  off[2] = off[1] + ptc_align_up(g * hid * c * 4, 256);
  off[3] = off[2] + ptc_align_up(g * hid * 4, 256);
  return off[3] + ptc_align_up(g * (size_t)c * 4, 256);
}
extern "C" size_t ptc_mlp_bwd_workspace_bytes(int64_t n, int c) {
  size_t off[4];
  return mlp_part_bytes(n > 0 ? n : 1, c, off);
}

int ptc_mlp_bwd_deferred(const void* dm, const void* x, int64_t n, int c, int dtype, const void* w1, const float* b1, const void* w2t, void* dx,
                         float* dw1, float* db1, float* dw2, float* db2, void* workspace, size_t workspace_bytes, ptc_stream_t stream,
                         PtcWgradJob* job_fc1, PtcWgradJob* job_fc2) {
  PTC_REQUIRE(n >= 0 && ptc_mlp_supported(c, dtype), PTC_EUNSUPPORTED, "ptc_mlp_bwd: c=%d dtype=%d (C = 32 | 64, 16-bit features)", c, dtype);
  PTC_REQUIRE((uint64_t)n * (uint64_t)c * 2 <= PTC_BUF_MAX_BYTES, PTC_EUNSUPPORTED, "ptc_mlp_bwd: input of 2 GiB or more");
  const int hid = 4 * c;
  hipStream_t s = (hipStream_t)stream;
  *job_fc1 = PtcWgradJob{nullptr, 0, 0, nullptr, nullptr, 0, nullptr};
  *job_fc2 = *job_fc1;
  if (n == 0) {
    if (dw1) PTC_HIP(hipMemsetAsync(dw1, 0, (size_t)hid * c * 4, s));
    if (dw2) PTC_HIP(hipMemsetAsync(dw2, 0, (size_t)hid * c * 4, s));
    if (db1) PTC_HIP(hipMemsetAsync(db1, 0, (size_t)hid * 4, s));
    if (db2) PTC_HIP(hipMemsetAsync(db2, 0, (size_t)c * 4, s));
    return PTC_OK;
  }
  PTC_REQUIRE(dm && x && w1 && w2t && dx && dw1 && dw2 && workspace, PTC_EINVAL, "ptc_mlp_bwd: null buffer");
  PTC_REQUIRE(((uintptr_t)dm % 16 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)w1 % 16 == 0) && ((uintptr_t)w2t % 16 == 0) && ((uintptr_t)dx % 16 == 0),
              PTC_EINVAL, "ptc_mlp_bwd: buffers must be 16-byte aligned");
  size_t off[4];
  PTC_REQUIRE(workspace_bytes >= mlp_part_bytes(n, c, off), PTC_EWORKSPACE, "ptc_mlp_bwd: workspace too small");
  const int grid = mlp_grid_bwd(n);
  char* ws = (char*)workspace;
  const MlpPartials part{(float*)(ws + off[0]), (float*)(ws + off[1]), (float*)(ws + off[2]), (float*)(ws + off[3])};
#define MLP_BWD(T, CC)                                                                                                          \
  {                                                                                                                             \
    auto kern = mlp_bwd_kernel<T, CC>;                                                                                          \
    const size_t lds = MlpBwdLds<CC>::total;                                                                                    \
    PTC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));    \
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(MLPB_THREADS), lds, s, (const T*)dm, (const T*)x, (const T*)w1, b1, (const T*)w2t, n, \
                       (uint32_t)((uint64_t)n * c * 2), (T*)dx, part);                                                          \
  }
  if (dtype == PTC_BF16) { if (c == 64) MLP_BWD(bf16_t, 64) else MLP_BWD(bf16_t, 32) }
  else { if (c == 64) MLP_BWD(f16_t, 64) else MLP_BWD(f16_t, 32) }
#undef MLP_BWD
  PTC_CHECK_LAUNCH("mlp_bwd_kernel");
  // splits = grid even when it is 1: the kernel always writes partials
  *job_fc1 = PtcWgradJob{part.w1, grid, (int64_t)hid * c, dw1, db1 ? part.b1 : nullptr, (int64_t)hid, db1};
  *job_fc2 = PtcWgradJob{part.w2, grid, (int64_t)hid * c, dw2, db2 ? part.b2 : nullptr, (int64_t)c, db2};
  return PTC_OK;
}

extern "C" int ptc_mlp_bwd(const void* dm, const void* x, int64_t n, int c, int dtype, const void* w1, const float* b1, const void* w2t, void* dx,
                           float* dw1, float* db1, float* dw2, float* db2, void* workspace, size_t workspace_bytes, ptc_stream_t stream) {
  PtcWgradJob jobs[2];
  const int rc = ptc_mlp_bwd_deferred(dm, x, n, c, dtype, w1, b1, w2t, dx, dw1, db1, dw2, db2, workspace, workspace_bytes, stream, &jobs[0], &jobs[1]);
  if (rc != PTC_OK) return rc;
  return ptc_wgrad_reduce_jobs(jobs, 2, stream);
}
