// spconv.hip -- sparse convolution compute on MFMA: out[o,:] = bias + sum_k W_k . in[nbr[k][o],:]
//
// Replaces the feature path of spconv's SubMConv3d / SparseConv3d / SparseInverseConv3d (third
// party, un-vendored; call sites ptv3m1:278-284,499-506 and spconv_unet_v1m1_base.py:43-68,
// 114-121,137-144,173-179) with ONE output-stationary implicit-GEMM kernel over the gather table
// built by rulebook.hip.  No atomics: every output row is produced by exactly one wave in a
// fixed k order, so results are bit-reproducible run to run.
//
// gfx950 mapping
//   * wave64 MFMA 16x16x32 (bf16/f16) or 16x16x4 (exact f32); roles are SWAPPED -- A = W_k tile
//     (rows = output channels), B = gathered input rows -- so that a lane ends up holding 4
//     CONSECUTIVE output channels of one output row (one 8/16-byte store instead of four 2-byte).
//   * B fragments are gathered straight from HBM/L2 into registers: lane (row r, k-group g) loads
//     16 contiguous bytes of input row nbr[k][r]; the 4 k-groups of a row cover 64 contiguous
//     bytes.  The contraction index inside one MFMA may be permuted freely as long as A and B
//     agree, which is what makes "16 contiguous bytes per lane" legal for both dtypes.
//   * W_k slices ([NT x KC] elements, <= 34 KB) are staged through LDS once per (k, channel chunk)
//     per 128-row workgroup; row pitch KC+16 B makes the ds_read_b128 fragment reads conflict-free.
//   * a workgroup skips offset k entirely when none of its 128 rows has that neighbour.
// Roofline: bytes = N_in*C_in*e + N_out*C_out*e + 4*kv*N_out (table) + kv*C_in*C_out*e,
// flops = 2*P*C_in*C_out; HBM-bound for C <= 64, MFMA-bound above.
#include "ptc_common.h"
#include "spconv_internal.h"

#include "mma.h"
#include "wgrad2.h"

#define SC_ROWS 128          // output rows per workgroup (4 waves x 2 sub-tiles x 16)
#define SC_LDS_BYTES 34816   // NT<=128 rows x (KC + 16 B) : (128+8)*2*128 = (64+4)*4*128

// ------------------------------------------------------------------------------------------------
// forward / dgrad / inverse conv
// ------------------------------------------------------------------------------------------------
// Output-channel tiles are processed in groups of G in {4,2,1} tiles.  Inside a group the weight
// rows are PERMUTED across the G MFMA tiles -- tile tt, A-row r  <->  channel 4G*(r>>2) + 4*tt + (r&3)
// -- so that after the MFMAs lane (row, g) holds 4G CONSECUTIVE output channels (8G / 16G bytes):
// one or two 16-byte stores per lane and 128 contiguous bytes per output row, instead of four
// scattered 8-byte pieces.  The permutation is applied while W is staged into LDS (row p(n)).
template <int NTILES> struct TileGroups {
  // group start / size of tile t
  static __host__ __device__ constexpr int gsize(int t) { return (NTILES - (t & ~3)) >= 4 ? 4 : (((NTILES & 3) - ((t & 3) & ~1)) >= 2 ? 2 : 1); }
  static __host__ __device__ constexpr int gstart(int t) { return (NTILES - (t & ~3)) >= 4 ? (t & ~3) : (((NTILES & 3) - ((t & 3) & ~1)) >= 2 ? ((t & ~3) + ((t & 3) & ~1)) : t); }
};

// LDS row that holds natural weight row n (0 <= n < NTILES*16)
template <int NTILES>
__device__ __forceinline__ int lds_row_of_channel(int n) {
  const int t0 = n >> 4;                       // tile the channel would naturally live in
  const int gs = TileGroups<NTILES>::gstart(t0), G = TileGroups<NTILES>::gsize(t0);
  const int local = n - 16 * gs;               // channel inside the group, [0, 16G)
  const int gq = local / (4 * G), rem = local - gq * 4 * G;
  const int tt = rem >> 2, e = rem & 3;
  return 16 * (gs + tt) + 4 * gq + e;
}

// accumulator initialisation = bias of the channel each accumulator element will be stored to (the
// inverse of the epilogue mapping): the bias add costs nothing and, above all, no global load sits
// between the last MFMA and the stores (loads return in order: a bias load in the epilogue would
// queue behind the rows prefetched for the next tile).
template <int NTILES>
__device__ __forceinline__ void sc_bias_regs(const float* __restrict__ bias, int n0, int g, f32x4 (&b)[NTILES]) {
#pragma unroll
  for (int t = 0; t < NTILES; ++t) {
    const int gs = TileGroups<NTILES>::gstart(t), G = TileGroups<NTILES>::gsize(t);
    const int ch = n0 + 16 * gs + 4 * G * g + 4 * (t - gs);
    b[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (bias) b[t] = *reinterpret_cast<const f32x4*>(bias + ch);
  }
}

// epilogue shared by the forward kernels: for a group of G tiles starting at gs, lane (row r, group g)
// holds channels n0 + 16*gs + 4G*g + 4*tt + e (tt < G, e < 4) = 4G consecutive channels of its row.
// two fp32 values -> one 32-bit word of two 16-bit features (RNE)
template <typename T> __device__ __forceinline__ uint32_t sc_pack2(float lo, float hi);
template <> __device__ __forceinline__ uint32_t sc_pack2<bf16_t>(float lo, float hi) { return ptc_pack_bf16x2(lo, hi); }
template <> __device__ __forceinline__ uint32_t sc_pack2<f16_t>(float lo, float hi) {
  const _Float16 a = (_Float16)lo, b = (_Float16)hi;
  return (uint32_t)(*reinterpret_cast<const uint16_t*>(&a)) | ((uint32_t)(*reinterpret_cast<const uint16_t*>(&b)) << 16);
}

template <typename T, int NTILES>
__device__ __forceinline__ void sc_epilogue(f32x4 (&acc)[2][NTILES], const float* __restrict__ bias, T* __restrict__ out,
                                            int64_t rowA, int64_t rowB, int64_t n_out, int c_out, int n0, int g) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int64_t row = s ? rowB : rowA;
    if (row >= n_out) continue;
#pragma unroll
    for (int t = 0; t < NTILES; ++t) {
      const int gs = TileGroups<NTILES>::gstart(t), G = TileGroups<NTILES>::gsize(t);
      if (t != gs) continue;                      // one store sequence per group
      const int ch0 = n0 + 16 * gs + 4 * G * g;
      f32x4 v[4];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        v[tt] = acc[s][(gs + tt) < NTILES ? (gs + tt) : t];
        if (bias && tt < G) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(bias + ch0 + 4 * tt);
          v[tt] += b;
        }
      }
      T* dst = out + row * c_out + ch0;
      if constexpr (sizeof(T) == 2) {
        uint32_t pk[8];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          pk[2 * tt] = sc_pack2<T>(v[tt][0], v[tt][1]);
          pk[2 * tt + 1] = sc_pack2<T>(v[tt][2], v[tt][3]);
        }
        if (G == 4) {
          reinterpret_cast<uint4*>(dst)[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
          reinterpret_cast<uint4*>(dst)[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
        } else if (G == 2) {
          reinterpret_cast<uint4*>(dst)[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        } else {
          reinterpret_cast<uint2*>(dst)[0] = make_uint2(pk[0], pk[1]);
        }
      } else {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
          if (tt < G) reinterpret_cast<f32x4*>(dst)[tt] = v[tt];
      }
    }
  }
}

template <typename T, int NTILES>
__global__ void __launch_bounds__(256)
spconv_fwd_kernel(const T* __restrict__ in, const T* __restrict__ w, const float* __restrict__ bias,
                  const int32_t* __restrict__ nbr, int64_t n_out, int kv, int c_in, int c_out, T* __restrict__ out) {
  using M = Mma<T>;
  constexpr int EPL = M::EPL, KS = M::KS;
  constexpr int KC = 16 * EPL;          // channel chunk staged in LDS (128 for 16-bit, 64 for f32)
  constexpr int PITCH = KC + EPL;       // +16 B pad: conflict-free ds_read_b128
  constexpr int NT = NTILES * 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SC_LDS_BYTES];
  T* wl = reinterpret_cast<T*>(smem);

  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * SC_ROWS + wave * 32;
  const int n0 = blockIdx.y * NT;
  const int64_t rowA = row0 + r, rowB = row0 + 16 + r;

  f32x4 acc[2][NTILES];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int t = 0; t < NTILES; ++t) acc[s][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int k = 0; k < kv; ++k) {
    // nbr == nullptr: identity table (dense row-wise GEMM: Linear layers, 1x1x1 convs)
    int32_t ia, ib;
    if (nbr) {
      ia = rowA < n_out ? nbr[(int64_t)k * n_out + rowA] : -1;
      ib = rowB < n_out ? nbr[(int64_t)k * n_out + rowB] : -1;
    } else {
      ia = rowA < n_out ? (int32_t)rowA : -1;
      ib = rowB < n_out ? (int32_t)rowB : -1;
    }
    // barrier (protects the LDS tile of the previous k) + workgroup-wide "any neighbour at k"
    if (!__syncthreads_or((ia >= 0) | (ib >= 0))) continue;
    for (int c0 = 0; c0 < c_in; c0 += KC) {
      const int kc = (c_in - c0) < KC ? (c_in - c0) : KC;
      if (c0 > 0) __syncthreads();
      // stage W[n0 .. n0+NT)[k][c0 .. c0+kc) -> LDS row p(n), [NT][PITCH]
      const int vpr = kc / EPL;  // 16-byte vectors per row
      for (int q = threadIdx.x; q < NT * vpr; q += 256) {
        const int n = q / vpr, cc = q - n * vpr;
        *reinterpret_cast<uint4*>(wl + lds_row_of_channel<NTILES>(n) * PITCH + cc * EPL) =
            *reinterpret_cast<const uint4*>(w + ((int64_t)(n0 + n) * kv + k) * c_in + c0 + cc * EPL);
      }
      __syncthreads();
      const int nks = (kc + KS - 1) / KS;
      for (int ks = 0; ks < nks; ++ks) {
        const int kk = ks * KS + g * EPL;
        const bool kval = kk < kc;
        typename M::frag fa = M::zero(), fb = M::zero();
        if (kval && ia >= 0) fa = ld_frag<T>(in + (int64_t)ia * c_in + c0 + kk);
        if (kval && ib >= 0) fb = ld_frag<T>(in + (int64_t)ib * c_in + c0 + kk);
#pragma unroll
        for (int t = 0; t < NTILES; ++t) {
          typename M::frag fw = M::zero();
          if (kval) fw = ld_frag<T>(wl + (t * 16 + r) * PITCH + kk);
          acc[0][t] = M::mma(fw, fa, acc[0][t]);
          acc[1][t] = M::mma(fw, fb, acc[1][t]);
        }
      }
    }
  }
  sc_epilogue<T, NTILES>(acc, bias, out, rowA, rowB, n_out, c_out, n0, g);
}

template <typename T, int NTILES>
static int launch_fwd(const void* in, const void* w, const float* bias, const int32_t* nbr, int64_t n_out, int kv,
                      int c_in, int c_out, void* out, hipStream_t s) {
  dim3 grid((unsigned)ptc_cdiv(n_out, SC_ROWS), (unsigned)(c_out / (NTILES * 16)));
  hipLaunchKernelGGL((spconv_fwd_kernel<T, NTILES>), grid, dim3(256), 0, s, (const T*)in, (const T*)w, bias, nbr, n_out,
                     kv, c_in, c_out, (T*)out);
  PTC_CHECK_LAUNCH("spconv_fwd_kernel");
  return PTC_OK;
}

template <typename T>
static int dispatch_fwd(const void* in, const void* w, const float* bias, const int32_t* nbr, int64_t n_out, int kv,
                        int c_in, int c_out, void* out, hipStream_t s) {
  // output-channel tile = largest of {128, 96, 64, 48, 32, 16} dividing c_out
  if (c_out % 128 == 0) return launch_fwd<T, 8>(in, w, bias, nbr, n_out, kv, c_in, c_out, out, s);
  if (c_out % 96 == 0) return launch_fwd<T, 6>(in, w, bias, nbr, n_out, kv, c_in, c_out, out, s);
  if (c_out % 64 == 0) return launch_fwd<T, 4>(in, w, bias, nbr, n_out, kv, c_in, c_out, out, s);
  if (c_out % 48 == 0) return launch_fwd<T, 3>(in, w, bias, nbr, n_out, kv, c_in, c_out, out, s);
  if (c_out % 32 == 0) return launch_fwd<T, 2>(in, w, bias, nbr, n_out, kv, c_in, c_out, out, s);
  return launch_fwd<T, 1>(in, w, bias, nbr, n_out, kv, c_in, c_out, out, s);
}

#include "fwd2.h"
#include "fwd2_joint.h"
#include "gemm3.h"
#include "wgrad3.h"
#include "conv3.h"
#include "conv5.h"
#include "conv7.h"
#include "conv8.h"
#include "wgrad7.h"

extern "C" int ptc_spconv_fwd(const void* in, int64_t n_in, const void* weight, const float* bias, const int32_t* nbr,
                              int64_t n_out, int kv, int c_in, int c_out, int dtype, void* out, ptc_stream_t stream) {
  PTC_REQUIRE(n_in >= 0 && n_out >= 0 && kv >= 1, PTC_EINVAL, "ptc_spconv_fwd: bad sizes");
  PTC_REQUIRE(c_in >= 8 && c_in % 8 == 0, PTC_EUNSUPPORTED, "ptc_spconv_fwd: c_in=%d must be a multiple of 8", c_in);
  PTC_REQUIRE(c_out >= 16 && c_out % 16 == 0, PTC_EUNSUPPORTED, "ptc_spconv_fwd: c_out=%d must be a multiple of 16", c_out);
  if (n_out == 0) return PTC_OK;
  PTC_REQUIRE(weight && out && (n_in == 0 || in), PTC_EINVAL, "ptc_spconv_fwd: null buffer");
  PTC_REQUIRE(nbr || (kv == 1 && n_in >= n_out), PTC_EINVAL, "ptc_spconv_fwd: nbr may be NULL only for kv == 1 (identity table)");
  PTC_REQUIRE(((uintptr_t)in % 16 == 0) && ((uintptr_t)weight % 16 == 0) && ((uintptr_t)out % 16 == 0), PTC_EINVAL,
              "ptc_spconv_fwd: buffers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  // the second- and third-generation kernels gather through raw buffer loads (32-bit offsets, < 2 GiB tensors)
  const bool buf_ok = (uint64_t)n_in * (uint64_t)c_in * ptc_dtype_size(dtype) <= PTC_BUF_MAX_BYTES;
  if (buf_ok && kv <= 2 && (nbr || kv == 1) && ptc_gemm3_supported(dtype, kv, c_in, c_out)) {     // the Linear layers of the deep stages (gemm3.h)
    return ptc_gemm3_launch(dtype, in, n_in, weight, bias, nbr, n_out, kv, c_in, c_out, out, s, 0, nullptr, nullptr);
  }
  if (buf_ok && conv5_supported(dtype, kv, c_in, c_out, nbr, n_in)) {
    if (dtype == PTC_BF16) return launch_conv5<bf16_t>(in, n_in, weight, bias, nbr, n_out, kv, c_in, c_out, out, s);
    return launch_conv5<f16_t>(in, n_in, weight, bias, nbr, n_out, kv, c_in, c_out, out, s);
  }
  if (buf_ok && (conv3_supported(dtype, kv, c_in, c_out, nbr) || conv3_dense_supported(dtype, kv, c_in, c_out, nbr))) {
    if (dtype == PTC_BF16) return launch_conv3<bf16_t>(in, n_in, weight, bias, nbr, n_out, kv, c_in, c_out, out, s);
    return launch_conv3<f16_t>(in, n_in, weight, bias, nbr, n_out, kv, c_in, c_out, out, s);
  }
  if (buf_ok && fwd2_supported(dtype, kv, c_in) && (nbr || kv == 1)) {
    if (dtype == PTC_BF16) return dispatch_fwd2<bf16_t>(in, n_in, weight, bias, nbr, n_out, kv, c_in, c_out, out, s);
    return dispatch_fwd2<f16_t>(in, n_in, weight, bias, nbr, n_out, kv, c_in, c_out, out, s);
  }
  PTC_DISPATCH_DTYPE(dtype, T, return dispatch_fwd<T>(in, weight, bias, nbr, n_out, kv, c_in, c_out, out, s));
  return PTC_OK;
}

extern "C" int ptc_spconv_fwd_blk(const void* in, int64_t n_in, const void* weight, const float* bias, const int32_t* nbr,
                                  const void* tab, const int32_t* hid, const int32_t* hcnt, int bm, int hcap, int64_t n_out, int kv,
                                  int c_in, int c_out, int dtype, void* out, ptc_stream_t stream) {
  const bool buf_ok = (uint64_t)n_in * (uint64_t)c_in * ptc_dtype_size(dtype) <= PTC_BUF_MAX_BYTES;
  // round 6: the wide shapes (c_in >= 96) on the block-staged kernel with streamed weights (conv8.h) -- OPT-IN (PTC_CONV8=1): correct
  // (tests/test_gpu_kernels.py::test_spconv_fwd_block_staged_wide) but measured slower than conv3's global gathers at every shape of the two
  // backbones (128 -> 96 at N = 819200: 820 vs 561 us; 256 -> 256 at N = 12115: 123 vs 100 us; profiles/r06_j_conv8_stages.txt, the ablation
  // in r06_k_conv8_ablation.txt and DESIGN 4.2 say why)
  const char* c8e = getenv("PTC_CONV8");
  const bool conv8_on = !(c8e && c8e[0] == '0');
  if (conv8_on && buf_ok && tab && hid && hcnt && nbr && n_in == n_out && conv8_supported(dtype, kv, c_in, c_out, bm, hcap, n_out)) {
    PTC_REQUIRE(weight && out && in, PTC_EINVAL, "ptc_spconv_fwd_blk: null buffer");
    PTC_REQUIRE(((uintptr_t)in % 16 == 0) && ((uintptr_t)weight % 16 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)tab % 16 == 0), PTC_EINVAL,
                "ptc_spconv_fwd_blk: buffers must be 16-byte aligned");
    if (dtype == PTC_BF16)
      return launch_conv8<bf16_t>(in, n_in, weight, bias, nbr, (const uint16_t*)tab, hid, hcnt, hcap, n_out, c_in, c_out, out, (hipStream_t)stream);
    return launch_conv8<f16_t>(in, n_in, weight, bias, nbr, (const uint16_t*)tab, hid, hcnt, hcap, n_out, c_in, c_out, out, (hipStream_t)stream);
  }
  if (!buf_ok || !tab || !hid || !hcnt || !nbr || n_in != n_out || !conv7_supported(dtype, kv, c_in, c_out, bm, hcap, n_out))
    return ptc_spconv_fwd(in, n_in, weight, bias, nbr, n_out, kv, c_in, c_out, dtype, out, stream);
  PTC_REQUIRE(weight && out && in, PTC_EINVAL, "ptc_spconv_fwd_blk: null buffer");
  PTC_REQUIRE(((uintptr_t)in % 16 == 0) && ((uintptr_t)weight % 16 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)tab % 16 == 0), PTC_EINVAL,
              "ptc_spconv_fwd_blk: buffers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  // conv7 (conv7.hip) over the blocks whose halo fits, then conv5 (its 128-row workgroups are the same blocks) over the others: a
  // workgroup of the second launch whose block conv7 served returns at once.  (Tried: the overflow path inside conv7 -- global gathers
  // with the same wave roles.  The second code path costs the 64-channel kernel its register allocation: 512 registers + 572 bytes
  // of scratch.)
  const int rc = ptc_conv7_launch(dtype, in, n_in, weight, bias, (const uint16_t*)tab, hid, hcnt, n_out, c_in, out, s);
  if (rc != PTC_OK) return rc;
  if (dtype == PTC_BF16) return launch_conv5<bf16_t>(in, n_in, weight, bias, nbr, n_out, 27, c_in, c_in, out, s, hcnt);
  return launch_conv5<f16_t>(in, n_in, weight, bias, nbr, n_out, 27, c_in, c_in, out, s, hcnt);
}

// Dense row-wise GEMM with an MLP epilogue (see fwd2.h): epilogue 1 = out: h, aux_out: GELU(h);
// epilogue 2 = out: acc * GELU'(aux_in).  16-bit features, c_in <= 256 (the persistent linear2 kernel).
extern "C" int ptc_linear_supported_ex(int c_in, int c_out, int dtype) {
  // (the GELU epilogues are instantiated for 128-wide output tiles only: hidden widths 4 C with C a multiple of 32)
  if (ptc_gemm3_supported(dtype, 1, c_in, c_out)) return 1;                          // gemm3.h: any c_in % 64 == 0 from 128 up
  return dtype != PTC_F32 && c_in % 8 == 0 && c_in <= 256 && c_out % 128 == 0;   // and n * c_in * 2 < 2 GiB (checked per call)
}
extern "C" int ptc_linear_fwd_ex(const void* in, int64_t n, const void* weight, const float* bias, int c_in, int c_out, int dtype,
                                 int epilogue, const void* aux_in, void* out, void* aux_out, ptc_stream_t stream) {
  PTC_REQUIRE(n >= 0 && ptc_linear_supported_ex(c_in, c_out, dtype), PTC_EUNSUPPORTED, "ptc_linear_fwd_ex: c_in=%d c_out=%d dtype=%d",
              c_in, c_out, dtype);
  PTC_REQUIRE(epilogue == 1 || epilogue == 2, PTC_EINVAL, "ptc_linear_fwd_ex: epilogue %d", epilogue);
  PTC_REQUIRE((uint64_t)n * (uint64_t)c_in * 2 <= PTC_BUF_MAX_BYTES, PTC_EUNSUPPORTED, "ptc_linear_fwd_ex: input of 2 GiB or more");
  if (n == 0) return PTC_OK;
  PTC_REQUIRE(in && weight && out && (epilogue == 1 ? aux_out != nullptr : aux_in != nullptr), PTC_EINVAL, "ptc_linear_fwd_ex: null buffer");
  hipStream_t s = (hipStream_t)stream;
  if (ptc_gemm3_supported(dtype, 1, c_in, c_out)) {
    return ptc_gemm3_launch(dtype, in, n, weight, bias, nullptr, n, 1, c_in, c_out, out, s, epilogue, aux_in, aux_out);
  }
  if (dtype == PTC_BF16) return dispatch_fwd2<bf16_t>(in, n, weight, bias, nullptr, n, 1, c_in, c_out, out, s, epilogue, aux_in, aux_out);
  return dispatch_fwd2<f16_t>(in, n, weight, bias, nullptr, n, 1, c_in, c_out, out, s, epilogue, aux_in, aux_out);
}

// A Linear with the residual joint behind it in its epilogue (fwd2_joint.h): z = a + row_scale * (in W^T + b), y = LN_B(z) | cast(z).
extern "C" int ptc_linear_joint_supported(int c_in, int c_out, int dtype) { return linear_joint_supported(dtype, c_in, c_out) ? 1 : 0; }
static int linear_joint_impl(const void* in, int64_t n_in, const void* weight, const float* bias, const int32_t* nbr, int64_t n_out, int c_in, int c_out,
                             int dtype, const F2Joint& J, ptc_stream_t stream);
extern "C" int ptc_linear_joint_fwd(const void* in, int64_t n_in, const void* weight, const float* bias, const int32_t* nbr, int64_t n_out,
                                    int c_in, int c_out, int dtype, const float* a, const float* row_scale, const float* gB, const float* bB,
                                    float epsB, int normB, float* z, void* y, float* statB, ptc_stream_t stream) {
  const F2Joint J{a, row_scale, gB, bB, epsB, normB, z, y, statB, nullptr, nullptr, 0.f, 0, nullptr, nullptr, 0};
  return linear_joint_impl(in, n_in, weight, bias, nbr, n_out, c_in, c_out, dtype, J, stream);
}
// ... with the branch operand normalised first: z = a + LN_A(in W^T + b), y = LN_B(z); u_out = the Linear's output (read by the backward)
extern "C" int ptc_linear_norm_joint_fwd(const void* in, int64_t n_in, const void* weight, const float* bias, int64_t n_out, int c_in, int c_out, int dtype,
                                         const float* gA, const float* bA, float epsA, const void* a, int a_dtype, const float* gB, const float* bB, float epsB,
                                         int normB, void* u_out, float* z, void* y, float* statA, float* statB, ptc_stream_t stream) {
  PTC_REQUIRE(n_out == 0 || (u_out && statA), PTC_EINVAL, "ptc_linear_norm_joint_fwd: null buffer");
  PTC_REQUIRE((uintptr_t)u_out % 16 == 0, PTC_EINVAL, "ptc_linear_norm_joint_fwd: buffers must be 16-byte aligned");
  PTC_REQUIRE(a_dtype == PTC_F32 || a_dtype == PTC_BF16 || a_dtype == PTC_F16, PTC_EINVAL, "ptc_linear_norm_joint_fwd: a_dtype %d", a_dtype);
  const F2Joint J{(const float*)a, nullptr, gB, bB, epsB, normB, z, y, statB, gA, bA, epsA, 1, statA, u_out, a_dtype == PTC_BF16 ? 1 : (a_dtype == PTC_F16 ? 2 : 0)};
  return linear_joint_impl(in, n_in, weight, bias, nullptr, n_out, c_in, c_out, dtype, J, stream);
}
static int linear_joint_impl(const void* in, int64_t n_in, const void* weight, const float* bias, const int32_t* nbr, int64_t n_out, int c_in, int c_out,
                             int dtype, const F2Joint& J, ptc_stream_t stream) {
  const float* a = J.a;
  float* z = J.z;
  void* y = J.y;
  const int normB = J.normB;
  float* statB = J.statB;
  PTC_REQUIRE(n_out >= 0 && n_in >= 0 && linear_joint_supported(dtype, c_in, c_out), PTC_EUNSUPPORTED,
              "ptc_linear_joint_fwd: c_in=%d c_out=%d dtype=%d", c_in, c_out, dtype);
  PTC_REQUIRE((uint64_t)n_in * (uint64_t)c_in * 2 <= PTC_BUF_MAX_BYTES, PTC_EUNSUPPORTED, "ptc_linear_joint_fwd: input of 2 GiB or more");
  if (n_out == 0) return PTC_OK;
  PTC_REQUIRE(in && weight && a && z, PTC_EINVAL, "ptc_linear_joint_fwd: null buffer");
  PTC_REQUIRE(!(normB && y) || statB, PTC_EINVAL, "ptc_linear_joint_fwd: missing statistics buffer");
  PTC_REQUIRE(((uintptr_t)in % 16 == 0) && ((uintptr_t)weight % 16 == 0) && ((uintptr_t)a % 16 == 0) && ((uintptr_t)z % 16 == 0) &&
              ((uintptr_t)y % 16 == 0), PTC_EINVAL, "ptc_linear_joint_fwd: buffers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
#define LJ_DISPATCH(T)                                                                             \
  if (c_out == 32) return launch_linear_joint<T, 2>(in, n_in, weight, bias, nbr, n_out, c_in, J, s);  \
  if (c_out == 64) return launch_linear_joint<T, 4>(in, n_in, weight, bias, nbr, n_out, c_in, J, s);  \
  return launch_linear_joint<T, 8>(in, n_in, weight, bias, nbr, n_out, c_in, J, s);
  if (dtype == PTC_BF16) { LJ_DISPATCH(bf16_t) }
  LJ_DISPATCH(f16_t)
#undef LJ_DISPATCH
}

// ------------------------------------------------------------------------------------------------
// wgrad: dw[co][k][ci] = sum_o dout[o][co] * in[nbr[k][o]][ci]
// grid = (splits, kv, channel tiles of 64x64).  The contraction runs over ROWS, so both operands
// are transposed on their way into LDS ([channel][row], rows contiguous) and read back as
// k-contiguous ds_read_b128 fragments.  Partials per split -> deterministic reduction kernel.
// ------------------------------------------------------------------------------------------------
#define WG_RO 64  // rows per staged chunk
#define WG_CT 64  // channel tile (both co and ci)

template <typename T>
__global__ void __launch_bounds__(256)
spconv_wgrad_kernel(const T* __restrict__ in, const T* __restrict__ dout, const int32_t* __restrict__ nbr,
                    int64_t n_out, int kv, int c_in, int c_out, int64_t rows_per_split, int ci_tiles,
                    float* __restrict__ partial, float* __restrict__ bias_partial) {
  using M = Mma<T>;
  constexpr int EPL = M::EPL, KS = M::KS;
  constexpr int PITCH = WG_RO + EPL;  // elements; rows of 16-byte multiples
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * WG_CT * (WG_RO + 8) * 4];
  T* dT = reinterpret_cast<T*>(smem);                 // [WG_CT][PITCH]  dout^T
  T* iT = dT + WG_CT * PITCH;                         // [WG_CT][PITCH]  gathered in^T

  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int k = blockIdx.y;
  const int co0 = (blockIdx.z / ci_tiles) * WG_CT, ci0 = (blockIdx.z % ci_tiles) * WG_CT;
  const int64_t begin = (int64_t)blockIdx.x * rows_per_split;
  int64_t end = begin + rows_per_split;
  if (end > n_out) end = n_out;

  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // bias gradient = column sums of dout: one extra MFMA per k-step against an all-ones B operand,
  // done by the workgroups of table row 0 / input-channel tile 0 only
  const bool do_bias = bias_partial != nullptr && k == 0 && (blockIdx.z % ci_tiles) == 0;
  f32x4 acc_b = (f32x4){0.f, 0.f, 0.f, 0.f};
  typename M::frag ones;
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    if (sizeof(T) == 4) reinterpret_cast<float*>(&ones)[e] = 1.0f;
    else if (std::is_same<T, bf16_t>::value) reinterpret_cast<uint16_t*>(&ones)[e] = 0x3F80;
    else reinterpret_cast<uint16_t*>(&ones)[e] = 0x3C00;
  }

  for (int64_t chunk = begin; chunk < end; chunk += WG_RO) {
    __syncthreads();  // previous chunk fully consumed
    if (sizeof(T) == 2) {
      // thread -> (row pair p, channel vector cv): two rows packed into one 32-bit LDS store
      const int p = threadIdx.x & 31, cv = threadIdx.x >> 5;  // cv in [0,8): 8 x 8 = 64 channels
      const int64_t ra = chunk + 2 * p, rb = ra + 1;
      uint4 da = {0, 0, 0, 0}, db = {0, 0, 0, 0}, xa = {0, 0, 0, 0}, xb = {0, 0, 0, 0};
      const bool cov = (co0 + cv * 8) < c_out, civ = (ci0 + cv * 8) < c_in;
      if (ra < end) {
        if (cov) da = *reinterpret_cast<const uint4*>(dout + ra * c_out + co0 + cv * 8);
        const int32_t j = nbr ? nbr[(int64_t)k * n_out + ra] : (int32_t)ra;
        if (civ && j >= 0) xa = *reinterpret_cast<const uint4*>(in + (int64_t)j * c_in + ci0 + cv * 8);
      }
      if (rb < end) {
        if (cov) db = *reinterpret_cast<const uint4*>(dout + rb * c_out + co0 + cv * 8);
        const int32_t j = nbr ? nbr[(int64_t)k * n_out + rb] : (int32_t)rb;
        if (civ && j >= 0) xb = *reinterpret_cast<const uint4*>(in + (int64_t)j * c_in + ci0 + cv * 8);
      }
      const uint16_t* pa = reinterpret_cast<const uint16_t*>(&da);
      const uint16_t* pb = reinterpret_cast<const uint16_t*>(&db);
      const uint16_t* qa = reinterpret_cast<const uint16_t*>(&xa);
      const uint16_t* qb = reinterpret_cast<const uint16_t*>(&xb);
      uint32_t* d32 = reinterpret_cast<uint32_t*>(dT);
      uint32_t* i32 = reinterpret_cast<uint32_t*>(iT);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        d32[((cv * 8 + e) * PITCH) / 2 + p] = (uint32_t)pa[e] | ((uint32_t)pb[e] << 16);
        i32[((cv * 8 + e) * PITCH) / 2 + p] = (uint32_t)qa[e] | ((uint32_t)qb[e] << 16);
      }
    } else {
      // fp32: thread -> (row rr, channel vector cv), 4 passes cover 16 vectors x 4 floats
      const int rr = threadIdx.x & 63;
      const int64_t row = chunk + rr;
      const int32_t j = row < end ? (nbr ? nbr[(int64_t)k * n_out + row] : (int32_t)row) : -1;
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int cv = (threadIdx.x >> 6) + pass * 4;  // [0,16)
        float4 d = {0.f, 0.f, 0.f, 0.f}, x = {0.f, 0.f, 0.f, 0.f};
        if (row < end && (co0 + cv * 4) < c_out)
          d = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(dout) + row * c_out + co0 + cv * 4);
        if (j >= 0 && (ci0 + cv * 4) < c_in)
          x = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(in) + (int64_t)j * c_in + ci0 + cv * 4);
        float* df = reinterpret_cast<float*>(dT);
        float* xf = reinterpret_cast<float*>(iT);
        df[(cv * 4 + 0) * PITCH + rr] = d.x; df[(cv * 4 + 1) * PITCH + rr] = d.y;
        df[(cv * 4 + 2) * PITCH + rr] = d.z; df[(cv * 4 + 3) * PITCH + rr] = d.w;
        xf[(cv * 4 + 0) * PITCH + rr] = x.x; xf[(cv * 4 + 1) * PITCH + rr] = x.y;
        xf[(cv * 4 + 2) * PITCH + rr] = x.z; xf[(cv * 4 + 3) * PITCH + rr] = x.w;
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < WG_RO / KS; ++ks) {
      const int kk = ks * KS + g * EPL;
      const typename M::frag fa = ld_frag<T>(dT + (wave * 16 + r) * PITCH + kk);  // A: i = co, k = rows
      if (do_bias) acc_b = M::mma(fa, ones, acc_b);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const typename M::frag fb = ld_frag<T>(iT + (t * 16 + r) * PITCH + kk);    // B: j = ci
        acc[t] = M::mma(fa, fb, acc[t]);
      }
    }
  }
  if (do_bias && r == 0) {  // every column j of acc_b holds the same column sum
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int co = co0 + wave * 16 + g * 4 + e;
      if (co < c_out) bias_partial[(int64_t)blockIdx.x * c_out + co] = acc_b[e];
    }
  }
  // D[i = co][j = ci]: lane (j = r) holds co = wave*16 + g*4 + e
  float* pout = partial + (int64_t)blockIdx.x * c_out * kv * c_in;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int ci = ci0 + t * 16 + r;
    if (ci >= c_in) continue;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int co = co0 + wave * 16 + g * 4 + e;
      if (co < c_out) pout[((int64_t)co * kv + k) * c_in + ci] = acc[t][e];
    }
  }
}

// dw[i] = sum_p partial[p][i].  Threads (32 elements x 8 slices of the split axis): coalesced
// 128-byte reads, 8 partial sums in flight per element, LDS finish.  Fixed summation order ->
// bit-reproducible.
// One launch reduces the weight partials AND (second segment, blocks >= nb1) the bias partials.  A block
// owns 64 consecutive outputs (16 lanes x float4); 16 thread groups split the partial index, each keeps four
// independent 16-byte loads in flight (the first version walked the partials with one dependent 4-byte load
// at a time: 9.5 us per call, latency-bound), partial sums meet in LDS in a fixed order.
// FEW partials (<= WG_REDUCE_FEW; `few`, decided on the host, sizes the first segment in blocks of 1024 outputs): one float4 per thread, the
// partials walked in order.  The 16-group form above spends a 256-thread block on 64 outputs and keeps 15 of its 16 groups idle when there is
// ONE partial -- the block-staged convolution gradient of the deep stages (one or eight sequences, wgrad7_splits): 28 MB of partial "reduced"
// in 125 us at 512 channels, 43 us per stage-3 Block (profiles/r06_aw_step_sequence.txt).
#define WG_REDUCE_FEW 8
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ partial, int splits, int64_t count, float* __restrict__ dw, int nb1,
                                                  const float* __restrict__ partial2, int64_t count2, float* __restrict__ out2, int bid, bool few) {
  __shared__ float4 red[16][16];
  if (few && bid < nb1) {
    const int64_t i = (int64_t)bid * 1024 + (int64_t)threadIdx.x * 4;
    if (i >= count) return;
    if ((i + 3 < count) && ((count & 3) == 0)) {
      float4 a = *reinterpret_cast<const float4*>(partial + i);
      for (int p = 1; p < splits; ++p) {
        const float4 v = *reinterpret_cast<const float4*>(partial + (int64_t)p * count + i);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
      *reinterpret_cast<float4*>(dw + i) = a;
    } else {
      for (int64_t j = i; j < count && j < i + 4; ++j) {
        float a = partial[j];
        for (int p = 1; p < splits; ++p) a += partial[(int64_t)p * count + j];
        dw[j] = a;
      }
    }
    return;
  }
  if (bid >= nb1) {
    partial = partial2; count = count2; dw = out2;
  }
  const int64_t blk = bid >= nb1 ? (int64_t)bid - nb1 : (int64_t)bid;
  const int lane = threadIdx.x & 15, pg = threadIdx.x >> 4;   // 16 lanes x float4 = 64 outputs, 16 partial groups
  const int64_t i = blk * 64 + lane * 4;
  float4 acc[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < count) {
    const bool vec = (i + 3 < count) && ((count & 3) == 0);
    for (int p0 = pg; p0 < splits; p0 += 64) {
      if (vec) {
        // four loads in flight: unconditional, from a clamped partial (a load under a per-lane condition is waited for one by one)
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int p = p0 + 16 * u;
          v[u] = *reinterpret_cast<const float4*>(partial + (int64_t)(p < splits ? p : splits - 1) * count + i);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool live = p0 + 16 * u < splits;
          acc[u].x += live ? v[u].x : 0.f; acc[u].y += live ? v[u].y : 0.f; acc[u].z += live ? v[u].z : 0.f; acc[u].w += live ? v[u].w : 0.f;
        }
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int p = p0 + 16 * u;
          if (p < splits) {
            const float* src = partial + (int64_t)p * count + i;
            acc[u].x += src[0];
            acc[u].y += i + 1 < count ? src[1] : 0.f;
            acc[u].z += i + 2 < count ? src[2] : 0.f;
            acc[u].w += i + 3 < count ? src[3] : 0.f;
          }
        }
      }
    }
  }
  float4 t;
  t.x = (acc[0].x + acc[1].x) + (acc[2].x + acc[3].x);
  t.y = (acc[0].y + acc[1].y) + (acc[2].y + acc[3].y);
  t.z = (acc[0].z + acc[1].z) + (acc[2].z + acc[3].z);
  t.w = (acc[0].w + acc[1].w) + (acc[2].w + acc[3].w);
  red[pg][lane] = t;
  __syncthreads();
  if (pg == 0 && i < count) {
    float4 r = red[0][lane];
#pragma unroll
    for (int q = 1; q < 16; ++q) {
      const float4 v = red[q][lane];
      r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w;
    }
    dw[i] = r.x;
    if (i + 1 < count) dw[i + 1] = r.y;
    if (i + 2 < count) dw[i + 2] = r.z;
    if (i + 3 < count) dw[i + 3] = r.w;
  }
}

__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float* __restrict__ partial, int splits, int64_t count, float* __restrict__ dw, int nb1,
                    const float* __restrict__ partial2, int64_t count2, float* __restrict__ out2, int few) {
  wgrad_reduce_body(partial, splits, count, dw, nb1, partial2, count2, out2, (int)blockIdx.x, few != 0);
}

// the reductions of several weight gradients in one launch (the Block executor: six per Block backward): workgroup b belongs to the job
// whose block range holds b and does there exactly what wgrad_reduce_kernel does
struct WgradReduceMulti {
  int n;
  int start[PTC_WGRAD_JOBS_MAX + 1];   // first workgroup of job j; start[n] = grid
  int nb1[PTC_WGRAD_JOBS_MAX];
  int few[PTC_WGRAD_JOBS_MAX];
  PtcWgradJob job[PTC_WGRAD_JOBS_MAX];
};
__global__ void __launch_bounds__(256)
wgrad_reduce_multi_kernel(WgradReduceMulti m) {
  int j = 0;
#pragma unroll
  for (int q = 1; q < PTC_WGRAD_JOBS_MAX; ++q)
    if (q < m.n && (int)blockIdx.x >= m.start[q]) j = q;
  const PtcWgradJob& J = m.job[j];
  const bool alt = J.gate != nullptr && *J.gate != 0;      // (uniform) which producer ran: see PtcWgradJob
  wgrad_reduce_body(alt ? J.alt_partial : J.partial, alt ? J.alt_splits : J.splits, J.count, J.dw, m.nb1[j], J.bias_partial, J.c_out, J.dbias,
                    (int)blockIdx.x - m.start[j], m.few[j] != 0);
}

int ptc_wgrad_reduce_jobs(const PtcWgradJob* jobs, int n, ptc_stream_t stream) {
  PTC_REQUIRE(jobs && n >= 0 && n <= PTC_WGRAD_JOBS_MAX, PTC_EINVAL, "ptc_wgrad_reduce_jobs: %d jobs", n);
  WgradReduceMulti m;
  m.n = 0;
  int grid = 0;
  for (int q = 0; q < n; ++q) {
    if (jobs[q].splits <= 0) continue;
    const bool few = jobs[q].splits <= WG_REDUCE_FEW && (jobs[q].gate == nullptr || jobs[q].alt_splits <= WG_REDUCE_FEW);
    const int nb1 = (int)ptc_cdiv(jobs[q].count, few ? 1024 : 64), nb2 = jobs[q].dbias ? (int)ptc_cdiv(jobs[q].c_out, 64) : 0;
    m.start[m.n] = grid;
    m.nb1[m.n] = nb1;
    m.few[m.n] = few ? 1 : 0;
    m.job[m.n] = jobs[q];
    grid += nb1 + nb2;
    ++m.n;
  }
  if (m.n == 0) return PTC_OK;
  for (int q = m.n; q <= PTC_WGRAD_JOBS_MAX; ++q) m.start[q] = grid;
  hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, m);
  PTC_CHECK_LAUNCH("wgrad_reduce_multi_kernel");
  return PTC_OK;
}

static int launch_wgrad_reduce(const float* partial, int splits, int64_t count, float* dw, const float* bias_partial, int64_t c_out,
                               float* dbias, hipStream_t s) {
  const bool few = splits <= WG_REDUCE_FEW;
  const int nb1 = (int)ptc_cdiv(count, few ? 1024 : 64);
  const int nb2 = dbias ? (int)ptc_cdiv(c_out, 64) : 0;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(nb1 + nb2)), dim3(256), 0, s, partial, splits, count, dw, nb1, bias_partial,
                     c_out, dbias, few ? 1 : 0);
  PTC_CHECK_LAUNCH("wgrad_reduce_kernel");
  return PTC_OK;
}

static int wgrad_splits(int64_t n_out, int kv, int c_in, int c_out) {
  const int64_t tiles = ptc_cdiv(c_out, WG_CT) * ptc_cdiv(c_in, WG_CT);
  int64_t s = 2048 / (kv * tiles);
  const int64_t max_s = ptc_cdiv(n_out, 4 * WG_RO);  // at least 4 chunks per split
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  if (s > 256) s = 256;
  return (int)s;
}

extern "C" size_t ptc_spconv_wgrad_workspace_bytes(int64_t n_out, int kv, int c_in, int c_out) {
  // max over the fp32 path (v1, `splits` partials) and the 16-bit path (v2, one partial per workgroup)
  size_t splits = (size_t)wgrad_splits(n_out, kv, c_in, c_out);
  size_t gx = (size_t)w2_plan(n_out, kv, c_in, c_out, false).gx;
  const size_t gxb = (size_t)w2_plan(n_out, kv, c_in, c_out, true).gx;
  if (gxb > gx) gx = gxb;
  if (gx > splits) splits = gx;
  if (ptc_wgrad3_supported(PTC_BF16, kv, c_in, c_out, n_out)) {      // wgrad3.h (the deep stages' Linear layers): its own split count
    int s3, cps;
    ptc_wgrad3_plan(n_out, c_in, c_out, &s3, &cps);
    if ((size_t)s3 > splits) splits = (size_t)s3;
  }
  return ptc_align_up(splits * (size_t)c_out * kv * c_in * sizeof(float), 256) + ptc_align_up(splits * (size_t)c_out * sizeof(float), 256);
}

// the block-staged weight gradient keeps its partials (one per persistent workgroup sequence, wgrad7.h) BEHIND the region above, which
// stays wgrad2's: the two kernels of ptc_spconv_wgrad_blk are gated on a device-side word and must not share partials
static size_t wgrad_blk_extra_bytes(int64_t n_out, int kv, int c_in, int c_out) {
  if (!wgrad7_supported(PTC_BF16, kv, c_in, c_out, C7_BM, C7_HCAP, n_out)) return 0;
  return ptc_align_up((size_t)wgrad7_splits(n_out, c_in, c_out) * (size_t)c_out * kv * c_in * sizeof(float), 256);
}
extern "C" size_t ptc_spconv_wgrad_blk_workspace_bytes(int64_t n_out, int kv, int c_in, int c_out) {
  return ptc_spconv_wgrad_workspace_bytes(n_out, kv, c_in, c_out) + wgrad_blk_extra_bytes(n_out, kv, c_in, c_out);
}

template <typename T>
static int launch_wgrad(const void* in, const void* dout, const int32_t* nbr, int64_t n_out, int kv, int c_in,
                        int c_out, float* dw, float* dbias, void* ws, hipStream_t s) {
  const int splits = wgrad_splits(n_out, kv, c_in, c_out);
  float* bias_partial = dbias ? (float*)((char*)ws + ptc_align_up((size_t)splits * (size_t)c_out * kv * c_in * sizeof(float), 256)) : nullptr;
  const int ci_tiles = (int)ptc_cdiv(c_in, WG_CT), co_tiles = (int)ptc_cdiv(c_out, WG_CT);
  int64_t rps = ptc_cdiv(ptc_cdiv(n_out, splits), WG_RO) * WG_RO;
  dim3 grid((unsigned)splits, (unsigned)kv, (unsigned)(ci_tiles * co_tiles));
  hipLaunchKernelGGL((spconv_wgrad_kernel<T>), grid, dim3(256), 0, s, (const T*)in, (const T*)dout, nbr, n_out, kv,
                     c_in, c_out, rps, ci_tiles, splits > 1 ? (float*)ws : dw, bias_partial);
  PTC_CHECK_LAUNCH("spconv_wgrad_kernel");
  const int64_t count = (int64_t)c_out * kv * c_in;
  if (splits > 1) return launch_wgrad_reduce((const float*)ws, splits, count, dw, bias_partial, c_out, dbias, s);
  if (dbias) {  // single split: the weight partial went straight to dw, the bias partial still needs its copy-out
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)ptc_cdiv(c_out, 64)), dim3(256), 0, s, (const float*)bias_partial, splits,
                       (int64_t)c_out, dbias, 1 << 30, (const float*)nullptr, (int64_t)0, (float*)nullptr, 0);
    PTC_CHECK_LAUNCH("wgrad_reduce_kernel(bias)");
  }
  return PTC_OK;
}

// ---- v2 (16-bit features): wave-private staging + transposing LDS reads, see wgrad2.h -----------
template <typename T, int COT, int CIT, int KG>
static int launch_wgrad2_inst(const W2Plan& p, const void* in, int64_t n_in, const void* dout, const int32_t* nbr, int64_t n_out, int kv,
                              int c_in, int c_out, float* partial, float* bias_partial, hipStream_t s, const int32_t* gate = nullptr) {
  auto kern = wgrad2_kernel<T, COT, CIT, KG>;
  if (p.lds > 48 * 1024)
    PTC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
  const int nblocks = p.co_blocks * p.ci_blocks, total = p.gx * p.groups * nblocks;
  hipLaunchKernelGGL(kern, dim3((unsigned)(8 * ((total + 7) / 8))), dim3(256), p.lds, s, (const T*)in, (const T*)dout, nbr, n_out, kv,
                     c_in, c_out, ptc_cdiv(n_out, W2_ROWS), p.ci_blocks, partial, bias_partial, p.gx, p.groups, nblocks,
                     (uint32_t)((uint64_t)n_in * c_in * sizeof(T)), (uint32_t)((uint64_t)n_out * c_out * sizeof(T)), gate);
  PTC_CHECK_LAUNCH("wgrad2_kernel");
  return PTC_OK;
}

template <typename T>
static int launch_wgrad2(const void* in, int64_t n_in, const void* dout, const int32_t* nbr, int64_t n_out, int kv, int c_in, int c_out,
                         float* dw, float* dbias, void* ws, hipStream_t s, PtcWgradJob* defer = nullptr, const int32_t* gate = nullptr) {
  const W2Plan p = w2_plan(n_out, kv, c_in, c_out, dbias != nullptr);
  const int64_t count = (int64_t)c_out * kv * c_in;
  // gated (ptc_spconv_wgrad_blk): the partials always go to the workspace -- dw belongs to whichever producer the reduction picks
  float* partial = (p.gx > 1 || gate) ? (float*)ws : dw;
  float* bias_partial = nullptr;
  if (dbias) bias_partial = p.gx > 1 ? (float*)((char*)ws + ptc_align_up((size_t)p.gx * (size_t)count * sizeof(float), 256)) : dbias;
  int rc = PTC_EUNSUPPORTED;
#define W2_CASE(COT, CIT, KG)                                                                                         \
  if (p.cot == COT && p.cit == CIT && p.kg == KG)                                                                     \
    rc = launch_wgrad2_inst<T, COT, CIT, KG>(p, in, n_in, dout, nbr, n_out, kv, c_in, c_out, partial, bias_partial, s, gate);
  W2_CASE(2, 1, 1) W2_CASE(2, 2, 1) W2_CASE(2, 4, 1) W2_CASE(4, 1, 1) W2_CASE(4, 2, 1) W2_CASE(4, 4, 1)
  W2_CASE(6, 1, 1) W2_CASE(6, 2, 1) W2_CASE(6, 4, 1) W2_CASE(8, 1, 1) W2_CASE(8, 2, 1)   // (8, 4, 1): never planned (w2_plan caps 64-wide input tiles at 64 outputs), spilled
  W2_CASE(2, 1, 8) W2_CASE(2, 2, 4) W2_CASE(4, 2, 4) W2_CASE(2, 4, 4) W2_CASE(4, 4, 2) W2_CASE(6, 2, 2)
#undef W2_CASE
  if (rc != PTC_OK) {
    if (rc == PTC_EUNSUPPORTED) ptc_set_error("ptc_spconv_wgrad: no wgrad2 instance for tiles (%d,%d,%d)", p.cot, p.cit, p.kg);
    return rc;
  }
  if (defer) {   // the caller batches the reduction (ptc_wgrad_reduce_jobs)
    *defer = PtcWgradJob{partial, (p.gx > 1 || gate) ? p.gx : 0, count, dw, bias_partial, (int64_t)c_out, dbias};
    return PTC_OK;
  }
  if (p.gx > 1) return launch_wgrad_reduce(partial, p.gx, count, dw, bias_partial, c_out, dbias, s);
  return PTC_OK;
}

static int spconv_wgrad_impl(const void* in, int64_t n_in, const void* dout, const int32_t* nbr, int64_t n_out, int kv, int c_in, int c_out,
                             int dtype, float* dw, float* dbias, void* workspace, size_t workspace_bytes, ptc_stream_t stream, PtcWgradJob* defer);

// ---- wgrad3.h: the Linear layers of the deep stages (c_in, c_out multiples of 128), several weights per launch -------------------------
static bool wgrad3_eligible(const void* in, int64_t n_in, const void* dout, const int32_t* nbr, int64_t n_out, int kv, int c_in, int c_out, int dtype,
                            const float* dw, const void* workspace, size_t workspace_bytes) {
  return n_out > 0 && in && dout && dw && workspace && (dtype == PTC_BF16 || dtype == PTC_F16) && ptc_wgrad3_supported(dtype, kv, c_in, c_out, n_out) &&
         (nbr || n_in >= n_out) && (uint64_t)n_in * c_in * 2 <= PTC_BUF_MAX_BYTES && (uint64_t)n_out * c_out * 2 <= PTC_BUF_MAX_BYTES &&
         ((uintptr_t)in % 16 == 0) && ((uintptr_t)dout % 16 == 0) && workspace_bytes >= ptc_spconv_wgrad_workspace_bytes(n_out, kv, c_in, c_out);
}
// appends one weight to the group: its partials live in `workspace`; *job = the reduction still owed
static void wgrad3_add(W3Group& g, const void* in, int64_t n_in, const void* dout, const int32_t* nbr, int64_t n_out, int c_in, int c_out, float* dw,
                       float* dbias, void* workspace, PtcWgradJob* job) {
  int splits, cps;
  ptc_wgrad3_plan(n_out, c_in, c_out, &splits, &cps);
  const int64_t count = (int64_t)c_out * c_in;
  float* partial = (float*)workspace;
  float* bias_partial = dbias ? (float*)((char*)workspace + ptc_align_up((size_t)splits * (size_t)count * sizeof(float), 256)) : nullptr;
  const int tiles_i = c_in / 128, tiles = tiles_i * (c_out / 128);
  const int q = g.n++;
  g.p[q] = W3Problem{in, dout, nbr, n_out, c_in, c_out, partial, bias_partial, splits, cps, tiles, tiles_i, (uint32_t)((uint64_t)n_in * c_in * 2),
                     (uint32_t)((uint64_t)n_out * c_out * 2)};
  g.start[q + 1] = g.start[q] + splits * tiles;
  for (int k = q + 2; k <= W3_MAX; ++k) g.start[k] = g.start[q + 1];
  *job = PtcWgradJob{partial, splits, count, dw, bias_partial, (int64_t)c_out, dbias};
}

// the grouped launch: calls[i] that are wgrad2-eligible and planned on the (4, 4, 1) instance share one launch of wgrad2_group_kernel
template <typename T>
static int launch_wgrad2_group(const PtcWgradCall* const* cs, int m, PtcWgradJob* const* jobs, hipStream_t s) {
  W2Group<T> g;
  g.n = m;
  int grid = 0;
  size_t lds = 0;
  for (int q = 0; q < m; ++q) {
    const PtcWgradCall& c = *cs[q];
    const W2Plan p = w2_plan(c.n_out, c.kv, c.c_in, c.c_out, c.dbias != nullptr);
    const int64_t count = (int64_t)c.c_out * c.kv * c.c_in;
    float* partial = p.gx > 1 ? (float*)c.workspace : c.dw;
    float* bias_partial = nullptr;
    if (c.dbias) bias_partial = p.gx > 1 ? (float*)((char*)c.workspace + ptc_align_up((size_t)p.gx * (size_t)count * sizeof(float), 256)) : c.dbias;
    const int nblocks = p.co_blocks * p.ci_blocks, total = p.gx * p.groups * nblocks;
    g.start[q] = grid;
    g.p[q] = W2Problem<T>{(const T*)c.in, (const T*)c.dout, c.nbr, c.n_out, c.kv, c.c_in, c.c_out, ptc_cdiv(c.n_out, W2_ROWS), p.ci_blocks, partial,
                          bias_partial, p.gx, p.groups, nblocks, (uint32_t)((uint64_t)c.n_in * c.c_in * sizeof(T)),
                          (uint32_t)((uint64_t)c.n_out * c.c_out * sizeof(T))};
    grid += 8 * ((total + 7) / 8);            // every problem starts on a multiple of 8: its XCD-first numbering stays its own
    lds = p.lds;
    *jobs[q] = PtcWgradJob{partial, p.gx > 1 ? p.gx : 0, count, c.dw, bias_partial, (int64_t)c.c_out, c.dbias};
  }
  for (int q = m; q <= W2_GROUP_MAX; ++q) g.start[q] = grid;
  auto kern = wgrad2_group_kernel<T, 4, 4, 1>;
  if (lds > 48 * 1024) PTC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, g);
  PTC_CHECK_LAUNCH("wgrad2_group_kernel");
  return PTC_OK;
}

int ptc_spconv_wgrad_group(const PtcWgradCall* calls, int n, PtcWgradJob* jobs, ptc_stream_t stream) {
  PTC_REQUIRE(calls && jobs && n >= 0, PTC_EINVAL, "ptc_spconv_wgrad_group: null tables");
  const PtcWgradCall* grp[W2_GROUP_MAX];
  PtcWgradJob* gj[W2_GROUP_MAX];
  int m = 0, gdtype = -1;
  W3Group g3;
  g3.n = 0;
  for (int k = 0; k <= W3_MAX; ++k) g3.start[k] = 0;
  int g3dtype = -1;
  for (int i = 0; i < n; ++i) {
    const PtcWgradCall& c = calls[i];
    jobs[i] = PtcWgradJob{nullptr, 0, 0, nullptr, nullptr, 0, nullptr};
    bool grouped = false;
    if (g3.n < W3_MAX && (g3dtype < 0 || g3dtype == c.dtype) &&
        wgrad3_eligible(c.in, c.n_in, c.dout, c.nbr, c.n_out, c.kv, c.c_in, c.c_out, c.dtype, c.dw, c.workspace, c.workspace_bytes)) {
      wgrad3_add(g3, c.in, c.n_in, c.dout, c.nbr, c.n_out, c.c_in, c.c_out, c.dw, c.dbias, c.workspace, &jobs[i]);
      g3dtype = c.dtype;
      continue;
    }
    if (c.n_out > 0 && c.in && c.dout && c.dw && c.workspace && (c.dtype == PTC_BF16 || c.dtype == PTC_F16) && (c.nbr || (c.kv == 1 && c.n_in >= c.n_out)) &&
        c.c_in % 8 == 0 && c.c_out % 8 == 0 && (uint64_t)c.n_in * c.c_in * 2 <= PTC_BUF_MAX_BYTES && (uint64_t)c.n_out * c.c_out * 2 <= PTC_BUF_MAX_BYTES &&
        c.workspace_bytes >= ptc_spconv_wgrad_workspace_bytes(c.n_out, c.kv, c.c_in, c.c_out)) {
      const W2Plan p = w2_plan(c.n_out, c.kv, c.c_in, c.c_out, c.dbias != nullptr);
      if (p.cot == 4 && p.cit == 4 && p.kg == 1 && m < W2_GROUP_MAX && (gdtype < 0 || gdtype == c.dtype)) {
        grp[m] = &c; gj[m] = &jobs[i]; ++m; gdtype = c.dtype;
        grouped = true;
      }
    }
    if (!grouped) {   // its own launch (and every argument check of the plain entry point)
      const int rc = spconv_wgrad_impl(c.in, c.n_in, c.dout, c.nbr, c.n_out, c.kv, c.c_in, c.c_out, c.dtype, c.dw, c.dbias, c.workspace, c.workspace_bytes,
                                       stream, &jobs[i]);
      if (rc != PTC_OK) return rc;
    }
  }
  if (g3.n > 0) {
    const int rc = ptc_wgrad3_launch(g3dtype, g3, (hipStream_t)stream);
    if (rc != PTC_OK) return rc;
  }
  if (m == 1) {
    const PtcWgradCall& c = *grp[0];
    return spconv_wgrad_impl(c.in, c.n_in, c.dout, c.nbr, c.n_out, c.kv, c.c_in, c.c_out, c.dtype, c.dw, c.dbias, c.workspace, c.workspace_bytes, stream, gj[0]);
  }
  if (m > 1) return gdtype == PTC_BF16 ? launch_wgrad2_group<bf16_t>(grp, m, gj, (hipStream_t)stream) : launch_wgrad2_group<f16_t>(grp, m, gj, (hipStream_t)stream);
  return PTC_OK;
}

extern "C" int ptc_spconv_wgrad(const void* in, int64_t n_in, const void* dout, const int32_t* nbr, int64_t n_out,
                                int kv, int c_in, int c_out, int dtype, float* dw, float* dbias, void* workspace,
                                size_t workspace_bytes, ptc_stream_t stream) {
  return spconv_wgrad_impl(in, n_in, dout, nbr, n_out, kv, c_in, c_out, dtype, dw, dbias, workspace, workspace_bytes, stream, nullptr);
}

int ptc_spconv_wgrad_deferred(const void* in, int64_t n_in, const void* dout, const int32_t* nbr, int64_t n_out, int kv, int c_in, int c_out,
                              int dtype, float* dw, float* dbias, void* workspace, size_t workspace_bytes, ptc_stream_t stream,
                              PtcWgradJob* job) {
  PTC_REQUIRE(job != nullptr, PTC_EINVAL, "ptc_spconv_wgrad_deferred: null job");
  *job = PtcWgradJob{nullptr, 0, 0, nullptr, nullptr, 0, nullptr};
  return spconv_wgrad_impl(in, n_in, dout, nbr, n_out, kv, c_in, c_out, dtype, dw, dbias, workspace, workspace_bytes, stream, job);
}

// ---- block-staged weight gradient (wgrad7.h): 3^3 submanifold table, c_in = c_out = 32 | 64, 16-bit features, block tables of blocks.hip ----
// Two producers, gated on the table builder's device-side overflow counter (no host synchronisation): wgrad7 when every block's halo fits,
// wgrad2 over the whole tensor when one did not; the reduction sums the partials of whichever ran.  Shapes outside wgrad7's range take
// ptc_spconv_wgrad unchanged.
int ptc_spconv_wgrad_blk_deferred(const void* in, int64_t n_in, const void* dout, const int32_t* nbr, const void* tab, const int32_t* hid,
                                  const int32_t* hcnt, const int32_t* n_overflow, int bm, int hcap, int64_t n_out, int kv, int c_in, int c_out,
                                  int dtype, float* dw, void* workspace, size_t workspace_bytes, ptc_stream_t stream, PtcWgradJob* job) {
  PTC_REQUIRE(job != nullptr, PTC_EINVAL, "ptc_spconv_wgrad_blk: null job");
  *job = PtcWgradJob{nullptr, 0, 0, nullptr, nullptr, 0, nullptr};
  const bool buf_ok = (uint64_t)n_in * c_in * 2 <= PTC_BUF_MAX_BYTES && (uint64_t)n_out * c_out * 2 <= PTC_BUF_MAX_BYTES;
  if (!buf_ok || !tab || !hid || !hcnt || !n_overflow || !nbr || n_in != n_out || !wgrad7_supported(dtype, kv, c_in, c_out, bm, hcap, n_out))
    return spconv_wgrad_impl(in, n_in, dout, nbr, n_out, kv, c_in, c_out, dtype, dw, nullptr, workspace, workspace_bytes, stream, job);
  PTC_REQUIRE(in && dout && dw && workspace, PTC_EINVAL, "ptc_spconv_wgrad_blk: null buffer");
  PTC_REQUIRE(workspace_bytes >= ptc_spconv_wgrad_blk_workspace_bytes(n_out, kv, c_in, c_out), PTC_EWORKSPACE, "ptc_spconv_wgrad_blk: workspace too small");
  PTC_REQUIRE(((uintptr_t)in % 16 == 0) && ((uintptr_t)dout % 16 == 0) && ((uintptr_t)tab % 16 == 0), PTC_EINVAL,
              "ptc_spconv_wgrad_blk: buffers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  float* p7 = (float*)((char*)workspace + ptc_spconv_wgrad_workspace_bytes(n_out, kv, c_in, c_out));
  int rc = ptc_wgrad7_launch(dtype, in, dout, (const uint16_t*)tab, hid, hcnt, n_overflow, n_out, c_in, c_out, p7, s);
  if (rc != PTC_OK) return rc;
  PtcWgradJob j2;
  rc = dtype == PTC_BF16 ? launch_wgrad2<bf16_t>(in, n_in, dout, nbr, n_out, kv, c_in, c_out, dw, nullptr, workspace, s, &j2, n_overflow)
                         : launch_wgrad2<f16_t>(in, n_in, dout, nbr, n_out, kv, c_in, c_out, dw, nullptr, workspace, s, &j2, n_overflow);
  if (rc != PTC_OK) return rc;
  *job = PtcWgradJob{p7, wgrad7_splits(n_out, c_in, c_out), (int64_t)c_out * kv * c_in, dw, nullptr, (int64_t)c_out, nullptr, n_overflow, j2.partial, j2.splits};
  return PTC_OK;
}

extern "C" int ptc_spconv_wgrad_blk(const void* in, int64_t n_in, const void* dout, const int32_t* nbr, const void* tab, const int32_t* hid,
                                    const int32_t* hcnt, const int32_t* n_overflow, int bm, int hcap, int64_t n_out, int kv, int c_in, int c_out,
                                    int dtype, float* dw, void* workspace, size_t workspace_bytes, ptc_stream_t stream) {
  PtcWgradJob job;
  const int rc = ptc_spconv_wgrad_blk_deferred(in, n_in, dout, nbr, tab, hid, hcnt, n_overflow, bm, hcap, n_out, kv, c_in, c_out, dtype, dw, workspace,
                                               workspace_bytes, stream, &job);
  if (rc != PTC_OK) return rc;
  return ptc_wgrad_reduce_jobs(&job, 1, stream);
}

static int spconv_wgrad_impl(const void* in, int64_t n_in, const void* dout, const int32_t* nbr, int64_t n_out, int kv, int c_in, int c_out,
                             int dtype, float* dw, float* dbias, void* workspace, size_t workspace_bytes, ptc_stream_t stream, PtcWgradJob* defer) {
  PTC_REQUIRE(n_in >= 0 && n_out >= 0 && kv >= 1, PTC_EINVAL, "ptc_spconv_wgrad: bad sizes");
  PTC_REQUIRE(c_in >= 8 && c_in % 8 == 0 && c_out >= 8 && c_out % 8 == 0, PTC_EUNSUPPORTED,
              "ptc_spconv_wgrad: c_in=%d c_out=%d must be multiples of 8", c_in, c_out);
  PTC_REQUIRE(dw && workspace, PTC_EINVAL, "ptc_spconv_wgrad: null buffer");
  PTC_REQUIRE(workspace_bytes >= ptc_spconv_wgrad_workspace_bytes(n_out, kv, c_in, c_out), PTC_EWORKSPACE,
              "ptc_spconv_wgrad: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  if (n_out == 0) {
    PTC_HIP(hipMemsetAsync(dw, 0, (size_t)c_out * kv * c_in * sizeof(float), s));
    if (dbias) PTC_HIP(hipMemsetAsync(dbias, 0, (size_t)c_out * sizeof(float), s));
    return PTC_OK;
  }
  PTC_REQUIRE(in && dout, PTC_EINVAL, "ptc_spconv_wgrad: null buffer");
  PTC_REQUIRE(nbr || (kv == 1 && n_in >= n_out), PTC_EINVAL, "ptc_spconv_wgrad: nbr may be NULL only for kv == 1 (identity table)");
  // wgrad2 gathers through raw buffer loads (< 2 GiB operands); larger ones take the v1 kernel
  const bool buf_ok = (uint64_t)n_in * c_in * 2 <= PTC_BUF_MAX_BYTES && (uint64_t)n_out * c_out * 2 <= PTC_BUF_MAX_BYTES;
  if (wgrad3_eligible(in, n_in, dout, nbr, n_out, kv, c_in, c_out, dtype, dw, workspace, workspace_bytes)) {
    W3Group g3;
    g3.n = 0;
    for (int k = 0; k <= W3_MAX; ++k) g3.start[k] = 0;
    PtcWgradJob job;
    wgrad3_add(g3, in, n_in, dout, nbr, n_out, c_in, c_out, dw, dbias, workspace, &job);
    const int rc = ptc_wgrad3_launch(dtype, g3, s);
    if (rc != PTC_OK) return rc;
    if (defer) { *defer = job; return PTC_OK; }
    return launch_wgrad_reduce(job.partial, job.splits, job.count, dw, job.bias_partial, c_out, dbias, s);
  }
  if (dtype == PTC_BF16 && buf_ok) return launch_wgrad2<bf16_t>(in, n_in, dout, nbr, n_out, kv, c_in, c_out, dw, dbias, workspace, s, defer);
  if (dtype == PTC_F16 && buf_ok) return launch_wgrad2<f16_t>(in, n_in, dout, nbr, n_out, kv, c_in, c_out, dw, dbias, workspace, s, defer);
  PTC_DISPATCH_DTYPE(dtype, T, return launch_wgrad<T>(in, dout, nbr, n_out, kv, c_in, c_out, dw, dbias, workspace, s));
  return PTC_OK;
}
