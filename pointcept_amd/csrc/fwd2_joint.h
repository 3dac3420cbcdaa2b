// fwd2_joint.h -- a residual joint of the PTv3 Block in the EPILOGUE of the Linear that feeds it (round 4; included by spconv.hip).
//
// Block.forward (pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py:318-338) adds the output of `proj`
// (SerializedAttention, :219) and of the MLP's `fc2` (:246) to the fp32 residual stream, each time followed by the next LayerNorm
// (norm2, :305) or by the 16-bit copy the next convolution reads:
//     z = a + row_scale * (x W^T + b);   y = LN_B(z) (or the cast of z)
// As two launches (linear2_kernel, then add_norm_fwd_kernel of norm.hip) the 16-bit GEMM output makes a round trip through HBM: 4 of
// the 14 bytes per element the pair moves (profiles/r04_f_step_traffic.txt: residual joints + LayerNorm = 21 GB of the step's 103).
// Here the persistent Linear kernel (W resident in LDS, 128-row tiles, rows prefetched one tile ahead: linear2_kernel's loop) writes
// each 16-row half of a wave's output tile to its LDS slice as [row][channel] in the feature dtype -- the rounding the reference's
// autocast Linear applies -- and reads it back with add_norm_fwd_kernel's lane mapping (C / 8 consecutive lanes per row, 8 channels
// each): the joint's arithmetic is THAT kernel's, statement for statement (ln_common.h), so the fused and the two-launch forms
// produce the same bits.  Only joints whose branch operand is not normalised (the two above) are served: the backward of
// `x + LN(cpe)` needs the Linear's output itself.  c_out = 32 | 64 | 128 (one column block: the whole row in a workgroup), c_in <= 256.
#pragma once
#include "ln_common.h"

struct F2Joint {
  const float* a;          // residual stream [n, c] fp32
  const float* row_scale;  // DropPath row factors [n] or NULL
  const float* gB;         // LayerNorm B affine (NULL: none)
  const float* bB;
  float epsB;
  int normB;               // 1: y = LN_B(z); 0: y = z
  float* z;                // [n, c] fp32
  void* y;                 // [n, c] feature dtype, or NULL
  float* statB;            // [2][n] mean / rstd of LN_B, or NULL
  // joint `x + LN_A(Linear(.))` (the positional encoding, ptv3m1:318-321): the branch operand is normalised first; its backward reads the
  // Linear's output, so that is written too (u_out)
  const float* gA;
  const float* bA;
  float epsA;
  int normA;
  float* statA;            // [2][n]
  void* u_out;             // [n, c] feature dtype: the Linear's output (normA only)
  int a_kind;              // dtype of `a`: 0 fp32 (the stream), 1 bf16, 2 f16 (first Block of a stage: the pooling / unpooling output)
};

template <typename T, int NTILES, int S>
__global__ void __launch_bounds__(256)
linear2_joint_kernel(const T* __restrict__ in, const T* __restrict__ w, const float* __restrict__ bias, const int32_t* __restrict__ nbr,
                     int64_t n_out, int c_in, uint32_t in_bytes, F2Joint J) {
  using M = Mma<T>;
  static_assert(sizeof(T) == 2, "16-bit features only");
  const __amdgpu_buffer_rsrc_t in_buf = ptc_buf(in, in_bytes);
  constexpr int NT = NTILES * 16, LPR = NT / LN_VEC, RB = NT * 2, P = RB + 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int pitch = c_in + 8;
  T* wl = reinterpret_cast<T*>(smem);                                                       // [NT][pitch]
  float* bl = reinterpret_cast<float*>(smem + (((size_t)NT * pitch * 2 + 15) & ~(size_t)15));   // [NT] bias
  unsigned char* slice = reinterpret_cast<unsigned char*>(bl + NT) + (threadIdx.x >> 6) * f2_out_slice_bytes(NT);
  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int vpr = c_in >> 3;
  for (int q = threadIdx.x; q < NT * vpr; q += 256) {
    const int n = q / vpr, cc = q - n * vpr;
    *reinterpret_cast<uint4*>(wl + lds_row_of_channel<NTILES>(n) * pitch + cc * 8) = *reinterpret_cast<const uint4*>(w + (int64_t)n * c_in + cc * 8);
  }
  for (int q = threadIdx.x; q < NT; q += 256) bl[q] = bias ? bias[q] : 0.f;
  __syncthreads();
  // LayerNorm B affine of this lane's 8 channels in the read-back mapping (slot = lane % LPR)
  const int slot = lane % LPR;
  float gb[LN_VEC], bb[LN_VEC], ga[LN_VEC], ba[LN_VEC];
#pragma unroll
  for (int i = 0; i < LN_VEC; ++i) {
    gb[i] = (J.normB && J.gB) ? J.gB[slot * LN_VEC + i] : 1.f;
    bb[i] = (J.normB && J.bB) ? J.bB[slot * LN_VEC + i] : 0.f;
    ga[i] = (J.normA && J.gA) ? J.gA[slot * LN_VEC + i] : 1.f;
    ba[i] = (J.normA && J.bA) ? J.bA[slot * LN_VEC + i] : 0.f;
  }

  const int64_t tiles = (n_out + F2_ROWS - 1) / F2_ROWS;
  auto load_idx = [&](int64_t tile, int32_t& ia, int32_t& ib) {
    const int64_t rowA = tile * F2_ROWS + wave * 32 + r, rowB = rowA + 16;
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
    load_rows(na, nb, pa, pb);
    load_idx(tile + 2 * (int64_t)gridDim.x, na, nb);
    f32x4 acc[2][NTILES];
#pragma unroll
    for (int t = 0; t < NTILES; ++t) {
      const int gs = TileGroups<NTILES>::gstart(t), G = TileGroups<NTILES>::gsize(t);
      acc[0][t] = *reinterpret_cast<const f32x4*>(bl + 16 * gs + 4 * G * g + 4 * (t - gs));
      acc[1][t] = acc[0][t];
    }
    auto products = [&](auto full_c) {                  // (see linear2_kernel: unconditional W reads when c_in is a multiple of 32)
      constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const int col = s * 32 + g * 8;
        const T* wrow = wl + r * pitch + col;
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
    // ---- epilogue: the 16-row halves of this wave's 32 rows through its LDS slice, then the joint in add_norm_fwd_kernel's mapping
    const int64_t row0 = tile * F2_ROWS + wave * 32;
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) {
#pragma unroll
      for (int t = 0; t < NTILES; ++t) {
        const int gs = TileGroups<NTILES>::gstart(t), G = TileGroups<NTILES>::gsize(t);
        if (t != gs) continue;
        uint32_t pk[8];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          const f32x4 v = acc[sx][(gs + tt) < NTILES ? (gs + tt) : t];
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
      for (int it = 0; it < (F2_OUT_ROWS * LPR + 63) / 64; ++it) {
        const int q = it * 64 + lane, row = q / LPR;                 // row < 16 always: 16 * LPR is a multiple of 64 or below it
        const int64_t grow = row0 + sx * 16 + row;
        const bool ok = row < F2_OUT_ROWS && grow < n_out;
        float v[LN_VEC], rr[LN_VEC];
        ln_load8<T>(reinterpret_cast<const T*>(slice + (row < F2_OUT_ROWS ? row : 0) * P) + slot * LN_VEC, v);
        if (ok) {
          if (J.a_kind == 1) ln_load8<bf16_t>(reinterpret_cast<const bf16_t*>(J.a) + grow * NT + slot * LN_VEC, rr);
          else if (J.a_kind == 2) ln_load8<f16_t>(reinterpret_cast<const f16_t*>(J.a) + grow * NT + slot * LN_VEC, rr);
          else ln_load8<float>(J.a + grow * NT + slot * LN_VEC, rr);
        } else {
#pragma unroll
          for (int i = 0; i < LN_VEC; ++i) rr[i] = 0.f;
        }
        if (J.normA) {              // add_norm_fwd_kernel's normA branch: the same ln_normalize
          if (ok) ln_store8<T>(reinterpret_cast<T*>(J.u_out) + grow * NT + slot * LN_VEC, v);     // v holds T-representable values: exact
          float mean, rstd;
          ln_normalize<LPR>(v, J.epsA, ga, ba, mean, rstd);
          if (ok && slot == 0) { J.statA[grow] = mean; J.statA[n_out + grow] = rstd; }
        }
        const float sc = (J.row_scale && ok) ? J.row_scale[grow] : 1.f;
#pragma unroll
        for (int i = 0; i < LN_VEC; ++i) rr[i] = fmaf(sc, v[i], rr[i]);
        if (ok) ln_store8<float>(J.z + grow * NT + slot * LN_VEC, rr);
        if (J.y) {
          if (J.normB) {              // (the shuffles run in every lane: rows past the end carry zeros and store nothing)
            float mean, rstd;
            ln_normalize<LPR>(rr, J.epsB, gb, bb, mean, rstd);
            if (ok && slot == 0) { J.statB[grow] = mean; J.statB[n_out + grow] = rstd; }
          }
          if (ok) ln_store8<T>(reinterpret_cast<T*>(J.y) + grow * NT + slot * LN_VEC, rr);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();      // the slice is rewritten by the next half / tile
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (int s = 0; s < S; ++s) { ca[s] = pa[s]; cb[s] = pb[s]; }
  }
}

// the shapes of the two joints: `proj` (C -> C) and `fc2` (4 C -> C) of the 32 / 64 / 128-channel Blocks, contraction <= 256
static inline bool linear_joint_supported(int dtype, int c_in, int c_out) {
  return dtype != PTC_F32 && (c_out == 32 || c_out == 64 || c_out == 128) && (c_in == c_out || c_in == 4 * c_out) && c_in <= 256 &&
         (size_t)c_out * (c_in + 8) * 2 + (size_t)c_out * 4 + 16 + 4 * f2_out_slice_bytes(c_out) <= 64 * 1024;
}

template <typename T, int NTILES>
static int launch_linear_joint(const void* in, int64_t n_in, const void* w, const float* bias, const int32_t* nbr, int64_t n_out, int c_in,
                               const F2Joint& J, hipStream_t s) {
  constexpr int NT = NTILES * 16;
  const size_t lds = (((size_t)NT * (c_in + 8) * 2 + 15) & ~(size_t)15) + (size_t)NT * 4 + 4 * f2_out_slice_bytes(NT);
  const int64_t tiles = ptc_cdiv(n_out, F2_ROWS);
  int64_t per_cu = (160 * 1024) / (int64_t)lds;
#ifndef F2_PER_CU
#define F2_PER_CU 4
#endif
  per_cu = per_cu > F2_PER_CU ? F2_PER_CU : (per_cu < 1 ? 1 : per_cu);
  int64_t gx = 256 * per_cu;
  if (gx > tiles) gx = tiles;
  if (gx < 1) gx = 1;
  const int S = c_in / 32;
#define LJ_CASE(SS)                                                                                                              \
  case SS: if constexpr ((SS) <= 8) {                                                                                            \
    auto kern = linear2_joint_kernel<T, NTILES, SS>;                                                                              \
    if (lds > 48 * 1024)                                                                                                         \
      PTC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));     \
    hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(256), lds, s, (const T*)in, (const T*)w, bias, nbr, n_out, c_in,            \
                       (uint32_t)((uint64_t)n_in * c_in * sizeof(T)), J);                                                        \
  } break;
  switch (S) {          // S = NTILES / 2 (proj) or 2 NTILES (fc2): only those are built
    LJ_CASE((NTILES / 2))
    LJ_CASE((NTILES * 2 <= 8 ? NTILES * 2 : 9))
    default: ptc_set_error("linear joint: c_in=%d unsupported for %d output channels", c_in, NT); return PTC_EUNSUPPORTED;
  }
#undef LJ_CASE
  PTC_CHECK_LAUNCH("linear2_joint_kernel");
  return PTC_OK;
}
