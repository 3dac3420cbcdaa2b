// wgrad2.h -- weight gradient of the gather-table convolution / Linear layers for 16-bit features:
//     dw[co][k][ci] = sum_o dout[o][co] * in[nbr[k][o]][ci]        (contraction over ROWS)
// Both MFMA operands are needed "channel-major" (8 consecutive rows of one channel per lane) while
// memory is row-major.  v1 (spconv.hip) transposed through LDS with 2-byte stores and two workgroup
// barriers per 64-row chunk per k and ran at ~15 % of the HBM roofline.  v2:
//   * every WAVE is an independent worker: it owns 32-row steps (interleaved across workers so
//     neighbouring waves stream neighbouring rows), stages them in a wave-private LDS slice and
//     never meets a workgroup barrier in the main loop;
//   * rows are stored row-major in LDS with plain 16-byte stores, as [16-channel plane][32 rows][16]
//     sub-tiles (pitch 32 B), and read back TRANSPOSED by ds_read_b64_tr_b16 (gfx950): a 16-lane
//     group addresses a [4 rows][16 channels] block and lane j receives rows 0..3 of channel j
//     (mapping measured with tools/probe_gfx950.hip) -- two reads make one MFMA fragment, and the
//     half-wave footprint (8 rows x 32 B) covers all 64 banks exactly once;
//   * the dout fragments of a step are read once and reused for all KG table rows of the group;
//     the gathered `in` rows of table row k+1 are in flight (registers) while k is multiplied;
//   * accumulators for KG table rows stay in registers for the whole row range; the four waves of a
//     workgroup are summed through LDS once at the end, then per-workgroup partials go to the
//     deterministic reduction kernel (no atomics, bit-reproducible).
#pragma once
#include "mma.h"
#include <stdlib.h>

typedef short s16x4_t __attribute__((ext_vector_type(4)));

#define W2_ROWS 32  // rows per wave step
// Byte stride between the 16-channel planes of an image.  32 rows x 32 B = 1024 B would put every plane on the SAME
// banks (1024 = 4 x 256 B): the eight lanes that store one row (pieces 0..7 = planes 0..3) then collide 4-way on every
// ds_write_b128 -- rocprofv3 PMC at the dec0 shape: SQ_LDS_BANK_CONFLICT = 100.8 M cycles, 44 % of the kernel's
// CU-cycles (profiles/r02_a_conv_pmc_s0.json).  +64 B rotates consecutive planes by 16 banks: the eight stores of a
// lane group land on distinct 16-byte bank quads for every COT / CIT in use; reads touch one plane per instruction
// and keep their conflict-free pattern.
#define W2_PLANE 1088

// LDS byte offset of (row, 8-channel piece) inside a wave's [planes][32][16] image
__device__ __forceinline__ int w2_off(int row, int piece) { return (piece >> 1) * W2_PLANE + (row << 5) + ((piece & 1) << 4); }

// one MFMA fragment (8 contraction values = rows {4g..4g+3, 16+4g..16+4g+3} of channel `lane&15` of plane t)
template <typename T>
__device__ __forceinline__ typename Mma<T>::frag w2_frag(const unsigned char* img, int t, int lane) {
  const int lp = lane & 15, g = lane >> 4;
  const int row = 4 * g + (lp >> 2);
  const unsigned char* p = img + t * W2_PLANE + (row << 5) + ((lp & 3) << 3);
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p + 16 * 32));
  s16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  typename Mma<T>::frag out;
  __builtin_memcpy(&out, &f, sizeof(out));
  return out;
}

// orders a wave's own LDS stores before its following (cross-lane) LDS reads
__device__ __forceinline__ void w2_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// PIPE (step-ahead prefetch): while a wave multiplies step s out of its LDS slice, the dout rows and the
// gathered rows of ALL KG table rows of its next step are in flight in registers (COT + KG*CIT 16-byte
// loads per lane), and the table entries of the step after that are being fetched.  v2 without it had one
// table row of gathers in flight and two dependent global latencies (entries -> rows) at every step:
// 56 TF/s useful at C = 64 (profiles/r01_h).  PIPE needs (COT + KG*CIT) KB of LDS per wave and
// 4*(COT + KG*CIT) registers; instances whose accumulators leave no room keep the in-step gathers.
template <int COT, int CIT, int KG>
struct W2Pipe { static constexpr bool value = KG * COT * CIT * 4 + 4 * (COT + KG * CIT) + 4 * (COT + CIT) <= 208; };

template <typename T, int COT, int CIT, int KG>
__global__ void __launch_bounds__(256, 2)   // two waves per SIMD: <= 256 registers
wgrad2_kernel(const T* __restrict__ in, const T* __restrict__ dout, const int32_t* __restrict__ nbr, int64_t n_out, int kv,
              int c_in, int c_out, int64_t steps_total, int ci_blocks, float* __restrict__ partial,
              float* __restrict__ bias_partial, int gx, int groups, int nblocks, uint32_t in_bytes, uint32_t dout_bytes) {
  using M = Mma<T>;
  const __amdgpu_buffer_rsrc_t in_buf = ptc_buf(in, in_bytes), dout_buf = ptc_buf(dout, dout_bytes);
  constexpr bool PIPE = W2Pipe<COT, CIT, KG>::value;
  constexpr int NI = PIPE ? KG : 2;                    // gathered-row images per wave
  constexpr int WAVE_BYTES = (COT + NI * CIT) * W2_PLANE;  // dout image + `in` images
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned char* D = smem + wave * WAVE_BYTES;
  unsigned char* I0 = D + COT * W2_PLANE;
  // 1-D grid, XCD-first numbering (workgroup b runs on XCD b % 8): logical id l = (row worker * blocks + channel
  // block) * groups + table-row group, so the `groups` workgroups that stream the SAME dout rows (and gather
  // overlapping neighbourhoods) run on one XCD at the same time and share its L2 -- dout left HBM once per group
  // (14x for a 3^3 table) in the (x, y, z)-grid form.
  const int total = gx * groups * nblocks;
  const int per_xcd = (total + 7) >> 3;
  const int lid = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (lid >= total) return;
  const int bgrp = lid % groups, bz = (lid / groups) % nblocks, bx = lid / (groups * nblocks);
  const int k0 = bgrp * KG;
  const int nk = (kv - k0) < KG ? (kv - k0) : KG;
  const int co0 = (bz / ci_blocks) * COT * 16, ci0 = (bz % ci_blocks) * CIT * 16;
  const int64_t workers = (int64_t)gx * 4, worker = (int64_t)bx * 4 + wave;
  // the fused bias gradient (one extra MFMA against ones per dout fragment) exists in the KG == 1 instances
  // only -- the host plans KG = 1 whenever dbias is requested; grouped instances have no registers to spare
  const bool do_bias = KG == 1 && bias_partial != nullptr && bgrp == 0 && (bz % ci_blocks) == 0;

  f32x4 acc[KG][COT][CIT];
  f32x4 accb[COT];
#pragma unroll
  for (int kk = 0; kk < KG; ++kk)
#pragma unroll
    for (int a = 0; a < COT; ++a)
#pragma unroll
      for (int b = 0; b < CIT; ++b) acc[kk][a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < COT; ++a) accb[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
  typename M::frag ones;
  {
    uint16_t one = std::is_same<T, bf16_t>::value ? 0x3F80 : 0x3C00;
    uint16_t o8[8] = {one, one, one, one, one, one, one, one};
    __builtin_memcpy(&ones, o8, sizeof(ones));
  }

  // All prefetch loads are UNCONDITIONAL: feature rows through raw buffer loads (absent rows = out-of-range offsets =
  // zeros, mma.h), table entries from clamped addresses.  A load under an exec-masked branch makes the compiler fall
  // back to s_waitcnt vmcnt(0) at every use, which serialised the step-ahead prefetch against the MFMAs it was
  // meant to overlap (ISA of r01_aj).
  // table entries of the rows this lane stages, for every table row of the group
  auto load_idx = [&](int64_t s, int32_t (&ix)[KG][CIT]) {
    const int64_t r0 = s * W2_ROWS;
#pragma unroll
    for (int kk = 0; kk < KG; ++kk)
#pragma unroll
      for (int i = 0; i < CIT; ++i) {
        const int row = (i * 64 + lane) / (2 * CIT);
        const int64_t rr = r0 + row;
        const bool ok = kk < nk && rr < n_out && s < steps_total;
        int32_t j;
        if (nbr) {   // wave-uniform
          const int kc = kk < nk ? kk : nk - 1;
          j = nbr[(int64_t)(k0 + kc) * n_out + (rr < n_out ? rr : n_out - 1)];
        } else {
          j = (int32_t)rr;
        }
        ix[kk][i] = ok ? j : -1;
      }
  };
  auto load_dout = [&](int64_t s, uint4 (&pd)[COT]) {
    const int64_t r0 = s * W2_ROWS;
#pragma unroll
    for (int i = 0; i < COT; ++i) {
      const int v = i * 64 + lane, row = v / (2 * COT), piece = v % (2 * COT);
      const int64_t rr = r0 + row;
      const int ch = co0 + piece * 8;
      const bool ok = rr < n_out && ch < c_out && s < steps_total;
      pd[i] = ptc_buf_load16(dout_buf, ok ? ((uint32_t)rr * (uint32_t)c_out + (uint32_t)ch) * 2u : PTC_BUF_OOB);
    }
  };
  auto store_dout = [&](const uint4 (&pd)[COT]) {
#pragma unroll
    for (int i = 0; i < COT; ++i) {
      const int v = i * 64 + lane, row = v / (2 * COT), piece = v % (2 * COT);
      *reinterpret_cast<uint4*>(D + w2_off(row, piece)) = pd[i];
    }
  };
  auto gather = [&](const int32_t (&ixk)[CIT], uint4 (&pi)[CIT]) {
#pragma unroll
    for (int i = 0; i < CIT; ++i) {
      const int piece = (i * 64 + lane) % (2 * CIT);
      const int ch = ci0 + piece * 8;
      const bool ok = ixk[i] >= 0 && ch < c_in;
      pi[i] = ptc_buf_load16(in_buf, ok ? ((uint32_t)ixk[i] * (uint32_t)c_in + (uint32_t)ch) * 2u : PTC_BUF_OOB);
    }
  };
  auto store_in = [&](unsigned char* buf, const uint4 (&pi)[CIT]) {
#pragma unroll
    for (int i = 0; i < CIT; ++i) {
      const int v = i * 64 + lane, row = v / (2 * CIT), piece = v % (2 * CIT);
      *reinterpret_cast<uint4*>(buf + w2_off(row, piece)) = pi[i];
    }
  };

  if constexpr (PIPE) {
    int32_t idx[KG][CIT];
    uint4 pd[COT], pi[KG][CIT];
    load_idx(worker, idx);
    load_dout(worker, pd);
#pragma unroll
    for (int kk = 0; kk < KG; ++kk) gather(idx[kk], pi[kk]);
    load_idx(worker + workers, idx);
    for (int64_t s = worker; s < steps_total; s += workers) {
      // 1. step s lands in the wave's LDS slice
      store_dout(pd);
#pragma unroll
      for (int kk = 0; kk < KG; ++kk)
        if (kk < nk) store_in(I0 + kk * CIT * W2_PLANE, pi[kk]);
      // 2. step s + workers goes out, entries of the one after it too
      load_dout(s + workers, pd);
#pragma unroll
      for (int kk = 0; kk < KG; ++kk) gather(idx[kk], pi[kk]);
      load_idx(s + 2 * workers, idx);
      // 3. multiply step s
      w2_wave_sync();
      typename M::frag A[COT];
#pragma unroll
      for (int a = 0; a < COT; ++a) A[a] = w2_frag<T>(D, a, lane);
      if constexpr (KG == 1) {
        if (do_bias) {
#pragma unroll
          for (int a = 0; a < COT; ++a) accb[a] = M::mma(A[a], ones, accb[a]);
        }
      }
#pragma unroll
      for (int kk = 0; kk < KG; ++kk) {
        if (kk < nk) {
          typename M::frag B[CIT];
#pragma unroll
          for (int b = 0; b < CIT; ++b) B[b] = w2_frag<T>(I0 + kk * CIT * W2_PLANE, b, lane);
#pragma unroll
          for (int a = 0; a < COT; ++a)
#pragma unroll
            for (int b = 0; b < CIT; ++b) acc[kk][a][b] = M::mma(A[a], B[b], acc[kk][a][b]);
        }
      }
      w2_wave_sync();  // the slice is rewritten at the top of the next trip
    }
  } else {
    unsigned char* I[2] = {I0, I0 + CIT * W2_PLANE};
    for (int64_t s = worker; s < steps_total; s += workers) {
      int32_t idx[KG][CIT];
      load_idx(s, idx);
      uint4 pd[COT];
      load_dout(s, pd);
      store_dout(pd);
      uint4 pre[CIT];
      gather(idx[0], pre);
      w2_wave_sync();
      typename M::frag A[COT];
#pragma unroll
      for (int a = 0; a < COT; ++a) A[a] = w2_frag<T>(D, a, lane);
      if constexpr (KG == 1) {
        if (do_bias) {
#pragma unroll
          for (int a = 0; a < COT; ++a) accb[a] = M::mma(A[a], ones, accb[a]);
        }
      }
#pragma unroll
      for (int kk = 0; kk < KG; ++kk) {
        if (kk < nk) {
          unsigned char* buf = I[kk & 1];
          store_in(buf, pre);
          if (kk + 1 < KG && kk + 1 < nk) gather(idx[(kk + 1) < KG ? (kk + 1) : 0], pre);  // next table row goes out first
          w2_wave_sync();
          typename M::frag B[CIT];
#pragma unroll
          for (int b = 0; b < CIT; ++b) B[b] = w2_frag<T>(buf, b, lane);
#pragma unroll
          for (int a = 0; a < COT; ++a)
#pragma unroll
            for (int b = 0; b < CIT; ++b) acc[kk][a][b] = M::mma(A[a], B[b], acc[kk][a][b]);
        }
      }
    }
  }

  // ---- sum the four waves through LDS, write this workgroup's partial ------------------------------
  // D[i = co][j = ci]: lane (j = lane & 15, g = lane >> 4) holds co = 16 a + 4 g + e, ci = 16 b + j
  float* red = reinterpret_cast<float*>(smem);  // [4 waves][CIT][4][64] floats <= 16 KB
  float* pout = partial + (int64_t)bx * c_out * kv * c_in;
#pragma unroll
  for (int kk = 0; kk < KG; ++kk) {
#pragma unroll
    for (int a = 0; a < COT; ++a) {
      __syncthreads();
      if (kk < nk) {
#pragma unroll
        for (int b = 0; b < CIT; ++b)
#pragma unroll
          for (int e = 0; e < 4; ++e) red[((wave * CIT + b) * 4 + e) * 64 + lane] = acc[kk][a][b][e];
      }
      __syncthreads();
      if (kk < nk) {
#pragma unroll
        for (int i = 0; i < CIT; ++i) {
          const int q = i * 256 + threadIdx.x;  // (b, e, lane)
          const int ln = q & 63, e = (q >> 6) & 3, b = q >> 8;
          const float v = red[q] + red[CIT * 256 + q] + red[2 * CIT * 256 + q] + red[3 * CIT * 256 + q];
          const int co = co0 + 16 * a + 4 * (ln >> 4) + e, ci = ci0 + 16 * b + (ln & 15);
          if (co < c_out && ci < c_in) pout[((int64_t)co * kv + (k0 + kk)) * c_in + ci] = v;
        }
      }
    }
  }
  if (do_bias) {  // every column j of accb holds the same column sum; take j = 0
    __syncthreads();
    if ((lane & 15) == 0) {
#pragma unroll
      for (int a = 0; a < COT; ++a)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave * COT * 16 + 16 * a + 4 * (lane >> 4) + e] = accb[a][e];
    }
    __syncthreads();
    if ((int)threadIdx.x < COT * 16) {
      const int co = co0 + threadIdx.x;
      if (co < c_out)
        bias_partial[(int64_t)bx * c_out + co] = red[threadIdx.x] + red[COT * 16 + threadIdx.x] +
                                                         red[2 * COT * 16 + threadIdx.x] + red[3 * COT * 16 + threadIdx.x];
    }
  }
}

// ---- host-side plan ----------------------------------------------------------------------------
struct W2Plan {
  int cot, cit, kg;          // tiles per workgroup (x16 channels), table rows per group
  int co_blocks, ci_blocks, groups;
  int gx;                    // workgroups along the row axis (= number of partials)
  size_t lds;
};

static inline W2Plan w2_plan(int64_t n_out, int kv, int c_in, int c_out, bool want_bias = false) {
  // workgroups aimed at per launch and 32-row steps a workgroup must at least own (swept for the small weight gradients of the deep
  // stages in round 2: 1024 / 16 kept)
  constexpr int cot_max = 8, cit_max = 4, target_wgs = 1024, min_steps = 16;
  W2Plan p;
  // channel tiles: 64x64 accumulators by default; channel counts that are multiples of 32 but not of 64
  // (SpUNet's 96-channel decoder) take 32-wide input tiles / a 96-wide output tile so that no MFMA runs on padding
  p.cit = c_in <= 16 ? 1 : (c_in <= 32 ? 2 : (c_in % 64 == 0 ? 4 : 2));
  p.cot = c_out <= 32 ? 2 : (c_out <= 64 ? 4 : (c_out <= 96 ? 6 : 8));
  if (c_out > 96 && c_out % 96 == 0 && c_out % 64 != 0) p.cot = 6;
  if (p.cot == 6) p.cit = p.cit > 2 ? 2 : p.cit;
  if (p.cit == 4 && p.cot > 4) p.cot = 4;   // 64x64 accumulators: the wider tiles spill under the 256-register cap
  if (p.cot > cot_max) p.cot = cot_max;
  if (p.cit > cit_max) p.cit = cit_max;
  p.kg = 1;
  if (kv > 1 && !want_bias) {  // instantiated groups (register budget: KG*COT*CIT*4 accumulators)
    if (p.cot == 2 && p.cit == 1) p.kg = 16;
    else if (p.cot == 2 && p.cit == 2) p.kg = 4;   // (2,2,9) cannot hold the step-ahead prefetch in registers
    else if (p.cot == 4 && p.cit == 2) p.kg = 4;
    else if (p.cot == 2 && p.cit == 4) p.kg = 4;
    else if (p.cot == 4 && p.cit == 4) p.kg = 2;
    else if (p.cot == 6 && p.cit == 2) p.kg = 2;
  }
  p.co_blocks = (int)ptc_cdiv(c_out, p.cot * 16);
  p.ci_blocks = (int)ptc_cdiv(c_in, p.cit * 16);
  p.groups = (int)ptc_cdiv(kv, p.kg);
  const int64_t steps = ptc_cdiv(n_out, W2_ROWS);
  int64_t gx = (int64_t)target_wgs / ((int64_t)p.groups * p.co_blocks * p.ci_blocks);
  const int64_t max_gx = ptc_cdiv(steps, min_steps > 0 ? min_steps : 16);  // at least ~4 steps per wave
  if (gx > max_gx) gx = max_gx;
  if (gx > 512) gx = 512;
  if (gx < 1) gx = 1;
  p.gx = (int)gx;
  const bool pipe = p.kg * p.cot * p.cit * 4 + 4 * (p.cot + p.kg * p.cit) + 4 * (p.cot + p.cit) <= 208;   // = W2Pipe<cot,cit,kg>
  p.lds = (size_t)4 * (p.cot + (pipe ? p.kg : 2) * p.cit) * W2_PLANE;  // >= the 4*CIT KB of the final cross-wave sum
  return p;
}
