// attention_rpe.h -- window attention with PTv3's relative position bias, head_dim 16 (included by attention.hip).
//
// the reference's non-flash branch with enable_rpe=True
//   pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py:29-48 (RPE), :104-112 (get_rel_pos), :190-206:
//     attn = (q * scale) @ k^T + rpe(rel_pos);  softmax;  attn @ v
//     rel_pos[p, i, j, a] = grid_coord[i][a] - grid_coord[j][a]            (i = query, j = key, rows in serialized order)
//     rpe[p, h, i, j] = sum_a table[a * R + clamp(rel_pos, -B, B) + B][h]   (R = 2 B + 1, B = int((4 K)^(1/3) * 2))
// materialises [P, H, K, K] in fp32 (4 MB per patch-head at K = 1024) plus a [P, K, K, 3] int64 index tensor.  Here the bias is
// evaluated per (query, key) pair inside the tile loop from the two rows' packed coordinates (LDS) and the head's 3 R table
// entries (LDS, pre-multiplied by log2 e): nothing of size K^2 ever exists.  Same operand layout, LDS images and MFMA products as
// attention.hip; the loops are the plain in-order forms (online softmax in the forward: the bias moves the row maximum, so the
// norm bound of the fast path does not apply).  The table gradient d table[a R + idx][h] += dS[i][j] is accumulated in 2^-24 FIXED
// POINT: 64-bit integer atomics in LDS per workgroup (ds_add_u64 runs at full rate; ds_add_f32 is executed lane by lane on gfx950
// -- the float version of this kernel took 58 ms at the dec0 shape), one 64-bit global atomic per entry at the end, one tiny
// conversion launch.  Integer sums do not depend on the order of the additions: the table gradient is bit-reproducible too
// (the reference's index_select backward, atomicAdd on floats, is not).
// Per pair the bias costs ~14 VALU + 3 LDS reads against ~1.6 VALU for the rest of the tile: this branch is ~10x slower than
// the flash branch by construction, and ~K^2-memory-free, which is what makes K = 1024 with RPE runnable at all.

#define AR_THREADS AT_THREADS

// packed coordinates of a row: .x = x | y << 16, .y = z   (grid coordinates are < 2^16: serialization depth <= 16, structure.py:77)
__device__ __forceinline__ uint2 ar_pack(const int32_t* __restrict__ gc, int64_t row, bool valid) {
  uint2 c = {0u, 0u};
  if (valid) {
    const int32_t x = gc[row * 3], y = gc[row * 3 + 1], z = gc[row * 3 + 2];
    c.x = ((uint32_t)x & 0xffffu) | ((uint32_t)y << 16);
    c.y = (uint32_t)z;
  }
  return c;
}
// bias (exp2 domain) of query coordinates (qx, qy, qz) against the packed key coordinates kc;  tb = table + B (LDS, [3][R])
__device__ __forceinline__ float ar_bias(int qx, int qy, int qz, uint2 kc, const float* tb, int R, int B, int& ix, int& iy, int& iz) {
  const int kx = (int)(kc.x & 0xffffu), ky = (int)(kc.x >> 16), kz = (int)kc.y;
  ix = min(max(qx - kx, -B), B);
  iy = min(max(qy - ky, -B), B) + R;
  iz = min(max(qz - kz, -B), B) + 2 * R;
  return tb[ix] + tb[iy] + tb[iz];
}

#define AR_FIX_SCALE 16777216.f            // 2^24: resolution 6e-8, range +-5e11 per table entry
static size_t ar_extra_lds(int lp_max, int R) { return (size_t)lp_max * 8 + (size_t)((3 * R + 3) & ~3) * 4 * 3; }   // coords | table f32 | d table i64

// stage the coordinates of rows [0, Lp) and the head's table column (scaled by log2 e; second copy zeroed = gradient accumulator)
__device__ __forceinline__ void ar_stage(const int32_t* __restrict__ gc, int64_t a, int L, int Lp, const float* __restrict__ table,
                                         int H, int head, int R, uint2* coords, float* tl, unsigned long long* dtl) {
  for (int row = threadIdx.x; row < Lp; row += AR_THREADS) coords[row] = ar_pack(gc, a + row, row < L);
  for (int i = threadIdx.x; i < 3 * R; i += AR_THREADS) {
    tl[i] = table[(int64_t)i * H + head] * AT_LOG2E;
    if (dtl) dtl[i] = 0ull;
  }
}

// ------------------------------------------------------------------------------------------------ forward
// LDS: K row-major | V^T [17][pitch] | coords [lp_max] uint2 | table [3R] | (unused second table copy)
template <bool F16>
__global__ void __launch_bounds__(AR_THREADS, 2)
attn_rpe_fwd_kernel(const uint16_t* __restrict__ qkv, const int32_t* __restrict__ cu, const int32_t* __restrict__ gc,
                    const float* __restrict__ table, int R, int B, int H, float scale, int64_t total, int lp_max, int n_units,
                    uint16_t* __restrict__ out, float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int unit = at_unit(n_units);
  if (unit >= n_units) return;
  const int seq = unit / H, head = unit % H;
  const int a = cu[seq], L = cu[seq + 1] - a;
  if (L <= 0) return;
  const int Lp = (L + 31) & ~31, n_tiles = Lp >> 5;
  if (Lp > lp_max) {
    at_poison_rows<F16>(out + ((int64_t)a * H + head) * 16, (int64_t)H * 16, L, lse + (int64_t)head * total + a);
    return;
  }
  const int pitch = lp_max + 8;
  unsigned char* Ksm = smem;
  unsigned char* Vt = smem + (size_t)lp_max * 32;
  uint2* coords = reinterpret_cast<uint2*>(Vt + (size_t)17 * pitch * 2 + 64);
  float* tl = reinterpret_cast<float*>(coords + lp_max);
  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  const int64_t rs = (int64_t)3 * H * 16;
  stage_row_major<F16>(qkv + qkv_off(a, 1, H, head), rs, L, Lp, Ksm);
  stage_transposed<F16>(qkv + qkv_off(a, 2, H, head), rs, L, Lp, pitch, Vt);
  for (int key = threadIdx.x; key < Lp; key += AR_THREADS)
    reinterpret_cast<uint16_t*>(Vt + (size_t)16 * pitch * 2)[vt_pos(key)] = key < L ? (uint16_t)0x3F80 : (uint16_t)0;
  ar_stage(gc, a, L, Lp, table, H, head, R, coords, tl, nullptr);
  __syncthreads();

  const int col = lane & 31, h2 = lane >> 5;
  const float c = scale * AT_LOG2E;
  const float* tb = tl + B;
  const unsigned char* vbase = Vt + ((size_t)(col <= 16 ? col : (col & 15)) * pitch + 8 * h2) * 2;
  const unsigned char* kbase = Ksm + rm_off(col, h2);
  for (int qt = wave; qt < n_tiles; qt += AT_WAVES) {
    const int q = qt * 32 + col;
    const s16x8 qf = ld_global_frag<F16>(qkv + qkv_off(a + q, 0, H, head) + h2 * 8, q < L);
    const uint2 qc = coords[q];
    const int qx = (int)(qc.x & 0xffffu), qy = (int)(qc.x >> 16), qz = (int)qc.y;
    s16x8 qhi, qlo;
    split_scaled(qf, c, qhi, qlo);
    f32x16 acc = zero16();
    float m = -INFINITY;
    for (int kt = 0; kt < n_tiles; ++kt) {
      const s16x8 kf = *reinterpret_cast<const s16x8*>(kbase + kt * 1024);
      f32x16 s = mfma32(kf, qhi, zero16());
      s = mfma32(kf, qlo, s);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 32 + crow(r, h2);
        int ix, iy, iz;
        s[r] += ar_bias(qx, qy, qz, coords[key], tb, R, B, ix, iy, iz);
        if (key >= L) s[r] = -INFINITY;
      }
      float mt = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[r]);
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      const float m_new = fmaxf(m, mt);
      const float alpha = __builtin_amdgcn_exp2f(m - m_new);
      m = m_new;
#pragma unroll
      for (int r = 0; r < 9; ++r) acc[r] *= alpha;
      uint32_t pk[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        pk[i] = pack_bf16x2(__builtin_amdgcn_exp2f(s[2 * i] - m), __builtin_amdgcn_exp2f(s[2 * i + 1] - m));
      acc = mfma32(*reinterpret_cast<const s16x8*>(vbase + kt * 64), make_frag(pk[0], pk[1], pk[2], pk[3]), acc);
      acc = mfma32(*reinterpret_cast<const s16x8*>(vbase + kt * 64 + 32), make_frag(pk[4], pk[5], pk[6], pk[7]), acc);
    }
    const float l = __shfl(acc[8], col, 64);
    const float inv = 1.f / l;
    if (q < L) {
      uint16_t* o = out + ((int64_t)(a + q) * H + head) * 16;
      uint2 w0, w1;
      w0.x = at_out<F16>(pack_bf16x2(acc[0] * inv, acc[1] * inv)); w0.y = at_out<F16>(pack_bf16x2(acc[2] * inv, acc[3] * inv));
      w1.x = at_out<F16>(pack_bf16x2(acc[4] * inv, acc[5] * inv)); w1.y = at_out<F16>(pack_bf16x2(acc[6] * inv, acc[7] * inv));
      *reinterpret_cast<uint2*>(o + 4 * h2) = w0;
      *reinterpret_cast<uint2*>(o + 8 + 4 * h2) = w1;
      if (h2 == 0) lse[(int64_t)head * total + a + q] = m * AT_LN2 + __logf(l);
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward: dQ, delta, d table
// LDS: V row-major | K row-major | coords | table | d table
template <bool F16>
__global__ void __launch_bounds__(AR_THREADS, 2)
attn_rpe_bwd_dq_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ out, const uint16_t* __restrict__ dout,
                       const float* __restrict__ lse, const int32_t* __restrict__ cu, const int32_t* __restrict__ gc,
                       const float* __restrict__ table, int R, int B, int H, float scale, int64_t total, int lp_max, int n_units,
                       uint16_t* __restrict__ dqkv, float* __restrict__ delta, unsigned long long* __restrict__ dtable_fix) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int unit = at_unit(n_units);
  if (unit >= n_units) return;
  const int seq = unit / H, head = unit % H;
  const int a = cu[seq], L = cu[seq + 1] - a;
  if (L <= 0) return;
  const int Lp = (L + 31) & ~31, n_tiles = Lp >> 5;
  if (Lp > lp_max) {
    at_poison_rows<F16>(dqkv + qkv_off(a, 0, H, head), (int64_t)3 * H * 16, L, nullptr);
    return;
  }
  unsigned char* Vsm = smem;
  unsigned char* Ksm = smem + (size_t)lp_max * 32;
  uint2* coords = reinterpret_cast<uint2*>(smem + (size_t)lp_max * 64);
  float* tl = reinterpret_cast<float*>(coords + lp_max);
  unsigned long long* dtl = reinterpret_cast<unsigned long long*>(tl + ((3 * R + 3) & ~3));   // 16-byte aligned: lp_max * 72 + 16 k
  const int64_t rs = (int64_t)3 * H * 16;
  stage_row_major<F16>(qkv + qkv_off(a, 2, H, head), rs, L, Lp, Vsm);
  stage_row_major<F16>(qkv + qkv_off(a, 1, H, head), rs, L, Lp, Ksm);
  ar_stage(gc, a, L, Lp, table, H, head, R, coords, tl, dtl);
  __syncthreads();

  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  const int col = lane & 31, h2 = lane >> 5;
  const float c = scale * AT_LOG2E;
  const TrAddr ta = tr_addr(lane);
  const int rmo = rm_off(col, h2);
  const float* tb = tl + B;
  unsigned long long* dtb = dtl + B;
  for (int qt = wave; qt < n_tiles; qt += AT_WAVES) {
    const int q = qt * 32 + col;
    const bool qv = q < L;
    const s16x8 qf = ld_global_frag<F16>(qkv + qkv_off(a + q, 0, H, head) + h2 * 8, qv);
    const int64_t orow = ((int64_t)(a + q) * H + head) * 16 + h2 * 8;
    const s16x8 dof = ld_global_frag<F16>(dout + orow, qv);
    const s16x8 of = ld_global_frag<F16>(out + orow, qv);
    float dl = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) dl += bf16_bits_to_float((uint16_t)dof[j]) * bf16_bits_to_float((uint16_t)of[j]);
    dl += __shfl_xor(dl, 32, 64);
    const float l2 = qv ? lse[(int64_t)head * total + a + q] * AT_LOG2E : INFINITY;
    if (qv && h2 == 0) delta[(int64_t)head * total + a + q] = dl;
    const uint2 qc = coords[q];
    const int qx = (int)(qc.x & 0xffffu), qy = (int)(qc.x >> 16), qz = (int)qc.y;
    s16x8 qhi, qlo;
    split_scaled(qf, c, qhi, qlo);
    const f32x16 negl = splat16(-l2), negd = splat16(-dl);
    f32x16 acc = zero16();
    for (int kt = 0; kt < n_tiles; ++kt) {
      const s16x8 kf = *reinterpret_cast<const s16x8*>(Ksm + kt * 1024 + rmo);
      const s16x8 vf = *reinterpret_cast<const s16x8*>(Vsm + kt * 1024 + rmo);
      f32x16 s = mfma32(kf, qhi, negl);
      s = mfma32(kf, qlo, s);
      const f32x16 dp = mfma32(vf, dof, negd);
      float ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 32 + crow(r, h2);
        int ix, iy, iz;
        const float b = ar_bias(qx, qy, qz, coords[key], tb, R, B, ix, iy, iz);
        // keys >= L: k = 0 and v = 0 give a finite P; it must not reach the table gradient (dQ is safe: it multiplies K = 0)
        ds[r] = (qv && key < L) ? __builtin_amdgcn_exp2f(s[r] + b) * dp[r] : 0.f;
        const unsigned long long fx = (unsigned long long)__float2ll_rn(ds[r] * AR_FIX_SCALE);   // two's complement: adds of negatives wrap correctly
        if (fx != 0ull) {
          atomicAdd(dtb + ix, fx);
          atomicAdd(dtb + iy, fx);
          atomicAdd(dtb + iz, fx);
        }
      }
#pragma unroll
      for (int mm = 0; mm < 2; ++mm) {
        const s16x8 dsf = make_frag(pack_bf16x2(ds[8 * mm], ds[8 * mm + 1]), pack_bf16x2(ds[8 * mm + 2], ds[8 * mm + 3]),
                                    pack_bf16x2(ds[8 * mm + 4], ds[8 * mm + 5]), pack_bf16x2(ds[8 * mm + 6], ds[8 * mm + 7]));
        acc = mfma32(ld_tr_frag(Ksm, ta, kt * 32 + 16 * mm), dsf, acc);
      }
    }
    if (qv) {
      uint16_t* o = dqkv + qkv_off(a + q, 0, H, head);
      uint2 w0, w1;
      w0.x = at_out<F16>(pack_bf16x2(acc[0] * scale, acc[1] * scale)); w0.y = at_out<F16>(pack_bf16x2(acc[2] * scale, acc[3] * scale));
      w1.x = at_out<F16>(pack_bf16x2(acc[4] * scale, acc[5] * scale)); w1.y = at_out<F16>(pack_bf16x2(acc[6] * scale, acc[7] * scale));
      *reinterpret_cast<uint2*>(o + 4 * h2) = w0;
      *reinterpret_cast<uint2*>(o + 8 + 4 * h2) = w1;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * R; i += AR_THREADS)
    if (dtl[i] != 0ull) atomicAdd(dtable_fix + (int64_t)i * H + head, dtl[i]);
}

__global__ void attn_rpe_table_finish_kernel(const unsigned long long* __restrict__ fix, int64_t n, float* __restrict__ dtable) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dtable[i] = (float)((double)(long long)fix[i] * (1.0 / (double)AR_FIX_SCALE));
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
// LDS: Q row-major | dO row-major | aux | coords | table
template <bool F16>
__global__ void __launch_bounds__(AR_THREADS, 2)
attn_rpe_bwd_dkv_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dout, const float* __restrict__ lse,
                        const float* __restrict__ delta, const int32_t* __restrict__ cu, const int32_t* __restrict__ gc,
                        const float* __restrict__ table, int R, int B, int H, float scale, int64_t total, int lp_max, int n_units,
                        uint16_t* __restrict__ dqkv) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int unit = at_unit(n_units);
  if (unit >= n_units) return;
  const int seq = unit / H, head = unit % H;
  const int a = cu[seq], L = cu[seq + 1] - a;
  if (L <= 0) return;
  const int Lp = (L + 31) & ~31, n_tiles = Lp >> 5;
  if (Lp > lp_max) {
    at_poison_rows<F16>(dqkv + qkv_off(a, 1, H, head), (int64_t)3 * H * 16, L, nullptr);
    at_poison_rows<F16>(dqkv + qkv_off(a, 2, H, head), (int64_t)3 * H * 16, L, nullptr);
    return;
  }
  unsigned char* Qsm = smem;
  unsigned char* dOsm = smem + (size_t)lp_max * 32;
  uint2* aux = reinterpret_cast<uint2*>(smem + (size_t)lp_max * 64);
  uint2* coords = aux + lp_max;
  float* tl = reinterpret_cast<float*>(coords + lp_max);
  stage_row_major<F16>(qkv + qkv_off(a, 0, H, head), (int64_t)3 * H * 16, L, Lp, Qsm);
  stage_row_major<F16>(dout + ((int64_t)a * H + head) * 16, (int64_t)H * 16, L, Lp, dOsm);
  for (int q = threadIdx.x; q < Lp; q += AR_THREADS) {
    const float l2 = q < L ? lse[(int64_t)head * total + a + q] * AT_LOG2E : AT_PAD_LSE;
    const float dl = q < L ? delta[(int64_t)head * total + a + q] : 0.f;
    const uint32_t hi = pack_bf16x2(l2, dl);
    const uint32_t lo = pack_bf16x2(l2 - __uint_as_float(hi << 16), dl - __uint_as_float(hi & 0xffff0000u));
    uint2 w;
    w.x = (hi & 0xffffu) | (lo << 16);
    w.y = (hi >> 16) | (lo & 0xffff0000u);
    aux[q] = w;
  }
  ar_stage(gc, a, L, Lp, table, H, head, R, coords, tl, nullptr);
  __syncthreads();

  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  const int col = lane & 31, h2 = lane >> 5;
  const float c = scale * AT_LOG2E;
  const TrAddr ta = tr_addr(lane);
  const int rmo = rm_off(col, h2);
  const uint32_t m1 = 0xBF80BF80u;
  const s16x8 bS = make_frag(h2 == 0 ? m1 : 0u, 0u, 0u, 0u);
  const s16x8 bD = make_frag(0u, h2 == 0 ? m1 : 0u, 0u, 0u);
  const float* tb = tl + B;
  for (int kt = wave; kt < n_tiles; kt += AT_WAVES) {
    const int key = kt * 32 + col;
    const s16x8 kf = ld_global_frag<F16>(qkv + qkv_off(a + key, 1, H, head) + h2 * 8, key < L);
    const s16x8 vf = ld_global_frag<F16>(qkv + qkv_off(a + key, 2, H, head) + h2 * 8, key < L);
    const uint2 kc = coords[key];
    s16x8 khi, klo;
    split_scaled(kf, c, khi, klo);
    f32x16 dv = zero16(), dk = zero16();
    for (int qt = 0; qt < n_tiles; ++qt) {
      const s16x8 qf = *reinterpret_cast<const s16x8*>(Qsm + qt * 1024 + rmo);
      const s16x8 dof = *reinterpret_cast<const s16x8*>(dOsm + qt * 1024 + rmo);
      const uint2 ax = aux[qt * 32 + col];
      const s16x8 af = make_frag(ax.x, ax.y, 0u, 0u);
      f32x16 s = mfma32(af, bS, zero16());
      s = mfma32(qf, khi, s);
      s = mfma32(qf, klo, s);
      f32x16 dp = mfma32(af, bD, zero16());
      dp = mfma32(dof, vf, dp);
      float p[16], ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint2 qc = coords[qt * 32 + crow(r, h2)];                  // the QUERY of this register; the lane's key is kc
        int ix, iy, iz;
        const float b = ar_bias((int)(qc.x & 0xffffu), (int)(qc.x >> 16), (int)qc.y, kc, tb, R, B, ix, iy, iz);
        p[r] = __builtin_amdgcn_exp2f(s[r] + b);                          // padding queries: lse = 1e30 -> 0
        ds[r] = p[r] * dp[r];
      }
#pragma unroll
      for (int mm = 0; mm < 2; ++mm) {
        const s16x8 pf = make_frag(pack_bf16x2(p[8 * mm], p[8 * mm + 1]), pack_bf16x2(p[8 * mm + 2], p[8 * mm + 3]),
                                   pack_bf16x2(p[8 * mm + 4], p[8 * mm + 5]), pack_bf16x2(p[8 * mm + 6], p[8 * mm + 7]));
        const s16x8 dsf = make_frag(pack_bf16x2(ds[8 * mm], ds[8 * mm + 1]), pack_bf16x2(ds[8 * mm + 2], ds[8 * mm + 3]),
                                    pack_bf16x2(ds[8 * mm + 4], ds[8 * mm + 5]), pack_bf16x2(ds[8 * mm + 6], ds[8 * mm + 7]));
        dv = mfma32(pf, ld_tr_frag(dOsm, ta, qt * 32 + 16 * mm), dv);
        dk = mfma32(dsf, ld_tr_frag(Qsm, ta, qt * 32 + 16 * mm), dk);
      }
    }
    if (col < 16) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kk = kt * 32 + crow(r, h2);
        if (kk < L) {
          dqkv[qkv_off(a + kk, 1, H, head) + col] = (uint16_t)(at_out<F16>(pack_bf16x2(dk[r] * scale, 0.f)) & 0xffffu);
          dqkv[qkv_off(a + kk, 2, H, head) + col] = (uint16_t)(at_out<F16>(pack_bf16x2(dv[r], 0.f)) & 0xffffu);
        }
      }
    }
  }
}
