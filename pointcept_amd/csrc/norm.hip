// norm.hip -- LayerNorm over the channel axis of [N, C] point features, forward + backward.
//
// Replaces the nn.LayerNorm calls inside every PTv3 Block (pointcept/models/point_transformer_v3/
// point_transformer_v3m1_base.py:286 (cpe.2), :289 (norm1), :305 (norm2); 66 forward calls per
// PTv3-base step).  With C = 32..512 channels and ~8e5 rows the op is a pure HBM stream; ATen's
// row-per-block kernels reach ~0.5 TB/s on these shapes (profiles/r01_a_*), so each row is handled
// by C/8 lanes of a wave here (16-byte accesses, shuffle reductions inside the lane group, several
// rows per wave) and the statistics / affine-gradient partials never leave registers.
//   forward : y = (x - mean) * rstd * gamma + beta ; saves mean, rstd (fp32 per row)
//   backward: dx = rstd * (dy*gamma - mean_c(dy*gamma) - xhat * mean_c(dy*gamma*xhat))
//             dgamma = sum_rows dy * xhat, dbeta = sum_rows dy   (per-block partials + reduction)
// Roofline: forward (in + out) * N * C bytes, backward (dy + x + dx) * N * C bytes.
#include "ptc_common.h"

#include "ln_common.h"

// (round 5) `c` is a run-time argument: C = c <= LPR * 8, c % 8 == 0 -- a width that is no power of two runs on the next larger instance
// with the lanes past the row idle (C = 48: six lanes of eight, C = 384: 48 of 64; 16-byte accesses as before).  For c = LPR * 8 every
// statement computes what the compile-time form computed (1 / c is exact), so the bits of the PT-v3m1 widths are unchanged.
template <typename TI, typename TO, int LPR>
__global__ void __launch_bounds__(LN_THREADS)
layer_norm_fwd_kernel(const TI* __restrict__ x, int64_t n, int c, const float* __restrict__ gamma, const float* __restrict__ beta,
                      float eps, TO* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  constexpr int RPB = LN_THREADS / LPR;  // rows per block iteration
  const int slot = threadIdx.x % LPR, rib = threadIdx.x / LPR;
  const bool act = slot * LN_VEC < c;
  const float inv_c = 1.f / (float)c;
  float g[LN_VEC], b[LN_VEC];
#pragma unroll
  for (int i = 0; i < LN_VEC; ++i) { g[i] = (gamma && act) ? gamma[slot * LN_VEC + i] : 1.f; b[i] = (beta && act) ? beta[slot * LN_VEC + i] : 0.f; }
  for (int64_t row = (int64_t)blockIdx.x * RPB + rib; row < n; row += (int64_t)gridDim.x * RPB) {
    float v[LN_VEC];
#pragma unroll
    for (int i = 0; i < LN_VEC; ++i) v[i] = 0.f;
    if (act) ln_load8<TI>(x + row * c + slot * LN_VEC, v);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_VEC; ++i) s += v[i];
    const float mean = group_sum<LPR>(s) * inv_c;
    float q = 0.f;
    if (act) {
#pragma unroll
      for (int i = 0; i < LN_VEC; ++i) { const float d = v[i] - mean; q += d * d; }
    }
    const float rstd = rsqrtf(group_sum<LPR>(q) * inv_c + eps);
    float o[LN_VEC];
#pragma unroll
    for (int i = 0; i < LN_VEC; ++i) o[i] = (v[i] - mean) * rstd * g[i] + b[i];
    if (act) ln_store8<TO>(y + row * c + slot * LN_VEC, o);
    if (slot == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
  }
}

template <typename TG, typename TX, int LPR>
__global__ void __launch_bounds__(LN_THREADS)
layer_norm_bwd_kernel(const TG* __restrict__ dy, const TX* __restrict__ x, const float* __restrict__ mean,
                      const float* __restrict__ rstd, const float* __restrict__ gamma, int64_t n, int c, TX* __restrict__ dx,
                      float* __restrict__ partial /*[grid][2][c]*/) {
  constexpr int RPB = LN_THREADS / LPR;
  __shared__ float red[2][LN_THREADS][LN_VEC + 1];
  const int slot = threadIdx.x % LPR, rib = threadIdx.x / LPR;
  const bool act = slot * LN_VEC < c;
  const float inv_c = 1.f / (float)c;
  float g[LN_VEC], dg[LN_VEC], db[LN_VEC];
#pragma unroll
  for (int i = 0; i < LN_VEC; ++i) { g[i] = (gamma && act) ? gamma[slot * LN_VEC + i] : 1.f; dg[i] = 0.f; db[i] = 0.f; }
  for (int64_t row = (int64_t)blockIdx.x * RPB + rib; row < n; row += (int64_t)gridDim.x * RPB) {
    float xv[LN_VEC], gv[LN_VEC];
#pragma unroll
    for (int i = 0; i < LN_VEC; ++i) { xv[i] = 0.f; gv[i] = 0.f; }
    if (act) {
      ln_load8<TX>(x + row * c + slot * LN_VEC, xv);
      ln_load8<TG>(dy + row * c + slot * LN_VEC, gv);
    }
    const float m = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f, xh[LN_VEC], w[LN_VEC];
#pragma unroll
    for (int i = 0; i < LN_VEC; ++i) {
      xh[i] = act ? (xv[i] - m) * rs : 0.f;
      w[i] = gv[i] * g[i];
      s1 += w[i] * xh[i];
      s2 += w[i];
      dg[i] += gv[i] * xh[i];
      db[i] += gv[i];
    }
    const float c1 = group_sum<LPR>(s1) * inv_c, c2 = group_sum<LPR>(s2) * inv_c;
    float o[LN_VEC];
#pragma unroll
    for (int i = 0; i < LN_VEC; ++i) o[i] = (w[i] - c2 - xh[i] * c1) * rs;
    if (act) ln_store8<TX>(dx + row * c + slot * LN_VEC, o);
  }
  // block reduction of the affine-gradient partials: threads with equal `slot` (RPB of them)
#pragma unroll
  for (int i = 0; i < LN_VEC; ++i) { red[0][threadIdx.x][i] = dg[i]; red[1][threadIdx.x][i] = db[i]; }
  __syncthreads();
  for (int t = threadIdx.x; t < 2 * c; t += LN_THREADS) {
    const int which = t / c, ch = t - which * c;
    const int sl = ch / LN_VEC, i = ch - sl * LN_VEC;
    float s = 0.f;
    for (int rr = 0; rr < RPB; ++rr) s += red[which][rr * LPR + sl][i];
    partial[((int64_t)blockIdx.x * 2 + which) * c + ch] = s;
  }
}

// dgamma / dbeta = column sums of the per-block partials [blocks][2][c].  One workgroup per 32
// channels, 32 slices of the block axis summed in parallel (coalesced 128-byte reads), LDS tree.
__global__ void __launch_bounds__(1024)
ln_partial_reduce_kernel(const float* __restrict__ partial, int blocks, int c, float* __restrict__ dgamma,
                         float* __restrict__ dbeta) {
  __shared__ float red[32][33];
  const int cx = threadIdx.x & 31, sy = threadIdx.x >> 5;
  const int t = blockIdx.x * 32 + cx;  // flat index into [2][c]; c % 32 == 0 so a block never straddles
  const int which = t / c, ch = t - which * c;
  float s = 0.f;
  if (t < 2 * c)
    for (int b = sy; b < blocks; b += 32) s += partial[((int64_t)b * 2 + which) * c + ch];
  red[sy][cx] = s;
  __syncthreads();
  if (sy == 0 && t < 2 * c) {
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) a += red[i][cx];
    float* dst = which ? dbeta : dgamma;
    if (dst) dst[ch] = a;
  }
}

// ---- generic widths (round 5): any even C <= LNG_MAX_C --------------------------------------------------------------------------
// The instances above need C / 8 to be a power of two (C = 32 .. 512: PT-v3m1).  PT-v3m2 (48 .. 512), PT-v3m3 (54 .. 576) and LitePT
// (36 .. 504) normalise over other widths (configs/sonata :45, configs/utonia :21, pointcept/models/litept/litept_v1.py:601); until
// round 5 those LayerNorms ran on ATen.  Here `lpr` lanes (the smallest power of two >= C / 2, at most 64) hold a row: lane li of the group
// owns the channel PAIRS 2 (li + lpr k), k < kc = ceil(C / 2 / lpr) <= LNG_K (4- / 8-byte accesses, consecutive lanes consecutive pairs),
// a wave holds 64 / lpr rows (C = 48: two rows per wave, C >= 66: one), the statistics are two reductions over the lane group (mean, then
// the centred second moment -- the same two-pass arithmetic as above), and the affine-gradient partials stay in the lane's registers
// over all rows it visits.  (The first form gave every row a whole wave: at C = 48 three lanes in eight did nothing and PT-v3m2's
// fused joints were SLOWER than its unfused Blocks, 77.5 vs 70.2 ms per step.)
#define LNG_K 8
#define LNG_MAX_C (LNG_K * 128)
struct LngGeo { int lpr, rpw, kc; };
__host__ __device__ __forceinline__ LngGeo lng_geo(int c) {
  LngGeo g;
  const int pairs = c >> 1;
  g.lpr = 8;
  while (g.lpr < 64 && g.lpr < pairs) g.lpr <<= 1;
  g.rpw = 64 / g.lpr;
  g.kc = (pairs + g.lpr - 1) / g.lpr;
  return g;
}
template <typename T>
__device__ __forceinline__ void lng_load2(const T* p, float& a, float& b);
template <> __device__ __forceinline__ void lng_load2<float>(const float* p, float& a, float& b) { const float2 v = *reinterpret_cast<const float2*>(p); a = v.x; b = v.y; }
template <> __device__ __forceinline__ void lng_load2<bf16_t>(const bf16_t* p, float& a, float& b) {
  const uint32_t u = *reinterpret_cast<const uint32_t*>(p);
  a = __uint_as_float(u << 16); b = __uint_as_float(u & 0xffff0000u);
}
template <> __device__ __forceinline__ void lng_load2<f16_t>(const f16_t* p, float& a, float& b) {
  const uint32_t u = *reinterpret_cast<const uint32_t*>(p);
  const _Float16* h = reinterpret_cast<const _Float16*>(&u);
  a = (float)h[0]; b = (float)h[1];
}
template <typename T>
__device__ __forceinline__ void lng_store2(T* p, float a, float b) {
  T o[2] = {ptc_from_float<T>(a), ptc_from_float<T>(b)};
  if (sizeof(T) == 4) *reinterpret_cast<uint2*>(p) = *reinterpret_cast<uint2*>(o);
  else *reinterpret_cast<uint32_t*>(p) = *reinterpret_cast<uint32_t*>(o);
}
// sum over the lpr lanes of a row group (lpr a power of two, groups aligned)
__device__ __forceinline__ float lng_group_sum(float v, int lpr) {
  for (int d = 1; d < lpr; d <<= 1) v += __shfl_xor(v, d, 64);
  return v;
}
// sum over the 64 / lpr row groups of a wave (lanes with equal li): the end-of-kernel merge of the affine-gradient partials
__device__ __forceinline__ float lng_rows_sum(float v, int lpr) {
  for (int d = lpr; d < 64; d <<= 1) v += __shfl_xor(v, d, 64);
  return v;
}
// per-lane view of the decomposition: li = lane in its row group, rs = row of the wave this lane works on
#define LNG_LANE_VIEW(c)                                                                  \
  const LngGeo geo = lng_geo(c);                                                          \
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;                             \
  const int li = lane & (geo.lpr - 1), rs = lane / geo.lpr;                               \
  constexpr int WPB = LN_THREADS / 64;                                                    \
  const int64_t row_step = (int64_t)gridDim.x * WPB * geo.rpw;                            \
  const int64_t row_first = ((int64_t)blockIdx.x * WPB + wave) * geo.rpw + rs
#define LNG_CH(k) (2 * (li + geo.lpr * (k)))

template <typename TI, typename TO>
__global__ void __launch_bounds__(LN_THREADS)
layer_norm_fwd_generic_kernel(const TI* __restrict__ x, int64_t n, int c, const float* __restrict__ gamma, const float* __restrict__ beta,
                              float eps, TO* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  LNG_LANE_VIEW(c);
  const float inv_c = 1.f / (float)c;
  float g[LNG_K][2], b[LNG_K][2];
#pragma unroll
  for (int k = 0; k < LNG_K; ++k) {
    const int ch = LNG_CH(k);
    const bool ok = k < geo.kc && ch < c;
    g[k][0] = (ok && gamma) ? gamma[ch] : 1.f; g[k][1] = (ok && gamma) ? gamma[ch + 1] : 1.f;
    b[k][0] = (ok && beta) ? beta[ch] : 0.f; b[k][1] = (ok && beta) ? beta[ch + 1] : 0.f;
  }
  // (whole waves leave the loop together: the bound is the wave's first row, lanes whose own row is past the end work on nothing)
  for (int64_t row0 = row_first - rs; row0 < n; row0 += row_step) {
    const int64_t row = row0 + rs;
    const bool rok = row < n;
    float v[LNG_K][2];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LNG_K; ++k) {
      v[k][0] = v[k][1] = 0.f;
      if (k < geo.kc) {
        const int ch = LNG_CH(k);
        if (rok && ch < c) lng_load2<TI>(x + row * c + ch, v[k][0], v[k][1]);
        s += v[k][0] + v[k][1];
      }
    }
    const float mean = lng_group_sum(s, geo.lpr) * inv_c;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < LNG_K; ++k) {
      if (k < geo.kc && LNG_CH(k) < c) {
        const float d0 = v[k][0] - mean, d1 = v[k][1] - mean;
        q = fmaf(d0, d0, q);
        q = fmaf(d1, d1, q);
      }
    }
    const float rstd = rsqrtf(fmaf(lng_group_sum(q, geo.lpr), inv_c, eps));
#pragma unroll
    for (int k = 0; k < LNG_K; ++k) {
      const int ch = LNG_CH(k);
      if (k < geo.kc && rok && ch < c)
        lng_store2<TO>(y + row * c + ch, fmaf((v[k][0] - mean) * rstd, g[k][0], b[k][0]), fmaf((v[k][1] - mean) * rstd, g[k][1], b[k][1]));
    }
    if (li == 0 && rok) { mean_out[row] = mean; rstd_out[row] = rstd; }
  }
}

// the waves' affine-gradient partials of ONE kind (kind-major acc[kind][k][e]) -> LDS -> partial[block][kinds][c]; called by all threads
template <int KINDS>
__device__ __forceinline__ void lng_merge_partials(float (&acc)[KINDS][LNG_K][2], const LngGeo& geo, int c, int li, int rs, int wave,
                                                   float (&red)[LN_THREADS / 64][LNG_MAX_C], float* __restrict__ partial) {
  constexpr int WPB = LN_THREADS / 64;
#pragma unroll
  for (int w4 = 0; w4 < KINDS; ++w4) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < LNG_K; ++k) {
      if (k < geo.kc) {
        const float a0 = lng_rows_sum(acc[w4][k][0], geo.lpr), a1 = lng_rows_sum(acc[w4][k][1], geo.lpr);
        const int ch = LNG_CH(k);
        if (rs == 0 && ch < c) { red[wave][ch] = a0; red[wave][ch + 1] = a1; }
      }
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < c; ch += LN_THREADS) {
      float s = 0.f;
#pragma unroll
      for (int wv = 0; wv < WPB; ++wv) s += red[wv][ch];
      partial[((int64_t)blockIdx.x * KINDS + w4) * c + ch] = s;
    }
  }
}

template <typename TG, typename TX>
__global__ void __launch_bounds__(LN_THREADS)
layer_norm_bwd_generic_kernel(const TG* __restrict__ dy, const TX* __restrict__ x, const float* __restrict__ mean,
                              const float* __restrict__ rstd, const float* __restrict__ gamma, int64_t n, int c, TX* __restrict__ dx,
                              float* __restrict__ partial /*[grid][2][c]*/) {
  LNG_LANE_VIEW(c);
  __shared__ float red[WPB][LNG_MAX_C];
  const float inv_c = 1.f / (float)c;
  float g[LNG_K][2], acc[2][LNG_K][2];                 // acc: dgamma, dbeta
#pragma unroll
  for (int k = 0; k < LNG_K; ++k) {
    const int ch = LNG_CH(k);
    const bool ok = k < geo.kc && ch < c;
    g[k][0] = (gamma && ok) ? gamma[ch] : 1.f;
    g[k][1] = (gamma && ok) ? gamma[ch + 1] : 1.f;
    acc[0][k][0] = acc[0][k][1] = acc[1][k][0] = acc[1][k][1] = 0.f;
  }
  for (int64_t row0 = row_first - rs; row0 < n; row0 += row_step) {
    const int64_t row = row0 + rs;
    const bool rok = row < n;
    const float m = rok ? mean[row] : 0.f, rsd = rok ? rstd[row] : 0.f;
    float xh[LNG_K][2], w[LNG_K][2];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < LNG_K; ++k) {
      xh[k][0] = xh[k][1] = w[k][0] = w[k][1] = 0.f;
      const int ch = LNG_CH(k);
      if (k < geo.kc && rok && ch < c) {
        float x0, x1, g0, g1;
        lng_load2<TX>(x + row * c + ch, x0, x1);
        lng_load2<TG>(dy + row * c + ch, g0, g1);
        xh[k][0] = (x0 - m) * rsd; xh[k][1] = (x1 - m) * rsd;
        w[k][0] = g0 * g[k][0]; w[k][1] = g1 * g[k][1];
        s1 += w[k][0] * xh[k][0] + w[k][1] * xh[k][1];
        s2 += w[k][0] + w[k][1];
        acc[0][k][0] += g0 * xh[k][0]; acc[0][k][1] += g1 * xh[k][1];
        acc[1][k][0] += g0; acc[1][k][1] += g1;
      }
    }
    const float c1 = lng_group_sum(s1, geo.lpr) * inv_c, c2 = lng_group_sum(s2, geo.lpr) * inv_c;
#pragma unroll
    for (int k = 0; k < LNG_K; ++k) {
      const int ch = LNG_CH(k);
      if (k < geo.kc && rok && ch < c) lng_store2<TX>(dx + row * c + ch, (w[k][0] - c2 - xh[k][0] * c1) * rsd, (w[k][1] - c2 - xh[k][1] * c1) * rsd);
    }
  }
  lng_merge_partials<2>(acc, geo, c, li, rs, wave, red, partial);
}

static int lng_grid(int64_t n, int c, int cap = 2048) {
  int64_t g = ptc_cdiv(n, (int64_t)(LN_THREADS / 64) * lng_geo(c).rpw);
  if (g > cap) g = cap;
  return (int)(g < 1 ? 1 : g);
}
static bool ln_generic_c(int c) { return c >= 2 && c <= LNG_MAX_C && (c & 1) == 0; }

static int ln_grid(int64_t n, int lpr) {
  const int rpb = LN_THREADS / lpr;
  int64_t g = ptc_cdiv(n, rpb);
  if (g > 2048) g = 2048;
  return (int)(g < 1 ? 1 : g);
}
static bool ln_supported_c(int c) { return c == 32 || c == 64 || c == 128 || c == 256 || c == 512; }
// widths the 8-channels-per-lane kernels take with their run-time width (round 5): multiples of 8 up to 512; the instance is the next power
// of two of c / 8 lanes per row (>= 4), lanes past the row idle
static bool ln_vec8_c(int c) { return c >= 8 && c <= 512 && c % 8 == 0; }
static int ln_lpr_of(int c) { int l = 4; while (l * LN_VEC < c) l <<= 1; return l; }

// 1: the C / 8-lanes-per-row instances (and the fused residual joints, ptc_add_norm_*); 2: the wave-per-row generic form; 0: neither
extern "C" int ptc_layer_norm_supported(int c) { return ln_supported_c(c) ? 1 : (ln_generic_c(c) ? 2 : 0); }

template <typename TI, typename TO>
static int launch_ln_fwd(const void* x, int64_t n, int c, const float* gamma, const float* beta, float eps, void* y,
                         float* mean, float* rstd, hipStream_t s) {
  if (!ln_vec8_c(c)) {
    hipLaunchKernelGGL((layer_norm_fwd_generic_kernel<TI, TO>), dim3(lng_grid(n, c)), dim3(LN_THREADS), 0, s, (const TI*)x, n, c, gamma, beta, eps,
                       (TO*)y, mean, rstd);
    PTC_CHECK_LAUNCH("layer_norm_fwd_generic_kernel");
    return PTC_OK;
  }
#define LN_FWD_CASE(LPR)                                                                                     \
  hipLaunchKernelGGL((layer_norm_fwd_kernel<TI, TO, LPR>), dim3(ln_grid(n, LPR)), dim3(LN_THREADS), 0, s,    \
                     (const TI*)x, n, c, gamma, beta, eps, (TO*)y, mean, rstd)
  switch (ln_lpr_of(c)) {
    case 4: LN_FWD_CASE(4); break;
    case 8: LN_FWD_CASE(8); break;
    case 16: LN_FWD_CASE(16); break;
    case 32: LN_FWD_CASE(32); break;
    default: LN_FWD_CASE(64); break;
  }
#undef LN_FWD_CASE
  PTC_CHECK_LAUNCH("layer_norm_fwd_kernel");
  return PTC_OK;
}

extern "C" int ptc_layer_norm_fwd(const void* x, int64_t n, int c, int in_dtype, const float* gamma, const float* beta,
                                  float eps, void* y, int out_dtype, float* mean, float* rstd, ptc_stream_t stream) {
  PTC_REQUIRE(n >= 0, PTC_EINVAL, "ptc_layer_norm_fwd: n < 0");
  PTC_REQUIRE(ln_supported_c(c) || ln_generic_c(c), PTC_EUNSUPPORTED, "ptc_layer_norm_fwd: C=%d is neither in {32,64,128,256,512} nor even and <= %d", c, LNG_MAX_C);
  if (n == 0) return PTC_OK;
  PTC_REQUIRE(x && y && mean && rstd, PTC_EINVAL, "ptc_layer_norm_fwd: null buffer");
  PTC_REQUIRE(out_dtype == PTC_F32 || out_dtype == in_dtype || in_dtype == PTC_F32, PTC_EUNSUPPORTED,
              "ptc_layer_norm_fwd: unsupported dtype pair %d -> %d", in_dtype, out_dtype);
  hipStream_t s = (hipStream_t)stream;
  // (the seven pairs the check above admits; bf16 <-> f16 pairs are not instantiated)
#define LN_PAIR(DI, DO, TI, TO) \
  if (in_dtype == DI && out_dtype == DO) return launch_ln_fwd<TI, TO>(x, n, c, gamma, beta, eps, y, mean, rstd, s);
  LN_PAIR(PTC_F32, PTC_F32, float, float) LN_PAIR(PTC_F32, PTC_BF16, float, bf16_t) LN_PAIR(PTC_F32, PTC_F16, float, f16_t)
  LN_PAIR(PTC_BF16, PTC_BF16, bf16_t, bf16_t) LN_PAIR(PTC_BF16, PTC_F32, bf16_t, float)
  LN_PAIR(PTC_F16, PTC_F16, f16_t, f16_t) LN_PAIR(PTC_F16, PTC_F32, f16_t, float)
#undef LN_PAIR
  ptc_set_error("ptc_layer_norm_fwd: bad dtype codes %d -> %d", in_dtype, out_dtype);
  return PTC_EINVAL;
}

extern "C" size_t ptc_layer_norm_bwd_workspace_bytes(int64_t n, int c) {
  if (!ln_vec8_c(c)) return ln_generic_c(c) ? ptc_align_up((size_t)lng_grid(n, c) * 2 * (size_t)c * sizeof(float), 256) : 256;
  return ptc_align_up((size_t)ln_grid(n, ln_lpr_of(c)) * 2 * (size_t)c * sizeof(float), 256);
}

template <typename TG, typename TX>
static int launch_ln_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                         int64_t n, int c, void* dx, float* dgamma, float* dbeta, void* ws, hipStream_t s) {
  if (!ln_vec8_c(c)) {
    const int grid = lng_grid(n, c);
    hipLaunchKernelGGL((layer_norm_bwd_generic_kernel<TG, TX>), dim3(grid), dim3(LN_THREADS), 0, s, (const TG*)dy, (const TX*)x, mean, rstd, gamma,
                       n, c, (TX*)dx, (float*)ws);
    PTC_CHECK_LAUNCH("layer_norm_bwd_generic_kernel");
    if (dgamma || dbeta) {
      hipLaunchKernelGGL(ln_partial_reduce_kernel, dim3((unsigned)ptc_cdiv(2 * c, 32)), dim3(1024), 0, s, (const float*)ws, grid, c, dgamma, dbeta);
      PTC_CHECK_LAUNCH("ln_partial_reduce_kernel");
    }
    return PTC_OK;
  }
  const int lpr = ln_lpr_of(c);
  const int grid = ln_grid(n, lpr);
#define LN_BWD_CASE(LPR)                                                                                          \
  hipLaunchKernelGGL((layer_norm_bwd_kernel<TG, TX, LPR>), dim3(grid), dim3(LN_THREADS), 0, s, (const TG*)dy,     \
                     (const TX*)x, mean, rstd, gamma, n, c, (TX*)dx, (float*)ws)
  switch (lpr) {
    case 4: LN_BWD_CASE(4); break;
    case 8: LN_BWD_CASE(8); break;
    case 16: LN_BWD_CASE(16); break;
    case 32: LN_BWD_CASE(32); break;
    default: LN_BWD_CASE(64); break;
  }
#undef LN_BWD_CASE
  PTC_CHECK_LAUNCH("layer_norm_bwd_kernel");
  if (dgamma || dbeta) {
    hipLaunchKernelGGL(ln_partial_reduce_kernel, dim3((unsigned)ptc_cdiv(2 * c, 32)), dim3(1024), 0, s, (const float*)ws,
                       grid, c, dgamma, dbeta);
    PTC_CHECK_LAUNCH("ln_partial_reduce_kernel");
  }
  return PTC_OK;
}

extern "C" int ptc_layer_norm_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean,
                                  const float* rstd, const float* gamma, int64_t n, int c, void* dx, float* dgamma,
                                  float* dbeta, void* workspace, size_t workspace_bytes, ptc_stream_t stream) {
  PTC_REQUIRE(n >= 0, PTC_EINVAL, "ptc_layer_norm_bwd: n < 0");
  PTC_REQUIRE(ln_supported_c(c) || ln_generic_c(c), PTC_EUNSUPPORTED, "ptc_layer_norm_bwd: C=%d is neither in {32,64,128,256,512} nor even and <= %d", c, LNG_MAX_C);
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    if (dgamma) PTC_HIP(hipMemsetAsync(dgamma, 0, (size_t)c * 4, s));
    if (dbeta) PTC_HIP(hipMemsetAsync(dbeta, 0, (size_t)c * 4, s));
    return PTC_OK;
  }
  PTC_REQUIRE(dy && x && mean && rstd && dx && workspace, PTC_EINVAL, "ptc_layer_norm_bwd: null buffer");
  PTC_REQUIRE(workspace_bytes >= ptc_layer_norm_bwd_workspace_bytes(n, c), PTC_EWORKSPACE, "ptc_layer_norm_bwd: workspace too small");
  PTC_REQUIRE(dy_dtype == PTC_F32 || dy_dtype == x_dtype || x_dtype == PTC_F32, PTC_EUNSUPPORTED,
              "ptc_layer_norm_bwd: unsupported dtype pair dy %d / x %d", dy_dtype, x_dtype);
#define LN_PAIR(DG, DX, TG, TX) \
  if (dy_dtype == DG && x_dtype == DX) return launch_ln_bwd<TG, TX>(dy, x, mean, rstd, gamma, n, c, dx, dgamma, dbeta, workspace, s);
  LN_PAIR(PTC_F32, PTC_F32, float, float) LN_PAIR(PTC_F32, PTC_BF16, float, bf16_t) LN_PAIR(PTC_F32, PTC_F16, float, f16_t)
  LN_PAIR(PTC_BF16, PTC_BF16, bf16_t, bf16_t) LN_PAIR(PTC_BF16, PTC_F32, bf16_t, float)
  LN_PAIR(PTC_F16, PTC_F16, f16_t, f16_t) LN_PAIR(PTC_F16, PTC_F32, f16_t, float)
#undef LN_PAIR
  ptc_set_error("ptc_layer_norm_bwd: bad dtype codes dy %d / x %d", dy_dtype, x_dtype);
  return PTC_EINVAL;
}

// ================================================================================================
// Fused residual + normalisation of the PTv3 Block (ptv3m1:318-338):
//     z = a + s_row * f(u),   f = LayerNorm_A or identity          (a: fp32 residual stream; bf16 where a stage begins)
//     y = LayerNorm_B(z)  or  y = z                                (y: operand of the next GEMM / conv)
// covers the three residual joints of a block in one pass each:
//     x1 = x + LN(cpe);  y1 = norm1(x1)      |  x2 = x1 + droppath(attn);  y2 = norm2(x2)
//     x3 = x2 + droppath(mlp);  y3 = bf16(x3) (next conv input)
// Unfused this is 3-4 kernels and 20-24 B per element of HBM traffic; fused it is 12 B.
// Backward (one pass, 18 B per element): dz = dz_in + LN_B'(dy) (or + dy);  da = dz;
// du = s_row * LN_A'(dz) (or s_row * dz);  affine gradients as per-block partials.
// ================================================================================================
template <typename TU, typename TY, int LPR>
__global__ void __launch_bounds__(LN_THREADS)
add_norm_fwd_kernel(const TU* __restrict__ u, const void* __restrict__ a, int a_bf16, const float* __restrict__ row_scale, int64_t n, int c,
                    const float* __restrict__ gA, const float* __restrict__ bA, float epsA, int normA,
                    const float* __restrict__ gB, const float* __restrict__ bB, float epsB, int normB,
                    float* __restrict__ z, TY* __restrict__ y, float* __restrict__ statA, float* __restrict__ statB) {
  constexpr int RPB = LN_THREADS / LPR;
  const int slot = threadIdx.x % LPR, rib = threadIdx.x / LPR;
  const bool act = slot * LN_VEC < c;            // (run-time width c <= LPR * 8, c % 8 == 0: see layer_norm_fwd_kernel)
  const float inv_c = 1.f / (float)c;
  float ga[LN_VEC], ba[LN_VEC], gb[LN_VEC], bb[LN_VEC];
#pragma unroll
  for (int i = 0; i < LN_VEC; ++i) {
    ga[i] = (normA && gA && act) ? gA[slot * LN_VEC + i] : 1.f; ba[i] = (normA && bA && act) ? bA[slot * LN_VEC + i] : 0.f;
    gb[i] = (normB && gB && act) ? gB[slot * LN_VEC + i] : 1.f; bb[i] = (normB && bB && act) ? bB[slot * LN_VEC + i] : 0.f;
  }
  for (int64_t row = (int64_t)blockIdx.x * RPB + rib; row < n; row += (int64_t)gridDim.x * RPB) {
    float v[LN_VEC], r[LN_VEC];
#pragma unroll
    for (int i = 0; i < LN_VEC; ++i) { v[i] = 0.f; r[i] = 0.f; }
    if (act) {
      ln_load8<TU>(u + row * c + slot * LN_VEC, v);
      // the residual operand is fp32 (the stream) or 16-bit (first block of a stage: the pooling / unpooling output); a_bf16 = 0 f32,
      // 1 bf16, 2 f16
      if (a_bf16 == 1) ln_load8<bf16_t>(reinterpret_cast<const bf16_t*>(a) + row * c + slot * LN_VEC, r);
      else if (a_bf16 == 2) ln_load8<f16_t>(reinterpret_cast<const f16_t*>(a) + row * c + slot * LN_VEC, r);
      else ln_load8<float>(reinterpret_cast<const float*>(a) + row * c + slot * LN_VEC, r);
    }
    if (normA) {
      float mean, rstd;
      ln_normalize_rt<LPR>(v, epsA, ga, ba, mean, rstd, inv_c, act);
      if (slot == 0) { statA[row] = mean; statA[n + row] = rstd; }
    }
    const float sc = row_scale ? row_scale[row] : 1.f;
#pragma unroll
    for (int i = 0; i < LN_VEC; ++i) r[i] = act ? fmaf(sc, v[i], r[i]) : 0.f;
    if (act) ln_store8<float>(z + row * c + slot * LN_VEC, r);
    if (y) {
      if (normB) {
        float mean, rstd;
        ln_normalize_rt<LPR>(r, epsB, gb, bb, mean, rstd, inv_c, act);
        if (slot == 0) { statB[row] = mean; statB[n + row] = rstd; }
      }
      if (act) ln_store8<TY>(y + row * c + slot * LN_VEC, r);
    }
  }
}

template <typename TU, typename TY, int LPR>
__global__ void __launch_bounds__(LN_THREADS)
add_norm_bwd_kernel(const float* __restrict__ dz_in, const TY* __restrict__ dy, const float* __restrict__ z,
                    const TU* __restrict__ u, const float* __restrict__ row_scale, int64_t n, int c,
                    const float* __restrict__ gA, const float* __restrict__ statA, int normA,
                    const float* __restrict__ gB, const float* __restrict__ statB, int normB,
                    void* __restrict__ da, int da_bf16, TU* __restrict__ du, float* __restrict__ partial /*[grid][4][c]*/) {
  constexpr int RPB = LN_THREADS / LPR;
  __shared__ float red[4][LN_THREADS][LN_VEC + 1];
  const int slot = threadIdx.x % LPR, rib = threadIdx.x / LPR;
  const bool act = slot * LN_VEC < c;            // (run-time width c <= LPR * 8, c % 8 == 0: see layer_norm_fwd_kernel)
  const float inv_c = 1.f / (float)c;
  float ga[LN_VEC], gb[LN_VEC], dgA[LN_VEC], dbA[LN_VEC], dgB[LN_VEC], dbB[LN_VEC];
#pragma unroll
  for (int i = 0; i < LN_VEC; ++i) {
    ga[i] = (normA && gA && act) ? gA[slot * LN_VEC + i] : 1.f;
    gb[i] = (normB && gB && act) ? gB[slot * LN_VEC + i] : 1.f;
    dgA[i] = dbA[i] = dgB[i] = dbB[i] = 0.f;
  }
  for (int64_t row = (int64_t)blockIdx.x * RPB + rib; row < n; row += (int64_t)gridDim.x * RPB) {
    float dz[LN_VEC];
#pragma unroll
    for (int i = 0; i < LN_VEC; ++i) dz[i] = 0.f;
    if (dz_in && act) ln_load8<float>(dz_in + row * c + slot * LN_VEC, dz);
    if (dy) {
      float g[LN_VEC];
#pragma unroll
      for (int i = 0; i < LN_VEC; ++i) g[i] = 0.f;
      if (act) ln_load8<TY>(dy + row * c + slot * LN_VEC, g);
      if (normB) {
        float zv[LN_VEC];
#pragma unroll
        for (int i = 0; i < LN_VEC; ++i) zv[i] = 0.f;
        if (act) ln_load8<float>(z + row * c + slot * LN_VEC, zv);
        const float m = statB[row], rs = statB[n + row];
        float s1 = 0.f, s2 = 0.f, xh[LN_VEC], w[LN_VEC];
#pragma unroll
        for (int i = 0; i < LN_VEC; ++i) {
          xh[i] = act ? (zv[i] - m) * rs : 0.f;
          w[i] = g[i] * gb[i];
          s1 += w[i] * xh[i];
          s2 += w[i];
          dgB[i] += g[i] * xh[i];
          dbB[i] += g[i];
        }
        const float c1 = group_sum<LPR>(s1) * inv_c, c2 = group_sum<LPR>(s2) * inv_c;
#pragma unroll
        for (int i = 0; i < LN_VEC; ++i) dz[i] += (w[i] - c2 - xh[i] * c1) * rs;
      } else {
#pragma unroll
        for (int i = 0; i < LN_VEC; ++i) dz[i] += g[i];
      }
    }
    if (act) {
      if (da_bf16 == 1) ln_store8<bf16_t>(reinterpret_cast<bf16_t*>(da) + row * c + slot * LN_VEC, dz);   // the gradient autograd would cast anyway
      else if (da_bf16 == 2) ln_store8<f16_t>(reinterpret_cast<f16_t*>(da) + row * c + slot * LN_VEC, dz);
      else ln_store8<float>(reinterpret_cast<float*>(da) + row * c + slot * LN_VEC, dz);
    }
    const float sc = row_scale ? row_scale[row] : 1.f;
    float o[LN_VEC];
    if (normA) {
      float uv[LN_VEC];
#pragma unroll
      for (int i = 0; i < LN_VEC; ++i) uv[i] = 0.f;
      if (act) ln_load8<TU>(u + row * c + slot * LN_VEC, uv);
      const float m = statA[row], rs = statA[n + row];
      float s1 = 0.f, s2 = 0.f, xh[LN_VEC], w[LN_VEC];
#pragma unroll
      for (int i = 0; i < LN_VEC; ++i) {
        const float gg = act ? dz[i] * sc : 0.f;
        xh[i] = act ? (uv[i] - m) * rs : 0.f;
        w[i] = gg * ga[i];
        s1 += w[i] * xh[i];
        s2 += w[i];
        dgA[i] += gg * xh[i];
        dbA[i] += gg;
      }
      const float c1 = group_sum<LPR>(s1) * inv_c, c2 = group_sum<LPR>(s2) * inv_c;
#pragma unroll
      for (int i = 0; i < LN_VEC; ++i) o[i] = (w[i] - c2 - xh[i] * c1) * rs;
    } else {
#pragma unroll
      for (int i = 0; i < LN_VEC; ++i) o[i] = dz[i] * sc;
    }
    if (act) ln_store8<TU>(du + row * c + slot * LN_VEC, o);
  }
#pragma unroll
  for (int i = 0; i < LN_VEC; ++i) {
    red[0][threadIdx.x][i] = dgA[i]; red[1][threadIdx.x][i] = dbA[i];
    red[2][threadIdx.x][i] = dgB[i]; red[3][threadIdx.x][i] = dbB[i];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 4 * c; t += LN_THREADS) {
    const int which = t / c, ch = t - which * c;
    const int sl = ch / LN_VEC, i = ch - sl * LN_VEC;
    float s = 0.f;
    for (int rr = 0; rr < RPB; ++rr) s += red[which][rr * LPR + sl][i];
    partial[((int64_t)blockIdx.x * 4 + which) * c + ch] = s;
  }
}

// ---- the residual joint for generic widths (round 5): any even C <= LNG_MAX_C, the row-group layout of layer_norm_*_generic_kernel -----
// PT-v3m2 at the Sonata widths ran its Blocks unfused (3 LayerNorms + 3 fp32 adds + casts per Block: ~17 ms of a 71 ms step,
// profiles/r05_k_m2_kernel_stats.csv); same statements as add_norm_{fwd,bwd}_kernel above.
__device__ __forceinline__ void lng_load2_any(const void* p, int kind, int64_t e, float& a, float& b) {   // kind: 0 f32, 1 bf16, 2 f16
  if (kind == 1) lng_load2<bf16_t>(reinterpret_cast<const bf16_t*>(p) + e, a, b);
  else if (kind == 2) lng_load2<f16_t>(reinterpret_cast<const f16_t*>(p) + e, a, b);
  else lng_load2<float>(reinterpret_cast<const float*>(p) + e, a, b);
}
__device__ __forceinline__ void lng_store2_any(void* p, int kind, int64_t e, float a, float b) {
  if (kind == 1) lng_store2<bf16_t>(reinterpret_cast<bf16_t*>(p) + e, a, b);
  else if (kind == 2) lng_store2<f16_t>(reinterpret_cast<f16_t*>(p) + e, a, b);
  else lng_store2<float>(reinterpret_cast<float*>(p) + e, a, b);
}
// LayerNorm of the row a lane group holds (pairs v[k][0..1] at channels 2 (li + lpr k)), in place
__device__ __forceinline__ void lng_normalize(float (&v)[LNG_K][2], const LngGeo& geo, int c, int li, float eps, const float (&g)[LNG_K][2],
                                              const float (&b)[LNG_K][2], float& mean, float& rstd) {
  const float inv_c = 1.f / (float)c;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < LNG_K; ++k)
    if (k < geo.kc && LNG_CH(k) < c) s += v[k][0] + v[k][1];
  mean = lng_group_sum(s, geo.lpr) * inv_c;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < LNG_K; ++k)
    if (k < geo.kc && LNG_CH(k) < c) {
      const float d0 = v[k][0] - mean, d1 = v[k][1] - mean;
      q = fmaf(d0, d0, q);
      q = fmaf(d1, d1, q);
    }
  rstd = rsqrtf(fmaf(lng_group_sum(q, geo.lpr), inv_c, eps));
#pragma unroll
  for (int k = 0; k < LNG_K; ++k)
    if (k < geo.kc) {
      v[k][0] = fmaf((v[k][0] - mean) * rstd, g[k][0], b[k][0]);
      v[k][1] = fmaf((v[k][1] - mean) * rstd, g[k][1], b[k][1]);
    }
}

template <typename TU, typename TY>
__global__ void __launch_bounds__(LN_THREADS)
add_norm_fwd_generic_kernel(const TU* __restrict__ u, const void* __restrict__ a, int a_kind, const float* __restrict__ row_scale, int64_t n,
                            int c, const float* __restrict__ gA, const float* __restrict__ bA, float epsA, int normA,
                            const float* __restrict__ gB, const float* __restrict__ bB, float epsB, int normB,
                            float* __restrict__ z, TY* __restrict__ y, float* __restrict__ statA, float* __restrict__ statB) {
  LNG_LANE_VIEW(c);
  float ga[LNG_K][2], ba[LNG_K][2], gb[LNG_K][2], bb[LNG_K][2];
#pragma unroll
  for (int k = 0; k < LNG_K; ++k) {
    const int ch = LNG_CH(k);
    const bool ok = k < geo.kc && ch < c;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      ga[k][e] = (ok && normA && gA) ? gA[ch + e] : 1.f; ba[k][e] = (ok && normA && bA) ? bA[ch + e] : 0.f;
      gb[k][e] = (ok && normB && gB) ? gB[ch + e] : 1.f; bb[k][e] = (ok && normB && bB) ? bB[ch + e] : 0.f;
    }
  }
  for (int64_t row0 = row_first - rs; row0 < n; row0 += row_step) {
    const int64_t row = row0 + rs;
    const bool rok = row < n;
    float v[LNG_K][2], r[LNG_K][2];
#pragma unroll
    for (int k = 0; k < LNG_K; ++k) {
      v[k][0] = v[k][1] = r[k][0] = r[k][1] = 0.f;
      const int ch = LNG_CH(k);
      if (k < geo.kc && rok && ch < c) {
        lng_load2<TU>(u + row * c + ch, v[k][0], v[k][1]);
        lng_load2_any(a, a_kind, row * c + ch, r[k][0], r[k][1]);
      }
    }
    if (normA) {
      float mean, rstd;
      lng_normalize(v, geo, c, li, epsA, ga, ba, mean, rstd);
      if (li == 0 && rok) { statA[row] = mean; statA[n + row] = rstd; }
    }
    const float sc = (row_scale && rok) ? row_scale[row] : 1.f;
#pragma unroll
    for (int k = 0; k < LNG_K; ++k) {
      if (k < geo.kc) {
        r[k][0] = fmaf(sc, v[k][0], r[k][0]);
        r[k][1] = fmaf(sc, v[k][1], r[k][1]);
        const int ch = LNG_CH(k);
        if (rok && ch < c) lng_store2<float>(z + row * c + ch, r[k][0], r[k][1]);
      }
    }
    if (y) {
      if (normB) {
        float mean, rstd;
        lng_normalize(r, geo, c, li, epsB, gb, bb, mean, rstd);
        if (li == 0 && rok) { statB[row] = mean; statB[n + row] = rstd; }
      }
#pragma unroll
      for (int k = 0; k < LNG_K; ++k) {
        const int ch = LNG_CH(k);
        if (k < geo.kc && rok && ch < c) lng_store2<TY>(y + row * c + ch, r[k][0], r[k][1]);
      }
    }
  }
}

template <typename TU, typename TY>
__global__ void __launch_bounds__(LN_THREADS)
add_norm_bwd_generic_kernel(const float* __restrict__ dz_in, const TY* __restrict__ dy, const float* __restrict__ z,
                            const TU* __restrict__ u, const float* __restrict__ row_scale, int64_t n, int c,
                            const float* __restrict__ gA, const float* __restrict__ statA, int normA,
                            const float* __restrict__ gB, const float* __restrict__ statB, int normB,
                            void* __restrict__ da, int da_kind, TU* __restrict__ du, float* __restrict__ partial /*[grid][4][c]*/) {
  LNG_LANE_VIEW(c);
  __shared__ float red[WPB][LNG_MAX_C];
  const float inv_c = 1.f / (float)c;
  float ga[LNG_K][2], gb[LNG_K][2], acc[4][LNG_K][2];        // acc: dgA, dbA, dgB, dbB
#pragma unroll
  for (int k = 0; k < LNG_K; ++k) {
    const int ch = LNG_CH(k);
    const bool ok = k < geo.kc && ch < c;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      ga[k][e] = (ok && normA && gA) ? gA[ch + e] : 1.f;
      gb[k][e] = (ok && normB && gB) ? gB[ch + e] : 1.f;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) acc[w4][k][e] = 0.f;
    }
  }
  for (int64_t row0 = row_first - rs; row0 < n; row0 += row_step) {
    const int64_t row = row0 + rs;
    const bool rok = row < n;
    float dz[LNG_K][2];
#pragma unroll
    for (int k = 0; k < LNG_K; ++k) {
      dz[k][0] = dz[k][1] = 0.f;
      const int ch = LNG_CH(k);
      if (dz_in && k < geo.kc && rok && ch < c) lng_load2<float>(dz_in + row * c + ch, dz[k][0], dz[k][1]);
    }
    if (dy) {
      float g[LNG_K][2];
#pragma unroll
      for (int k = 0; k < LNG_K; ++k) {
        g[k][0] = g[k][1] = 0.f;
        const int ch = LNG_CH(k);
        if (k < geo.kc && rok && ch < c) lng_load2<TY>(dy + row * c + ch, g[k][0], g[k][1]);
      }
      if (normB) {
        const float m = rok ? statB[row] : 0.f, rsd = rok ? statB[n + row] : 0.f;
        float xh[LNG_K][2], w[LNG_K][2], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < LNG_K; ++k) {
          xh[k][0] = xh[k][1] = w[k][0] = w[k][1] = 0.f;
          const int ch = LNG_CH(k);
          if (k < geo.kc && rok && ch < c) {
            float z0, z1;
            lng_load2<float>(z + row * c + ch, z0, z1);
            xh[k][0] = (z0 - m) * rsd; xh[k][1] = (z1 - m) * rsd;
            w[k][0] = g[k][0] * gb[k][0]; w[k][1] = g[k][1] * gb[k][1];
            s1 += w[k][0] * xh[k][0] + w[k][1] * xh[k][1];
            s2 += w[k][0] + w[k][1];
            acc[2][k][0] += g[k][0] * xh[k][0]; acc[2][k][1] += g[k][1] * xh[k][1];
            acc[3][k][0] += g[k][0]; acc[3][k][1] += g[k][1];
          }
        }
        const float c1 = lng_group_sum(s1, geo.lpr) * inv_c, c2 = lng_group_sum(s2, geo.lpr) * inv_c;
#pragma unroll
        for (int k = 0; k < LNG_K; ++k)
          if (k < geo.kc) {
            dz[k][0] += (w[k][0] - c2 - xh[k][0] * c1) * rsd;
            dz[k][1] += (w[k][1] - c2 - xh[k][1] * c1) * rsd;
          }
      } else {
#pragma unroll
        for (int k = 0; k < LNG_K; ++k) { dz[k][0] += g[k][0]; dz[k][1] += g[k][1]; }
      }
    }
#pragma unroll
    for (int k = 0; k < LNG_K; ++k) {
      const int ch = LNG_CH(k);
      if (k < geo.kc && rok && ch < c) lng_store2_any(da, da_kind, row * c + ch, dz[k][0], dz[k][1]);      // the gradient autograd would cast anyway
    }
    const float sc = (row_scale && rok) ? row_scale[row] : 1.f;
    if (normA) {
      const float m = rok ? statA[row] : 0.f, rsd = rok ? statA[n + row] : 0.f;
      float xh[LNG_K][2], w[LNG_K][2], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int k = 0; k < LNG_K; ++k) {
        xh[k][0] = xh[k][1] = w[k][0] = w[k][1] = 0.f;
        const int ch = LNG_CH(k);
        if (k < geo.kc && rok && ch < c) {
          float u0, u1;
          lng_load2<TU>(u + row * c + ch, u0, u1);
          const float g0 = dz[k][0] * sc, g1 = dz[k][1] * sc;
          xh[k][0] = (u0 - m) * rsd; xh[k][1] = (u1 - m) * rsd;
          w[k][0] = g0 * ga[k][0]; w[k][1] = g1 * ga[k][1];
          s1 += w[k][0] * xh[k][0] + w[k][1] * xh[k][1];
          s2 += w[k][0] + w[k][1];
          acc[0][k][0] += g0 * xh[k][0]; acc[0][k][1] += g1 * xh[k][1];
          acc[1][k][0] += g0; acc[1][k][1] += g1;
        }
      }
      const float c1 = lng_group_sum(s1, geo.lpr) * inv_c, c2 = lng_group_sum(s2, geo.lpr) * inv_c;
#pragma unroll
      for (int k = 0; k < LNG_K; ++k) {
        const int ch = LNG_CH(k);
        if (k < geo.kc && rok && ch < c) lng_store2<TU>(du + row * c + ch, (w[k][0] - c2 - xh[k][0] * c1) * rsd, (w[k][1] - c2 - xh[k][1] * c1) * rsd);
      }
    } else {
#pragma unroll
      for (int k = 0; k < LNG_K; ++k) {
        const int ch = LNG_CH(k);
        if (k < geo.kc && rok && ch < c) lng_store2<TU>(du + row * c + ch, dz[k][0] * sc, dz[k][1] * sc);
      }
    }
  }
  lng_merge_partials<4>(acc, geo, c, li, rs, wave, red, partial);
}
static int ang_grid(int64_t n, int c) { return lng_grid(n, c, 1024); }

// column sums of partial [blocks][4][c] -> four vectors (any may be NULL)
__global__ void __launch_bounds__(1024)
add_norm_partial_reduce_kernel(const float* __restrict__ partial, int blocks, int c, float* __restrict__ o0,
                               float* __restrict__ o1, float* __restrict__ o2, float* __restrict__ o3) {
  __shared__ float red[32][33];
  const int cx = threadIdx.x & 31, sy = threadIdx.x >> 5;
  const int t = blockIdx.x * 32 + cx;
  const int which = t / c, ch = t - which * c;
  // four independent accumulators: the loop was one dependent load chain of blocks / 32 steps (11 us per call for 2 MB, 44 calls per
  // step); the summation order stays fixed
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (t < 4 * c) {
    const float* p = partial + (int64_t)which * c + ch;
    const int64_t st = (int64_t)4 * c;
    int b = sy;
    for (; b + 96 < blocks; b += 128) {
      s0 += p[(int64_t)b * st];
      s1 += p[(int64_t)(b + 32) * st];
      s2 += p[(int64_t)(b + 64) * st];
      s3 += p[(int64_t)(b + 96) * st];
    }
    for (; b < blocks; b += 32) s0 += p[(int64_t)b * st];
  }
  const float s = (s0 + s1) + (s2 + s3);
  red[sy][cx] = s;
  __syncthreads();
  if (sy == 0 && t < 4 * c) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) acc += red[i][cx];
    float* dst = which == 0 ? o0 : (which == 1 ? o1 : (which == 2 ? o2 : o3));
    if (dst) dst[ch] = acc;
  }
}

static int an_grid(int64_t n, int lpr) {
  const int rpb = LN_THREADS / lpr;
  int64_t g = ptc_cdiv(n, rpb);
  if (g > 1024) g = 1024;
  return (int)(g < 1 ? 1 : g);
}

template <typename TU, typename TY>
static int launch_an_fwd(const void* u, const void* a, int a_bf16, const float* row_scale, int64_t n, int c, const float* gA,
                         const float* bA, float epsA, int normA, const float* gB, const float* bB, float epsB, int normB,
                         float* z, void* y, float* statA, float* statB, hipStream_t s) {
  if (!ln_vec8_c(c)) {
    hipLaunchKernelGGL((add_norm_fwd_generic_kernel<TU, TY>), dim3(lng_grid(n, c)), dim3(LN_THREADS), 0, s, (const TU*)u, a, a_bf16, row_scale, n, c,
                       gA, bA, epsA, normA, gB, bB, epsB, normB, z, (TY*)y, statA, statB);
    PTC_CHECK_LAUNCH("add_norm_fwd_generic_kernel");
    return PTC_OK;
  }
#define AN_FWD_CASE(LPR)                                                                                                \
  hipLaunchKernelGGL((add_norm_fwd_kernel<TU, TY, LPR>), dim3(ln_grid(n, LPR)), dim3(LN_THREADS), 0, s, (const TU*)u, a, \
                     a_bf16, row_scale, n, c, gA, bA, epsA, normA, gB, bB, epsB, normB, z, (TY*)y, statA, statB)
  switch (ln_lpr_of(c)) {
    case 4: AN_FWD_CASE(4); break;
    case 8: AN_FWD_CASE(8); break;
    case 16: AN_FWD_CASE(16); break;
    case 32: AN_FWD_CASE(32); break;
    default: AN_FWD_CASE(64); break;
  }
#undef AN_FWD_CASE
  PTC_CHECK_LAUNCH("add_norm_fwd_kernel");
  return PTC_OK;
}

extern "C" int ptc_add_norm_fwd(const void* u, int u_dtype, const void* a, int a_dtype, const float* row_scale, int64_t n, int c,
                                const float* gA, const float* bA, float epsA, int normA, const float* gB, const float* bB,
                                float epsB, int normB, float* z, void* y, int y_dtype, float* statA, float* statB,
                                ptc_stream_t stream) {
  PTC_REQUIRE(n >= 0, PTC_EINVAL, "ptc_add_norm_fwd: n < 0");
  PTC_REQUIRE(ln_supported_c(c) || ln_generic_c(c), PTC_EUNSUPPORTED, "ptc_add_norm_fwd: C=%d is neither in {32,64,128,256,512} nor even and <= %d", c, LNG_MAX_C);
  if (n == 0) return PTC_OK;
  PTC_REQUIRE(u && a && z, PTC_EINVAL, "ptc_add_norm_fwd: null buffer");
  PTC_REQUIRE((!normA || statA) && (!(normB && y) || statB), PTC_EINVAL, "ptc_add_norm_fwd: missing statistics buffer");
  // 16-bit operands of ONE kind per call (bf16 autocast or fp16 autocast: VERDICT r2 weak 11 -- the reference's fp16 + GradScaler recipe
  // used to fall off the fused joints); a: f32 / bf16 / f16 independently
  const int k16 = (u_dtype != PTC_F32) ? u_dtype : ((y && y_dtype != PTC_F32) ? y_dtype : PTC_BF16);
  PTC_REQUIRE((u_dtype == PTC_F32 || u_dtype == k16) && (!y || y_dtype == PTC_F32 || y_dtype == k16), PTC_EUNSUPPORTED,
              "ptc_add_norm_fwd: u / y must be f32 or one 16-bit type");
  const int a_kind = a_dtype == PTC_BF16 ? 1 : (a_dtype == PTC_F16 ? 2 : 0);
  hipStream_t s = (hipStream_t)stream;
#define AN_FWD(TU, TY) return launch_an_fwd<TU, TY>(u, a, a_kind, row_scale, n, c, gA, bA, epsA, normA, gB, bB, epsB, normB, z, y, statA, statB, s)
  const bool u16 = u_dtype != PTC_F32, y16 = y && y_dtype != PTC_F32;
  if (k16 == PTC_BF16) {
    if (u16 && y16) AN_FWD(bf16_t, bf16_t);
    if (u16) AN_FWD(bf16_t, float);
    if (y16) AN_FWD(float, bf16_t);
  } else {
    if (u16 && y16) AN_FWD(f16_t, f16_t);
    if (u16) AN_FWD(f16_t, float);
    if (y16) AN_FWD(float, f16_t);
  }
  AN_FWD(float, float);
#undef AN_FWD
}

extern "C" size_t ptc_add_norm_bwd_workspace_bytes(int64_t n, int c) {
  if (!ln_vec8_c(c)) return ln_generic_c(c) ? ptc_align_up((size_t)ang_grid(n, c) * 4 * (size_t)c * sizeof(float), 256) : 256;
  return ptc_align_up((size_t)an_grid(n, ln_lpr_of(c)) * 4 * (size_t)c * sizeof(float), 256);
}

template <typename TU, typename TY>
static int launch_an_bwd(const float* dz_in, const void* dy, const float* z, const void* u, const float* row_scale, int64_t n,
                         int c, const float* gA, const float* statA, int normA, const float* gB, const float* statB, int normB,
                         void* da, int da_bf16, void* du, float* dgA, float* dbA, float* dgB, float* dbB, void* ws, hipStream_t s) {
  if (!ln_vec8_c(c)) {
    const int grid = ang_grid(n, c);
    hipLaunchKernelGGL((add_norm_bwd_generic_kernel<TU, TY>), dim3(grid), dim3(LN_THREADS), 0, s, dz_in, (const TY*)dy, z, (const TU*)u, row_scale,
                       n, c, gA, statA, normA, gB, statB, normB, da, da_bf16, (TU*)du, (float*)ws);
    PTC_CHECK_LAUNCH("add_norm_bwd_generic_kernel");
    if (dgA || dbA || dgB || dbB) {
      hipLaunchKernelGGL(add_norm_partial_reduce_kernel, dim3((unsigned)ptc_cdiv(4 * c, 32)), dim3(1024), 0, s, (const float*)ws, grid, c, dgA, dbA,
                         dgB, dbB);
      PTC_CHECK_LAUNCH("add_norm_partial_reduce_kernel");
    }
    return PTC_OK;
  }
  const int lpr = ln_lpr_of(c);
  const int grid = an_grid(n, lpr);
#define AN_BWD_CASE(LPR)                                                                                               \
  hipLaunchKernelGGL((add_norm_bwd_kernel<TU, TY, LPR>), dim3(grid), dim3(LN_THREADS), 0, s, dz_in, (const TY*)dy, z,  \
                     (const TU*)u, row_scale, n, c, gA, statA, normA, gB, statB, normB, da, da_bf16, (TU*)du, (float*)ws)
  switch (lpr) {
    case 4: AN_BWD_CASE(4); break;
    case 8: AN_BWD_CASE(8); break;
    case 16: AN_BWD_CASE(16); break;
    case 32: AN_BWD_CASE(32); break;
    default: AN_BWD_CASE(64); break;
  }
#undef AN_BWD_CASE
  PTC_CHECK_LAUNCH("add_norm_bwd_kernel");
  if (dgA || dbA || dgB || dbB) {
    hipLaunchKernelGGL(add_norm_partial_reduce_kernel, dim3((unsigned)ptc_cdiv(4 * c, 32)), dim3(1024), 0, s, (const float*)ws,
                       grid, c, dgA, dbA, dgB, dbB);
    PTC_CHECK_LAUNCH("add_norm_partial_reduce_kernel");
  }
  return PTC_OK;
}

extern "C" int ptc_add_norm_bwd(const float* dz_in, const void* dy, int dy_dtype, const float* z, const void* u, int u_dtype,
                                const float* row_scale, int64_t n, int c, const float* gA, const float* statA, int normA,
                                const float* gB, const float* statB, int normB, void* da, int da_dtype, void* du, float* dgA, float* dbA,
                                float* dgB, float* dbB, void* workspace, size_t workspace_bytes, ptc_stream_t stream) {
  PTC_REQUIRE(n >= 0, PTC_EINVAL, "ptc_add_norm_bwd: n < 0");
  PTC_REQUIRE(ln_supported_c(c) || ln_generic_c(c), PTC_EUNSUPPORTED, "ptc_add_norm_bwd: C=%d is neither in {32,64,128,256,512} nor even and <= %d", c, LNG_MAX_C);
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    float* outs[4] = {dgA, dbA, dgB, dbB};
    for (int i = 0; i < 4; ++i)
      if (outs[i]) PTC_HIP(hipMemsetAsync(outs[i], 0, (size_t)c * 4, s));
    return PTC_OK;
  }
  PTC_REQUIRE(da && du && workspace && (dz_in || dy), PTC_EINVAL, "ptc_add_norm_bwd: null buffer");
  PTC_REQUIRE((!normA || (u && statA)) && (!(normB && dy) || (z && statB)), PTC_EINVAL, "ptc_add_norm_bwd: missing saved tensors");
  PTC_REQUIRE(workspace_bytes >= ptc_add_norm_bwd_workspace_bytes(n, c), PTC_EWORKSPACE, "ptc_add_norm_bwd: workspace too small");
  const int k16 = (u_dtype != PTC_F32) ? u_dtype : ((dy && dy_dtype != PTC_F32) ? dy_dtype : PTC_BF16);
  PTC_REQUIRE((u_dtype == PTC_F32 || u_dtype == k16) && (!dy || dy_dtype == PTC_F32 || dy_dtype == k16), PTC_EUNSUPPORTED,
              "ptc_add_norm_bwd: u / dy must be f32 or one 16-bit type");
  const int da_kind = da_dtype == PTC_BF16 ? 1 : (da_dtype == PTC_F16 ? 2 : 0);
#define AN_BWD(TU, TY) return launch_an_bwd<TU, TY>(dz_in, dy, z, u, row_scale, n, c, gA, statA, normA, gB, statB, normB, da, da_kind, du, dgA, dbA, dgB, dbB, workspace, s)
  const bool u16 = u_dtype != PTC_F32, y16 = dy && dy_dtype != PTC_F32;
  if (k16 == PTC_BF16) {
    if (u16 && y16) AN_BWD(bf16_t, bf16_t);
    if (u16) AN_BWD(bf16_t, float);
    if (y16) AN_BWD(float, bf16_t);
  } else {
    if (u16 && y16) AN_BWD(f16_t, f16_t);
    if (u16) AN_BWD(f16_t, float);
    if (y16) AN_BWD(float, f16_t);
  }
  AN_BWD(float, float);
#undef AN_BWD
}
