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

#define LN_THREADS 256
#define LN_VEC 8  // channels per lane

template <typename T>
__device__ __forceinline__ void ln_load8(const T* p, float (&v)[LN_VEC]);
template <>
__device__ __forceinline__ void ln_load8<float>(const float* p, float (&v)[LN_VEC]) {
  const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <>
__device__ __forceinline__ void ln_load8<bf16_t>(const bf16_t* p, float (&v)[LN_VEC]) {
  const uint4 u = *reinterpret_cast<const uint4*>(p);
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
template <>
__device__ __forceinline__ void ln_load8<f16_t>(const f16_t* p, float (&v)[LN_VEC]) {
  const uint4 u = *reinterpret_cast<const uint4*>(p);
  const _Float16* h = reinterpret_cast<const _Float16*>(&u);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)h[i];
}
template <typename T>
__device__ __forceinline__ void ln_store8(T* p, const float (&v)[LN_VEC]) {
  T o[LN_VEC];
#pragma unroll
  for (int i = 0; i < LN_VEC; ++i) o[i] = ptc_from_float<T>(v[i]);
  if (sizeof(T) == 4) { reinterpret_cast<uint4*>(p)[0] = reinterpret_cast<uint4*>(o)[0]; reinterpret_cast<uint4*>(p)[1] = reinterpret_cast<uint4*>(o)[1]; }
  else reinterpret_cast<uint4*>(p)[0] = reinterpret_cast<uint4*>(o)[0];
}

// sum over the LPR lanes of a row group (LPR power of two, groups aligned)
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int d = 1; d < LPR; d <<= 1) v += __shfl_xor(v, d, 64);
  return v;
}

template <typename TI, typename TO, int LPR>
__global__ void __launch_bounds__(LN_THREADS)
layer_norm_fwd_kernel(const TI* __restrict__ x, int64_t n, const float* __restrict__ gamma, const float* __restrict__ beta,
                      float eps, TO* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  constexpr int C = LPR * LN_VEC;
  constexpr int RPB = LN_THREADS / LPR;  // rows per block iteration
  const int slot = threadIdx.x % LPR, rib = threadIdx.x / LPR;
  float g[LN_VEC], b[LN_VEC];
#pragma unroll
  for (int i = 0; i < LN_VEC; ++i) { g[i] = gamma ? gamma[slot * LN_VEC + i] : 1.f; b[i] = beta ? beta[slot * LN_VEC + i] : 0.f; }
  for (int64_t row = (int64_t)blockIdx.x * RPB + rib; row < n; row += (int64_t)gridDim.x * RPB) {
    float v[LN_VEC];
    ln_load8<TI>(x + row * C + slot * LN_VEC, v);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_VEC; ++i) s += v[i];
    const float mean = group_sum<LPR>(s) * (1.f / C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_VEC; ++i) { const float d = v[i] - mean; q += d * d; }
    const float rstd = rsqrtf(group_sum<LPR>(q) * (1.f / C) + eps);
    float o[LN_VEC];
#pragma unroll
    for (int i = 0; i < LN_VEC; ++i) o[i] = (v[i] - mean) * rstd * g[i] + b[i];
    ln_store8<TO>(y + row * C + slot * LN_VEC, o);
    if (slot == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
  }
}

template <typename TG, typename TX, int LPR>
__global__ void __launch_bounds__(LN_THREADS)
layer_norm_bwd_kernel(const TG* __restrict__ dy, const TX* __restrict__ x, const float* __restrict__ mean,
                      const float* __restrict__ rstd, const float* __restrict__ gamma, int64_t n, TX* __restrict__ dx,
                      float* __restrict__ partial /*[grid][2][C]*/) {
  constexpr int C = LPR * LN_VEC;
  constexpr int RPB = LN_THREADS / LPR;
  __shared__ float red[2][LN_THREADS][LN_VEC + 1];
  const int slot = threadIdx.x % LPR, rib = threadIdx.x / LPR;
  float g[LN_VEC], dg[LN_VEC], db[LN_VEC];
#pragma unroll
  for (int i = 0; i < LN_VEC; ++i) { g[i] = gamma ? gamma[slot * LN_VEC + i] : 1.f; dg[i] = 0.f; db[i] = 0.f; }
  for (int64_t row = (int64_t)blockIdx.x * RPB + rib; row < n; row += (int64_t)gridDim.x * RPB) {
    float xv[LN_VEC], gv[LN_VEC];
    ln_load8<TX>(x + row * C + slot * LN_VEC, xv);
    ln_load8<TG>(dy + row * C + slot * LN_VEC, gv);
    const float m = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f, xh[LN_VEC], w[LN_VEC];
#pragma unroll
    for (int i = 0; i < LN_VEC; ++i) {
      xh[i] = (xv[i] - m) * rs;
      w[i] = gv[i] * g[i];
      s1 += w[i] * xh[i];
      s2 += w[i];
      dg[i] += gv[i] * xh[i];
      db[i] += gv[i];
    }
    const float c1 = group_sum<LPR>(s1) * (1.f / C), c2 = group_sum<LPR>(s2) * (1.f / C);
    float o[LN_VEC];
#pragma unroll
    for (int i = 0; i < LN_VEC; ++i) o[i] = (w[i] - c2 - xh[i] * c1) * rs;
    ln_store8<TX>(dx + row * C + slot * LN_VEC, o);
  }
  // block reduction of the affine-gradient partials: threads with equal `slot` (RPB of them)
#pragma unroll
  for (int i = 0; i < LN_VEC; ++i) { red[0][threadIdx.x][i] = dg[i]; red[1][threadIdx.x][i] = db[i]; }
  __syncthreads();
  for (int t = threadIdx.x; t < 2 * C; t += LN_THREADS) {
    const int which = t / C, ch = t - which * C;
    const int sl = ch / LN_VEC, i = ch - sl * LN_VEC;
    float s = 0.f;
    for (int rr = 0; rr < RPB; ++rr) s += red[which][rr * LPR + sl][i];
    partial[((int64_t)blockIdx.x * 2 + which) * C + ch] = s;
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

static int ln_grid(int64_t n, int lpr) {
  const int rpb = LN_THREADS / lpr;
  int64_t g = ptc_cdiv(n, rpb);
  if (g > 2048) g = 2048;
  return (int)(g < 1 ? 1 : g);
}
static bool ln_supported_c(int c) { return c == 32 || c == 64 || c == 128 || c == 256 || c == 512; }

extern "C" int ptc_layer_norm_supported(int c) { return ln_supported_c(c) ? 1 : 0; }

template <typename TI, typename TO>
static int launch_ln_fwd(const void* x, int64_t n, int c, const float* gamma, const float* beta, float eps, void* y,
                         float* mean, float* rstd, hipStream_t s) {
#define LN_FWD_CASE(LPR)                                                                                     \
  hipLaunchKernelGGL((layer_norm_fwd_kernel<TI, TO, LPR>), dim3(ln_grid(n, LPR)), dim3(LN_THREADS), 0, s,    \
                     (const TI*)x, n, gamma, beta, eps, (TO*)y, mean, rstd)
  switch (c / LN_VEC) {
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
  PTC_REQUIRE(ln_supported_c(c), PTC_EUNSUPPORTED, "ptc_layer_norm_fwd: C=%d not in {32,64,128,256,512}", c);
  if (n == 0) return PTC_OK;
  PTC_REQUIRE(x && y && mean && rstd, PTC_EINVAL, "ptc_layer_norm_fwd: null buffer");
  PTC_REQUIRE(out_dtype == PTC_F32 || out_dtype == in_dtype || in_dtype == PTC_F32, PTC_EUNSUPPORTED,
              "ptc_layer_norm_fwd: unsupported dtype pair %d -> %d", in_dtype, out_dtype);
  hipStream_t s = (hipStream_t)stream;
  PTC_DISPATCH_DTYPE(in_dtype, TI, {
    if (out_dtype == PTC_F32) return launch_ln_fwd<TI, float>(x, n, c, gamma, beta, eps, y, mean, rstd, s);
    if (out_dtype == PTC_BF16) return launch_ln_fwd<TI, bf16_t>(x, n, c, gamma, beta, eps, y, mean, rstd, s);
    return launch_ln_fwd<TI, f16_t>(x, n, c, gamma, beta, eps, y, mean, rstd, s);
  });
  return PTC_OK;
}

extern "C" size_t ptc_layer_norm_bwd_workspace_bytes(int64_t n, int c) {
  if (!ln_supported_c(c)) return 256;
  return ptc_align_up((size_t)ln_grid(n, c / LN_VEC) * 2 * (size_t)c * sizeof(float), 256);
}

template <typename TG, typename TX>
static int launch_ln_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                         int64_t n, int c, void* dx, float* dgamma, float* dbeta, void* ws, hipStream_t s) {
  const int lpr = c / LN_VEC;
  const int grid = ln_grid(n, lpr);
#define LN_BWD_CASE(LPR)                                                                                          \
  hipLaunchKernelGGL((layer_norm_bwd_kernel<TG, TX, LPR>), dim3(grid), dim3(LN_THREADS), 0, s, (const TG*)dy,     \
                     (const TX*)x, mean, rstd, gamma, n, (TX*)dx, (float*)ws)
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
  PTC_REQUIRE(ln_supported_c(c), PTC_EUNSUPPORTED, "ptc_layer_norm_bwd: C=%d not in {32,64,128,256,512}", c);
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
  PTC_DISPATCH_DTYPE(x_dtype, TX, {
    if (dy_dtype == PTC_F32) return launch_ln_bwd<float, TX>(dy, x, mean, rstd, gamma, n, c, dx, dgamma, dbeta, workspace, s);
    if (dy_dtype == PTC_BF16) return launch_ln_bwd<bf16_t, TX>(dy, x, mean, rstd, gamma, n, c, dx, dgamma, dbeta, workspace, s);
    return launch_ln_bwd<f16_t, TX>(dy, x, mean, rstd, gamma, n, c, dx, dgamma, dbeta, workspace, s);
  });
  return PTC_OK;
}
