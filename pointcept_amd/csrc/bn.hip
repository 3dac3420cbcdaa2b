// bn.hip -- BatchNorm1d over the rows of [N, C] point features with the following activation fused.
//
// Where the reference has `Linear/conv -> BatchNorm1d(eps=1e-3, momentum=0.01) -> GELU` (PTv3:
// Embedding ptv3m1:485-515, SerializedPooling :371-444, SerializedUnpooling :447-482, norm built at :581)
// or `conv -> BatchNorm1d -> ReLU` (SpUNet: spconv_unet_v1m1_base.py:49-68,110-121,137-146,173-181), ATen runs
// six elementwise / reduction kernels per site and step (statistics, transform, activation; activation
// backward, reduce, elementwise) and materialises the pre-activation.  Here:
//   forward  : bn_reduce (one read of x -> per-workgroup (sum, sum of squares) partials) -> bn_finish (fp64
//              sums in a fixed order, running statistics, per-channel scale / shift) -> bn_apply (y = act(x*a+b))
//   backward : bn_bwd_reduce (x, dy -> sum dz, sum dz*xhat with dz = dy*act'(z), z RECOMPUTED from x)
//              -> bn_bwd_finish -> bn_bwd_apply (dx = a*(dz - mean(dz) - xhat*mean(dz*xhat)))
// 8 instead of 13 passes over [N, C], nothing saved but x, mean and rstd, fixed reduction order
// (bit-reproducible), statistics in fp32/fp64 whatever the feature dtype.  Per-GPU statistics, as in
// the reference (sync_bn = False).
#include "ptc_common.h"

#define BN_THREADS 256
#define BN_MAX_GROUPS 1024
enum { BN_ACT_NONE = 0, BN_ACT_GELU = 1, BN_ACT_RELU = 2 };

template <typename T> struct BnVec;
template <> struct BnVec<float> { static constexpr int V = 4; };
template <> struct BnVec<bf16_t> { static constexpr int V = 8; };
template <> struct BnVec<f16_t> { static constexpr int V = 8; };

// A thread's lane is V consecutive channels = 16 bytes by default (BnVec<T>::V).  Round 6: channel counts that are no multiple of that
// (PT-v3m3 54 / 108, LitePT 36 / 252: configs/utonia/*:21, litept_v1.py:601) take 8- or 4-byte lanes (V halved once or twice) through the
// SAME kernels -- they ran torch.nn.BatchNorm1d until round 5, the last library operator behind a module mirror.
template <int BYTES> struct BnRaw;
template <> struct BnRaw<16> { using type = uint4; };
template <> struct BnRaw<8> { using type = uint2; };
template <> struct BnRaw<4> { using type = uint32_t; };
template <typename T, int V = BnVec<T>::V>
__device__ __forceinline__ void bn_load(const T* p, float (&v)[V]) {
  using R = typename BnRaw<V * (int)sizeof(T)>::type;
  const R raw = *reinterpret_cast<const R*>(p);
  T tmp[V];
  __builtin_memcpy(tmp, &raw, sizeof(R));
#pragma unroll
  for (int j = 0; j < V; ++j) v[j] = ptc_to_float(tmp[j]);
}
template <typename T, int V = BnVec<T>::V>
__device__ __forceinline__ void bn_store(T* p, const float (&v)[V]) {
  using R = typename BnRaw<V * (int)sizeof(T)>::type;
  T tmp[V];
#pragma unroll
  for (int j = 0; j < V; ++j) tmp[j] = ptc_from_float<T>(v[j]);
  R raw;
  __builtin_memcpy(&raw, tmp, sizeof(R));
  *reinterpret_cast<R*>(p) = raw;
}

__device__ __forceinline__ float bn_act(float z, int act) {
  if (act == BN_ACT_GELU) return ptc_gelu(z);   // nn.GELU() (erf form; ptc_common.h)
  if (act == BN_ACT_RELU) return z > 0.f ? z : 0.f;
  return z;
}
__device__ __forceinline__ float bn_act_grad(float z, int act) {
  if (act == BN_ACT_GELU) return ptc_gelu_grad(z);
  if (act == BN_ACT_RELU) return z > 0.f ? 1.f : 0.f;
  return 1.f;
}

// Row / column-group decomposition shared by all kernels: a thread owns V consecutive channels (16 bytes)
// of rows r, r + rpi, ... ; cgs = C / V column groups, rpi = 256 / cgs rows per sweep.
struct BnMap { int cgs, rpi; };
template <typename T, int V = BnVec<T>::V> __host__ __device__ __forceinline__ BnMap bn_map(int c) {
  BnMap m;
  m.cgs = c / V;
  m.rpi = BN_THREADS / m.cgs;
  return m;
}

// ---- per-channel two-value reduction over rows: workgroup partials --------------------------------
// MODE 0: (sum x, sum x^2);  MODE 1: (sum dz, sum dz * xhat).  A thread adds <= ~100 values in fp32; the
// <= 1024 workgroup partials are then summed in fp64 by the finish kernels.
// the saved statistics of a layer as the backward kernels read them: a = gamma * rstd, b = beta - mean * a are rebuilt per thread
// (4 loads per channel, once per kernel) instead of by a launch of their own in front of every backward (59 per SpUNet step)
struct BnStat { const float* gamma; const float* beta; const float* mean; const float* rstd; };
__device__ __forceinline__ void bn_stat_coef(const BnStat& st, int ch, float& a, float& b, float& mu, float& rs) {
  rs = st.rstd[ch];
  mu = st.mean[ch];
  a = (st.gamma ? st.gamma[ch] : 1.f) * rs;
  b = __builtin_fmaf(-mu, a, st.beta ? st.beta[ch] : 0.f);
}

template <typename T, typename TD, int MODE, int V = BnVec<T>::V>
__global__ void __launch_bounds__(BN_THREADS)
bn_reduce_kernel(const T* __restrict__ x, const TD* __restrict__ dy, BnStat st, int64_t n, int c,
                 int act, int64_t rows_per_group, float* __restrict__ partial, const T* __restrict__ res = nullptr) {
  __shared__ float red[BN_THREADS][2 * V + 1];
  const BnMap m = bn_map<T, V>(c);
  const int cg = threadIdx.x % m.cgs, r = threadIdx.x / m.cgs;
  const int64_t row_lo = (int64_t)blockIdx.x * rows_per_group;
  const int64_t row_hi = (row_lo + rows_per_group) < n ? (row_lo + rows_per_group) : n;
  float s1[V], s2[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  float a[V], b[V], mu[V], rs[V];
  if (MODE == 1 && r < m.rpi) {
#pragma unroll
    for (int j = 0; j < V; ++j) {
      bn_stat_coef(st, cg * V + j, a[j], b[j], mu[j], rs[j]);
    }
  }
  if (r < m.rpi) {
    for (int64_t row = row_lo + r; row < row_hi; row += m.rpi) {
      float xv[V];
      bn_load<T, V>(x + row * c + cg * V, xv);
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < V; ++j) { s1[j] += xv[j]; s2[j] = fmaf(xv[j], xv[j], s2[j]); }
      } else {
        float dv[V];
        if constexpr (sizeof(TD) == sizeof(T)) {
          bn_load<TD, V>(dy + row * c + cg * V, dv);
        } else {                                       // 16-bit x with fp32 dy (or the reverse): element loads
#pragma unroll
          for (int j = 0; j < V; ++j) dv[j] = ptc_to_float(dy[row * c + cg * V + j]);
        }
        float rv[V];
#pragma unroll
        for (int j = 0; j < V; ++j) rv[j] = 0.f;
        if (res) bn_load<T, V>(res + row * c + cg * V, rv);          // y = act(BN(x) + res): the pre-activation includes the residual
#pragma unroll
        for (int j = 0; j < V; ++j) {
          const float dz = dv[j] * bn_act_grad(fmaf(xv[j], a[j], b[j]) + rv[j], act);
          s1[j] += dz;
          s2[j] = fmaf(dz, (xv[j] - mu[j]) * rs[j], s2[j]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < V; ++j) { red[threadIdx.x][j] = s1[j]; red[threadIdx.x][V + j] = s2[j]; }
  __syncthreads();
  // thread t < C sums its channel over the rpi row slots in a fixed order
  for (int ch = threadIdx.x; ch < c; ch += BN_THREADS) {
    const int g = ch / V, j = ch % V;
    float t1 = 0.f, t2 = 0.f;
    for (int rr = 0; rr < m.rpi; ++rr) {
      t1 += red[rr * m.cgs + g][j];
      t2 += red[rr * m.cgs + g][V + j];
    }
    float* out = partial + ((int64_t)blockIdx.x * c + ch) * 2;
    out[0] = t1;
    out[1] = t2;
  }
}

// Sum of the <= 1024 workgroup partials of one channel, in fp64 and in a fixed order.  A workgroup owns 16
// channels; 16 "parts" of 16 lanes walk the groups with stride 16, 8 independent loads in flight per lane
// (a single thread per channel walking all groups serially was latency-bound: 117 us per call, r01_p),
// then the parts are combined through LDS by part 0.
#define BN_FIN_CH 16
__device__ __forceinline__ bool bn_sum_partials(const float* __restrict__ partial, int groups, int c, double& s1, double& s2, int& ch) {
  __shared__ double fin[BN_THREADS / BN_FIN_CH][BN_FIN_CH][2];
  const int lc = threadIdx.x % BN_FIN_CH, part = threadIdx.x / BN_FIN_CH;
  constexpr int PARTS = BN_THREADS / BN_FIN_CH;
  ch = blockIdx.x * BN_FIN_CH + lc;
  double a1[8], a2[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) { a1[u] = 0.0; a2[u] = 0.0; }
  if (ch < c) {
    for (int g0 = part; g0 < groups; g0 += PARTS * 8) {
      // (all eight loads issued before the first add: unconditional, from a clamped address -- a load under a per-lane condition
      //  is waited for one by one; these finish kernels were 10 us of pure latency per BatchNorm layer and direction)
      float2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int g = g0 + u * PARTS;
        v[u] = *reinterpret_cast<const float2*>(partial + ((int64_t)(g < groups ? g : groups - 1) * c + ch) * 2);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool live = g0 + u * PARTS < groups;
        a1[u] += live ? (double)v[u].x : 0.0;
        a2[u] += live ? (double)v[u].y : 0.0;
      }
    }
  }
  fin[part][lc][0] = ((a1[0] + a1[1]) + (a1[2] + a1[3])) + ((a1[4] + a1[5]) + (a1[6] + a1[7]));
  fin[part][lc][1] = ((a2[0] + a2[1]) + (a2[2] + a2[3])) + ((a2[4] + a2[5]) + (a2[6] + a2[7]));
  __syncthreads();
  if (part != 0 || ch >= c) return false;
  s1 = 0.0; s2 = 0.0;
  for (int p = 0; p < PARTS; ++p) { s1 += fin[p][lc][0]; s2 += fin[p][lc][1]; }
  return true;
}

// coef layout (fp32, 4*C): [0,C) a = gamma*rstd | [C,2C) b = beta - mean*a | [2C,3C) mean | [3C,4C) rstd
__global__ void __launch_bounds__(BN_THREADS)
bn_finish_kernel(const float* __restrict__ partial, int groups, int64_t n, int c,
                 const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum, int training,
                 float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ coef,
                 float* __restrict__ save_mean, float* __restrict__ save_rstd) {
  double s1 = 0.0, s2 = 0.0;
  int ch;
  if (training) {
    if (!bn_sum_partials(partial, groups, c, s1, s2, ch)) return;
  } else {
    ch = blockIdx.x * BN_FIN_CH + threadIdx.x;
    if (threadIdx.x >= BN_FIN_CH || ch >= c) return;
  }
  double mean, var;
  if (training) {
    const double cnt = (double)n;
    mean = s1 / cnt;
    var = s2 / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    if (running_mean) running_mean[ch] = (float)((1.0 - momentum) * running_mean[ch] + momentum * mean);
    if (running_var) running_var[ch] = (float)((1.0 - momentum) * running_var[ch] + momentum * (n > 1 ? var * cnt / (cnt - 1.0) : var));
  } else {
    mean = running_mean[ch];
    var = running_var[ch];
  }
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float a = (gamma ? gamma[ch] : 1.f) * rstd;
  coef[ch] = a;
  coef[c + ch] = (beta ? beta[ch] : 0.f) - (float)mean * a;
  coef[2 * c + ch] = (float)mean;
  coef[3 * c + ch] = rstd;
  if (save_mean) save_mean[ch] = (float)mean;
  if (save_rstd) save_rstd[ch] = rstd;
}

// Elementwise passes: a thread keeps ONE column group (its V channels' coefficients live in registers for the
// whole kernel) and walks rows r, r + rows_per_sweep, ...  (The first version re-derived the column group per
// element and re-loaded 6 coefficients per channel per element: 290 us for 819200 x 64, VALU/L1-bound.)
template <typename T, typename TY, int V = BnVec<T>::V>
__global__ void __launch_bounds__(BN_THREADS)
bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ coef, int64_t n, int c, int act, TY* __restrict__ y,
                const T* __restrict__ res = nullptr) {
  const BnMap m = bn_map<T, V>(c);
  const int cg = threadIdx.x % m.cgs, r = threadIdx.x / m.cgs;
  if (r >= m.rpi) return;
  float a[V], b[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { a[j] = coef[cg * V + j]; b[j] = coef[c + cg * V + j]; }
  const int64_t sweep = (int64_t)gridDim.x * m.rpi;
  for (int64_t row = (int64_t)blockIdx.x * m.rpi + r; row < n; row += sweep) {
    const int64_t e = row * c + cg * V;
    float xv[V];
    bn_load<T, V>(x + e, xv);
    float rv[V];
#pragma unroll
    for (int j = 0; j < V; ++j) rv[j] = 0.f;
    if (res) bn_load<T, V>(res + e, rv);
#pragma unroll
    for (int j = 0; j < V; ++j) xv[j] = bn_act(fmaf(xv[j], a[j], b[j]) + rv[j], act);
    if constexpr (sizeof(TY) == sizeof(T)) {
      bn_store<TY, V>(reinterpret_cast<TY*>(y) + e, xv);
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) y[e + j] = ptc_from_float<TY>(xv[j]);
    }
  }
}

// bcoef (fp32, 2*C): [0,C) c1 = a*S1/N | [C,2C) c2 = a*S2/N  (both 0 when the statistics were not batch statistics)
__global__ void __launch_bounds__(BN_THREADS)
bn_bwd_finish_kernel(const float* __restrict__ partial, int groups, int64_t n, int c, BnStat st, int training,
                     float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ bcoef) {
  double s1, s2;
  int ch;
  if (!bn_sum_partials(partial, groups, c, s1, s2, ch)) return;
  if (dgamma) dgamma[ch] = (float)s2;
  if (dbeta) dbeta[ch] = (float)s1;
  const double a = (st.gamma ? st.gamma[ch] : 1.f) * st.rstd[ch];
  bcoef[ch] = training ? (float)(a * s1 / (double)n) : 0.f;
  bcoef[c + ch] = training ? (float)(a * s2 / (double)n) : 0.f;
}

template <typename T, typename TD, int V = BnVec<T>::V>
__global__ void __launch_bounds__(BN_THREADS)
bn_bwd_apply_kernel(const T* __restrict__ x, const TD* __restrict__ dy, BnStat st, const float* __restrict__ bcoef,
                    int64_t n, int c, int act, T* __restrict__ dx, const T* __restrict__ res = nullptr, T* __restrict__ dres = nullptr) {
  const BnMap m = bn_map<T, V>(c);
  const int cg = threadIdx.x % m.cgs, r = threadIdx.x / m.cgs;
  if (r >= m.rpi) return;
  float a[V], b[V], mu[V], rs[V], c1[V], c2[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const int ch = cg * V + j;
    bn_stat_coef(st, ch, a[j], b[j], mu[j], rs[j]);
    c1[j] = bcoef[ch]; c2[j] = bcoef[c + ch];
  }
  const int64_t sweep = (int64_t)gridDim.x * m.rpi;
  for (int64_t row = (int64_t)blockIdx.x * m.rpi + r; row < n; row += sweep) {
    const int64_t e = row * c + cg * V;
    float xv[V], dv[V];
    bn_load<T, V>(x + e, xv);
    if constexpr (sizeof(TD) == sizeof(T)) {
      bn_load<TD, V>(dy + e, dv);
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) dv[j] = ptc_to_float(dy[e + j]);
    }
    float rv[V];
#pragma unroll
    for (int j = 0; j < V; ++j) rv[j] = 0.f;
    if (res) bn_load<T, V>(res + e, rv);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float dz = dv[j] * bn_act_grad(fmaf(xv[j], a[j], b[j]) + rv[j], act);
      const float xhat = (xv[j] - mu[j]) * rs[j];
      xv[j] = a[j] * dz - c1[j] - xhat * c2[j];
      rv[j] = dz;                                     // the residual branch receives the activation's gradient unchanged
    }
    bn_store<T, V>(dx + e, xv);
    if (dres) bn_store<T, V>(dres + e, rv);
  }
}

// column sums of [n, c] (bias gradients): the same two-level reduction, second moment ignored
__global__ void __launch_bounds__(BN_THREADS)
bn_colsum_finish_kernel(const float* __restrict__ partial, int groups, int c, float* __restrict__ out) {
  double s1, s2;
  int ch;
  if (!bn_sum_partials(partial, groups, c, s1, s2, ch)) return;
  out[ch] = (float)s1;
}

// ---- host side ---------------------------------------------------------------------------------
struct BnPlan { int groups; int64_t rows_per_group; };
static BnPlan bn_plan(int64_t n, int rpi) {
  BnPlan p;
  int64_t g = ptc_cdiv(n, (int64_t)rpi * 8);            // >= 8 sweeps per workgroup
  if (g > BN_MAX_GROUPS) g = BN_MAX_GROUPS;
  if (g < 1) g = 1;
  p.rows_per_group = ptc_cdiv(n, g);
  p.groups = (int)ptc_cdiv(n, p.rows_per_group);
  if (p.groups < 1) p.groups = 1;
  return p;
}

// channels per lane for (c, dtype): the widest of 16 / 8 / 4 bytes that divides the row and leaves <= 256 column groups; 0: unsupported
static int bn_lane(int c, int dtype) {
  const int v0 = dtype == PTC_F32 ? 4 : 8;
  for (int v = v0; v >= v0 / 4 && v >= 1; v >>= 1)
    if (c >= v && c % v == 0 && c / v <= BN_THREADS) return v;
  return 0;
}
extern "C" int ptc_batch_norm_supported(int c, int dtype) { return bn_lane(c, dtype) != 0; }
// narrow lanes exist for the same-dtype tensor pairs only (what autocast and fp32 runs produce); the mixed pairs keep 16-byte lanes
static bool bn_full_lane(int c, int dtype) { return bn_lane(c, dtype) == (dtype == PTC_F32 ? 4 : 8); }

// workspace: partial [BN_MAX_GROUPS][C][2] | coef [4C] | bcoef [2C]   (fp32)
static size_t bn_partial_bytes(int c) { return ptc_align_up((size_t)BN_MAX_GROUPS * c * 2 * sizeof(float), 256); }
extern "C" size_t ptc_batch_norm_workspace_bytes(int64_t n, int c) {
  (void)n;
  return bn_partial_bytes(c) + ptc_align_up((size_t)6 * c * sizeof(float), 256);
}

template <typename T, typename TY, int V = BnVec<T>::V>
static int bn_fwd_typed(const void* x, int64_t n, int c, const float* gamma, const float* beta, float eps, float momentum,
                        int training, float* running_mean, float* running_var, int act, void* y, float* save_mean,
                        float* save_rstd, char* ws, hipStream_t s, const void* res = nullptr) {
  float* partial = (float*)ws;
  float* coef = (float*)(ws + bn_partial_bytes(c));
  const BnMap m = bn_map<T, V>(c);
  const BnPlan p = bn_plan(n, m.rpi);
  if (training) {
    hipLaunchKernelGGL((bn_reduce_kernel<T, T, 0, V>), dim3(p.groups), dim3(BN_THREADS), 0, s, (const T*)x, (const T*)nullptr,
                       BnStat{nullptr, nullptr, nullptr, nullptr}, n, c, act, p.rows_per_group, partial);
    PTC_CHECK_LAUNCH("bn_reduce_kernel<stats>");
  }
  hipLaunchKernelGGL(bn_finish_kernel, dim3((unsigned)ptc_cdiv(c, BN_FIN_CH)), dim3(BN_THREADS), 0, s, partial, p.groups,
                     n, c, gamma, beta, eps, momentum, training, running_mean, running_var, coef, save_mean, save_rstd);
  PTC_CHECK_LAUNCH("bn_finish_kernel");
  int64_t grid = ptc_cdiv(n, (int64_t)m.rpi * 4);        // >= 4 rows per thread
  if (grid > 256 * 16) grid = 256 * 16;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((bn_apply_kernel<T, TY, V>), dim3((unsigned)grid), dim3(BN_THREADS), 0, s, (const T*)x, coef, n, c, act, (TY*)y, (const T*)res);
  PTC_CHECK_LAUNCH("bn_apply_kernel");
  return PTC_OK;
}

static int bn_act_fwd_impl(const void* x, const void* res, int64_t n, int c, int dtype, const float* gamma, const float* beta, float eps,
                           float momentum, int training, float* running_mean, float* running_var, int act, void* y,
                           int y_dtype, float* save_mean, float* save_rstd, void* workspace, size_t workspace_bytes,
                           ptc_stream_t stream) {
  PTC_REQUIRE(n >= 0 && ptc_batch_norm_supported(c, dtype), PTC_EUNSUPPORTED, "ptc_batch_norm_act_fwd: n=%lld c=%d dtype=%d unsupported",
              (long long)n, c, dtype);
  PTC_REQUIRE(act >= 0 && act <= 2, PTC_EINVAL, "ptc_batch_norm_act_fwd: bad activation %d", act);
  PTC_REQUIRE(training || (running_mean && running_var), PTC_EINVAL, "ptc_batch_norm_act_fwd: eval mode needs running statistics");
  PTC_REQUIRE(y_dtype == dtype || y_dtype == PTC_F32 || dtype == PTC_F32, PTC_EUNSUPPORTED, "ptc_batch_norm_act_fwd: dtype pair");
  if (n == 0) return PTC_OK;
  PTC_REQUIRE(x && y && workspace, PTC_EINVAL, "ptc_batch_norm_act_fwd: null buffer");
  PTC_REQUIRE(workspace_bytes >= ptc_batch_norm_workspace_bytes(n, c), PTC_EWORKSPACE, "ptc_batch_norm_act_fwd: workspace too small");
  PTC_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0), PTC_EINVAL, "ptc_batch_norm_act_fwd: buffers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  PTC_REQUIRE((uintptr_t)res % 16 == 0, PTC_EINVAL, "ptc_batch_norm_add_act_fwd: buffers must be 16-byte aligned");
#define BN_FWD(T, TY) return bn_fwd_typed<T, TY>(x, n, c, gamma, beta, eps, momentum, training, running_mean, running_var, act, y, save_mean, save_rstd, ws, s, res)
#define BN_FWD_V(T, VV) return bn_fwd_typed<T, T, VV>(x, n, c, gamma, beta, eps, momentum, training, running_mean, running_var, act, y, save_mean, save_rstd, ws, s, res)
  if (!bn_full_lane(c, dtype)) {
    const int v = bn_lane(c, dtype);
    PTC_REQUIRE(y_dtype == dtype, PTC_EUNSUPPORTED, "ptc_batch_norm_act_fwd: c=%d needs %d-channel lanes, which exist for y_dtype == dtype only", c, v);
    if (dtype == PTC_F32) { if (v == 2) BN_FWD_V(float, 2); BN_FWD_V(float, 1); }
    if (dtype == PTC_BF16) { if (v == 4) BN_FWD_V(bf16_t, 4); BN_FWD_V(bf16_t, 2); }
    if (v == 4) BN_FWD_V(f16_t, 4);
    BN_FWD_V(f16_t, 2);
  }
#undef BN_FWD_V
  if (dtype == PTC_F32 && y_dtype == PTC_F32) BN_FWD(float, float);
  if (dtype == PTC_BF16 && y_dtype == PTC_BF16) BN_FWD(bf16_t, bf16_t);
  if (dtype == PTC_F16 && y_dtype == PTC_F16) BN_FWD(f16_t, f16_t);
  if (dtype == PTC_BF16 && y_dtype == PTC_F32) BN_FWD(bf16_t, float);
  if (dtype == PTC_F16 && y_dtype == PTC_F32) BN_FWD(f16_t, float);
  if (dtype == PTC_F32 && y_dtype == PTC_BF16) BN_FWD(float, bf16_t);
  if (dtype == PTC_F32 && y_dtype == PTC_F16) BN_FWD(float, f16_t);
#undef BN_FWD
  ptc_set_error("ptc_batch_norm_act_fwd: unsupported dtype pair (%d, %d)", dtype, y_dtype);
  return PTC_EUNSUPPORTED;
}

extern "C" int ptc_batch_norm_act_fwd(const void* x, int64_t n, int c, int dtype, const float* gamma, const float* beta, float eps,
                                      float momentum, int training, float* running_mean, float* running_var, int act, void* y,
                                      int y_dtype, float* save_mean, float* save_rstd, void* workspace, size_t workspace_bytes,
                                      ptc_stream_t stream) {
  return bn_act_fwd_impl(x, nullptr, n, c, dtype, gamma, beta, eps, momentum, training, running_mean, running_var, act, y, y_dtype, save_mean,
                         save_rstd, workspace, workspace_bytes, stream);
}
// y = act(BN(x) + res): the tail of a residual block (spconv_unet_v1m1_base.py:79-83: out = bn2(conv2(.)); out = relu(out + residual)) in
// the BatchNorm's own apply pass -- the add and the activation are two more elementwise passes over [N, C] in the reference.  res has x's
// dtype; the statistics are those of x alone.
extern "C" int ptc_batch_norm_add_act_fwd(const void* x, const void* res, int64_t n, int c, int dtype, const float* gamma, const float* beta,
                                          float eps, float momentum, int training, float* running_mean, float* running_var, int act,
                                          void* y, int y_dtype, float* save_mean, float* save_rstd, void* workspace,
                                          size_t workspace_bytes, ptc_stream_t stream) {
  PTC_REQUIRE(res || n == 0, PTC_EINVAL, "ptc_batch_norm_add_act_fwd: null residual");
  return bn_act_fwd_impl(x, res, n, c, dtype, gamma, beta, eps, momentum, training, running_mean, running_var, act, y, y_dtype, save_mean,
                         save_rstd, workspace, workspace_bytes, stream);
}

template <typename T, typename TD, int V = BnVec<T>::V>
static int bn_bwd_typed(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean, const float* rstd,
                        int64_t n, int c, int training, int act, void* dx, float* dgamma, float* dbeta, char* ws, hipStream_t s,
                        const void* res = nullptr, void* dres = nullptr) {
  float* partial = (float*)ws;
  float* coef = (float*)(ws + bn_partial_bytes(c));
  float* bcoef = coef + 4 * c;
  const BnMap m = bn_map<T, V>(c);
  const BnPlan p = bn_plan(n, m.rpi);
  const BnStat st{gamma, beta, mean, rstd};
  hipLaunchKernelGGL((bn_reduce_kernel<T, TD, 1, V>), dim3(p.groups), dim3(BN_THREADS), 0, s, (const T*)x, (const TD*)dy, st, n, c, act,
                     p.rows_per_group, partial, (const T*)res);
  PTC_CHECK_LAUNCH("bn_reduce_kernel<bwd>");
  hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3((unsigned)ptc_cdiv(c, BN_FIN_CH)), dim3(BN_THREADS), 0, s, partial, p.groups, n, c, st,
                     training, dgamma, dbeta, bcoef);
  PTC_CHECK_LAUNCH("bn_bwd_finish_kernel");
  int64_t grid = ptc_cdiv(n, (int64_t)m.rpi * 4);
  if (grid > 256 * 16) grid = 256 * 16;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((bn_bwd_apply_kernel<T, TD, V>), dim3((unsigned)grid), dim3(BN_THREADS), 0, s, (const T*)x, (const TD*)dy, st, bcoef, n, c,
                     act, (T*)dx, (const T*)res, (T*)dres);
  PTC_CHECK_LAUNCH("bn_bwd_apply_kernel");
  return PTC_OK;
}

static int bn_act_bwd_impl(const void* dy, int dy_dtype, const void* x, const void* res, int x_dtype, const float* gamma, const float* beta,
                           const float* save_mean, const float* save_rstd, int64_t n, int c, int training, int act,
                           void* dx, void* dres, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                           ptc_stream_t stream) {
  PTC_REQUIRE(n >= 0 && ptc_batch_norm_supported(c, x_dtype), PTC_EUNSUPPORTED, "ptc_batch_norm_act_bwd: n=%lld c=%d dtype=%d unsupported",
              (long long)n, c, x_dtype);
  PTC_REQUIRE(act >= 0 && act <= 2, PTC_EINVAL, "ptc_batch_norm_act_bwd: bad activation %d", act);
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    if (dgamma) PTC_HIP(hipMemsetAsync(dgamma, 0, (size_t)c * sizeof(float), s));
    if (dbeta) PTC_HIP(hipMemsetAsync(dbeta, 0, (size_t)c * sizeof(float), s));
    return PTC_OK;
  }
  PTC_REQUIRE(dy && x && dx && save_mean && save_rstd && workspace, PTC_EINVAL, "ptc_batch_norm_act_bwd: null buffer");
  PTC_REQUIRE(workspace_bytes >= ptc_batch_norm_workspace_bytes(n, c), PTC_EWORKSPACE, "ptc_batch_norm_act_bwd: workspace too small");
  char* ws = (char*)workspace;
  PTC_REQUIRE(((uintptr_t)res % 16 == 0) && ((uintptr_t)dres % 16 == 0), PTC_EINVAL, "ptc_batch_norm_add_act_bwd: buffers must be 16-byte aligned");
#define BN_BWD(T, TD) return bn_bwd_typed<T, TD>(dy, x, gamma, beta, save_mean, save_rstd, n, c, training, act, dx, dgamma, dbeta, ws, s, res, dres)
#define BN_BWD_V(T, VV) return bn_bwd_typed<T, T, VV>(dy, x, gamma, beta, save_mean, save_rstd, n, c, training, act, dx, dgamma, dbeta, ws, s, res, dres)
  if (!bn_full_lane(c, x_dtype)) {
    const int v = bn_lane(c, x_dtype);
    PTC_REQUIRE(dy_dtype == x_dtype, PTC_EUNSUPPORTED, "ptc_batch_norm_act_bwd: c=%d needs %d-channel lanes, which exist for dy_dtype == x_dtype only", c, v);
    if (x_dtype == PTC_F32) { if (v == 2) BN_BWD_V(float, 2); BN_BWD_V(float, 1); }
    if (x_dtype == PTC_BF16) { if (v == 4) BN_BWD_V(bf16_t, 4); BN_BWD_V(bf16_t, 2); }
    if (v == 4) BN_BWD_V(f16_t, 4);
    BN_BWD_V(f16_t, 2);
  }
#undef BN_BWD_V
  if (x_dtype == PTC_F32 && dy_dtype == PTC_F32) BN_BWD(float, float);
  if (x_dtype == PTC_BF16 && dy_dtype == PTC_BF16) BN_BWD(bf16_t, bf16_t);
  if (x_dtype == PTC_F16 && dy_dtype == PTC_F16) BN_BWD(f16_t, f16_t);
  if (x_dtype == PTC_BF16 && dy_dtype == PTC_F32) BN_BWD(bf16_t, float);
  if (x_dtype == PTC_F16 && dy_dtype == PTC_F32) BN_BWD(f16_t, float);
  if (x_dtype == PTC_F32 && dy_dtype == PTC_BF16) BN_BWD(float, bf16_t);
  if (x_dtype == PTC_F32 && dy_dtype == PTC_F16) BN_BWD(float, f16_t);
#undef BN_BWD
  ptc_set_error("ptc_batch_norm_act_bwd: unsupported dtype pair (%d, %d)", x_dtype, dy_dtype);
  return PTC_EUNSUPPORTED;
}

extern "C" int ptc_batch_norm_act_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* gamma, const float* beta,
                                      const float* save_mean, const float* save_rstd, int64_t n, int c, int training, int act,
                                      void* dx, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                                      ptc_stream_t stream) {
  return bn_act_bwd_impl(dy, dy_dtype, x, nullptr, x_dtype, gamma, beta, save_mean, save_rstd, n, c, training, act, dx, nullptr, dgamma, dbeta,
                         workspace, workspace_bytes, stream);
}
// backward of ptc_batch_norm_add_act_fwd: dres (x's dtype) = dy * act'(BN(x) + res), dx / dgamma / dbeta the BatchNorm backward of that
extern "C" int ptc_batch_norm_add_act_bwd(const void* dy, int dy_dtype, const void* x, const void* res, int x_dtype, const float* gamma,
                                          const float* beta, const float* save_mean, const float* save_rstd, int64_t n, int c, int training,
                                          int act, void* dx, void* dres, float* dgamma, float* dbeta, void* workspace,
                                          size_t workspace_bytes, ptc_stream_t stream) {
  PTC_REQUIRE((res && dres) || n == 0, PTC_EINVAL, "ptc_batch_norm_add_act_bwd: null residual buffers");
  return bn_act_bwd_impl(dy, dy_dtype, x, res, x_dtype, gamma, beta, save_mean, save_rstd, n, c, training, act, dx, dres, dgamma, dbeta,
                         workspace, workspace_bytes, stream);
}

// out[c] (fp32) = sum over rows of x[n, c].  Replaces `grad.float().sum(0)` (the bias gradient of the CPE
// convolutions, ptv3m1:278-284: a bf16 -> fp32 copy of [N, C] plus an ATen reduction, 1.8 ms per step in r01_y)
// by one read of x.  workspace: ptc_batch_norm_workspace_bytes(n, c).
extern "C" int ptc_column_sum(const void* x, int64_t n, int c, int dtype, float* out, void* workspace, size_t workspace_bytes,
                              ptc_stream_t stream) {
  PTC_REQUIRE(n >= 0 && ptc_batch_norm_supported(c, dtype), PTC_EUNSUPPORTED, "ptc_column_sum: n=%lld c=%d dtype=%d unsupported",
              (long long)n, c, dtype);
  PTC_REQUIRE(out != nullptr, PTC_EINVAL, "ptc_column_sum: null output");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    PTC_HIP(hipMemsetAsync(out, 0, (size_t)c * sizeof(float), s));
    return PTC_OK;
  }
  PTC_REQUIRE(x && workspace, PTC_EINVAL, "ptc_column_sum: null buffer");
  PTC_REQUIRE(workspace_bytes >= ptc_batch_norm_workspace_bytes(n, c), PTC_EWORKSPACE, "ptc_column_sum: workspace too small");
  float* partial = (float*)workspace;
#define BN_CS(T) BN_CS_V(T, BnVec<T>::V)
#define BN_CS_V(T, VV)                                                                                                         \
  {                                                                                                                            \
    const BnMap m = bn_map<T, VV>(c);                                                                                          \
    const BnPlan p = bn_plan(n, m.rpi);                                                                                        \
    hipLaunchKernelGGL((bn_reduce_kernel<T, T, 0, VV>), dim3(p.groups), dim3(BN_THREADS), 0, s, (const T*)x, (const T*)nullptr, \
                       BnStat{nullptr, nullptr, nullptr, nullptr}, n, c, 0, p.rows_per_group, partial);                        \
    PTC_CHECK_LAUNCH("bn_reduce_kernel<colsum>");                                                                              \
    hipLaunchKernelGGL(bn_colsum_finish_kernel, dim3((unsigned)ptc_cdiv(c, BN_FIN_CH)), dim3(BN_THREADS), 0, s, partial, p.groups, c, out); \
    PTC_CHECK_LAUNCH("bn_colsum_finish_kernel");                                                                               \
    return PTC_OK;                                                                                                             \
  }
  if (!bn_full_lane(c, dtype)) {
    const int v = bn_lane(c, dtype);
    if (dtype == PTC_F32) { if (v == 2) BN_CS_V(float, 2) BN_CS_V(float, 1) }
    if (dtype == PTC_BF16) { if (v == 4) BN_CS_V(bf16_t, 4) BN_CS_V(bf16_t, 2) }
    if (v == 4) BN_CS_V(f16_t, 4)
    BN_CS_V(f16_t, 2)
  }
  if (dtype == PTC_F32) BN_CS(float)
  if (dtype == PTC_BF16) BN_CS(bf16_t)
  BN_CS(f16_t)
#undef BN_CS
#undef BN_CS_V
}
