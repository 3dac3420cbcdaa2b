// lovasz.hip -- Lovasz-Softmax loss (multiclass, classes = "present", whole batch) forward + gradient.
//
// Replaces LovaszLoss(mode="multiclass", ignore_index) of pointcept/models/losses/lovasz.py:118-146 (_lovasz_softmax_flat),
// :22-33 (_lovasz_grad), :149-166 (_flatten_probas) as configured at configs/scannet/semseg-pt-v3m1-0-base.py:49-52 and
// called from pointcept/models/default.py:78-84.  The reference loops over the classes present in the labels and, for
// each, runs softmax column -> |fg - p| -> torch.sort(descending) -> cumsum -> Jaccard differences -> dot: 20 sorts
// of N floats plus ~12 elementwise launches per class.  Here the whole loss is six launches around ONE segmented
// radix sort of the [C, N] error matrix (the row-batched sort of scan_sort.hip that also orders the serialization
// curves):
//   1. lovasz_keys      : per point softmax in fp32 straight from the (strided, 16-bit) head output; key[c][i] =
//                         (0x3f800000 - bits(|fg - p_c|)) << 1 | fg: errors lie in [0, 1], so the 30-bit integer ascends as the
//                         error descends.  Ignored points get error 0 (they sort to the tail and multiply their
//                         Jaccard step by 0).  Class populations are counted with integer atomics (exact, order free).
//   2. ptc_sort_keys_ex : C rows of N keys, bits [1, 31): 4 passes; the foreground flag in bit 0 is a payload the sort carries along,
//                         and the sorted key words come back with the order (round 5: the first form gathered target[order[t]] for the
//                         flag and keys[order[t]] for the error -- two random 8-byte gathers per slot, 143 + 452 us at 819200 x 20,
//                         1.8 GB of sector traffic in lovasz_step for 0.5 GB of operands, profiles/r04_zy_step_traffic.txt).
//   3. lovasz_fg        : foreground flag of every sorted slot (bit 0 of its key); ptc_exclusive_scan_i32 over the flat [C*N] flags.
//   4. lovasz_step      : Jaccard step of slot i, computed EXACTLY from the integer counts instead of as the difference
//                         of two nearly equal quotients (lovasz.py:31-32): with I = fg still to come, U = union so far,
//                         step = 1/U for a foreground slot and I/(U (U-1)) for a background slot.  Accumulates
//                         error * step per workgroup (fixed order, no float atomics) and writes the gradient w.r.t.
//                         the probability back to the point: g[c][src] = -+ step / n_present.
//   5. lovasz_finish    : sums the partials in a fixed order -> loss.
//   6. lovasz_dlogits   : softmax backward per point, dz = p * (g - <g, p>), 0 for ignored points.
// All of it is HBM-bound streaming work (~190 B per (point, class) slot); bit-reproducible.
#include "ptc_common.h"
#include "voxel_keys.h"
#include "loss_rows.h"

#define LV_THREADS 256
#define LV_MAX_C 64
#define LV_ONE 0x3f800000u
#define LV_STEP_BLOCKS 4096

// key word of one (class, point) slot: bits [1, 31) ascend as the error descends, bit 0 = the slot is foreground (its point carries this
// class) -- below the sorted bit range, carried along by the sort
__device__ __forceinline__ int64_t lv_key(float e, bool fg) { return (int64_t)(((uint64_t)(LV_ONE - __float_as_uint(e)) << 1) | (fg ? 1u : 0u)); }
__device__ __forceinline__ float lv_key_error(int64_t key) { return __uint_as_float(LV_ONE - (uint32_t)((uint64_t)key >> 1)); }

template <typename T>
__global__ void __launch_bounds__(LV_THREADS)
lovasz_keys_kernel(const T* __restrict__ logits, int64_t row_stride, const int64_t* __restrict__ target, int64_t n, int c,
                   int64_t ignore_index, int64_t* __restrict__ keys, int32_t* __restrict__ class_count) {
  __shared__ int32_t cnt[LV_MAX_C];
  if (threadIdx.x < LV_MAX_C) cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * LV_THREADS + threadIdx.x;
  if (i < n) {
    const T* row = logits + i * row_stride;
    const int64_t t = target[i];
    const bool valid = t != ignore_index && t >= 0 && t < c;
    float m = -INFINITY;
    for (int j = 0; j < c; ++j) m = fmaxf(m, ptc_to_float(row[j]));
    float ssum = 0.f;
    for (int j = 0; j < c; ++j) ssum += __expf(ptc_to_float(row[j]) - m);
    const float inv = 1.f / ssum;
    for (int j = 0; j < c; ++j) {
      const float p = __expf(ptc_to_float(row[j]) - m) * inv;
      float e = valid ? (j == t ? 1.f - p : p) : 0.f;
      e = fminf(fmaxf(e, 0.f), 1.f);
      keys[(int64_t)j * n + i] = lv_key(e, valid && j == t);
    }
    if (valid) atomicAdd(&cnt[(int)t], 1);
  }
  __syncthreads();
  if (threadIdx.x < c && cnt[threadIdx.x] != 0) atomicAdd(&class_count[threadIdx.x], cnt[threadIdx.x]);
}

// C <= LR_CP: the row in registers (loss_rows.h); exp(x - m) is evaluated once per class instead of twice (the same call on the same
// operands: the same value)
template <typename T, int VB>
__global__ void __launch_bounds__(LV_THREADS)
lovasz_keys_rows_kernel(const T* __restrict__ logits, int64_t row_stride, const int64_t* __restrict__ target, int64_t n, int c,
                        int64_t ignore_index, int64_t* __restrict__ keys, int32_t* __restrict__ class_count) {
  __shared__ int32_t cnt[LV_MAX_C];
  if (threadIdx.x < LV_MAX_C) cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * LV_THREADS + threadIdx.x;
  if (i < n) {
    float v[LR_CP];
    lr_load_row<T, VB>(logits + i * row_stride, c, v);
    const int64_t t = target[i];
    const bool valid = t != ignore_index && t >= 0 && t < c;
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < LR_CP; ++j) if (j < c) m = fmaxf(m, v[j]);
    float ssum = 0.f;
#pragma unroll
    for (int j = 0; j < LR_CP; ++j) if (j < c) { v[j] = __expf(v[j] - m); ssum += v[j]; }
    const float inv = 1.f / ssum;
#pragma unroll
    for (int j = 0; j < LR_CP; ++j) {
      if (j < c) {
        const float p = v[j] * inv;
        float e = valid ? (j == t ? 1.f - p : p) : 0.f;
        e = fminf(fmaxf(e, 0.f), 1.f);
        keys[(int64_t)j * n + i] = lv_key(e, valid && j == t);
      }
    }
    if (valid) atomicAdd(&cnt[(int)t], 1);
  }
  __syncthreads();
  if (threadIdx.x < c && cnt[threadIdx.x] != 0) atomicAdd(&class_count[threadIdx.x], cnt[threadIdx.x]);
}

__global__ void __launch_bounds__(LV_THREADS)
lovasz_fg_kernel(const int64_t* __restrict__ sorted_keys, int64_t total, int32_t* __restrict__ fg) {
  const int64_t t = (int64_t)blockIdx.x * LV_THREADS + threadIdx.x;
  if (t < total) fg[t] = (int32_t)(sorted_keys[t] & 1);
}

// one slot of lovasz_step: Jaccard step, the slot's share of the loss, the gradient w.r.t. the probability scattered back to the point.
// ONE fp64 division per slot: the two branches of ptc_lovasz_step (voxel_keys.h) share it through selects of numerator and denominator
// (same operands, same quotient), and the 1 / n_present of the gradient is a multiplication by the reciprocal the workgroup computed once.
__device__ __forceinline__ double lv_step_slot(int64_t key, int32_t src, int32_t cum_fg_excl, int32_t i, int32_t gts, double inv_present,
                                               float* __restrict__ gprob_row) {
  const int f = (int)(key & 1);
  const int32_t cum_fg = cum_fg_excl + f;              // inclusive
  const int32_t cum_bg = (i + 1) - cum_fg;
  const double U = (double)gts + (double)cum_bg, I = (double)(gts - cum_fg);
  const double step = (f ? 1.0 : I) / (f ? U : U * (U - 1.0));   // exact Jaccard difference, ptc_lovasz_step's two quotients
  const float e = lv_key_error(key);
  float g = (float)(step * inv_present);
  g = f ? -g : g;                                      // d|fg - p| / dp
  gprob_row[src] = g;
  return (double)e * step;
}

// grid (G, c): class row blockIdx.y, G workgroups stride over its n sorted slots.  The first form of this kernel was ONE flat grid over the
// c n slots: `row = t / n` is a 64-bit integer division per slot (a ~100-instruction software sequence) and the step took three fp64 divisions
// (both Jaccard branches + step / n_present): 338 us at 819200 x 20 for 460 MB of streams -- compute, not the 4-byte scatter, bound it
// (profiles/r05_s_lovasz_xcd_rows.txt had already shown that the write-backs do not).  Rows with no foreground (gts == 0: class absent) give
// zero gradient and contribute nothing.
__global__ void __launch_bounds__(LV_THREADS)
lovasz_step_kernel(const int64_t* __restrict__ sorted_keys, const int64_t* __restrict__ order,
                   const int64_t* __restrict__ fg_scan, const int32_t* __restrict__ class_count, int64_t n, int c,
                   float* __restrict__ gprob, double* __restrict__ partial) {
  __shared__ double red[LV_THREADS / 64];
  __shared__ int n_present_s;
  if (threadIdx.x == 0) {
    int np = 0;
    for (int j = 0; j < c; ++j) np += class_count[j] > 0 ? 1 : 0;
    n_present_s = np;
  }
  __syncthreads();
  const double inv_present = 1.0 / (double)n_present_s;
  const int row = (int)blockIdx.y;
  const int64_t base = (int64_t)row * n;
  const int32_t gts = class_count[row], nn = (int32_t)n;
  const int64_t* __restrict__ kr = sorted_keys + base;
  const int64_t* __restrict__ orr = order + base;
  const int64_t* __restrict__ sr = fg_scan + base;
  float* __restrict__ gr = gprob + base;
  double contrib = 0.0;
  if (gts > 0) {
    const int64_t scan0 = sr[0];
    for (int32_t i = (int32_t)(blockIdx.x * LV_THREADS + threadIdx.x); i < nn; i += (int32_t)(gridDim.x * LV_THREADS))
      contrib += lv_step_slot(kr[i], (int32_t)orr[i], (int32_t)(sr[i] - scan0), i, gts, inv_present, gr);
  } else {
    for (int32_t i = (int32_t)(blockIdx.x * LV_THREADS + threadIdx.x); i < nn; i += (int32_t)(gridDim.x * LV_THREADS)) gr[i] = 0.f;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) contrib += __shfl_xor(contrib, o, 64);
  if (ptc_lane() == 0) red[threadIdx.x >> 6] = contrib;
  __syncthreads();
  if (threadIdx.x == 0) partial[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(LV_THREADS)
lovasz_finish_kernel(const double* __restrict__ partial, int64_t n_partial, const int32_t* __restrict__ class_count, int c,
                     float* __restrict__ loss) {
  __shared__ double red[LV_THREADS];
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < n_partial; i += LV_THREADS) acc += partial[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = LV_THREADS / 2; s >= 1; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int np = 0;
    for (int j = 0; j < c; ++j) np += class_count[j] > 0 ? 1 : 0;
    loss[0] = np > 0 ? (float)(red[0] / (double)np) : 0.f;
  }
}

template <typename T>
__global__ void __launch_bounds__(LV_THREADS)
lovasz_dlogits_kernel(const T* __restrict__ logits, int64_t row_stride, const int64_t* __restrict__ target,
                      const float* __restrict__ gprob, int64_t n, int c, int64_t ignore_index, float* __restrict__ dlogits) {
  const int64_t i = (int64_t)blockIdx.x * LV_THREADS + threadIdx.x;
  if (i >= n) return;
  const T* row = logits + i * row_stride;
  float* drow = dlogits + i * (int64_t)c;
  const int64_t t = target[i];
  const bool valid = t != ignore_index && t >= 0 && t < c;
  if (!valid) {
    for (int j = 0; j < c; ++j) drow[j] = 0.f;
    return;
  }
  float m = -INFINITY;
  for (int j = 0; j < c; ++j) m = fmaxf(m, ptc_to_float(row[j]));
  float ssum = 0.f;
  for (int j = 0; j < c; ++j) ssum += __expf(ptc_to_float(row[j]) - m);
  const float inv = 1.f / ssum;
  float dot = 0.f;
  for (int j = 0; j < c; ++j) dot += gprob[(int64_t)j * n + i] * (__expf(ptc_to_float(row[j]) - m) * inv);
  for (int j = 0; j < c; ++j) {
    const float p = __expf(ptc_to_float(row[j]) - m) * inv;
    drow[j] = p * (gprob[(int64_t)j * n + i] - dot);
  }
}

// C <= LR_CP: row in registers, the gradient rows out through LDS (loss_rows.h); the arithmetic is lovasz_dlogits_kernel's
template <typename T, int VB>
__global__ void __launch_bounds__(LV_THREADS)
lovasz_dlogits_rows_kernel(const T* __restrict__ logits, int64_t row_stride, const int64_t* __restrict__ target,
                           const float* __restrict__ gprob, int64_t n, int c, int64_t ignore_index, float* __restrict__ dlogits, int a16) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // [256][c] fp32: this workgroup's chunk of dlogits
  const int64_t row0 = (int64_t)blockIdx.x * LV_THREADS, i = row0 + threadIdx.x;
  if (i < n) {
    float* drow = reinterpret_cast<float*>(smem) + (int)threadIdx.x * c;
    const int64_t t = target[i];
    const bool valid = t != ignore_index && t >= 0 && t < c;
    if (!valid) {
#pragma unroll
      for (int j = 0; j < LR_CP; ++j) if (j < c) drow[j] = 0.f;
    } else {
      float v[LR_CP], gp[LR_CP];
      lr_load_row<T, VB>(logits + i * row_stride, c, v);
#pragma unroll
      for (int j = 0; j < LR_CP; ++j) gp[j] = j < c ? gprob[(int64_t)j * n + i] : 0.f;
      float m = -INFINITY;
#pragma unroll
      for (int j = 0; j < LR_CP; ++j) if (j < c) m = fmaxf(m, v[j]);
      float ssum = 0.f;
#pragma unroll
      for (int j = 0; j < LR_CP; ++j) if (j < c) { v[j] = __expf(v[j] - m); ssum += v[j]; }
      const float inv = 1.f / ssum;
      float dot = 0.f;
#pragma unroll
      for (int j = 0; j < LR_CP; ++j) if (j < c) dot += gp[j] * (v[j] * inv);
#pragma unroll
      for (int j = 0; j < LR_CP; ++j) {
        if (j < c) {
          const float p = v[j] * inv;
          drow[j] = p * (gp[j] - dot);
        }
      }
    }
  }
  __syncthreads();
  const int64_t rows = (n - row0) < LV_THREADS ? (n - row0) : LV_THREADS;
  lr_copy_out<float>(smem, dlogits + row0 * c, (int)rows * c, a16 != 0);
}

struct LvLayout {
  size_t keys, order, fg, scan, gprob, partial, count, sort_ws, scan_ws, total;
  int64_t n_partial;
};
static LvLayout lv_layout(int64_t n, int c) {
  LvLayout L;
  const size_t nc = (size_t)(n > 0 ? n : 1) * (size_t)c;
  int64_t g = ptc_cdiv(n > 0 ? n : 1, LV_THREADS);           // lovasz_step: G workgroups per class row, one partial each
  if (g > LV_STEP_BLOCKS / c) g = LV_STEP_BLOCKS / c;
  L.n_partial = (g > 0 ? g : 1) * c;
  size_t o = 0;
  L.keys = o; o += ptc_align_up(nc * 8, 256);
  L.order = o; o += ptc_align_up(nc * 8, 256);
  L.fg = o; o += ptc_align_up(nc * 4, 256);
  L.scan = o; o += ptc_align_up(nc * 8, 256);
  L.gprob = o; o += ptc_align_up(nc * 4, 256);
  L.partial = o; o += ptc_align_up((size_t)L.n_partial * 8, 256);
  L.count = o; o += 256;
  L.sort_ws = o; o += ptc_align_up(ptc_sort_keys_workspace_bytes(n, c), 256);
  L.scan_ws = o; o += ptc_align_up(ptc_exclusive_scan_workspace_bytes((int64_t)nc), 256);
  L.total = o;
  return L;
}

extern "C" size_t ptc_lovasz_softmax_workspace_bytes(int64_t n, int c) {
  if (n < 0 || c < 1 || c > LV_MAX_C) return 0;
  return lv_layout(n, c).total;
}

extern "C" int ptc_lovasz_softmax(const void* logits, int64_t row_stride, const int64_t* target, int64_t n, int c, int dtype,
                                  int64_t ignore_index, float* loss, float* dlogits, void* workspace,
                                  size_t workspace_bytes, ptc_stream_t stream) {
  PTC_REQUIRE(n >= 0 && c >= 1 && row_stride >= c, PTC_EINVAL, "ptc_lovasz_softmax: bad sizes");
  PTC_REQUIRE(c <= LV_MAX_C, PTC_EUNSUPPORTED, "ptc_lovasz_softmax: c=%d > %d classes", c, LV_MAX_C);
  PTC_REQUIRE(n < (1ll << 31), PTC_EUNSUPPORTED, "ptc_lovasz_softmax: n >= 2^31");
  PTC_REQUIRE(loss != nullptr, PTC_EINVAL, "ptc_lovasz_softmax: null loss");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    PTC_HIP(hipMemsetAsync(loss, 0, sizeof(float), s));
    return PTC_OK;
  }
  PTC_REQUIRE(logits && target && dlogits && workspace, PTC_EINVAL, "ptc_lovasz_softmax: null buffer");
  const LvLayout L = lv_layout(n, c);
  PTC_REQUIRE(workspace_bytes >= L.total, PTC_EWORKSPACE, "ptc_lovasz_softmax: workspace %zu < %zu", workspace_bytes, L.total);
  char* ws = (char*)workspace;
  int64_t* keys = (int64_t*)(ws + L.keys);
  int64_t* order = (int64_t*)(ws + L.order);
  int32_t* fg = (int32_t*)(ws + L.fg);
  int64_t* scan = (int64_t*)(ws + L.scan);
  float* gprob = (float*)(ws + L.gprob);
  double* partial = (double*)(ws + L.partial);
  int32_t* count = (int32_t*)(ws + L.count);
  const int64_t nc = n * (int64_t)c;
  const unsigned grid_n = (unsigned)ptc_cdiv(n, LV_THREADS), grid_nc = (unsigned)ptc_cdiv(nc, LV_THREADS);

  const bool rows = c <= LR_CP;                      // the row-in-registers kernels (loss_rows.h)
  const int vb = lr_vec_bytes(logits, row_stride, c, ptc_dtype_size(dtype));
  PTC_HIP(hipMemsetAsync(count, 0, 256, s));
  if (rows) {
    PTC_DISPATCH_DTYPE(dtype, T, LR_DISPATCH_VB(vb, VB, hipLaunchKernelGGL((lovasz_keys_rows_kernel<T, VB>), dim3(grid_n), dim3(LV_THREADS), 0, s,
                                                                         (const T*)logits, row_stride, target, n, c, ignore_index, keys, count)))
  } else {
    PTC_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(lovasz_keys_kernel<T>, dim3(grid_n), dim3(LV_THREADS), 0, s, (const T*)logits,
                                                     row_stride, target, n, c, ignore_index, keys, count))
  }
  PTC_CHECK_LAUNCH("lovasz_keys_kernel");
  // the sorted key words replace the unsorted ones in place (the sort reads `keys` in its first pass only)
  int rc = ptc_sort_keys_ex(keys, n, c, 1, 31, order, nullptr, keys, ws + L.sort_ws, L.scan_ws - L.sort_ws, stream);
  if (rc != PTC_OK) return rc;
  hipLaunchKernelGGL(lovasz_fg_kernel, dim3(grid_nc), dim3(LV_THREADS), 0, s, keys, nc, fg);
  PTC_CHECK_LAUNCH("lovasz_fg_kernel");
  rc = ptc_exclusive_scan_i32(fg, nc, scan, ws + L.scan_ws, L.total - L.scan_ws, stream);
  if (rc != PTC_OK) return rc;
  hipLaunchKernelGGL(lovasz_step_kernel, dim3((unsigned)(L.n_partial / c), (unsigned)c), dim3(LV_THREADS), 0, s, keys, order, scan, count, n, c, gprob,
                     partial);
  PTC_CHECK_LAUNCH("lovasz_step_kernel");
  hipLaunchKernelGGL(lovasz_finish_kernel, dim3(1), dim3(LV_THREADS), 0, s, partial, L.n_partial, count, c, loss);
  PTC_CHECK_LAUNCH("lovasz_finish_kernel");
  if (rows) {
    const int a16 = ((uintptr_t)dlogits & 15) == 0;
    const size_t lds = (size_t)LV_THREADS * c * sizeof(float);
    PTC_DISPATCH_DTYPE(dtype, T, LR_DISPATCH_VB(vb, VB, hipLaunchKernelGGL((lovasz_dlogits_rows_kernel<T, VB>), dim3(grid_n), dim3(LV_THREADS), lds, s,
                                                                         (const T*)logits, row_stride, target, gprob, n, c, ignore_index, dlogits,
                                                                         a16)))
  } else {
    PTC_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(lovasz_dlogits_kernel<T>, dim3(grid_n), dim3(LV_THREADS), 0, s, (const T*)logits,
                                                     row_stride, target, gprob, n, c, ignore_index, dlogits))
  }
  PTC_CHECK_LAUNCH("lovasz_dlogits_kernel");
  return PTC_OK;
}
