// rows.hip -- HBM-bound feature-row movers of the PTv3 path:
//   * row gather (feat[order], feat[inverse], feat[pooling_inverse]; ptv3m1:188,216,478) with an
//     optional second source row (gather-form backward of the padded gather: no atomics),
//   * CSR segmented reduce with the gather fused (torch_scatter.segment_csr(src[indices], idx_ptr)
//     at ptv3m1:416-421) and its backward.
// Roofline: (rows_in + rows_out) * C * e bytes + 8 B index per row.
// Every thread moves one 16-byte chunk (8 x 16-bit or 4 x fp32 channels); consecutive threads
// cover consecutive chunks of a row, so reads inside a gathered row and all writes are coalesced.
#include "ptc_common.h"

template <typename T> struct Vec16;  // 16-byte chunk of T
template <> struct Vec16<float> { static constexpr int N = 4; float v[4]; };
template <> struct Vec16<bf16_t> { static constexpr int N = 8; bf16_t v[8]; };
template <> struct Vec16<f16_t> { static constexpr int N = 8; f16_t v[8]; };

template <typename T>
__device__ __forceinline__ Vec16<T> load16(const T* p) {
  Vec16<T> r;
  *reinterpret_cast<uint4*>(&r) = *reinterpret_cast<const uint4*>(p);
  return r;
}
template <typename T>
__device__ __forceinline__ void store16(T* p, const Vec16<T>& r) {
  *reinterpret_cast<uint4*>(p) = *reinterpret_cast<const uint4*>(&r);
}

// ------------------------------------------------------------------------------------------------
// gather rows
// ------------------------------------------------------------------------------------------------
template <typename T, bool VEC>
__global__ void __launch_bounds__(256)
gather_rows_kernel(const T* __restrict__ src, const int64_t* __restrict__ idx, const int64_t* __restrict__ idx2,
                   int64_t n_out, int c, T* __restrict__ out) {
  constexpr int W = VEC ? Vec16<T>::N : 1;
  const int cpr = c / W;  // chunks per row
  const int64_t total = n_out * cpr;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t row = t / cpr;
    const int ch = (int)(t - row * cpr) * W;
    const int64_t s1 = idx[row];
    const int64_t s2 = idx2 ? idx2[row] : -1;
    if (VEC) {
      Vec16<T> a;
      if (s1 >= 0) a = load16(src + s1 * c + ch);
      else { for (int j = 0; j < W; ++j) a.v[j] = ptc_from_float<T>(0.f); }
      if (s2 >= 0) {
        Vec16<T> b = load16(src + s2 * c + ch);
#pragma unroll
        for (int j = 0; j < W; ++j) a.v[j] = ptc_from_float<T>(ptc_to_float(a.v[j]) + ptc_to_float(b.v[j]));
      }
      store16(out + row * c + ch, a);
    } else {
      float a = s1 >= 0 ? ptc_to_float(src[s1 * c + ch]) : 0.f;
      if (s2 >= 0) a += ptc_to_float(src[s2 * c + ch]);
      out[row * c + ch] = ptc_from_float<T>(a);
    }
  }
}

// out[i] = addend[i] + src[idx[i]] (one rounding, as the reference's `parent.feat + point.feat[inverse]`, ptv3m1:478): SerializedUnpooling's
// gather and add in one pass instead of a gather kernel and an ATen add over [N, C] (round 6)
template <typename T>
__global__ void __launch_bounds__(256)
gather_rows_add_kernel(const T* __restrict__ src, const int64_t* __restrict__ idx, const T* __restrict__ addend, int64_t n_out, int c,
                       T* __restrict__ out) {
  constexpr int W = Vec16<T>::N;
  const int cpr = c / W;
  const int64_t total = n_out * cpr;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t row = t / cpr;
    const int ch = (int)(t - row * cpr) * W;
    const int64_t s1 = idx[row];
    Vec16<T> a = load16(addend + row * c + ch);
    if (s1 >= 0) {
      const Vec16<T> b = load16(src + s1 * c + ch);
#pragma unroll
      for (int j = 0; j < W; ++j) a.v[j] = ptc_from_float<T>(ptc_to_float(a.v[j]) + ptc_to_float(b.v[j]));
    }
    store16(out + row * c + ch, a);
  }
}

extern "C" int ptc_gather_rows_add(const void* src, int64_t n_src, const int64_t* idx, const void* addend, int64_t n_out, int c, int dtype,
                                   void* out, ptc_stream_t stream) {
  PTC_REQUIRE(n_src >= 0 && n_out >= 0 && c >= 1, PTC_EINVAL, "ptc_gather_rows_add: bad sizes");
  if (n_out == 0) return PTC_OK;
  PTC_REQUIRE(src && idx && addend && out, PTC_EINVAL, "ptc_gather_rows_add: null buffer");
  const int w = dtype == PTC_F32 ? 4 : 8;
  PTC_REQUIRE(c % w == 0 && (uintptr_t)src % 16 == 0 && (uintptr_t)addend % 16 == 0 && (uintptr_t)out % 16 == 0, PTC_EUNSUPPORTED,
              "ptc_gather_rows_add: rows of whole 16-byte lanes only (c=%d)", c);
  int64_t grid = ptc_cdiv(n_out * (c / w), 256);
  if (grid > 256 * 32) grid = 256 * 32;
  PTC_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((gather_rows_add_kernel<T>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const T*)src, idx,
                                                  (const T*)addend, n_out, c, (T*)out));
  PTC_CHECK_LAUNCH("gather_rows_add_kernel");
  return PTC_OK;
}

template <typename T>
static int launch_gather_rows(const void* src, const int64_t* idx, const int64_t* idx2, int64_t n_out, int c,
                              void* out, hipStream_t s) {
  const bool vec = (c % Vec16<T>::N) == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)out % 16 == 0);
  const int W = vec ? Vec16<T>::N : 1;
  const int64_t total = n_out * (c / W);
  int64_t grid = ptc_cdiv(total, 256);
  if (grid > 256 * 32) grid = 256 * 32;
  if (vec)
    hipLaunchKernelGGL((gather_rows_kernel<T, true>), dim3((unsigned)grid), dim3(256), 0, s, (const T*)src, idx, idx2, n_out, c, (T*)out);
  else
    hipLaunchKernelGGL((gather_rows_kernel<T, false>), dim3((unsigned)grid), dim3(256), 0, s, (const T*)src, idx, idx2, n_out, c, (T*)out);
  PTC_CHECK_LAUNCH("gather_rows_kernel");
  return PTC_OK;
}

extern "C" int ptc_gather_rows(const void* src, int64_t n_src, const int64_t* idx, const int64_t* idx2,
                               int64_t n_out, int c, int dtype, void* out, ptc_stream_t stream) {
  PTC_REQUIRE(n_src >= 0 && n_out >= 0 && c >= 1, PTC_EINVAL, "ptc_gather_rows: bad sizes");
  if (n_out == 0) return PTC_OK;
  PTC_REQUIRE(src && idx && out, PTC_EINVAL, "ptc_gather_rows: null buffer");
  PTC_DISPATCH_DTYPE(dtype, T, return launch_gather_rows<T>(src, idx, idx2, n_out, c, out, (hipStream_t)stream));
  return PTC_OK;
}

// ------------------------------------------------------------------------------------------------
// segment_csr forward / backward
// ------------------------------------------------------------------------------------------------
template <typename T, bool VEC, int REDUCE>
__global__ void __launch_bounds__(256)
segment_csr_fwd_kernel(const T* __restrict__ src, const int64_t* __restrict__ perm, const int64_t* __restrict__ indptr,
                       int64_t n_seg, int c, T* __restrict__ out, int32_t* __restrict__ arg_out) {
  constexpr int W = VEC ? Vec16<T>::N : 1;
  const int cpr = c / W;
  const int64_t total = n_seg * cpr;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t seg = t / cpr;
    const int ch = (int)(t - seg * cpr) * W;
    const int64_t r0 = indptr[seg], r1 = indptr[seg + 1];
    float acc[W];
    int32_t arg[W];
#pragma unroll
    for (int j = 0; j < W; ++j) {
      acc[j] = (REDUCE == PTC_REDUCE_MAX) ? -INFINITY : (REDUCE == PTC_REDUCE_MIN ? INFINITY : 0.f);
      arg[j] = -1;
    }
    for (int64_t r = r0; r < r1; ++r) {
      const int64_t p = perm ? perm[r] : r;
      float v[W];
      if (VEC) {
        Vec16<T> a = load16(src + p * c + ch);
#pragma unroll
        for (int j = 0; j < W; ++j) v[j] = ptc_to_float(a.v[j]);
      } else {
        v[0] = ptc_to_float(src[p * c + ch]);
      }
#pragma unroll
      for (int j = 0; j < W; ++j) {
        if (REDUCE == PTC_REDUCE_MAX) { if (v[j] > acc[j] || arg[j] < 0) { acc[j] = v[j]; arg[j] = (int32_t)p; } }
        else if (REDUCE == PTC_REDUCE_MIN) { if (v[j] < acc[j] || arg[j] < 0) { acc[j] = v[j]; arg[j] = (int32_t)p; } }
        else acc[j] += v[j];
      }
    }
    const float cnt = (float)(r1 - r0);
#pragma unroll
    for (int j = 0; j < W; ++j) {
      if (REDUCE == PTC_REDUCE_MEAN) acc[j] = (r1 > r0) ? acc[j] / cnt : 0.f;
      if ((REDUCE == PTC_REDUCE_MAX || REDUCE == PTC_REDUCE_MIN) && r1 <= r0) acc[j] = 0.f;  // empty segment -> 0
    }
    if (VEC) {
      Vec16<T> o;
#pragma unroll
      for (int j = 0; j < W; ++j) o.v[j] = ptc_from_float<T>(acc[j]);
      store16(out + seg * c + ch, o);
    } else {
      out[seg * c + ch] = ptc_from_float<T>(acc[0]);
    }
    if (arg_out && (REDUCE == PTC_REDUCE_MAX || REDUCE == PTC_REDUCE_MIN)) {
#pragma unroll
      for (int j = 0; j < W; ++j) arg_out[seg * c + ch + j] = arg[j];
    }
  }
}

template <typename T, int REDUCE>
static int launch_segment_fwd(const void* src, const int64_t* perm, const int64_t* indptr, int64_t n_seg, int c,
                              void* out, int32_t* arg_out, hipStream_t s) {
  const bool vec = (c % Vec16<T>::N) == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)out % 16 == 0);
  const int W = vec ? Vec16<T>::N : 1;
  int64_t grid = ptc_cdiv(n_seg * (c / W), 256);
  if (grid > 256 * 32) grid = 256 * 32;
  if (vec)
    hipLaunchKernelGGL((segment_csr_fwd_kernel<T, true, REDUCE>), dim3((unsigned)grid), dim3(256), 0, s, (const T*)src, perm, indptr, n_seg, c, (T*)out, arg_out);
  else
    hipLaunchKernelGGL((segment_csr_fwd_kernel<T, false, REDUCE>), dim3((unsigned)grid), dim3(256), 0, s, (const T*)src, perm, indptr, n_seg, c, (T*)out, arg_out);
  PTC_CHECK_LAUNCH("segment_csr_fwd_kernel");
  return PTC_OK;
}

extern "C" int ptc_segment_csr_fwd(const void* src, const int64_t* perm, const int64_t* indptr, int64_t n_seg,
                                   int c, int dtype, int reduce, void* out, int32_t* arg_out, ptc_stream_t stream) {
  PTC_REQUIRE(n_seg >= 0 && c >= 1, PTC_EINVAL, "ptc_segment_csr_fwd: bad sizes");
  PTC_REQUIRE(reduce >= 0 && reduce <= 3, PTC_EINVAL, "ptc_segment_csr_fwd: bad reduce %d", reduce);
  if (n_seg == 0) return PTC_OK;
  PTC_REQUIRE(src && indptr && out, PTC_EINVAL, "ptc_segment_csr_fwd: null buffer");
  hipStream_t s = (hipStream_t)stream;
  PTC_DISPATCH_DTYPE(dtype, T, {
    switch (reduce) {
      case PTC_REDUCE_SUM: return launch_segment_fwd<T, PTC_REDUCE_SUM>(src, perm, indptr, n_seg, c, out, arg_out, s);
      case PTC_REDUCE_MEAN: return launch_segment_fwd<T, PTC_REDUCE_MEAN>(src, perm, indptr, n_seg, c, out, arg_out, s);
      case PTC_REDUCE_MAX: return launch_segment_fwd<T, PTC_REDUCE_MAX>(src, perm, indptr, n_seg, c, out, arg_out, s);
      default: return launch_segment_fwd<T, PTC_REDUCE_MIN>(src, perm, indptr, n_seg, c, out, arg_out, s);
    }
  });
  return PTC_OK;
}

// backward: one thread per (segment, chunk) walks the segment's rows and writes EVERY member row
// (value or zero), so no pre-zeroing and no atomics are needed when perm covers all source rows.
template <typename T, bool VEC, int REDUCE>
__global__ void __launch_bounds__(256)
segment_csr_bwd_kernel(const T* __restrict__ gout, const int64_t* __restrict__ perm, const int64_t* __restrict__ indptr,
                       const int32_t* __restrict__ arg, int64_t n_seg, int c, T* __restrict__ gsrc) {
  constexpr int W = VEC ? Vec16<T>::N : 1;
  const int cpr = c / W;
  const int64_t total = n_seg * cpr;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t seg = t / cpr;
    const int ch = (int)(t - seg * cpr) * W;
    const int64_t r0 = indptr[seg], r1 = indptr[seg + 1];
    float g[W];
    int32_t a[W];
    if (VEC) {
      Vec16<T> gv = load16(gout + seg * c + ch);
#pragma unroll
      for (int j = 0; j < W; ++j) g[j] = ptc_to_float(gv.v[j]);
    } else {
      g[0] = ptc_to_float(gout[seg * c + ch]);
    }
    if (REDUCE == PTC_REDUCE_MEAN) {
      const float inv = (r1 > r0) ? 1.f / (float)(r1 - r0) : 0.f;
#pragma unroll
      for (int j = 0; j < W; ++j) g[j] *= inv;
    }
    if (REDUCE == PTC_REDUCE_MAX || REDUCE == PTC_REDUCE_MIN) {
#pragma unroll
      for (int j = 0; j < W; ++j) a[j] = arg[seg * c + ch + j];
    }
    for (int64_t r = r0; r < r1; ++r) {
      const int64_t p = perm ? perm[r] : r;
      if (VEC) {
        Vec16<T> o;
#pragma unroll
        for (int j = 0; j < W; ++j) {
          float v = g[j];
          if (REDUCE == PTC_REDUCE_MAX || REDUCE == PTC_REDUCE_MIN) v = (a[j] == (int32_t)p) ? g[j] : 0.f;
          o.v[j] = ptc_from_float<T>(v);
        }
        store16(gsrc + p * c + ch, o);
      } else {
        float v = g[0];
        if (REDUCE == PTC_REDUCE_MAX || REDUCE == PTC_REDUCE_MIN) v = (a[0] == (int32_t)p) ? g[0] : 0.f;
        gsrc[p * c + ch] = ptc_from_float<T>(v);
      }
    }
  }
}

template <typename T, int REDUCE>
static int launch_segment_bwd(const void* gout, const int64_t* perm, const int64_t* indptr, const int32_t* arg,
                              int64_t n_seg, int c, void* gsrc, hipStream_t s) {
  const bool vec = (c % Vec16<T>::N) == 0 && ((uintptr_t)gout % 16 == 0) && ((uintptr_t)gsrc % 16 == 0);
  const int W = vec ? Vec16<T>::N : 1;
  int64_t grid = ptc_cdiv(n_seg * (c / W), 256);
  if (grid > 256 * 32) grid = 256 * 32;
  if (vec)
    hipLaunchKernelGGL((segment_csr_bwd_kernel<T, true, REDUCE>), dim3((unsigned)grid), dim3(256), 0, s, (const T*)gout, perm, indptr, arg, n_seg, c, (T*)gsrc);
  else
    hipLaunchKernelGGL((segment_csr_bwd_kernel<T, false, REDUCE>), dim3((unsigned)grid), dim3(256), 0, s, (const T*)gout, perm, indptr, arg, n_seg, c, (T*)gsrc);
  PTC_CHECK_LAUNCH("segment_csr_bwd_kernel");
  return PTC_OK;
}

extern "C" int ptc_segment_csr_bwd(const void* grad_out, const int64_t* perm, const int64_t* indptr,
                                   const int32_t* arg, int64_t n_seg, int64_t n_src, int c, int dtype, int reduce,
                                   void* grad_src, ptc_stream_t stream) {
  PTC_REQUIRE(n_seg >= 0 && n_src >= 0 && c >= 1, PTC_EINVAL, "ptc_segment_csr_bwd: bad sizes");
  PTC_REQUIRE(reduce >= 0 && reduce <= 3, PTC_EINVAL, "ptc_segment_csr_bwd: bad reduce %d", reduce);
  if (n_seg == 0) return PTC_OK;
  PTC_REQUIRE(grad_out && indptr && grad_src, PTC_EINVAL, "ptc_segment_csr_bwd: null buffer");
  PTC_REQUIRE(!(reduce == PTC_REDUCE_MAX || reduce == PTC_REDUCE_MIN) || arg, PTC_EINVAL, "ptc_segment_csr_bwd: max/min needs arg");
  hipStream_t s = (hipStream_t)stream;
  PTC_DISPATCH_DTYPE(dtype, T, {
    switch (reduce) {
      case PTC_REDUCE_SUM: return launch_segment_bwd<T, PTC_REDUCE_SUM>(grad_out, perm, indptr, arg, n_seg, c, grad_src, s);
      case PTC_REDUCE_MEAN: return launch_segment_bwd<T, PTC_REDUCE_MEAN>(grad_out, perm, indptr, arg, n_seg, c, grad_src, s);
      case PTC_REDUCE_MAX: return launch_segment_bwd<T, PTC_REDUCE_MAX>(grad_out, perm, indptr, arg, n_seg, c, grad_src, s);
      default: return launch_segment_bwd<T, PTC_REDUCE_MIN>(grad_out, perm, indptr, arg, n_seg, c, grad_src, s);
    }
  });
  return PTC_OK;
}

// ------------------------------------------------------------------------------------------------
// Backward-pass weight layouts, all weights of the model in ONE launch.
// The input-gradient GEMMs read W as [c_in][taps][c_out] (taps mirrored for a submanifold convolution run over its own
// table, functional._SparseConv; plain transpose for nn.Linear; the same matrix repeated for a kv = 2 gather-fused Linear).
// Produced per layer these were 118 transposes + 23 index_selects per training step (profiles/r02_r_trace_copies.txt); here every
// stale layout is refreshed together, right after the multi-tensor weight cast, from a descriptor table the caller keeps on the
// device:  desc[e] = { src, dst, c_out, taps_src, c_in, taps_dst | mode << 32 },  prefix[e] = first output element of entry e.
//   dst[ci][j][co] = src[co][m(j)][ci],   mode 0: m(j) = taps_src - 1 - j (mirror; plain transpose when taps = 1)
//                                          mode 1: m(j) = 0 (replicate)     mode 2: m(j) = j
// 16-bit elements.  One thread per output element: coalesced stores, strided loads of L2-resident weights.
__global__ void __launch_bounds__(256)
weight_layouts_kernel(const int64_t* __restrict__ desc, const int64_t* __restrict__ prefix, int n, int64_t total) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  int lo = 0, hi = n - 1;                       // largest e with prefix[e] <= t
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (prefix[mid] <= t) lo = mid; else hi = mid - 1;
  }
  const int64_t* d = desc + (int64_t)lo * 6;
  const uint16_t* src = reinterpret_cast<const uint16_t*>(d[0]);
  uint16_t* dst = reinterpret_cast<uint16_t*>(d[1]);
  const int64_t co_n = d[2], ks = d[3], ci_n = d[4];
  const int64_t kd = d[5] & 0xffffffffll;
  const int mode = (int)(d[5] >> 32);
  const int64_t o = t - prefix[lo];
  const int64_t co = o % co_n, j = (o / co_n) % kd, ci = o / (co_n * kd);
  const int64_t m = mode == 0 ? ks - 1 - j : (mode == 1 ? 0 : j);
  dst[o] = src[(co * ks + m) * ci_n + ci];
}

extern "C" int ptc_weight_layouts(const int64_t* desc, const int64_t* prefix, int n, int64_t total, ptc_stream_t stream) {
  PTC_REQUIRE(n >= 0 && total >= 0, PTC_EINVAL, "ptc_weight_layouts: bad sizes");
  if (n == 0 || total == 0) return PTC_OK;
  PTC_REQUIRE(desc && prefix, PTC_EINVAL, "ptc_weight_layouts: null buffer");
  hipLaunchKernelGGL(weight_layouts_kernel, dim3((unsigned)ptc_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, desc, prefix, n, total);
  PTC_CHECK_LAUNCH("weight_layouts_kernel");
  return PTC_OK;
}

// ------------------------------------------------------------------------------------------------
// fp32 master weights -> 16-bit shadows, all stale ones of the model in ONE launch.
// torch._foreach_copy_ does this in ~29 multi-tensor launches of ~14 us (486 tensors, 46 M elements: 0.40 ms per step,
// profiles/r02_ag_bench_kernel_stats.csv) -- ten times the 35 us the 276 MB take at the HBM rate.  One thread converts one UNIT of
// 8 consecutive elements of one tensor: desc[e] = { src (fp32), dst (16 bit), n_elements }, prefix[e] = first unit of entry e
// (every entry is rounded up to whole units), total_units = prefix[n].
template <typename TO>
__global__ void __launch_bounds__(256)
cast_many_kernel(const int64_t* __restrict__ desc, const int64_t* __restrict__ prefix, int n, int64_t total_units) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < total_units; u += stride) {
    int lo = 0, hi = n - 1;                       // largest e with prefix[e] <= u
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (prefix[mid] <= u) lo = mid; else hi = mid - 1;
    }
    const int64_t* d = desc + (int64_t)lo * 3;
    const float* src = reinterpret_cast<const float*>(d[0]);
    TO* dst = reinterpret_cast<TO*>(d[1]);
    const int64_t cnt = d[2], base = (u - prefix[lo]) * 8;
    if (base + 8 <= cnt && (((uintptr_t)(src + base)) & 15) == 0 && (((uintptr_t)(dst + base)) & 15) == 0) {
      const float4 a = reinterpret_cast<const float4*>(src + base)[0], b = reinterpret_cast<const float4*>(src + base)[1];
      TO v[8] = {ptc_from_float<TO>(a.x), ptc_from_float<TO>(a.y), ptc_from_float<TO>(a.z), ptc_from_float<TO>(a.w),
                 ptc_from_float<TO>(b.x), ptc_from_float<TO>(b.y), ptc_from_float<TO>(b.z), ptc_from_float<TO>(b.w)};
      *reinterpret_cast<uint4*>(dst + base) = *reinterpret_cast<const uint4*>(v);
    } else {
      for (int e = 0; e < 8; ++e)
        if (base + e < cnt) dst[base + e] = ptc_from_float<TO>(src[base + e]);
    }
  }
}

extern "C" int ptc_cast_many(const int64_t* desc, const int64_t* prefix, int n, int64_t total_units, int dst_dtype, ptc_stream_t stream) {
  PTC_REQUIRE(n >= 0 && total_units >= 0, PTC_EINVAL, "ptc_cast_many: bad sizes");
  PTC_REQUIRE(dst_dtype == PTC_BF16 || dst_dtype == PTC_F16, PTC_EUNSUPPORTED, "ptc_cast_many: the shadows are bf16 or f16");
  if (n == 0 || total_units == 0) return PTC_OK;
  PTC_REQUIRE(desc && prefix, PTC_EINVAL, "ptc_cast_many: null buffer");
  int64_t grid = ptc_cdiv(total_units, 256);
  if (grid > 65536) grid = 65536;
  if (dst_dtype == PTC_BF16)
    hipLaunchKernelGGL(cast_many_kernel<bf16_t>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, desc, prefix, n, total_units);
  else
    hipLaunchKernelGGL(cast_many_kernel<f16_t>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, desc, prefix, n, total_units);
  PTC_CHECK_LAUNCH("cast_many_kernel");
  return PTC_OK;
}

