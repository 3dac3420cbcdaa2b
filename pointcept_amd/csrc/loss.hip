// loss.hip -- the two ends of the step that sat on slow generic reductions (profiles/r01_k):
//   * ptc_coord_max : max over points of grid_coord per axis (Point.serialization derives the curve depth
//     from it, pointcept/models/utils/structure.py:74; Point.sparsify the sparse shape, :136-138).
//     ATen's int64 max-reduce over [819200,3] took 505 us; this is one 20 MB streaming pass.
//   * ptc_cross_entropy_{fwd,bwd} : CrossEntropyLoss(ignore_index) over seg logits [N, C<=1024]
//     (pointcept/models/losses/misc.py CrossEntropyLoss as configured at
//     configs/scannet/semseg-pt-v3m1-0-base.py:49-52, called from pointcept/models/default.py:78-84).
//     One thread per point: log-sum-exp in fp32 straight from the (possibly strided, bf16) head output;
//     per-workgroup (loss, count) partials are summed by the host wrapper in a fixed order
//     (deterministic, no float atomics); backward writes softmax - onehot scaled by a DEVICE scalar
//     (grad / count), so the mean never needs a host sync.  ATen: 0.95 ms + 0.61 ms per step.
#include "ptc_common.h"
#include "loss_rows.h"

template <typename CoordT>
__global__ void __launch_bounds__(256)
coord_max_kernel(const CoordT* __restrict__ gc, int64_t n, unsigned long long* __restrict__ out3) {
  // the maximum is taken over the UNSIGNED (sign-extended) values: a negative coordinate is larger than every legal
  // one, so it surfaces as a negative int64 in out3 and the host-side range check (ops.check_coord_range) rejects it
  // -- the voxel hash packs coordinates into 18-bit fields and would otherwise alias it silently
  unsigned long long m0 = 0, m1 = 0, m2 = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const unsigned long long x = (unsigned long long)(long long)gc[3 * i], y = (unsigned long long)(long long)gc[3 * i + 1],
                             z = (unsigned long long)(long long)gc[3 * i + 2];
    m0 = x > m0 ? x : m0;
    m1 = y > m1 ? y : m1;
    m2 = z > m2 ? z : m2;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const unsigned long long a = __shfl_xor(m0, o, 64), b = __shfl_xor(m1, o, 64), c = __shfl_xor(m2, o, 64);
    m0 = a > m0 ? a : m0;
    m1 = b > m1 ? b : m1;
    m2 = c > m2 ? c : m2;
  }
  // one set of atomics per WORKGROUP (the first version issued them per wave: 24576 64-bit atomics on three addresses serialised in
  // the L2 -- 114 us for a 20 MB read, profiles/r03_y_bench_kernel_stats.csv); integer max: order-independent, exact
  __shared__ unsigned long long wm[4][3];
  const int wave = threadIdx.x >> 6;
  if (ptc_lane() == 0) { wm[wave][0] = m0; wm[wave][1] = m1; wm[wave][2] = m2; }
  __syncthreads();
  if (threadIdx.x < 3) {
    unsigned long long m = wm[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < 4; ++w) m = wm[w][threadIdx.x] > m ? wm[w][threadIdx.x] : m;
    atomicMax(out3 + threadIdx.x, m);
  }
}

extern "C" int ptc_coord_max(const void* grid_coord, int coord_is_i64, int64_t n, int64_t* out3, ptc_stream_t stream) {
  PTC_REQUIRE(n >= 0, PTC_EINVAL, "ptc_coord_max: n < 0");
  PTC_REQUIRE(out3 != nullptr, PTC_EINVAL, "ptc_coord_max: null output");
  hipStream_t s = (hipStream_t)stream;
  PTC_HIP(hipMemsetAsync(out3, 0, 3 * sizeof(int64_t), s));
  if (n == 0) return PTC_OK;
  PTC_REQUIRE(grid_coord != nullptr, PTC_EINVAL, "ptc_coord_max: null buffer");
  int64_t grid = ptc_cdiv(n, 256 * 4);
  if (grid > 512) grid = 512;
  if (coord_is_i64)
    hipLaunchKernelGGL(coord_max_kernel<int64_t>, dim3((unsigned)grid), dim3(256), 0, s, (const int64_t*)grid_coord, n,
                       (unsigned long long*)out3);
  else
    hipLaunchKernelGGL(coord_max_kernel<int32_t>, dim3((unsigned)grid), dim3(256), 0, s, (const int32_t*)grid_coord, n,
                       (unsigned long long*)out3);
  PTC_CHECK_LAUNCH("coord_max_kernel");
  return PTC_OK;
}

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
ce_fwd_kernel(const T* __restrict__ logits, int64_t row_stride, const int64_t* __restrict__ target, int64_t n, int c,
              int64_t ignore_index, float* __restrict__ lse, float* __restrict__ partial) {
  __shared__ float red[2][4];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float loss = 0.f, cnt = 0.f;
  if (i < n) {
    const T* row = logits + i * row_stride;
    float m = -INFINITY;
    for (int j = 0; j < c; ++j) m = fmaxf(m, ptc_to_float(row[j]));
    float ssum = 0.f;
    for (int j = 0; j < c; ++j) ssum += __expf(ptc_to_float(row[j]) - m);
    const float l = m + __logf(ssum);
    lse[i] = l;
    const int64_t t = target[i];
    if (t != ignore_index && t >= 0 && t < c) {
      loss = l - ptc_to_float(row[t]);
      cnt = 1.f;
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    loss += __shfl_xor(loss, o, 64);
    cnt += __shfl_xor(cnt, o, 64);
  }
  const int wave = threadIdx.x >> 6;
  if (ptc_lane() == 0) { red[0][wave] = loss; red[1][wave] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * (int64_t)blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    partial[2 * (int64_t)blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
ce_bwd_kernel(const T* __restrict__ logits, int64_t row_stride, const int64_t* __restrict__ target, const float* __restrict__ lse,
              const float* __restrict__ scale, int64_t n, int c, int64_t ignore_index, T* __restrict__ dlogits, int64_t drow_stride) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const T* row = logits + i * row_stride;
  T* drow = dlogits + i * drow_stride;
  const int64_t t = target[i];
  const bool valid = t != ignore_index && t >= 0 && t < c;
  const float sc = valid ? scale[0] : 0.f;
  const float l = lse[i];
  for (int j = 0; j < c; ++j) {
    const float p = __expf(ptc_to_float(row[j]) - l);
    drow[j] = ptc_from_float<T>(sc * (p - (j == t ? 1.f : 0.f)));
  }
}

// C <= LR_CP: the row in registers, one vector load pass; the gradient rows leave through LDS (loss_rows.h).  Same arithmetic, same bits.
template <typename T, int VB>
__global__ void __launch_bounds__(LR_THREADS)
ce_fwd_rows_kernel(const T* __restrict__ logits, int64_t row_stride, const int64_t* __restrict__ target, int64_t n, int c,
                   int64_t ignore_index, float* __restrict__ lse, float* __restrict__ partial) {
  __shared__ float red[2][4];
  const int64_t i = (int64_t)blockIdx.x * LR_THREADS + threadIdx.x;
  float loss = 0.f, cnt = 0.f;
  if (i < n) {
    float v[LR_CP];
    lr_load_row<T, VB>(logits + i * row_stride, c, v);
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < LR_CP; ++j) if (j < c) m = fmaxf(m, v[j]);
    float ssum = 0.f;
#pragma unroll
    for (int j = 0; j < LR_CP; ++j) if (j < c) ssum += __expf(v[j] - m);
    const float l = m + __logf(ssum);
    lse[i] = l;
    const int64_t t = target[i];
    if (t != ignore_index && t >= 0 && t < c) {
      loss = l - lr_pick(v, (int)t);
      cnt = 1.f;
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    loss += __shfl_xor(loss, o, 64);
    cnt += __shfl_xor(cnt, o, 64);
  }
  const int wave = threadIdx.x >> 6;
  if (ptc_lane() == 0) { red[0][wave] = loss; red[1][wave] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * (int64_t)blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    partial[2 * (int64_t)blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

template <typename T, int VB>
__global__ void __launch_bounds__(LR_THREADS)
ce_bwd_rows_kernel(const T* __restrict__ logits, int64_t row_stride, const int64_t* __restrict__ target, const float* __restrict__ lse,
                   const float* __restrict__ scale, int64_t n, int c, int64_t ignore_index, T* __restrict__ dlogits, int a16) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // [256][c] T: this workgroup's chunk of dlogits (dense rows)
  const int64_t row0 = (int64_t)blockIdx.x * LR_THREADS, i = row0 + threadIdx.x;
  if (i < n) {
    float v[LR_CP];
    lr_load_row<T, VB>(logits + i * row_stride, c, v);
    const int64_t t = target[i];
    const bool valid = t != ignore_index && t >= 0 && t < c;
    const float sc = valid ? scale[0] : 0.f;
    const float l = lse[i];
    T* drow = reinterpret_cast<T*>(smem) + (int)threadIdx.x * c;
#pragma unroll
    for (int j = 0; j < LR_CP; ++j) {
      if (j < c) {
        const float p = __expf(v[j] - l);
        drow[j] = ptc_from_float<T>(sc * (p - (j == t ? 1.f : 0.f)));
      }
    }
  }
  __syncthreads();
  const int64_t rows = (n - row0) < LR_THREADS ? (n - row0) : LR_THREADS;
  lr_copy_out<T>(smem, dlogits + row0 * c, (int)rows * c, a16 != 0);
}

extern "C" int64_t ptc_cross_entropy_partials(int64_t n) { return n > 0 ? ptc_cdiv(n, 256) : 1; }

extern "C" int ptc_cross_entropy_fwd(const void* logits, int64_t row_stride, const int64_t* target, int64_t n, int c, int dtype,
                                     int64_t ignore_index, float* lse, float* partial, ptc_stream_t stream) {
  PTC_REQUIRE(n >= 0 && c >= 1 && c <= 1024 && row_stride >= c, PTC_EINVAL, "ptc_cross_entropy_fwd: bad sizes");
  PTC_REQUIRE(partial != nullptr, PTC_EINVAL, "ptc_cross_entropy_fwd: null buffer");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    PTC_HIP(hipMemsetAsync(partial, 0, 2 * sizeof(float), s));
    return PTC_OK;
  }
  PTC_REQUIRE(logits && target && lse, PTC_EINVAL, "ptc_cross_entropy_fwd: null buffer");
  const unsigned grid = (unsigned)ptc_cdiv(n, 256);
  if (c <= LR_CP) {
    const int vb = lr_vec_bytes(logits, row_stride, c, ptc_dtype_size(dtype));
    PTC_DISPATCH_DTYPE(dtype, T, LR_DISPATCH_VB(vb, VB, hipLaunchKernelGGL((ce_fwd_rows_kernel<T, VB>), dim3(grid), dim3(LR_THREADS), 0, s,
                                                                         (const T*)logits, row_stride, target, n, c, ignore_index, lse, partial)));
    PTC_CHECK_LAUNCH("ce_fwd_rows_kernel");
    return PTC_OK;
  }
  PTC_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(ce_fwd_kernel<T>, dim3(grid), dim3(256), 0, s, (const T*)logits, row_stride, target,
                                                   n, c, ignore_index, lse, partial));
  PTC_CHECK_LAUNCH("ce_fwd_kernel");
  return PTC_OK;
}

extern "C" int ptc_cross_entropy_bwd(const void* logits, int64_t row_stride, const int64_t* target, const float* lse,
                                     const float* scale, int64_t n, int c, int dtype, int64_t ignore_index, void* dlogits,
                                     int64_t drow_stride, ptc_stream_t stream) {
  PTC_REQUIRE(n >= 0 && c >= 1 && c <= 1024 && row_stride >= c && drow_stride >= c, PTC_EINVAL, "ptc_cross_entropy_bwd: bad sizes");
  if (n == 0) return PTC_OK;
  PTC_REQUIRE(logits && target && lse && scale && dlogits, PTC_EINVAL, "ptc_cross_entropy_bwd: null buffer");
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)ptc_cdiv(n, 256);
  if (c <= LR_CP && drow_stride == c) {       // dense gradient rows: the workgroup's chunk is contiguous
    const int vb = lr_vec_bytes(logits, row_stride, c, ptc_dtype_size(dtype));
    const int a16 = ((uintptr_t)dlogits & 15) == 0;
    const size_t lds = (size_t)LR_THREADS * c * ptc_dtype_size(dtype);
    PTC_DISPATCH_DTYPE(dtype, T, LR_DISPATCH_VB(vb, VB, hipLaunchKernelGGL((ce_bwd_rows_kernel<T, VB>), dim3(grid), dim3(LR_THREADS), lds, s,
                                                                         (const T*)logits, row_stride, target, lse, scale, n, c, ignore_index,
                                                                         (T*)dlogits, a16)));
    PTC_CHECK_LAUNCH("ce_bwd_rows_kernel");
    return PTC_OK;
  }
  PTC_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(ce_bwd_kernel<T>, dim3(grid), dim3(256), 0, s, (const T*)logits, row_stride, target,
                                                   lse, scale, n, c, ignore_index, (T*)dlogits, drow_stride));
  PTC_CHECK_LAUNCH("ce_bwd_kernel");
  return PTC_OK;
}
