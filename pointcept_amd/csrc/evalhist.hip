// evalhist.hip -- the evaluation tail of the segmentation step, one pass.
//
// Replaces, for `SemSegEvaluator` (pointcept/engines/hooks/evaluator.py:139-152):
//     pred = seg_logits.max(1)[1];  pred = pred[inverse];  intersection_and_union_gpu(pred, segment, K, ignore)
// where intersection_and_union_gpu (pointcept/utils/misc.py:57-69) masks the ignored points and takes three
// torch.histc over [0, K-1] -- arg-max kernel, gather, compare / index kernels and three histogram launches on the
// reference; here every evaluated point reads its (inverse-mapped) logit row once and bumps three integer counters:
//     hist[0][c] = #{pred == target == c}   hist[1][c] = #{pred == c, target not ignored}   hist[2][c] = #{target == c}
// (area_union = hist[1] + hist[2] - hist[0], misc.py:68).  Integer atomics on per-workgroup LDS copies, then on the
// output: order-independent, exact.  arg-max ties resolve to the LOWEST class index.
#include "ptc_common.h"

#define EH_MAX_K 1024

template <typename T>
__global__ void __launch_bounds__(256)
seg_hist_kernel(const T* __restrict__ logits, int64_t row_stride, int c, const int64_t* __restrict__ pred_in,
                const int64_t* __restrict__ inverse, const int64_t* __restrict__ target, int64_t m, int64_t n_rows, int k,
                int64_t ignore_index, unsigned long long* __restrict__ hist) {
  extern __shared__ unsigned int lh[];          // [3][k]
  for (int i = threadIdx.x; i < 3 * k; i += 256) lh[i] = 0u;
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
    const int64_t t = target[i];
    if (t == ignore_index) continue;                      // misc.py:62: output[target == ignore] = ignore, then outside every bin
    int64_t p;
    if (pred_in) {
      p = pred_in[i];
    } else {
      int64_t row = inverse ? inverse[i] : i;
      if (row < 0 || row >= n_rows) continue;
      const T* r = logits + row * row_stride;
      float best = ptc_to_float(r[0]);
      int arg = 0;
      for (int j = 1; j < c; ++j) {
        const float v = ptc_to_float(r[j]);
        if (v > best) { best = v; arg = j; }
      }
      p = arg;
    }
    if (p >= 0 && p < k) {
      atomicAdd(&lh[k + (int)p], 1u);
      if (p == t) atomicAdd(&lh[(int)p], 1u);
    }
    if (t >= 0 && t < k) atomicAdd(&lh[2 * k + (int)t], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * k; i += 256)
    if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
}

extern "C" int ptc_seg_eval_hist(const void* logits, int dtype, int64_t row_stride, int c, const int64_t* pred, const int64_t* inverse,
                                 const int64_t* target, int64_t m, int64_t n_rows, int k, int64_t ignore_index, int64_t* hist3k,
                                 ptc_stream_t stream) {
  PTC_REQUIRE(m >= 0 && k >= 1 && k <= EH_MAX_K, PTC_EUNSUPPORTED, "ptc_seg_eval_hist: k=%d not in [1,%d]", k, EH_MAX_K);
  PTC_REQUIRE(hist3k != nullptr, PTC_EINVAL, "ptc_seg_eval_hist: null output");
  PTC_REQUIRE((logits != nullptr) != (pred != nullptr), PTC_EINVAL, "ptc_seg_eval_hist: give logits OR predictions");
  PTC_REQUIRE(pred == nullptr || inverse == nullptr, PTC_EINVAL, "ptc_seg_eval_hist: predictions are already per evaluated point");
  PTC_REQUIRE(logits == nullptr || (c >= 1 && row_stride >= c && n_rows >= 0), PTC_EINVAL, "ptc_seg_eval_hist: bad logits shape");
  hipStream_t s = (hipStream_t)stream;
  PTC_HIP(hipMemsetAsync(hist3k, 0, (size_t)3 * k * sizeof(int64_t), s));
  if (m == 0) return PTC_OK;
  PTC_REQUIRE(target != nullptr, PTC_EINVAL, "ptc_seg_eval_hist: null target");
  int64_t grid = ptc_cdiv(m, 256 * 8);
  if (grid > 1024) grid = 1024;
  if (grid < 1) grid = 1;
  const size_t lds = (size_t)3 * k * sizeof(unsigned int);
  if (logits) {
    PTC_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(seg_hist_kernel<T>, dim3((unsigned)grid), dim3(256), lds, s, (const T*)logits, row_stride, c,
                                                    (const int64_t*)nullptr, inverse, target, m, n_rows, k, ignore_index,
                                                    (unsigned long long*)hist3k));
  } else {
    hipLaunchKernelGGL(seg_hist_kernel<float>, dim3((unsigned)grid), dim3(256), lds, s, (const float*)nullptr, (int64_t)0, 0, pred,
                       (const int64_t*)nullptr, target, m, (int64_t)0, k, ignore_index, (unsigned long long*)hist3k);
  }
  PTC_CHECK_LAUNCH("seg_hist_kernel");
  return PTC_OK;
}
