// pad_maps.h -- closed forms of SerializedAttention.get_padding_and_inverse
// (pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py:114-170), shared by the
// device kernel (maps.hip) and the host probe library (CPU unit checks of product code).
//
// Per scene i with n_i points and patch size K:
//   padded_i = n_i                 if n_i <= K      (ptv3m1:135-136: short scenes stay unpadded)
//            = ceil(n_i/K)*K       otherwise        (ptv3m1:126-133)
//   r = n_i mod K.  If padded_i != n_i the tail slots [padded_i-K+r, padded_i) of the last patch
//   repeat the slots K earlier (ptv3m1:144-154), i.e. sorted ranks [n_i-K, padded_i-K).
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define PTC_HD2 __host__ __device__ __forceinline__
#else
#define PTC_HD2 static inline
#endif

PTC_HD2 int64_t ptc_padded_len(int64_t n_i, int64_t K) {
  return (n_i > K) ? ((n_i + K - 1) / K) * K : n_i;
}
PTC_HD2 int64_t ptc_num_seq(int64_t n_i, int64_t K) {
  // torch.arange(off_pad[i], off_pad[i+1], K) (ptv3m1:156-164): ceil(padded/K) starts, 0 if empty
  int64_t p = ptc_padded_len(n_i, K);
  return (p + K - 1) / K;
}
// padded local slot -> local sorted rank
PTC_HD2 int64_t ptc_pad_local(int64_t local, int64_t n_i, int64_t K) {
  const int64_t padded = ptc_padded_len(n_i, K);
  if (padded != n_i) {
    const int64_t r = n_i % K;
    if (local >= padded - K + r) local -= K;
  }
  return local;
}
// local sorted rank -> second padded local slot holding it, or -1
PTC_HD2 int64_t ptc_dup_local(int64_t rank, int64_t n_i, int64_t K) {
  const int64_t padded = ptc_padded_len(n_i, K);
  if (padded != n_i && rank >= n_i - K && rank < padded - K) return rank + K;
  return -1;
}
// index of the scene containing position p, given cumulative ends e[0..B) (e[-1] = 0):
// smallest i with p < e[i]
PTC_HD2 int ptc_find_scene(const int64_t* ends, int B, int64_t p) {
  int lo = 0, hi = B - 1;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (p < ends[mid]) hi = mid; else lo = mid + 1;
  }
  return lo;
}
