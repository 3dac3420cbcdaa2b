// loss_rows.h -- how the per-point loss kernels (loss.hip: cross entropy; lovasz.hip: keys / dlogits) move a point's row of class
// scores (round 5).  One thread owns one point.  The first form read and wrote its row element by element: 20 two-byte loads per pass
// over the row, three passes, every load instruction touching 64 different sectors (one per lane), and 20 four-byte stores per
// thread at an 80-byte stride -- ce_bwd 129 us and lovasz_dlogits 324 us for 100 - 165 MB of traffic (0.5 - 1.3 TB/s,
// profiles/r05_p_bench_kernel_stats.csv).  For C <= LR_CP:
//   * the row is loaded ONCE into registers with the widest loads its alignment allows (16 or 8 bytes: a [:, :C] view of the head's
//     32-column GEMM output has 64-byte rows; a dense [N, 20] bf16 tensor 40-byte rows) and every pass runs over registers;
//   * rows that leave a kernel (dlogits) are written by their owners into an LDS image of the workgroup's CONTIGUOUS output chunk
//     (256 rows x C values) and copied out with 16-byte stores, lane after lane.
// The arithmetic (order of the max / sum / exp calls) is the element-wise kernels', so the results are the same bits; wider rows
// (C > LR_CP: CrossEntropyLoss serves up to 1024 classes) keep the element-wise kernels.
#pragma once

#define LR_CP 32
#define LR_THREADS 256

// VB = bytes per load instruction: 16 | 8 | 0 (element-wise).  Elements past c inside the last vector are loaded (the host checked that
// they lie inside the row's stride) and left in v: every consumer loops `j < c`.
template <typename T, int VB>
__device__ __forceinline__ void lr_load_row(const T* __restrict__ row, int c, float (&v)[LR_CP]) {
  if constexpr (VB == 0) {
#pragma unroll
    for (int j = 0; j < LR_CP; ++j) v[j] = j < c ? ptc_to_float(row[j]) : 0.f;
  } else {
    constexpr int E = VB / (int)sizeof(T);
#pragma unroll
    for (int q = 0; q < LR_CP / E; ++q) {
      if (q * E < c) {
        __attribute__((aligned(16))) T tmp[E];
        if constexpr (VB == 16) *reinterpret_cast<uint4*>(tmp) = reinterpret_cast<const uint4*>(row)[q];
        else *reinterpret_cast<uint2*>(tmp) = reinterpret_cast<const uint2*>(row)[q];
#pragma unroll
        for (int i = 0; i < E; ++i) v[q * E + i] = ptc_to_float(tmp[i]);
      } else {
#pragma unroll
        for (int i = 0; i < E; ++i) v[q * E + i] = 0.f;
      }
    }
  }
}

// v[t] for a run-time t (a register array cannot be indexed dynamically without going through scratch)
__device__ __forceinline__ float lr_pick(const float (&v)[LR_CP], int t) {
  float r = 0.f;
#pragma unroll
  for (int j = 0; j < LR_CP; ++j) r = j == t ? v[j] : r;
  return r;
}

// the workgroup's output chunk [rows x c] of OT, assembled in LDS (row-major, dense: the layout of the output itself), to global memory.
// Every thread of the workgroup calls this after the barrier that follows the owners' writes.  `a16`: dst is 16-byte aligned.
template <typename OT>
__device__ __forceinline__ void lr_copy_out(const unsigned char* tile, OT* __restrict__ dst, int elems, bool a16) {
  int done = 0;
  if (a16) {
    const int nv = (int)((size_t)elems * sizeof(OT) >> 4);
    for (int q = threadIdx.x; q < nv; q += LR_THREADS) reinterpret_cast<uint4*>(dst)[q] = reinterpret_cast<const uint4*>(tile)[q];
    done = nv * (16 / (int)sizeof(OT));
  }
  for (int e = done + (int)threadIdx.x; e < elems; e += LR_THREADS) dst[e] = reinterpret_cast<const OT*>(tile)[e];
}

// widest load the rows of a [n, c] matrix with this base address and row stride (in elements) allow
static inline int lr_vec_bytes(const void* base, int64_t row_stride, int c, size_t esize) {
  const int cand[2] = {16, 8};
  for (int k = 0; k < 2; ++k) {
    const int vb = cand[k], e = vb / (int)esize;
    if (((uintptr_t)base % (uintptr_t)vb) == 0 && ((size_t)row_stride * esize) % (size_t)vb == 0 && (int64_t)((c + e - 1) / e) * e <= row_stride)
      return vb;
  }
  return 0;
}

#define LR_DISPATCH_VB(vb, VB, ...)                          \
  switch (vb) {                                              \
    case 16: { constexpr int VB = 16; __VA_ARGS__; } break;  \
    case 8: { constexpr int VB = 8; __VA_ARGS__; } break;    \
    default: { constexpr int VB = 0; __VA_ARGS__; } break;   \
  }
