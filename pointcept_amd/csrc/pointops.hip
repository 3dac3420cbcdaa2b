// pointops.hip -- nearest-neighbour query and farthest point sampling over offset-batched point clouds
// (libs/pointops, used by the evaluators / testers to carry predictions between point sets --
// pointcept/engines/hooks/evaluator.py:569, engines/test.py:1201 -- and by the SSL heads, sonata_v1m1_base.py:320).
//
// ptc_knn_query replaces knn_query_cuda (libs/pointops/src/knn_query/knn_query_cuda_kernel.cu:60-108, python wrapper
// libs/pointops/functions/query.py:7-26): for every query point the `nsample` nearest points of ITS scene, ascending
// distance, idx -1 / dist 1e5 (= sqrt(1e10)) where the scene has fewer points.  The reference walks each query's scene
// from global memory with a heap in local memory; here a workgroup of 256 queries streams the scene through LDS in
// 1024-point tiles (coalesced loads, broadcast reads) and keeps each query's k best in REGISTERS as a sorted list with
// compile-time indices (k in {1,4,8,16,32,64}; 128 spills to scratch).  Equal distances: the lower point index
// wins / comes first (the reference's heap leaves that order implementation-defined).
// ptc_farthest_point_sampling replaces farthest_point_sampling_cuda (src/sampling/sampling_cuda_kernel.cu:15-122,
// functions/sampling.py:7-24): one 1024-thread workgroup per scene, m_b sequential arg-max rounds over the running
// min-distance array; first pick = first point of the scene; equal distances: the lower index wins.
// Distances are ((dx*dx + dy*dy) + dz*dz) in fp32 with every operation rounded (no FMA contraction), so the CPU
// oracle reproduces them bit for bit.
#include "ptc_common.h"

#define KNN_THREADS 256
#define KNN_TILE 1024

__device__ __forceinline__ float po_dist2(float ax, float ay, float az, float bx, float by, float bz) {
#pragma clang fp contract(off)      // HIP's __fmul_rn / __fadd_rn are plain operators: without this they fuse into FMAs
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
  return (xx + yy) + zz;
}

// smallest i with p < ends[i] (ends ascending, p < ends[b-1])
__device__ __forceinline__ int po_scene_of(const int* __restrict__ ends, int b, int64_t p) {
  int lo = 0, hi = b - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (p < ends[mid]) hi = mid; else lo = mid + 1;
  }
  return lo;
}

template <int K>
__global__ void __launch_bounds__(KNN_THREADS)
knn_query_kernel(const float* __restrict__ xyz, const int* __restrict__ offset, const float* __restrict__ new_xyz,
                 const int* __restrict__ new_offset, int b, int64_t m, int nsample, int32_t* __restrict__ idx,
                 float* __restrict__ dist) {
  __shared__ float sx[KNN_TILE], sy[KNN_TILE], sz[KNN_TILE];
  __shared__ int s_lo, s_hi;
  const int64_t q = (int64_t)blockIdx.x * KNN_THREADS + threadIdx.x;
  const bool valid = q < m;
  int start = 0, end = 0;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (valid) {
    const int s = po_scene_of(new_offset, b, q);
    start = s ? offset[s - 1] : 0;
    end = offset[s];
    qx = new_xyz[3 * q]; qy = new_xyz[3 * q + 1]; qz = new_xyz[3 * q + 2];
  }
  // queries are ordered by scene: the block's candidate range runs from the first query's scene to the last one's
  if (threadIdx.x == 0) s_lo = start;
  const int64_t q_last = ((int64_t)blockIdx.x * KNN_THREADS + KNN_THREADS - 1 < m) ? (int64_t)blockIdx.x * KNN_THREADS + KNN_THREADS - 1 : m - 1;
  if (q == q_last) s_hi = end;
  __syncthreads();
  const int lo = s_lo, hi = s_hi;

  float bd[K];
  int bi[K];
#pragma unroll
  for (int j = 0; j < K; ++j) { bd[j] = 1e10f; bi[j] = -1; }

  for (int t0 = lo; t0 < hi; t0 += KNN_TILE) {
    __syncthreads();
    const int cnt = (hi - t0) < KNN_TILE ? (hi - t0) : KNN_TILE;
    for (int e = threadIdx.x; e < cnt; e += KNN_THREADS) {
      const float* p = xyz + 3 * (int64_t)(t0 + e);
      sx[e] = p[0]; sy[e] = p[1]; sz[e] = p[2];
    }
    __syncthreads();
    const int i0 = (start > t0 ? start : t0) - t0;
    const int i1 = ((end < t0 + cnt ? end : t0 + cnt)) - t0;
    for (int e = i0; e < i1; ++e) {
      const float d2 = po_dist2(qx, qy, qz, sx[e], sy[e], sz[e]);
      if (d2 < bd[K - 1]) {
        int pos = 0;                                  // insert after every entry <= d2: ascending (distance, index)
#pragma unroll
        for (int j = 0; j < K; ++j) pos += (bd[j] <= d2) ? 1 : 0;
#pragma unroll
        for (int j = K - 1; j > 0; --j)
          if (j > pos) { bd[j] = bd[j - 1]; bi[j] = bi[j - 1]; }
#pragma unroll
        for (int j = 0; j < K; ++j)
          if (j == pos) { bd[j] = d2; bi[j] = t0 + e; }
      }
    }
  }
  if (valid) {
#pragma unroll
    for (int j = 0; j < K; ++j)
      if (j < nsample) {
        idx[q * nsample + j] = bi[j];
        dist[q * nsample + j] = (float)sqrt((double)bd[j]);   // correctly rounded fp32 root (53 >= 2*24+2 bits): == host sqrt
      }
  }
}

extern "C" int ptc_knn_query(const float* xyz, const int32_t* offset, const float* new_xyz, const int32_t* new_offset, int b,
                             int64_t n, int64_t m, int nsample, int32_t* idx, float* dist, ptc_stream_t stream) {
  PTC_REQUIRE(n >= 0 && m >= 0 && b >= 1, PTC_EINVAL, "ptc_knn_query: bad sizes");
  PTC_REQUIRE(nsample >= 1 && nsample <= 128, PTC_EUNSUPPORTED, "ptc_knn_query: nsample=%d not in [1,128]", nsample);
  PTC_REQUIRE(n < (1ll << 31), PTC_EUNSUPPORTED, "ptc_knn_query: n >= 2^31");
  if (m == 0) return PTC_OK;
  PTC_REQUIRE(offset && new_xyz && new_offset && idx && dist && (n == 0 || xyz), PTC_EINVAL, "ptc_knn_query: null buffer");
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)ptc_cdiv(m, KNN_THREADS);
#define KNN_LAUNCH(KK) hipLaunchKernelGGL(knn_query_kernel<KK>, dim3(grid), dim3(KNN_THREADS), 0, s, xyz, offset, new_xyz, new_offset, b, m, nsample, idx, dist)
  if (nsample == 1) KNN_LAUNCH(1);
  else if (nsample <= 4) KNN_LAUNCH(4);
  else if (nsample <= 8) KNN_LAUNCH(8);
  else if (nsample <= 16) KNN_LAUNCH(16);
  else if (nsample <= 32) KNN_LAUNCH(32);
  else if (nsample <= 64) KNN_LAUNCH(64);
  else KNN_LAUNCH(128);
#undef KNN_LAUNCH
  PTC_CHECK_LAUNCH("knn_query_kernel");
  return PTC_OK;
}

// ------------------------------------------------------------------------------------------------
#define FPS_THREADS 1024

__global__ void __launch_bounds__(FPS_THREADS)
fps_kernel(const float* __restrict__ xyz, const int* __restrict__ offset, const int* __restrict__ new_offset,
           float* __restrict__ tmp, int32_t* __restrict__ idx) {
  __shared__ float w_best[FPS_THREADS / 64];
  __shared__ int w_arg[FPS_THREADS / 64];
  __shared__ int s_old;
  const int bid = blockIdx.x, tid = threadIdx.x;
  const int start_n = bid ? offset[bid - 1] : 0, end_n = offset[bid];
  const int start_m = bid ? new_offset[bid - 1] : 0, end_m = new_offset[bid];
  if (end_m <= start_m || end_n <= start_n) return;
  for (int k = start_n + tid; k < end_n; k += FPS_THREADS) tmp[k] = 1e10f;
  if (tid == 0) idx[start_m] = start_n;
  int old = start_n;
  __syncthreads();
  for (int j = start_m + 1; j < end_m; ++j) {
    const float x1 = xyz[3 * (int64_t)old], y1 = xyz[3 * (int64_t)old + 1], z1 = xyz[3 * (int64_t)old + 2];
    float best = -1.f;
    int arg = start_n;
    for (int k = start_n + tid; k < end_n; k += FPS_THREADS) {
      const float d = po_dist2(xyz[3 * (int64_t)k], xyz[3 * (int64_t)k + 1], xyz[3 * (int64_t)k + 2], x1, y1, z1);
      const float t = fminf(d, tmp[k]);
      tmp[k] = t;
      if (t > best) { best = t; arg = k; }            // ascending k: the lowest index among equals stays
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oa = __shfl_xor(arg, o, 64);
      if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    }
    if ((tid & 63) == 0) { w_best[tid >> 6] = best; w_arg[tid >> 6] = arg; }
    __syncthreads();
    if (tid < 64) {
      float bb = tid < FPS_THREADS / 64 ? w_best[tid] : -2.f;
      int aa = tid < FPS_THREADS / 64 ? w_arg[tid] : 0x7fffffff;
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) {
        const float ob = __shfl_xor(bb, o, 64);
        const int oa = __shfl_xor(aa, o, 64);
        if (ob > bb || (ob == bb && oa < aa)) { bb = ob; aa = oa; }
      }
      if (tid == 0) { s_old = aa; idx[j] = aa; }
    }
    __syncthreads();
    old = s_old;
  }
}

extern "C" int ptc_farthest_point_sampling(const float* xyz, const int32_t* offset, const int32_t* new_offset, int b, int64_t n,
                                           float* tmp, int32_t* idx, ptc_stream_t stream) {
  PTC_REQUIRE(b >= 1 && n >= 0, PTC_EINVAL, "ptc_farthest_point_sampling: bad sizes");
  PTC_REQUIRE(n < (1ll << 31), PTC_EUNSUPPORTED, "ptc_farthest_point_sampling: n >= 2^31");
  if (n == 0) return PTC_OK;
  PTC_REQUIRE(xyz && offset && new_offset && tmp && idx, PTC_EINVAL, "ptc_farthest_point_sampling: null buffer");
  hipLaunchKernelGGL(fps_kernel, dim3((unsigned)b), dim3(FPS_THREADS), 0, (hipStream_t)stream, xyz, offset, new_offset, tmp, idx);
  PTC_CHECK_LAUNCH("fps_kernel");
  return PTC_OK;
}


// ------------------------------------------------------------------------------------------------
// ball query / random ball query
// ------------------------------------------------------------------------------------------------
// ptc_ball_query replaces ball_query_cuda (libs/pointops/src/ball_query/ball_query_cuda_kernel.cu:59-123, wrapper
// functions/query.py:78-113): for every query, the points p of ITS scene with d2 <= 1e-5 or min_r^2 <= d2 < max_r^2
// (:98), sorted by ascending distance (:106), all of them when there are <= nsample (padding idx -1 / dist2 1e10,
// :107-115), else nsample of them at the uniformly spaced ranks int(i * float(count) / nsample) (:117-122).  The
// reference keeps 2048 candidates per THREAD in local memory and heap-sorts them; here ONE WAVE serves a query: the scene
// is scanned 64 points at a time (coalesced), in-range candidates are appended to a wave-private LDS list through a
// ballot / prefix count (the first BQ_CAP = 2048 in index order: the reference's array bound), the list is bitonic-
// sorted by (distance, index) in LDS, and the selected ranks are written out.  Two deliberate deviations, both in the
// documentation of pointops_api.ball_query: equal distances come out in ascending index order (heap sort leaves it
// unspecified), and in the sub-sampled branch dist2 holds the candidate's DISTANCE (the reference stores its INDEX there,
// :120 `dist2[i] = candi_idx[index]` -- a bug not to copy).
// ptc_random_ball_query replaces random_ball_query_cuda (src/random_ball_query/...kernel.cu:58-108): the first nsample
// in-range points in the order of a caller-supplied permutation `order` of every scene's points (wrapper
// functions/query.py:29-75 draws it with torch.randperm); wave-cooperative scan with early exit.
#define BQ_CAP 2048
#define BQ_WAVES 4

__global__ void __launch_bounds__(BQ_WAVES * 64)
ball_query_kernel(const float* __restrict__ xyz, const int* __restrict__ offset, const float* __restrict__ new_xyz,
                  const int* __restrict__ new_offset, int b, int64_t m, int nsample, float min_r2, float max_r2,
                  int32_t* __restrict__ idx, float* __restrict__ dist2) {
  __shared__ float cd[BQ_WAVES][BQ_CAP];
  __shared__ int ci[BQ_WAVES][BQ_CAP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t q = (int64_t)blockIdx.x * BQ_WAVES + wave;
  if (q >= m) return;                              // whole wave
  float* D = cd[wave];
  int* I = ci[wave];
  const int sc = po_scene_of(new_offset, b, q);
  const int start = sc ? offset[sc - 1] : 0, end = offset[sc];
  const float qx = new_xyz[3 * q], qy = new_xyz[3 * q + 1], qz = new_xyz[3 * q + 2];
  int cnt = 0;
  for (int base = start; base < end && cnt < BQ_CAP; base += 64) {
    const int p = base + lane;
    float d = 0.f;
    bool in = false;
    if (p < end) {
      d = po_dist2(qx, qy, qz, xyz[3 * (int64_t)p], xyz[3 * (int64_t)p + 1], xyz[3 * (int64_t)p + 2]);
      in = d <= 1e-5f || (d >= min_r2 && d < max_r2);
    }
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(in);
    const int pos = cnt + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
    if (in && pos < BQ_CAP) {
      D[pos] = d;
      I[pos] = p;
    }
    cnt += __builtin_popcountll(mask);
  }
  if (cnt > BQ_CAP) cnt = BQ_CAP;
  int P = 2;
  while (P < cnt) P <<= 1;
  for (int i = cnt + lane; i < P; i += 64) {       // padding sorts last
    D[i] = 3.0e38f;
    I[i] = 0x7fffffff;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int k2 = 2; k2 <= P; k2 <<= 1)
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = lane; i < P; i += 64) {
        const int x = i ^ j;
        if (x > i) {
          const float da = D[i], db = D[x];
          const int ia = I[i], ib = I[x];
          const bool gt = da > db || (da == db && ia > ib);
          if (gt == ((i & k2) == 0)) {
            D[i] = db; D[x] = da;
            I[i] = ib; I[x] = ia;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  int32_t* oi = idx + q * nsample;
  float* od = dist2 + q * nsample;
  if (cnt <= nsample) {
    for (int i = lane; i < nsample; i += 64) {
      oi[i] = i < cnt ? I[i] : -1;
      od[i] = i < cnt ? D[i] : 1e10f;
    }
  } else {
    const float sep = (float)cnt / (float)nsample;            // :117, fp32 as the reference
    for (int i = lane; i < nsample; i += 64) {
      const int r = (int)(sep * (float)i);
      oi[i] = I[r];
      od[i] = D[r];
    }
  }
}

__global__ void __launch_bounds__(256)
random_ball_query_kernel(const float* __restrict__ xyz, const int* __restrict__ offset, const float* __restrict__ new_xyz,
                         const int* __restrict__ new_offset, const int* __restrict__ order, int b, int64_t m, int nsample,
                         float min_r2, float max_r2, int32_t* __restrict__ idx, float* __restrict__ dist2) {
  const int lane = threadIdx.x & 63;
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= m) return;
  const int sc = po_scene_of(new_offset, b, q);
  const int start = sc ? offset[sc - 1] : 0, end = offset[sc];
  const float qx = new_xyz[3 * q], qy = new_xyz[3 * q + 1], qz = new_xyz[3 * q + 2];
  int32_t* oi = idx + q * nsample;
  float* od = dist2 + q * nsample;
  int cnt = 0;
  for (int base = start; base < end && cnt < nsample; base += 64) {
    const int i = base + lane;
    float d = 0.f;
    int p = -1;
    bool in = false;
    if (i < end) {
      p = order[i];
      d = po_dist2(qx, qy, qz, xyz[3 * (int64_t)p], xyz[3 * (int64_t)p + 1], xyz[3 * (int64_t)p + 2]);
      in = d <= 1e-5f || (d >= min_r2 && d < max_r2);
    }
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(in);
    const int pos = cnt + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
    if (in && pos < nsample) {
      oi[pos] = p;
      od[pos] = d;
    }
    cnt += __builtin_popcountll(mask);
  }
  if (cnt > nsample) cnt = nsample;
  for (int i = cnt + lane; i < nsample; i += 64) {
    oi[i] = -1;
    od[i] = 1e10f;
  }
}

extern "C" int ptc_ball_query(const float* xyz, const int32_t* offset, const float* new_xyz, const int32_t* new_offset, const int32_t* order,
                              int b, int64_t n, int64_t m, int nsample, float min_radius, float max_radius, int32_t* idx, float* dist2,
                              ptc_stream_t stream) {
  PTC_REQUIRE(n >= 0 && m >= 0 && b >= 1 && nsample >= 1, PTC_EINVAL, "ptc_ball_query: bad sizes");
  PTC_REQUIRE(min_radius < max_radius, PTC_EINVAL, "ptc_ball_query: min_radius must be < max_radius");     // query.py:45,93
  PTC_REQUIRE(n < (1ll << 31), PTC_EUNSUPPORTED, "ptc_ball_query: n >= 2^31");
  if (m == 0) return PTC_OK;
  PTC_REQUIRE(offset && new_xyz && new_offset && idx && dist2 && (n == 0 || xyz), PTC_EINVAL, "ptc_ball_query: null buffer");
  hipStream_t s = (hipStream_t)stream;
  const float mn = min_radius * min_radius, mx = max_radius * max_radius;
  if (order)
    hipLaunchKernelGGL(random_ball_query_kernel, dim3((unsigned)ptc_cdiv(m, 4)), dim3(256), 0, s, xyz, offset, new_xyz, new_offset, order, b, m,
                       nsample, mn, mx, idx, dist2);
  else
    hipLaunchKernelGGL(ball_query_kernel, dim3((unsigned)ptc_cdiv(m, BQ_WAVES)), dim3(BQ_WAVES * 64), 0, s, xyz, offset, new_xyz, new_offset, b, m,
                       nsample, mn, mx, idx, dist2);
  PTC_CHECK_LAUNCH("ball_query_kernel");
  return PTC_OK;
}
