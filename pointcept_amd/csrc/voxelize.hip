// voxelize.hip -- the integer front end of GridSample (the step immediately before the hot path).
//
// Replaces, for device-resident point clouds, the numpy prologue of GridSample.__call__
// (pointcept/datasets/transform.py:867-875):
//     scaled = coord / grid_size;  grid = floor(scaled).astype(int);  grid -= grid.min(0);  key = fnv_hash_vec(grid)
// (fnv_hash_vec :997-1011: h = 14695981039346656037; for each axis: h *= 1099511628211; h ^= g).  The rest of the
// transform -- argsort(key), unique / inverse / count, the per-voxel representative -- runs on the radix sort of
// scan_sort.hip and the cluster maps of maps.hip (ptc_sort_keys, ptc_pool_maps_count / _fill): the same
// sort / flag / scan / fill machinery that orders the serialization curves and builds the pooling clusters.
//
// Arithmetic: numpy divides the float32 coordinates by a float64 0-d array, i.e. in float64 (NEP 50), and floors the
// float64 quotient; the kernel does exactly that (double division, floor), so the voxel of every point is bit-exact.
// Two streaming passes (12 B in + 24 B out, then 24 B in + 24 B + 8 B out per point); the minimum is an integer
// atomic (exact, order free).
#include "ptc_common.h"
#include "voxel_keys.h"

__global__ void __launch_bounds__(256)
voxel_floor_kernel(const float* __restrict__ coord, int64_t n, double grid_size, int64_t* __restrict__ grid,
                   long long* __restrict__ min3) {
  long long m0 = LLONG_MAX, m1 = LLONG_MAX, m2 = LLONG_MAX;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const long long x = ptc_voxel_floor(coord[3 * i], grid_size);
    const long long y = ptc_voxel_floor(coord[3 * i + 1], grid_size);
    const long long z = ptc_voxel_floor(coord[3 * i + 2], grid_size);
    grid[3 * i] = x;
    grid[3 * i + 1] = y;
    grid[3 * i + 2] = z;
    m0 = x < m0 ? x : m0;
    m1 = y < m1 ? y : m1;
    m2 = z < m2 ? z : m2;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const long long a = __shfl_xor(m0, o, 64), b = __shfl_xor(m1, o, 64), c = __shfl_xor(m2, o, 64);
    m0 = a < m0 ? a : m0;
    m1 = b < m1 ? b : m1;
    m2 = c < m2 ? c : m2;
  }
  if (ptc_lane() == 0) {
    atomicMin(min3 + 0, m0);
    atomicMin(min3 + 1, m1);
    atomicMin(min3 + 2, m2);
  }
}

__global__ void __launch_bounds__(256)
voxel_key_kernel(int64_t* __restrict__ grid, int64_t n, const long long* __restrict__ min3, int64_t* __restrict__ key) {
  const long long m0 = min3[0], m1 = min3[1], m2 = min3[2];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const unsigned long long x = (unsigned long long)(grid[3 * i] - m0), y = (unsigned long long)(grid[3 * i + 1] - m1),
                             z = (unsigned long long)(grid[3 * i + 2] - m2);
    grid[3 * i] = (int64_t)x;
    grid[3 * i + 1] = (int64_t)y;
    grid[3 * i + 2] = (int64_t)z;
    key[i] = (int64_t)ptc_fnv3(x, y, z);
  }
}

extern "C" int ptc_voxel_keys(const float* coord, int64_t n, double grid_size, int64_t* grid_coord, int64_t* min_coord3,
                              int64_t* key, ptc_stream_t stream) {
  PTC_REQUIRE(n >= 0, PTC_EINVAL, "ptc_voxel_keys: n < 0");
  PTC_REQUIRE(grid_size > 0.0, PTC_EINVAL, "ptc_voxel_keys: grid_size must be positive");
  PTC_REQUIRE(min_coord3 != nullptr, PTC_EINVAL, "ptc_voxel_keys: null min_coord3");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    PTC_HIP(hipMemsetAsync(min_coord3, 0, 3 * sizeof(int64_t), s));
    return PTC_OK;
  }
  PTC_REQUIRE(coord && grid_coord && key, PTC_EINVAL, "ptc_voxel_keys: null buffer");
  static const long long init[3] = {LLONG_MAX, LLONG_MAX, LLONG_MAX};
  PTC_HIP(hipMemcpyAsync(min_coord3, init, sizeof(init), hipMemcpyHostToDevice, s));
  int64_t grid = ptc_cdiv(n, 256 * 4);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(voxel_floor_kernel, dim3((unsigned)grid), dim3(256), 0, s, coord, n, grid_size, grid_coord, (long long*)min_coord3);
  PTC_CHECK_LAUNCH("voxel_floor_kernel");
  hipLaunchKernelGGL(voxel_key_kernel, dim3((unsigned)grid), dim3(256), 0, s, grid_coord, n, (const long long*)min_coord3, key);
  PTC_CHECK_LAUNCH("voxel_key_kernel");
  return PTC_OK;
}
