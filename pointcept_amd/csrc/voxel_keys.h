// voxel_keys.h -- per-point arithmetic of the GridSample front end and per-slot arithmetic of the Lovasz loss, shared
// by the device kernels (voxelize.hip, lovasz.hip) and the host probe library (CPU unit checks of product code).
#pragma once
#include <math.h>
#include <stdint.h>
#if defined(__HIPCC__)
#define PTC_HD4 __host__ __device__ __forceinline__
#else
#define PTC_HD4 static inline
#endif

// floor(coord / grid_size) with the division in float64, as numpy promotes `float32 array / float64 0-d array`
// (pointcept/datasets/transform.py:867-868)
PTC_HD4 long long ptc_voxel_floor(float c, double grid_size) { return (long long)floor((double)c / grid_size); }

// fnv_hash_vec (transform.py:997-1011): multiply by the FNV prime FIRST, then xor the coordinate word
PTC_HD4 unsigned long long ptc_fnv3(unsigned long long x, unsigned long long y, unsigned long long z) {
  unsigned long long h = 14695981039346656037ull;
  h *= 1099511628211ull; h ^= x;
  h *= 1099511628211ull; h ^= y;
  h *= 1099511628211ull; h ^= z;
  return h;
}

// Jaccard step of one sorted slot of the Lovasz extension (pointcept/models/losses/lovasz.py:22-33), from integer
// counts: gts = foreground total of the class, cum_fg / cum_bg = foreground / background slots up to and including this
// one.  jaccard_i - jaccard_{i-1} with U = gts + cum_bg (union so far), I = gts - cum_fg (foreground still to come):
// a foreground slot lowers I by one at constant U -> 1/U; a background slot raises U by one at constant I -> I/(U(U-1)).
PTC_HD4 double ptc_lovasz_step(long long gts, long long cum_fg, long long cum_bg, int is_fg) {
  const double U = (double)(gts + cum_bg), I = (double)(gts - cum_fg);
  return is_fg ? 1.0 / U : I / (U * (U - 1.0));
}
