// sfc_keys.h -- scalar space-filling-curve key functions shared by the device kernel
// (serialize.hip) and the host probe library (host_probe.cpp, CPU unit checks of product code).
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define PTC_HD __host__ __device__ __forceinline__
#else
#define PTC_HD static inline
#endif

// spread the low 16 bits of v so that bit i lands on bit 3i
PTC_HD uint64_t ptc_part1by2(uint64_t v) {
  v &= 0xffffull;
  v = (v | (v << 32)) & 0x001f00000000ffffull;
  v = (v | (v << 16)) & 0x001f0000ff0000ffull;
  v = (v | (v << 8)) & 0x100f00f00f00f00full;
  v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}

// z_order.py:40-50 : key = sum_i x_i<<(3i+2) | y_i<<(3i+1) | z_i<<(3i)
PTC_HD uint64_t ptc_morton3(uint32_t x, uint32_t y, uint32_t z) {
  return (ptc_part1by2(x) << 2) | (ptc_part1by2(y) << 1) | ptc_part1by2(z);
}

// hilbert.py:150-192, scalar form.  X0 is "dim 0".
PTC_HD uint64_t ptc_hilbert3(uint32_t X0, uint32_t X1, uint32_t X2, int depth) {
  for (int q = depth - 1; q >= 0; --q) {
    const uint32_t Q = 1u << q, P = Q - 1u;
    // dim 0 : "else" branch is a no-op for i == 0
    if (X0 & Q) X0 ^= P;
    // dim 1
    if (X1 & Q) X0 ^= P;
    else { uint32_t t = (X0 ^ X1) & P; X0 ^= t; X1 ^= t; }
    // dim 2
    if (X2 & Q) X0 ^= P;
    else { uint32_t t = (X0 ^ X2) & P; X0 ^= t; X2 ^= t; }
  }
  // interleave MSB-first, dim 0 most significant of each triple (hilbert.py:172)
  uint64_t g = (ptc_part1by2(X0) << 2) | (ptc_part1by2(X1) << 1) | ptc_part1by2(X2);
  // Gray -> binary: prefix XOR from the MSB (hilbert.py:175, gray2binary :69-88)
  g ^= g >> 1; g ^= g >> 2; g ^= g >> 4; g ^= g >> 8; g ^= g >> 16; g ^= g >> 32;
  return g;
}

