// voxel_hash.h -- voxel coordinate packing + hash shared by rulebook.hip and the host probe.
// Coordinates: batch < 1023, 0 <= x,y,z < 2^18 (262144 voxels per axis; ScanNet at 2 cm spans
// ~200, nuScenes at 5 cm ~2600).  The all-ones word is reserved as the EMPTY slot marker.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define PTC_HD3 __host__ __device__ __forceinline__
#else
#define PTC_HD3 static inline
#endif

#define PTC_VOX_BITS 18
#define PTC_VOX_MAX (1 << PTC_VOX_BITS)
#define PTC_VOX_BATCH_MAX 1023
#define PTC_HASH_EMPTY 0xffffffffffffffffull

// lexicographic (b, x, y, z) order == numeric order of the packed word
PTC_HD3 uint64_t ptc_vox_pack(int b, int x, int y, int z) {
  return ((uint64_t)(uint32_t)b << (3 * PTC_VOX_BITS)) | ((uint64_t)(uint32_t)x << (2 * PTC_VOX_BITS)) |
         ((uint64_t)(uint32_t)y << PTC_VOX_BITS) | (uint64_t)(uint32_t)z;
}
PTC_HD3 bool ptc_vox_in_range(int x, int y, int z) {
  return (unsigned)x < (unsigned)PTC_VOX_MAX && (unsigned)y < (unsigned)PTC_VOX_MAX && (unsigned)z < (unsigned)PTC_VOX_MAX;
}
// 64-bit finalizer (murmur3 fmix64)
PTC_HD3 uint64_t ptc_vox_hash(uint64_t h) {
  h ^= h >> 33; h *= 0xff51afd7ed558ccdull;
  h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull;
  h ^= h >> 33;
  return h;
}

// Home slot of a packed voxel key.  (A block-local variant -- hash the 4x4x4 block, add the position
// inside it -- was tried for cache locality in r01: linear probing degenerates under such clustered homes,
// E[run length] 3.3 -> 32 on the indoor scenes and the build / lookups got 4-6x SLOWER.  Locality comes
// from the voxel-major lookup order in rulebook.hip instead.)
PTC_HD3 uint64_t ptc_vox_home(uint64_t key) { return ptc_vox_hash(key); }
