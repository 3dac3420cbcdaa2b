// rulebook.hip -- kernel maps ("rulebooks") for sparse convolution, built on device.
//
// Replaces what spconv (third-party, un-vendored) builds inside SubMConv3d / SparseConv3d /
// SparseInverseConv3d on the first use of an indice_key.  Call sites in the reference:
//   pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py:278-284 (CPE k=3), :499-506 (stem k=5)
//   pointcept/models/sparse_unet/spconv_unet_v1m1_base.py:43-68,114-121 (SubM), :137-144 (k2 s2), :173-179 (inverse)
// Canonical form: dense gather tables nbr[kv][n_out] (int32, -1 = none),
// so the convolution is output-stationary: no atomics, bit-reproducible accumulation order.
#include "ptc_common.h"
#include "voxel_hash.h"
#include <stdlib.h>

extern "C" int ptc_sort_keys(const int64_t*, int64_t, int, int, int, int64_t*, int64_t*, void*, size_t, ptc_stream_t);
extern "C" size_t ptc_sort_keys_workspace_bytes(int64_t, int);
extern "C" size_t ptc_exclusive_scan_workspace_bytes(int64_t);
extern "C" int ptc_exclusive_scan_i32(const int32_t*, int64_t, int64_t*, void*, size_t, ptc_stream_t);

// ---- voxel table: open addressing over BUCKETS of 2x2x2 voxels ------------------------------------
// One 64-byte bucket = { block key (b, x>>1, y>>1, z>>1) | 8 row indices, one per voxel of the block }.
// A 3^3 window touches at most 8 buckets and a 5^3 window 27 -- one cache line each -- where the
// voxel-per-slot table of r01 needed 27 / 125 unrelated lines (k = 5: 1.36 ms, 4 % of the HBM roofline).
// Buckets are placed by the murmur finalizer of the block key (uniform homes: a locality-preserving
// "block-local" home was tried and made linear probing 4-6x slower); at most one bucket per occupied block,
// and there are never more occupied blocks than voxels, so n_buckets = pow2 >= n keeps the load <= 1 and
// typically ~1/3.  Duplicate voxels: lowest row index wins (atomicMin).
struct __attribute__((aligned(64))) VoxBucket {
  unsigned long long key;
  unsigned int pad[2];
  unsigned int vals[8];
  unsigned int pad2[4];
};
static_assert(sizeof(VoxBucket) == 64, "bucket = one 64-byte line");

extern "C" int64_t ptc_hash_table_size(int64_t n) {   // number of buckets
  int64_t t = 1024;
  while (t < n) t <<= 1;
  return t;
}
extern "C" size_t ptc_hash_table_bytes(int64_t n) { return (size_t)ptc_hash_table_size(n) * sizeof(VoxBucket); }

__device__ __forceinline__ unsigned long long vox_block_key(int b, int x, int y, int z) { return ptc_vox_pack(b, x >> 1, y >> 1, z >> 1); }
__device__ __forceinline__ int vox_local(int x, int y, int z) { return ((x & 1) << 2) | ((y & 1) << 1) | (z & 1); }

__global__ void __launch_bounds__(256)
hash_insert_kernel(const int32_t* __restrict__ indices, int64_t n, VoxBucket* __restrict__ table, uint64_t mask) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int4 c = reinterpret_cast<const int4*>(indices)[i];
    const unsigned long long key = vox_block_key(c.x, c.y, c.z, c.w);
    uint64_t slot = ptc_vox_home(key) & mask;
    for (uint64_t probe = 0; probe <= mask; ++probe) {
      const unsigned long long prev = atomicCAS(&table[slot].key, (unsigned long long)PTC_HASH_EMPTY, key);
      if (prev == PTC_HASH_EMPTY || prev == key) {
        atomicMin(&table[slot].vals[vox_local(c.y, c.z, c.w)], (unsigned int)i);  // duplicate voxels: lowest row index wins
        break;
      }
      slot = (slot + 1) & mask;
    }
  }
}

extern "C" int ptc_hash_build(const int32_t* indices, int64_t n, void* table, size_t table_bytes, ptc_stream_t stream) {
  PTC_REQUIRE(n >= 0, PTC_EINVAL, "ptc_hash_build: n < 0");
  PTC_REQUIRE(n < (1ll << 31), PTC_EUNSUPPORTED, "ptc_hash_build: n >= 2^31");
  PTC_REQUIRE(table && table_bytes >= ptc_hash_table_bytes(n) && (n == 0 || indices), PTC_EINVAL, "ptc_hash_build: null / short buffer");
  PTC_REQUIRE((uintptr_t)table % 64 == 0, PTC_EINVAL, "ptc_hash_build: table must be 64-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int64_t nb = ptc_hash_table_size(n);
  PTC_HIP(hipMemsetAsync(table, 0xff, (size_t)nb * sizeof(VoxBucket), s));   // keys EMPTY, rows -1
  if (n == 0) return PTC_OK;
  int64_t grid = ptc_cdiv(n, 256);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(hash_insert_kernel, dim3((unsigned)grid), dim3(256), 0, s, indices, n, (VoxBucket*)table, (uint64_t)(nb - 1));
  PTC_CHECK_LAUNCH("hash_insert_kernel");
  return PTC_OK;
}

// One thread per voxel.  Its ks^3 window spans NB = ks/2 + 1 blocks per axis; each block is ONE bucket probe
// (key compare) and, on a hit, one 32-byte read of its 8 row indices, which are then scattered to the window
// offsets they belong to.  Consecutive lanes hold neighbouring voxels (curve order), so the handful of
// buckets of a window is shared by the whole wave and served by L1/L2.
// Round 2: the probes of a window are INDEPENDENT, the first version walked them one after the other
// (`#pragma unroll 1`): rocprofv3 PMC at 819200 voxels: 81 % of the wave cycles in s_waitcnt, 9 % issuing
// (profiles/r02_a_conv_pmc_s0.json) -- a chain of 8 (k = 3) or 27 (k = 5) dependent L2 latencies per voxel.  Now the
// home-slot keys of a GROUP of blocks (8 / 9) are requested together, collisions (rare at load <= 1/3: linear probing
// from the home slot) are resolved per block, then the 32-byte value reads of the group go out together.
template <int KS>
__global__ void __launch_bounds__(256)
rulebook_subm_kernel(const int32_t* __restrict__ indices, int64_t n, const VoxBucket* __restrict__ table, uint64_t mask,
                     int32_t* __restrict__ nbr) {
  constexpr int R = KS / 2, NB = R + 1, NBLK = NB * NB * NB;
  constexpr int GB = NBLK <= 9 ? NBLK : 9;                 // blocks per group: 1 | 8 | 9 (k = 5, 7: 3 / 8 groups... NBLK % GB == 0 for 1,8,27,64)
  constexpr int NG = (NBLK + GB - 1) / GB;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int4 c = reinterpret_cast<const int4*>(indices)[i];
    const int bx0 = (c.y - R) >> 1, by0 = (c.z - R) >> 1, bz0 = (c.w - R) >> 1;   // arithmetic shift: floor for negatives
    const int ex = c.y - 2 * bx0, ey = c.z - 2 * by0, ez = c.w - 2 * bz0;         // R or R + 1
#pragma unroll 1
    for (int grp = 0; grp < NG; ++grp) {
      unsigned long long key[GB], kk[GB];
      uint64_t slot[GB];
      bool live[GB];
#pragma unroll
      for (int q = 0; q < GB; ++q) {
        const int ob = grp * GB + q;
        const int o0 = ob / (NB * NB), o1 = (ob / NB) % NB, o2 = ob % NB;
        const int bx = bx0 + o0, by = by0 + o1, bz = bz0 + o2;
        live[q] = ob < NBLK && bx >= 0 && by >= 0 && bz >= 0 && bx < (PTC_VOX_MAX >> 1) && by < (PTC_VOX_MAX >> 1) && bz < (PTC_VOX_MAX >> 1);
        key[q] = ptc_vox_pack(c.x, bx, by, bz);
        slot[q] = ptc_vox_home(key[q]) & mask;
        kk[q] = table[slot[q]].key;                                   // all GB home-slot keys in flight (always in bounds)
      }
#pragma unroll
      for (int q = 0; q < GB; ++q) {
        bool found = live[q] && kk[q] == key[q];
        if (live[q] && !found && kk[q] != PTC_HASH_EMPTY) {           // collision at the home slot: walk on
          uint64_t sl = slot[q];
          for (uint64_t probe = 1; probe <= mask; ++probe) {
            sl = (sl + 1) & mask;
            const unsigned long long k2 = table[sl].key;
            if (k2 == key[q]) { found = true; slot[q] = sl; break; }
            if (k2 == PTC_HASH_EMPTY) break;
          }
        }
        live[q] = found;
      }
      uint4 v0[GB], v1[GB];
#pragma unroll
      for (int q = 0; q < GB; ++q) {
        const uint4* pv = reinterpret_cast<const uint4*>(table[slot[q]].vals);   // in bounds whether found or not
        v0[q] = pv[0];
        v1[q] = pv[1];
      }
#pragma unroll
      for (int q = 0; q < GB; ++q) {
        const int ob = grp * GB + q;
        if (ob >= NBLK) continue;
        const int o0 = ob / (NB * NB), o1 = (ob / NB) % NB, o2 = ob % NB;
        const uint4 none = make_uint4(~0u, ~0u, ~0u, ~0u);
        const uint4 a = live[q] ? v0[q] : none, b2 = live[q] ? v1[q] : none;
        const unsigned int vv[8] = {a.x, a.y, a.z, a.w, b2.x, b2.y, b2.z, b2.w};
#pragma unroll
        for (int cell = 0; cell < 8; ++cell) {
          const int d0 = 2 * o0 + (cell >> 2) - ex, d1 = 2 * o1 + ((cell >> 1) & 1) - ey, d2 = 2 * o2 + (cell & 1) - ez;
          if (d0 >= -R && d0 <= R && d1 >= -R && d1 <= R && d2 >= -R && d2 <= R) {
            const int k = ((d0 + R) * KS + (d1 + R)) * KS + (d2 + R);
            nbr[(int64_t)k * n + i] = (int32_t)vv[cell];
          }
        }
      }
    }
  }
}

extern "C" int ptc_rulebook_subm(const int32_t* indices, int64_t n, int ksize, const void* table, size_t table_bytes, int32_t* nbr,
                                 ptc_stream_t stream) {
  PTC_REQUIRE(n >= 0, PTC_EINVAL, "ptc_rulebook_subm: n < 0");
  PTC_REQUIRE(ksize >= 1 && ksize <= 7 && (ksize & 1), PTC_EUNSUPPORTED, "ptc_rulebook_subm: ksize %d (odd, <= 7)", ksize);
  if (n == 0) return PTC_OK;
  PTC_REQUIRE(indices && table && nbr && table_bytes >= ptc_hash_table_bytes(n), PTC_EINVAL, "ptc_rulebook_subm: null / short buffer");
  int64_t grid = ptc_cdiv(n, 256);
  if (grid > 256 * 64) grid = 256 * 64;
  const uint64_t mask = (uint64_t)(ptc_hash_table_size(n) - 1);
  const VoxBucket* tb = (const VoxBucket*)table;
  hipStream_t s = (hipStream_t)stream;
  switch (ksize) {
    case 1: hipLaunchKernelGGL(rulebook_subm_kernel<1>, dim3((unsigned)grid), dim3(256), 0, s, indices, n, tb, mask, nbr); break;
    case 3: hipLaunchKernelGGL(rulebook_subm_kernel<3>, dim3((unsigned)grid), dim3(256), 0, s, indices, n, tb, mask, nbr); break;
    case 5: hipLaunchKernelGGL(rulebook_subm_kernel<5>, dim3((unsigned)grid), dim3(256), 0, s, indices, n, tb, mask, nbr); break;
    default: hipLaunchKernelGGL(rulebook_subm_kernel<7>, dim3((unsigned)grid), dim3(256), 0, s, indices, n, tb, mask, nbr); break;
  }
  PTC_CHECK_LAUNCH("rulebook_subm_kernel");
  return PTC_OK;
}

// ------------------------------------------------------------------------------------------------
// strided k=2 s=2 conv: coarse sites = unique (b, x>>1, y>>1, z>>1), numbered by ascending (batch, Morton code of the coarse
// coordinate).  Round 6: the numbering was the lexicographic (b, x, y, z) key -- rows that follow each other were then neighbours along z
// only, and a block of 128 consecutive coarse rows named ~600 distinct input rows in its 3^3 neighbourhood (the block-staged kernels'
// LDS image holds 352 / 416) where a curve order names ~230 (tools/halo_stats.py --stride 2).  spconv's own numbering of the output
// sites is its hash table's insertion order -- no caller may rely on it; the oracle (oracle/ops.py down_rulebook) numbers the same way.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t down_spread3(uint32_t v) {          // bit i of a 21-bit value -> bit 3 i
  uint64_t x = v & 0x1fffffu;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}
__global__ void __launch_bounds__(256)
down_keys_kernel(const int32_t* __restrict__ indices, int64_t n, int cb, int64_t* __restrict__ keys) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int4 c = reinterpret_cast<const int4*>(indices)[i];
    // (b, Morton(x>>1, y>>1, z>>1)): bit i of x at 3 i + 2, of y at 3 i + 1, of z at 3 i; cb bits per axis
    keys[i] = (int64_t)((uint64_t)(uint32_t)c.x << (3 * cb) | down_spread3((uint32_t)(c.y >> 1)) << 2 | down_spread3((uint32_t)(c.z >> 1)) << 1 |
                        down_spread3((uint32_t)(c.w >> 1)));
  }
}

__global__ void __launch_bounds__(256)
down_flags_kernel(const int64_t* __restrict__ keys, const int64_t* __restrict__ order, int64_t n,
                  int32_t* __restrict__ flags) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride)
    flags[r] = (r == 0 || keys[order[r - 1]] != keys[order[r]]) ? 1 : 0;
}

__global__ void __launch_bounds__(256)
down_assign_kernel(const int64_t* __restrict__ order, const int32_t* __restrict__ flags,
                   const int64_t* __restrict__ excl, int64_t n, int32_t* __restrict__ out_of_in,
                   int64_t* __restrict__ n_out_dev) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) {
    const int64_t id = excl[r] + flags[r] - 1;
    out_of_in[order[r]] = (int32_t)id;
    if (r == n - 1) *n_out_dev = id + 1;
  }
}

struct DownLayout { size_t keys, order, flags, excl, scan, sort, total; };
static DownLayout down_layout(int64_t n) {
  DownLayout L;
  const size_t nn = (size_t)(n > 0 ? n : 1);
  size_t o = 0;
  L.keys = o; o += ptc_align_up(nn * 8, 256);
  L.order = o; o += ptc_align_up(nn * 8, 256);
  L.flags = o; o += ptc_align_up(nn * 4, 256);
  L.excl = o; o += ptc_align_up(nn * 8, 256);
  L.scan = o; o += ptc_exclusive_scan_workspace_bytes((int64_t)nn);
  L.sort = o; o += ptc_sort_keys_workspace_bytes((int64_t)nn, 1);
  L.total = o;
  return L;
}

extern "C" size_t ptc_rulebook_down_workspace_bytes(int64_t n_in) { return down_layout(n_in).total; }

extern "C" int ptc_rulebook_down_count(const int32_t* indices, int64_t n_in, int coord_bits, int batch_bits,
                                       int32_t* out_of_in, int64_t* n_out_dev, void* workspace,
                                       size_t workspace_bytes, ptc_stream_t stream) {
  PTC_REQUIRE(n_in >= 1, PTC_EINVAL, "ptc_rulebook_down_count: n_in=%lld", (long long)n_in);
  PTC_REQUIRE(coord_bits >= 1 && coord_bits <= PTC_VOX_BITS && batch_bits >= 0 && batch_bits <= 10, PTC_EINVAL,
              "ptc_rulebook_down_count: coord_bits=%d batch_bits=%d", coord_bits, batch_bits);
  PTC_REQUIRE(indices && out_of_in && n_out_dev && workspace, PTC_EINVAL, "ptc_rulebook_down_count: null buffer");
  const DownLayout L = down_layout(n_in);
  PTC_REQUIRE(workspace_bytes >= L.total, PTC_EWORKSPACE, "ptc_rulebook_down_count: workspace %zu < %zu", workspace_bytes, L.total);
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  int64_t* keys = (int64_t*)(ws + L.keys);
  int64_t* order = (int64_t*)(ws + L.order);
  int32_t* flags = (int32_t*)(ws + L.flags);
  int64_t* excl = (int64_t*)(ws + L.excl);
  int64_t grid = ptc_cdiv(n_in, 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(down_keys_kernel, dim3((unsigned)grid), dim3(256), 0, s, indices, n_in, coord_bits, keys);
  PTC_CHECK_LAUNCH("down_keys_kernel");
  // compact key: coord_bits per axis (host derives it from the spatial shape), so the sort only
  // runs ceil((3*coord_bits + batch_bits)/8) passes (27 bits -> 4 passes for ScanNet).
  int rc = ptc_sort_keys(keys, n_in, 1, 0, 3 * coord_bits + batch_bits, order, nullptr, ws + L.sort,
                         ptc_sort_keys_workspace_bytes(n_in, 1), stream);
  if (rc != PTC_OK) return rc;
  hipLaunchKernelGGL(down_flags_kernel, dim3((unsigned)grid), dim3(256), 0, s, keys, order, n_in, flags);
  PTC_CHECK_LAUNCH("down_flags_kernel");
  rc = ptc_exclusive_scan_i32(flags, n_in, excl, ws + L.scan, ptc_exclusive_scan_workspace_bytes(n_in), stream);
  if (rc != PTC_OK) return rc;
  hipLaunchKernelGGL(down_assign_kernel, dim3((unsigned)grid), dim3(256), 0, s, order, flags, excl, n_in, out_of_in, n_out_dev);
  PTC_CHECK_LAUNCH("down_assign_kernel");
  return PTC_OK;
}

__global__ void __launch_bounds__(256)
down_fill_kernel(const int32_t* __restrict__ indices, int64_t n_in, const int32_t* __restrict__ out_of_in,
                 int64_t n_out, int32_t* __restrict__ out_indices, unsigned int* __restrict__ nbr_down,
                 int32_t* __restrict__ nbr_up) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_in; j += stride) {
    const int4 c = reinterpret_cast<const int4*>(indices)[j];
    const int o = out_of_in[j];
    const int k = ((c.y & 1) << 2) | ((c.z & 1) << 1) | (c.w & 1);
    atomicMin(&nbr_down[(int64_t)k * n_out + o], (unsigned int)j);  // duplicate voxels: lowest row wins
    // every member writes the same coarse coordinate
    reinterpret_cast<int4*>(out_indices)[o] = make_int4(c.x, c.y >> 1, c.z >> 1, c.w >> 1);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) nbr_up[(int64_t)kk * n_in + j] = (kk == k) ? o : -1;
  }
}

extern "C" int ptc_rulebook_down_fill(const int32_t* indices, int64_t n_in, const int32_t* out_of_in, int64_t n_out,
                                      int32_t* out_indices, int32_t* nbr_down, int32_t* nbr_up, ptc_stream_t stream) {
  PTC_REQUIRE(n_in >= 1 && n_out >= 1 && n_out <= n_in, PTC_EINVAL, "ptc_rulebook_down_fill: bad sizes");
  PTC_REQUIRE(indices && out_of_in && out_indices && nbr_down && nbr_up, PTC_EINVAL, "ptc_rulebook_down_fill: null buffer");
  hipStream_t s = (hipStream_t)stream;
  PTC_HIP(hipMemsetAsync(nbr_down, 0xff, (size_t)n_out * 8 * 4, s));
  int64_t grid = ptc_cdiv(n_in, 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(down_fill_kernel, dim3((unsigned)grid), dim3(256), 0, s, indices, n_in, out_of_in, n_out,
                     out_indices, (unsigned int*)nbr_down, nbr_up);
  PTC_CHECK_LAUNCH("down_fill_kernel");
  return PTC_OK;
}
