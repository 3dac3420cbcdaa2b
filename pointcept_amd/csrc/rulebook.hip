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

extern "C" int ptc_sort_keys(const int64_t*, int64_t, int, int, int, int64_t*, int64_t*, void*, size_t, ptc_stream_t);
extern "C" size_t ptc_sort_keys_workspace_bytes(int64_t, int);
extern "C" size_t ptc_exclusive_scan_workspace_bytes(int64_t);
extern "C" int ptc_exclusive_scan_i32(const int32_t*, int64_t, int64_t*, void*, size_t, ptc_stream_t);

extern "C" int64_t ptc_hash_table_size(int64_t n) {
  int64_t t = 1024;
  while (t < 2 * n) t <<= 1;
  return t;
}

__global__ void __launch_bounds__(256)
hash_insert_kernel(const int32_t* __restrict__ indices, int64_t n, unsigned long long* __restrict__ keys,
                   unsigned int* __restrict__ vals, uint64_t mask) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int4 c = reinterpret_cast<const int4*>(indices)[i];
    const unsigned long long key = ptc_vox_pack(c.x, c.y, c.z, c.w);
    uint64_t slot = ptc_vox_home(key) & mask;
    for (uint64_t probe = 0; probe <= mask; ++probe) {
      const unsigned long long prev = atomicCAS(&keys[slot], (unsigned long long)PTC_HASH_EMPTY, key);
      if (prev == PTC_HASH_EMPTY || prev == key) {
        atomicMin(&vals[slot], (unsigned int)i);  // duplicate voxels: lowest row index wins
        break;
      }
      slot = (slot + 1) & mask;
    }
  }
}

__device__ __forceinline__ int32_t hash_lookup(const uint64_t* __restrict__ keys, const int32_t* __restrict__ vals,
                                               uint64_t mask, uint64_t key) {
  uint64_t slot = ptc_vox_home(key) & mask;
  for (uint64_t probe = 0; probe <= mask; ++probe) {
    const uint64_t k = keys[slot];
    if (k == key) return vals[slot];
    if (k == PTC_HASH_EMPTY) return -1;
    slot = (slot + 1) & mask;
  }
  return -1;
}

extern "C" int ptc_hash_build(const int32_t* indices, int64_t n, uint64_t* table_keys, int32_t* table_vals,
                              int64_t table_size, ptc_stream_t stream) {
  PTC_REQUIRE(n >= 0, PTC_EINVAL, "ptc_hash_build: n < 0");
  PTC_REQUIRE(table_size >= 2 * n && table_size >= 2 && (table_size & (table_size - 1)) == 0, PTC_EINVAL,
              "ptc_hash_build: table_size %lld must be a power of two >= 2n", (long long)table_size);
  PTC_REQUIRE(n < (1ll << 31), PTC_EUNSUPPORTED, "ptc_hash_build: n >= 2^31");
  PTC_REQUIRE(table_keys && table_vals && (n == 0 || indices), PTC_EINVAL, "ptc_hash_build: null buffer");
  hipStream_t s = (hipStream_t)stream;
  PTC_HIP(hipMemsetAsync(table_keys, 0xff, (size_t)table_size * 8, s));
  PTC_HIP(hipMemsetAsync(table_vals, 0xff, (size_t)table_size * 4, s));
  if (n == 0) return PTC_OK;
  int64_t grid = ptc_cdiv(n, 256);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(hash_insert_kernel, dim3((unsigned)grid), dim3(256), 0, s, indices, n,
                     (unsigned long long*)table_keys, (unsigned int*)table_vals, (uint64_t)(table_size - 1));
  PTC_CHECK_LAUNCH("hash_insert_kernel");
  return PTC_OK;
}

// One thread per voxel, all ks^3 probes.  Consecutive lanes hold voxels that are neighbours along the
// serialization curve, and neighbouring voxels share most of their windows (18 of 27 cells for a face
// neighbour): the cells a wave probes are re-probed by the same wave within microseconds and hit L1/L2,
// instead of being touched once per offset in 27 / 125 separate sweeps over the 24 MB table (the
// offset-major mapping of r01: 1.76 ms for k = 5, 3 % of the HBM roofline).  The ks probes along z of one (dx,dy) column are issued together (ks independent loads in
// flight); nbr[k][i] stores are coalesced across lanes for every k.
template <int KS>
__global__ void __launch_bounds__(256)
rulebook_subm_kernel(const int32_t* __restrict__ indices, int64_t n, const uint64_t* __restrict__ keys,
                     const int32_t* __restrict__ vals, uint64_t mask, int32_t* __restrict__ nbr) {
  constexpr int R = KS / 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int4 c = reinterpret_cast<const int4*>(indices)[i];
#pragma unroll 1
    for (int a = 0; a < KS * KS; ++a) {
      const int d0 = a / KS - R, d1 = a % KS - R;
      const int x = c.y + d0, y = c.z + d1;
      uint64_t key[KS], slot[KS], got[KS];
      bool ok[KS];
#pragma unroll
      for (int b = 0; b < KS; ++b) {
        const int z = c.w + b - R;
        ok[b] = ptc_vox_in_range(x, y, z);
        key[b] = ptc_vox_pack(c.x, x, y, z);
        slot[b] = ptc_vox_home(key[b]) & mask;
        got[b] = ok[b] ? keys[slot[b]] : PTC_HASH_EMPTY;
      }
#pragma unroll
      for (int b = 0; b < KS; ++b) {
        int32_t j = -1;
        if (got[b] == key[b]) {
          j = vals[slot[b]];
        } else if (got[b] != PTC_HASH_EMPTY) {   // collision: continue the linear probe
          uint64_t sl = (slot[b] + 1) & mask;
          for (uint64_t probe = 1; probe <= mask; ++probe) {
            const uint64_t kk = keys[sl];
            if (kk == key[b]) { j = vals[sl]; break; }
            if (kk == PTC_HASH_EMPTY) break;
            sl = (sl + 1) & mask;
          }
        }
        nbr[(int64_t)(a * KS + b) * n + i] = j;
      }
    }
  }
}

extern "C" int ptc_rulebook_subm(const int32_t* indices, int64_t n, int ksize, const uint64_t* table_keys,
                                 const int32_t* table_vals, int64_t table_size, int32_t* nbr, ptc_stream_t stream) {
  PTC_REQUIRE(n >= 0, PTC_EINVAL, "ptc_rulebook_subm: n < 0");
  PTC_REQUIRE(ksize >= 1 && ksize <= 7 && (ksize & 1), PTC_EUNSUPPORTED, "ptc_rulebook_subm: ksize %d (odd, <= 7)", ksize);
  PTC_REQUIRE(table_size >= 2 && (table_size & (table_size - 1)) == 0, PTC_EINVAL, "ptc_rulebook_subm: bad table_size");
  if (n == 0) return PTC_OK;
  PTC_REQUIRE(indices && table_keys && table_vals && nbr, PTC_EINVAL, "ptc_rulebook_subm: null buffer");
  int64_t grid = ptc_cdiv(n, 256);
  if (grid > 256 * 64) grid = 256 * 64;
  const uint64_t mask = (uint64_t)(table_size - 1);
  hipStream_t s = (hipStream_t)stream;
  switch (ksize) {
    case 1: hipLaunchKernelGGL(rulebook_subm_kernel<1>, dim3((unsigned)grid), dim3(256), 0, s, indices, n, table_keys, table_vals, mask, nbr); break;
    case 3: hipLaunchKernelGGL(rulebook_subm_kernel<3>, dim3((unsigned)grid), dim3(256), 0, s, indices, n, table_keys, table_vals, mask, nbr); break;
    case 5: hipLaunchKernelGGL(rulebook_subm_kernel<5>, dim3((unsigned)grid), dim3(256), 0, s, indices, n, table_keys, table_vals, mask, nbr); break;
    default: hipLaunchKernelGGL(rulebook_subm_kernel<7>, dim3((unsigned)grid), dim3(256), 0, s, indices, n, table_keys, table_vals, mask, nbr); break;
  }
  PTC_CHECK_LAUNCH("rulebook_subm_kernel");
  return PTC_OK;
}

// ------------------------------------------------------------------------------------------------
// strided k=2 s=2 conv: coarse sites = unique (b, x>>1, y>>1, z>>1), ascending packed key
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
down_keys_kernel(const int32_t* __restrict__ indices, int64_t n, int cb, int64_t* __restrict__ keys) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int4 c = reinterpret_cast<const int4*>(indices)[i];
    // lexicographic (b, x>>1, y>>1, z>>1) with cb bits per axis
    keys[i] = (int64_t)(((((uint64_t)(uint32_t)c.x << cb | (uint64_t)(uint32_t)(c.y >> 1)) << cb) |
                         (uint64_t)(uint32_t)(c.z >> 1)) << cb | (uint64_t)(uint32_t)(c.w >> 1));
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
