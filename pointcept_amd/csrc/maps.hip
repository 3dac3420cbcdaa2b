// maps.hip -- integer index maps of the PTv3 path, built on device without host loops:
//   * patch padding maps (pad / unpad / cu_seqlens / dup)   -- ptv3m1:114-170
//   * serialized pooling maps (cluster / idx_ptr / head / child codes) -- ptv3m1:383-398
// ("ptv3m1" = pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py)
#include "ptc_common.h"
#include "pad_maps.h"

#define PM_MAX_B 1024

// One launch: every block rebuilds the tiny per-scene prefix tables in LDS (B <= 1024), then
// threads cover max(n_pad, n, n_seq+1) positions.
__global__ void __launch_bounds__(256)
patch_pad_maps_kernel(const int64_t* __restrict__ offset, int B, int64_t K, int64_t n, int64_t n_pad,
                      int64_t n_seq, int64_t* __restrict__ pad, int64_t* __restrict__ unpad,
                      int32_t* __restrict__ cu_seqlens, int64_t* __restrict__ dup) {
  __shared__ int64_t s_off[PM_MAX_B];      // cumulative ends of the unpadded scenes
  __shared__ int64_t s_offpad[PM_MAX_B];   // cumulative ends of the padded scenes
  __shared__ int64_t s_seq[PM_MAX_B];      // cumulative ends of the per-scene sequence counts
  for (int i = threadIdx.x; i < B; i += blockDim.x) s_off[i] = offset[i];
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t prev = 0, accp = 0, accs = 0;
    for (int i = 0; i < B; ++i) {
      const int64_t ni = s_off[i] - prev;
      prev = s_off[i];
      accp += ptc_padded_len(ni, K);
      accs += ptc_num_seq(ni, K);
      s_offpad[i] = accp;
      s_seq[i] = accs;
    }
  }
  __syncthreads();
  int64_t work = n_pad > n ? n_pad : n;
  if (n_seq + 1 > work) work = n_seq + 1;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < work; t += stride) {
    if (t < n_pad) {
      const int i = ptc_find_scene(s_offpad, B, t);
      const int64_t o0 = i ? s_off[i - 1] : 0, p0 = i ? s_offpad[i - 1] : 0;
      const int64_t ni = s_off[i] - o0;
      pad[t] = o0 + ptc_pad_local(t - p0, ni, K);
    }
    if (t < n) {
      const int i = ptc_find_scene(s_off, B, t);
      const int64_t o0 = i ? s_off[i - 1] : 0, p0 = i ? s_offpad[i - 1] : 0;
      const int64_t ni = s_off[i] - o0;
      unpad[t] = t - o0 + p0;
      if (dup) {
        const int64_t d = ptc_dup_local(t - o0, ni, K);
        dup[t] = d < 0 ? -1 : p0 + d;
      }
    }
    if (t < n_seq) {
      const int i = ptc_find_scene(s_seq, B, t);
      const int64_t q0 = i ? s_seq[i - 1] : 0, p0 = i ? s_offpad[i - 1] : 0;
      cu_seqlens[t] = (int32_t)(p0 + (t - q0) * K);
    } else if (t == n_seq) {
      cu_seqlens[t] = (int32_t)n_pad;
    }
  }
}

extern "C" int ptc_patch_pad_maps(const int64_t* offset, int B, int patch, int64_t n, int64_t n_pad,
                                  int64_t n_seq, int64_t* pad, int64_t* unpad, int32_t* cu_seqlens,
                                  int64_t* dup, ptc_stream_t stream) {
  PTC_REQUIRE(B >= 1 && B <= PM_MAX_B, PTC_EUNSUPPORTED, "ptc_patch_pad_maps: B=%d not in [1,%d]", B, PM_MAX_B);
  PTC_REQUIRE(patch >= 1, PTC_EINVAL, "ptc_patch_pad_maps: patch=%d", patch);
  PTC_REQUIRE(n >= 0 && n_pad >= n && n_seq >= 0, PTC_EINVAL, "ptc_patch_pad_maps: bad sizes n=%lld n_pad=%lld n_seq=%lld",
              (long long)n, (long long)n_pad, (long long)n_seq);
  PTC_REQUIRE(n_pad < (1ll << 31), PTC_EUNSUPPORTED, "ptc_patch_pad_maps: n_pad does not fit int32 cu_seqlens");
  PTC_REQUIRE(offset && cu_seqlens && (n_pad == 0 || pad) && (n == 0 || unpad), PTC_EINVAL, "ptc_patch_pad_maps: null buffer");
  int64_t work = n_pad > n ? n_pad : n;
  if (n_seq + 1 > work) work = n_seq + 1;
  int64_t grid = ptc_cdiv(work, 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(patch_pad_maps_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, offset, B,
                     (int64_t)patch, n, n_pad, n_seq, pad, unpad, cu_seqlens, dup);
  PTC_CHECK_LAUNCH("patch_pad_maps_kernel");
  return PTC_OK;
}

// ------------------------------------------------------------------------------------------------
// Gather tables of one serialized-attention order, in ONE launch (the torch form is ~12 indexing / cast
// kernels per (stage, order): order[pad], unpad[inverse], dup[inverse], inv[gidx] == arange, where, 4 casts):
//   t_qkv_fwd [n_pad]   = order[pad[s]]                      padded slot -> point        (ptv3m1:184,188)
//   t_qkv_bwd [2][n]    = (unpad[inverse[p]], dup[inverse[p]])  point -> its slot(s)     (backward of the gather)
//   t_proj_fwd [n]      = unpad[inverse[p]]                   point -> primary slot       (ptv3m1:185,216)
//   t_proj_bwd [n_pad]  = point if slot is that point's primary slot else -1              (backward of the un-gather)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
attn_tables_kernel(const int64_t* __restrict__ order, const int64_t* __restrict__ inverse, const int64_t* __restrict__ pad,
                   const int64_t* __restrict__ unpad, const int64_t* __restrict__ dup, int64_t n, int64_t n_pad,
                   int32_t* __restrict__ t_qkv_fwd, int32_t* __restrict__ t_qkv_bwd, int32_t* __restrict__ t_proj_fwd,
                   int32_t* __restrict__ t_proj_bwd) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t work = n_pad > n ? n_pad : n;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < work; t += stride) {
    if (t < n_pad) {
      const int64_t point = order[pad[t]];
      t_qkv_fwd[t] = (int32_t)point;
      t_proj_bwd[t] = unpad[inverse[point]] == t ? (int32_t)point : -1;
    }
    if (t < n) {
      const int64_t rank = inverse[t];
      const int32_t slot = (int32_t)unpad[rank];
      t_qkv_bwd[t] = slot;
      t_qkv_bwd[n + t] = (int32_t)dup[rank];
      t_proj_fwd[t] = slot;
    }
  }
}

extern "C" int ptc_attn_tables(const int64_t* order, const int64_t* inverse, const int64_t* pad, const int64_t* unpad,
                               const int64_t* dup, int64_t n, int64_t n_pad, int32_t* t_qkv_fwd, int32_t* t_qkv_bwd,
                               int32_t* t_proj_fwd, int32_t* t_proj_bwd, ptc_stream_t stream) {
  PTC_REQUIRE(n >= 0 && n_pad >= n && n_pad < (1ll << 31), PTC_EINVAL, "ptc_attn_tables: bad sizes n=%lld n_pad=%lld", (long long)n,
              (long long)n_pad);
  if (n == 0) return PTC_OK;
  PTC_REQUIRE(order && inverse && pad && unpad && dup && t_qkv_fwd && t_qkv_bwd && t_proj_fwd && t_proj_bwd, PTC_EINVAL,
              "ptc_attn_tables: null buffer");
  int64_t grid = ptc_cdiv(n_pad, 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(attn_tables_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, order, inverse, pad, unpad, dup, n,
                     n_pad, t_qkv_fwd, t_qkv_bwd, t_proj_fwd, t_proj_bwd);
  PTC_CHECK_LAUNCH("attn_tables_kernel");
  return PTC_OK;
}

// ------------------------------------------------------------------------------------------------
// pooling maps
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
pool_flags_kernel(const int64_t* __restrict__ code0, const int64_t* __restrict__ order0, int64_t n, int shift,
                  int32_t* __restrict__ flags) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) {
    const int64_t c = code0[order0[r]] >> shift;
    flags[r] = (r == 0 || (code0[order0[r - 1]] >> shift) != c) ? 1 : 0;
  }
}

__global__ void __launch_bounds__(256)
pool_cluster_kernel(const int64_t* __restrict__ order0, const int32_t* __restrict__ flags,
                    const int64_t* __restrict__ excl, int64_t n, int64_t* __restrict__ cluster,
                    int64_t* __restrict__ n_cluster_out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) {
    const int64_t id = excl[r] + flags[r] - 1;   // inclusive scan - 1
    cluster[order0[r]] = id;
    if (r == n - 1) *n_cluster_out = id + 1;
  }
}

struct PoolLayout { size_t flags, excl, scan, total; };
static PoolLayout pool_layout(int64_t n) {
  PoolLayout L;
  const size_t nn = (size_t)(n > 0 ? n : 1);
  size_t o = 0;
  L.flags = o; o += ptc_align_up(nn * 4, 256);
  L.excl = o; o += ptc_align_up(nn * 8, 256);
  L.scan = o; o += ptc_exclusive_scan_workspace_bytes((int64_t)nn);
  L.total = o;
  return L;
}

extern "C" size_t ptc_pool_maps_workspace_bytes(int64_t n) { return pool_layout(n).total; }

extern "C" int ptc_pool_maps_count(const int64_t* code0, const int64_t* order0, int64_t n, int shift,
                                   int64_t* cluster, int64_t* n_cluster_out, void* workspace,
                                   size_t workspace_bytes, ptc_stream_t stream) {
  PTC_REQUIRE(n >= 1, PTC_EINVAL, "ptc_pool_maps_count: n=%lld", (long long)n);
  PTC_REQUIRE(shift >= 0 && shift < 64, PTC_EINVAL, "ptc_pool_maps_count: shift=%d", shift);
  PTC_REQUIRE(code0 && order0 && cluster && n_cluster_out && workspace, PTC_EINVAL, "ptc_pool_maps_count: null buffer");
  const PoolLayout L = pool_layout(n);
  PTC_REQUIRE(workspace_bytes >= L.total, PTC_EWORKSPACE, "ptc_pool_maps_count: workspace %zu < %zu", workspace_bytes, L.total);
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  int32_t* flags = (int32_t*)(ws + L.flags);
  int64_t* excl = (int64_t*)(ws + L.excl);
  int64_t grid = ptc_cdiv(n, 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(pool_flags_kernel, dim3((unsigned)grid), dim3(256), 0, s, code0, order0, n, shift, flags);
  PTC_CHECK_LAUNCH("pool_flags_kernel");
  int rc = ptc_exclusive_scan_i32(flags, n, excl, ws + L.scan, ptc_exclusive_scan_workspace_bytes(n), stream);
  if (rc != PTC_OK) return rc;
  hipLaunchKernelGGL(pool_cluster_kernel, dim3((unsigned)grid), dim3(256), 0, s, order0, flags, excl, n, cluster, n_cluster_out);
  PTC_CHECK_LAUNCH("pool_cluster_kernel");
  return PTC_OK;
}

__global__ void __launch_bounds__(256)
pool_fill_kernel(const int64_t* __restrict__ order0, const int64_t* __restrict__ cluster, int64_t n,
                 int64_t n_cluster, int64_t* __restrict__ idx_ptr, int64_t* __restrict__ head) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) {
    const int64_t p = order0[r];
    const int64_t c = cluster[p];
    // c < n_cluster: n_cluster may come from the host-side prefetch of the level sizes rather than from the count
    // kernel that numbered `cluster`; a stale value must not become an out-of-bounds write (the maps are then wrong
    // -- caught by the bit-exact tests -- but memory stays intact)
    if ((r == 0 || cluster[order0[r - 1]] != c) && c >= 0 && c < n_cluster) {
      idx_ptr[c] = r;   // ptv3m1:394
      head[c] = p;      // ptv3m1:396 (any member: all share code>>shift, grid_coord>>depth, batch)
    }
    if (r == n - 1) idx_ptr[n_cluster] = n;
  }
}

extern "C" int ptc_pool_maps_fill(const int64_t* order0, const int64_t* cluster, int64_t n, int64_t n_cluster,
                                  int64_t* idx_ptr, int64_t* head, ptc_stream_t stream) {
  PTC_REQUIRE(n >= 1 && n_cluster >= 1 && n_cluster <= n, PTC_EINVAL, "ptc_pool_maps_fill: n=%lld n_cluster=%lld",
              (long long)n, (long long)n_cluster);
  PTC_REQUIRE(order0 && cluster && idx_ptr && head, PTC_EINVAL, "ptc_pool_maps_fill: null buffer");
  int64_t grid = ptc_cdiv(n, 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(pool_fill_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, order0, cluster, n,
                     n_cluster, idx_ptr, head);
  PTC_CHECK_LAUNCH("pool_fill_kernel");
  return PTC_OK;
}

__global__ void __launch_bounds__(256)
pool_child_codes_kernel(const int64_t* __restrict__ code_in, int64_t n, int k, const int64_t* __restrict__ head,
                        int64_t n_cluster, int shift, int64_t* __restrict__ code_out) {
  const int64_t total = n_cluster * k;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t row = t / n_cluster, c = t - row * n_cluster;
    code_out[t] = code_in[row * n + head[c]] >> shift;
  }
}

extern "C" int ptc_pool_child_codes(const int64_t* code_in, int64_t n, int k, const int64_t* head,
                                    int64_t n_cluster, int shift, int64_t* code_out, ptc_stream_t stream) {
  PTC_REQUIRE(n >= 1 && k >= 1 && n_cluster >= 1, PTC_EINVAL, "ptc_pool_child_codes: bad sizes");
  PTC_REQUIRE(code_in && head && code_out, PTC_EINVAL, "ptc_pool_child_codes: null buffer");
  int64_t grid = ptc_cdiv(n_cluster * k, 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(pool_child_codes_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, code_in, n, k,
                     head, n_cluster, shift, code_out);
  PTC_CHECK_LAUNCH("pool_child_codes_kernel");
  return PTC_OK;
}

// ------------------------------------------------------------------------------------------------
// Cluster counts of EVERY pooling level in one pass over the stage-0 codes (one host sync instead of two per level).
// A cluster of level l is a cell of 2^d_l voxels per axis; all serialization curves are hierarchical (Morton bit
// interleave; Hilbert: tests/test_golden_cpu.py::test_hilbert_prefix_property), so `code >> shift_l` names that cell
// for whichever curve sits in row 0, and the number of cells per scene equals the number of changes of code >> shift_l
// along the sorted row.  counts[l][b] = clusters of scene b at level l.  Scenes are contiguous in sorted order, so a
// wave normally sees one scene: one integer atomic per (wave, level); exact and order independent.
// ------------------------------------------------------------------------------------------------
#define PL_MAX_LEVELS 8
struct PoolLevelShifts { int n; int shift[PL_MAX_LEVELS]; };

#define PL_CHUNK 512    // consecutive sorted positions per wave: scenes are contiguous, so a wave flushes ~once per level
__global__ void __launch_bounds__(256)
pool_level_counts_kernel(const int64_t* __restrict__ code0, const int64_t* __restrict__ order0, int64_t n, int batch_shift,
                         int n_batch, PoolLevelShifts lv, unsigned long long* __restrict__ counts) {
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = ptc_lane();
  const int64_t begin = wave * PL_CHUNK;
  if (begin >= n) return;                               // whole wave leaves together
  const int64_t end = begin + PL_CHUNK < n ? begin + PL_CHUNK : n;
  int cur_b = -1;                                       // scene the running counters belong to (wave-uniform)
  unsigned int run[PL_MAX_LEVELS];
#pragma unroll
  for (int l = 0; l < PL_MAX_LEVELS; ++l) run[l] = 0;
  for (int64_t base = begin; base < end; base += 64) {  // trip count is wave-uniform: ballots below are safe
    const int64_t r = base + lane;
    const bool valid = r < end;
    const uint64_t c = valid ? (uint64_t)code0[order0[r]] : 0ull;
    uint64_t prev = __shfl_up(c, 1, 64);                 // the sorted predecessor sits in the previous lane
    if (lane == 0) prev = r > 0 ? (uint64_t)code0[order0[r - 1]] : ~0ull;
    int b = valid ? (int)(c >> batch_shift) : -1;
    if (b >= n_batch) b = n_batch - 1;
    const int b0 = __shfl(b, 0, 64);
    const bool uniform = __all(!valid || b == b0);
    if (!uniform || b0 != cur_b) {                      // scene boundary: flush the running counters
      if (lane == 0 && cur_b >= 0) {
#pragma unroll
        for (int l = 0; l < PL_MAX_LEVELS; ++l)
          if (l < lv.n && run[l]) atomicAdd(&counts[(int64_t)l * n_batch + cur_b], (unsigned long long)run[l]);
      }
#pragma unroll
      for (int l = 0; l < PL_MAX_LEVELS; ++l) run[l] = 0;
      cur_b = uniform ? b0 : -1;
    }
#pragma unroll
    for (int l = 0; l < PL_MAX_LEVELS; ++l) {
      if (l < lv.n) {                                   // wave-uniform condition
        const bool head = valid && (r == 0 || (c >> lv.shift[l]) != (prev >> lv.shift[l]));
        if (uniform) run[l] += (unsigned int)__popcll(__ballot(head));
        else if (head) atomicAdd(&counts[(int64_t)l * n_batch + b], 1ull);
      }
    }
  }
  if (lane == 0 && cur_b >= 0) {
#pragma unroll
    for (int l = 0; l < PL_MAX_LEVELS; ++l)
      if (l < lv.n && run[l]) atomicAdd(&counts[(int64_t)l * n_batch + cur_b], (unsigned long long)run[l]);
  }
}

extern "C" int ptc_pool_level_counts(const int64_t* code0, const int64_t* order0, int64_t n, int batch_shift, int n_batch,
                                     const int* shifts, int n_levels, int64_t* counts, ptc_stream_t stream) {
  PTC_REQUIRE(n >= 0 && n_batch >= 1 && batch_shift >= 0 && batch_shift < 64, PTC_EINVAL, "ptc_pool_level_counts: bad sizes");
  PTC_REQUIRE(n_levels >= 1 && n_levels <= PL_MAX_LEVELS && shifts, PTC_EINVAL, "ptc_pool_level_counts: n_levels=%d not in [1,%d]",
              n_levels, PL_MAX_LEVELS);
  PTC_REQUIRE(counts != nullptr, PTC_EINVAL, "ptc_pool_level_counts: null output");
  hipStream_t s = (hipStream_t)stream;
  PTC_HIP(hipMemsetAsync(counts, 0, sizeof(int64_t) * (size_t)n_levels * (size_t)n_batch, s));
  if (n == 0) return PTC_OK;
  PTC_REQUIRE(code0 && order0, PTC_EINVAL, "ptc_pool_level_counts: null buffer");
  PoolLevelShifts lv;
  lv.n = n_levels;
  for (int l = 0; l < PL_MAX_LEVELS; ++l) {
    lv.shift[l] = l < n_levels ? shifts[l] : 0;
    PTC_REQUIRE(lv.shift[l] >= 0 && lv.shift[l] < 64, PTC_EINVAL, "ptc_pool_level_counts: shift %d", lv.shift[l]);
  }
  const int64_t grid = ptc_cdiv(ptc_cdiv(n, PL_CHUNK), 4);   // 4 waves per workgroup, PL_CHUNK positions per wave
  hipLaunchKernelGGL(pool_level_counts_kernel, dim3((unsigned)grid), dim3(256), 0, s, code0, order0, n, batch_shift, n_batch, lv,
                     (unsigned long long*)counts);
  PTC_CHECK_LAUNCH("pool_level_counts_kernel");
  return PTC_OK;
}
