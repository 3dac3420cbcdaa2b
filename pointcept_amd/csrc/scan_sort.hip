// scan_sort.hip -- device-wide exclusive scan and stable LSD radix sort of the serialization keys.
//
// Replaces torch.argsort(code) + the inverse-permutation scatter at
//   pointcept/models/utils/structure.py:93-100 and
//   pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py:399-406.
// Keys carry only 3*depth + bits(batch) significant bits (27 for ScanNet, 39 outdoors), so the
// sort runs ceil(bits/8) passes instead of 8.  Every pass is HBM-bound:
//   histogram (read keys) -> scan of [row][digit][block] counters -> stable scatter (read+write).
// Stability (canonical tie order = ascending original index) comes from
// wave-level match-any ranking with 64-bit ballots; no atomics touch the output order.
#include "ptc_common.h"

// ------------------------------------------------------------------------------------------------
// wave / block scan primitives (64-lane waves)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t wave_inclusive_scan_i64(int64_t v) {
  const int lane = ptc_lane();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int lo = __shfl_up((int)(uint32_t)(v & 0xffffffffll), d, 64);
    int hi = __shfl_up((int)(uint32_t)((uint64_t)v >> 32), d, 64);
    int64_t o = (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
    if (lane >= d) v += o;
  }
  return v;
}

#define SCAN_THREADS 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_THREADS * SCAN_ITEMS)

// block-wide exclusive scan of one int64 per thread; returns exclusive prefix, *total = block sum
__device__ __forceinline__ int64_t block_exclusive_scan_i64(int64_t v, int64_t* total, int64_t* smem /*[5]*/) {
  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  int64_t inc = wave_inclusive_scan_i64(v);
  if (lane == 63) smem[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t run = 0;
    for (int w = 0; w < SCAN_THREADS / 64; ++w) { int64_t t = smem[w]; smem[w] = run; run += t; }
    smem[4] = run;
  }
  __syncthreads();
  int64_t excl = inc - v + smem[wave];
  *total = smem[4];
  __syncthreads();
  return excl;
}

__global__ void __launch_bounds__(SCAN_THREADS)
scan_tile_sums_kernel(const int32_t* __restrict__ in, int64_t n, int64_t* __restrict__ tile_sums) {
  __shared__ int64_t smem[5];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int64_t s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) { int64_t i = base + j; if (i < n) s += in[i]; }
  int64_t total;
  block_exclusive_scan_i64(s, &total, smem);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// single block: exclusive scan of tile_sums in place (chunks of SCAN_THREADS)
__global__ void __launch_bounds__(SCAN_THREADS)
scan_tile_sums_scan_kernel(int64_t* __restrict__ tile_sums, int64_t n_tiles) {
  __shared__ int64_t smem[5];
  int64_t carry = 0;
  for (int64_t c = 0; c < n_tiles; c += SCAN_THREADS) {
    int64_t i = c + threadIdx.x;
    int64_t v = (i < n_tiles) ? tile_sums[i] : 0;
    int64_t total;
    int64_t ex = block_exclusive_scan_i64(v, &total, smem);
    if (i < n_tiles) tile_sums[i] = carry + ex;
    carry += total;
  }
}

__global__ void __launch_bounds__(SCAN_THREADS)
scan_apply_kernel(const int32_t* __restrict__ in, int64_t n, const int64_t* __restrict__ tile_offsets,
                  int64_t* __restrict__ out) {
  __shared__ int64_t smem[5];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int32_t v[SCAN_ITEMS];
  int64_t s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) { int64_t i = base + j; v[j] = (i < n) ? in[i] : 0; s += v[j]; }
  int64_t total;
  int64_t ex = block_exclusive_scan_i64(s, &total, smem) + tile_offsets[blockIdx.x];
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) { int64_t i = base + j; if (i < n) out[i] = ex; ex += v[j]; }
}

// n <= SCAN1_MAX: the whole scan in ONE workgroup of 1024 threads (16 items per thread and sweep, a carry across sweeps) -- one launch
// instead of three (the digit histograms of the deep stages' sorts, the pad / pooling maps); integer arithmetic: the result is the same
// whatever the grouping.
#define SCAN1_THREADS 1024
#define SCAN1_ITEMS 16
#define SCAN1_MAX (SCAN1_THREADS * SCAN1_ITEMS)   // one sweep: measured, a 4-sweep scan (25 us) is slower than the three launches it replaces (23 us)
__global__ void __launch_bounds__(SCAN1_THREADS)
scan_single_kernel(const int32_t* __restrict__ in, int64_t n, int64_t* __restrict__ out) {
  __shared__ int64_t wsum[SCAN1_THREADS / 64], wex[SCAN1_THREADS / 64 + 1];
  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  int64_t carry = 0;
  for (int64_t sweep = 0; sweep < n; sweep += (int64_t)SCAN1_THREADS * SCAN1_ITEMS) {
    const int64_t base = sweep + (int64_t)threadIdx.x * SCAN1_ITEMS;
    int32_t v[SCAN1_ITEMS];
    int64_t s = 0;
#pragma unroll
    for (int j = 0; j < SCAN1_ITEMS; ++j) { const int64_t i = base + j; v[j] = (i < n) ? in[i] : 0; s += v[j]; }
    const int64_t inc = wave_inclusive_scan_i64(s);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    if (wave == 0) {
      const int64_t w = lane < SCAN1_THREADS / 64 ? wsum[lane] : 0;
      const int64_t winc = wave_inclusive_scan_i64(w);
      if (lane < SCAN1_THREADS / 64) wex[lane] = winc - w;
      if (lane == SCAN1_THREADS / 64 - 1) wex[SCAN1_THREADS / 64] = winc;
    }
    __syncthreads();
    int64_t ex = carry + wex[wave] + inc - s;
#pragma unroll
    for (int j = 0; j < SCAN1_ITEMS; ++j) { const int64_t i = base + j; if (i < n) out[i] = ex; ex += v[j]; }
    carry += wex[SCAN1_THREADS / 64];
    __syncthreads();
  }
}

static int64_t scan_num_tiles(int64_t n) { return ptc_cdiv(n > 0 ? n : 1, SCAN_TILE); }

extern "C" size_t ptc_exclusive_scan_workspace_bytes(int64_t n) {
  return ptc_align_up((size_t)scan_num_tiles(n) * sizeof(int64_t), 256);
}

// internal: scan on stream with caller-provided workspace
static int exclusive_scan_i32(const int32_t* in, int64_t n, int64_t* out, void* ws, hipStream_t s) {
  if (n <= 0) return PTC_OK;
  if (n <= SCAN1_MAX) {
    hipLaunchKernelGGL(scan_single_kernel, dim3(1), dim3(SCAN1_THREADS), 0, s, in, n, out);
    PTC_CHECK_LAUNCH("scan_single_kernel");
    return PTC_OK;
  }
  int64_t tiles = scan_num_tiles(n);
  int64_t* tile_sums = (int64_t*)ws;
  hipLaunchKernelGGL(scan_tile_sums_kernel, dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, s, in, n, tile_sums);
  PTC_CHECK_LAUNCH("scan_tile_sums_kernel");
  hipLaunchKernelGGL(scan_tile_sums_scan_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, tile_sums, tiles);
  PTC_CHECK_LAUNCH("scan_tile_sums_scan_kernel");
  hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, s, in, n, tile_sums, out);
  PTC_CHECK_LAUNCH("scan_apply_kernel");
  return PTC_OK;
}

extern "C" int ptc_exclusive_scan_i32(const int32_t* in, int64_t n, int64_t* out, void* workspace,
                                      size_t workspace_bytes, ptc_stream_t stream) {
  PTC_REQUIRE(n >= 0, PTC_EINVAL, "ptc_exclusive_scan_i32: n < 0");
  if (n == 0) return PTC_OK;
  PTC_REQUIRE(in && out && workspace, PTC_EINVAL, "ptc_exclusive_scan_i32: null buffer");
  PTC_REQUIRE(workspace_bytes >= ptc_exclusive_scan_workspace_bytes(n), PTC_EWORKSPACE,
              "ptc_exclusive_scan_i32: workspace %zu < %zu", workspace_bytes, ptc_exclusive_scan_workspace_bytes(n));
  return exclusive_scan_i32(in, n, out, workspace, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// radix sort
// ------------------------------------------------------------------------------------------------
#define RS_THREADS 256
#define RS_WAVES (RS_THREADS / 64)
#define RS_ITERS 16                       // 64-element strips per wave
#define RS_TILE (RS_THREADS * RS_ITERS)   // 4096 keys per block
#define RS_RADIX 256
#ifndef RS_PACK
#define RS_PACK 1                           // 0: key and row index as separate arrays in every pass (timing A/B)
#endif

__device__ __forceinline__ uint32_t rs_digit(uint64_t key, int shift, uint32_t mask) {
  return (uint32_t)(key >> shift) & mask;
}

// hist[(row*256 + digit) * n_blocks + block]
__global__ void __launch_bounds__(RS_THREADS)
rs_histogram_kernel(const uint64_t* __restrict__ keys, int64_t n, int shift, uint32_t mask,
                    int n_blocks, int32_t* __restrict__ hist) {
  __shared__ int32_t lh[RS_RADIX];
  const int row = blockIdx.y, blk = blockIdx.x;
  lh[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t* k = keys + (int64_t)row * n;
  const int64_t base = (int64_t)blk * RS_TILE;
#pragma unroll 4
  for (int it = 0; it < RS_ITERS; ++it) {
    int64_t i = base + (int64_t)it * RS_THREADS + threadIdx.x;
    if (i < n) atomicAdd(&lh[rs_digit(k[i], shift, mask)], 1);
  }
  __syncthreads();
  hist[((int64_t)row * RS_RADIX + threadIdx.x) * n_blocks + blk] = lh[threadIdx.x];
}

// Stable scatter.  Element order inside a tile: wave w owns strips [w*RS_ITERS, (w+1)*RS_ITERS),
// strip s covers 64 consecutive keys, lane = position in strip.
// PACK (round 4): the row index rides in the low `idx_bits` bits of the key word itself -- word = (key bits [0, end_bit) << idx_bits) | row,
// built by the first pass in registers, sorted by its key bits only (`shift` then counts from bit idx_bits) -- so a pass moves 8 bytes
// per element instead of 12 and the tile needs no value array in LDS (39 KB instead of 55 KB per workgroup).  Used whenever
// end_bit + ceil(log2 n) <= 64: every sort of the model (28 + 20 bits at the bench size).  Same stable order, bit for bit.
template <bool FIRST, bool PACK>
__global__ void __launch_bounds__(RS_THREADS)
rs_scatter_kernel(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ idx_in, int64_t n,
                  int shift, uint32_t mask, int n_blocks, const int64_t* __restrict__ offsets,
                  uint64_t* __restrict__ keys_out, uint32_t* __restrict__ idx_out, int idx_bits, uint64_t key_mask) {
  __shared__ int32_t wave_hist[RS_WAVES][RS_RADIX];
  __shared__ int64_t glob[RS_RADIX];
  __shared__ uint64_t skeys[RS_TILE];      // the tile in digit order (32 KB + 16 KB)
  __shared__ uint32_t svals[PACK ? 1 : RS_TILE];
  const int row = blockIdx.y, blk = blockIdx.x;
  const int lane = ptc_lane(), wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < RS_WAVES * RS_RADIX; i += RS_THREADS) (&wave_hist[0][0])[i] = 0;
  // global offset of (row, digit, block), relative to the start of this row
  glob[threadIdx.x] = offsets[((int64_t)row * RS_RADIX + threadIdx.x) * n_blocks + blk] - (int64_t)row * n;
  __syncthreads();

  const uint64_t* k = keys_in + (int64_t)row * n;
  const uint32_t* ix = (FIRST || PACK) ? nullptr : idx_in + (int64_t)row * n;
  const int dshift = shift + ((PACK && FIRST) ? idx_bits : 0);    // FIRST packs in registers: the digit is then read from the packed word
  const int64_t base = (int64_t)blk * RS_TILE + (int64_t)wave * (RS_ITERS * 64);
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

  uint64_t key[RS_ITERS];
  uint32_t val[RS_ITERS];
  int32_t rank[RS_ITERS];
#pragma unroll
  for (int it = 0; it < RS_ITERS; ++it) {
    const int64_t i = base + (int64_t)it * 64 + lane;
    const bool valid = i < n;
    key[it] = valid ? k[i] : ~0ull;
    if (PACK && FIRST && valid) key[it] = ((key[it] & key_mask) << idx_bits) | (uint64_t)i;
    val[it] = (valid && !PACK) ? (FIRST ? (uint32_t)i : ix[i]) : 0u;
    const uint32_t d = rs_digit(key[it], dshift, mask);
    // match-any over the wave: lanes holding the same digit
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const uint64_t bal = __ballot(bit);
      peers &= bit ? bal : ~bal;
    }
    int32_t before = 0;
    if (valid) {
      before = wave_hist[wave][d];                       // count from earlier strips of this wave
      rank[it] = before + __popcll(peers & lt_mask);
    } else {
      rank[it] = 0;
    }
    // the lowest peer lane publishes the new running count (one writer per digit per wave) -- after EVERY lane of the wave has read
    // the old one: the wave barrier states the lockstep this step relies on (no instruction; it pins the order for the compiler)
    __builtin_amdgcn_wave_barrier();
    if (valid && (peers & lt_mask) == 0ull) wave_hist[wave][d] = before + __popcll(peers);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  // exclusive prefix over waves, per digit (thread t <-> digit t); `run` ends as this block's count of digit t
  int32_t run = 0;
#pragma unroll
  for (int w = 0; w < RS_WAVES; ++w) { int32_t c = wave_hist[w][threadIdx.x]; wave_hist[w][threadIdx.x] = run; run += c; }
  // Block-local exclusive scan of the digit counts -> start of every digit's run inside the tile.  The elements are first put
  // in digit order in LDS and then written out by position: a digit's elements of this tile are consecutive in the output, so
  // consecutive threads write consecutive addresses (the direct form wrote 8 + 4 bytes per lane to 256 scattered runs: 1.4 TB/s
  // on the low-digit passes of the serialization sort, profiles/r02_ag_bench_kernel_stats.csv).
  __shared__ int32_t lstart[RS_RADIX];
  __shared__ int32_t wsum[RS_WAVES];
  {
    int32_t incl = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t v = __shfl_up(incl, o, 64);
      if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int32_t base_w = 0;
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) base_w += (w < wave) ? wsum[w] : 0;
    lstart[threadIdx.x] = base_w + incl - run;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < RS_ITERS; ++it) {
    const int64_t i = base + (int64_t)it * 64 + lane;
    if (i < n) {
      const uint32_t d = rs_digit(key[it], dshift, mask);
      const int lp = lstart[d] + wave_hist[wave][d] + rank[it];
      skeys[lp] = key[it];
      if (!PACK) svals[lp] = val[it];
    }
  }
  __syncthreads();
  uint64_t* ko = keys_out + (int64_t)row * n;
  uint32_t* io = PACK ? nullptr : idx_out + (int64_t)row * n;
  const int64_t tile0 = (int64_t)blk * RS_TILE;
  const int cnt = (int)((n - tile0) < RS_TILE ? (n - tile0) : RS_TILE);
  for (int j = threadIdx.x; j < cnt; j += RS_THREADS) {
    const uint64_t kk = skeys[j];
    const uint32_t d = rs_digit(kk, dshift, mask);
    const int64_t dst = glob[d] + (j - lstart[d]);
    ko[dst] = kk;
    if (!PACK) io[dst] = svals[j];
  }
}

// `sorted_keys` (optional): the key words in sorted order -- bits [0, end_bit) of the caller's keys, i.e. INCLUDING whatever the caller
// keeps below begin_bit (a payload the sort carries along: the Lovasz loss rides its foreground flag there, lovasz.hip).  `plain` = the
// sorted key words of the unpacked path (or the input itself when the bit range is empty).
__global__ void __launch_bounds__(256)
rs_finalize_kernel(const uint32_t* __restrict__ idx, const uint64_t* __restrict__ packed, uint64_t idx_mask, int idx_bits, int64_t n, int k,
                   int64_t* __restrict__ order, int64_t* __restrict__ inverse, const uint64_t* plain,
                   int64_t* sorted_keys) {            // (`plain` and `sorted_keys` may be the same array: empty bit range, in-place caller)
  const int64_t total = n * k;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t row = t / n, i = t - row * n;
    const uint64_t word = packed ? packed[t] : 0ull;
    const int64_t src = packed ? (int64_t)(word & idx_mask) : (idx ? (int64_t)idx[t] : i);
    order[t] = src;
    if (inverse) inverse[row * n + src] = i;
    if (sorted_keys) sorted_keys[t] = (int64_t)(packed ? (word >> idx_bits) : plain[t]);
  }
}

static int rs_num_blocks(int64_t n) { return (int)ptc_cdiv(n > 0 ? n : 1, RS_TILE); }

struct RsLayout {
  size_t keysA, keysB, idxA, idxB, hist, offs, scan, total;
};
static RsLayout rs_layout(int64_t n, int k) {
  RsLayout L;
  const size_t nk = (size_t)(n > 0 ? n : 1) * (size_t)k;
  const size_t nh = (size_t)k * RS_RADIX * (size_t)rs_num_blocks(n);
  size_t o = 0;
  L.keysA = o; o += ptc_align_up(nk * 8, 256);
  L.keysB = o; o += ptc_align_up(nk * 8, 256);
  L.idxA = o; o += ptc_align_up(nk * 4, 256);
  L.idxB = o; o += ptc_align_up(nk * 4, 256);
  L.hist = o; o += ptc_align_up(nh * 4, 256);
  L.offs = o; o += ptc_align_up(nh * 8, 256);
  L.scan = o; o += ptc_exclusive_scan_workspace_bytes((int64_t)nh);
  L.total = o;
  return L;
}

extern "C" size_t ptc_sort_keys_workspace_bytes(int64_t n, int k) { return rs_layout(n, k).total; }

extern "C" int ptc_sort_keys(const int64_t* keys, int64_t n, int k, int begin_bit, int end_bit,
                             int64_t* order, int64_t* inverse, void* workspace, size_t workspace_bytes,
                             ptc_stream_t stream) {
  return ptc_sort_keys_ex(keys, n, k, begin_bit, end_bit, order, inverse, nullptr, workspace, workspace_bytes, stream);
}

// ptc_sort_keys + the sorted key words themselves (library-internal: lovasz.hip; `sorted_keys` may alias `keys`, which is read by the
// first pass only -- and by the last kernel at the same index it writes when the bit range is empty)
int ptc_sort_keys_ex(const int64_t* keys, int64_t n, int k, int begin_bit, int end_bit, int64_t* order, int64_t* inverse,
                     int64_t* sorted_keys, void* workspace, size_t workspace_bytes, ptc_stream_t stream) {
  PTC_REQUIRE(n >= 0 && k >= 1, PTC_EINVAL, "ptc_sort_keys: bad n=%lld k=%d", (long long)n, k);
  PTC_REQUIRE(begin_bit >= 0 && end_bit <= 64 && begin_bit <= end_bit, PTC_EINVAL,
              "ptc_sort_keys: bad bit range [%d,%d)", begin_bit, end_bit);
  PTC_REQUIRE(n < (1ll << 32), PTC_EUNSUPPORTED, "ptc_sort_keys: n >= 2^32");
  if (n == 0) return PTC_OK;
  PTC_REQUIRE(keys && order && workspace, PTC_EINVAL, "ptc_sort_keys: null buffer");
  const RsLayout L = rs_layout(n, k);
  PTC_REQUIRE(workspace_bytes >= L.total, PTC_EWORKSPACE, "ptc_sort_keys: workspace %zu < %zu", workspace_bytes, L.total);
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  uint64_t* kbuf[2] = {(uint64_t*)(ws + L.keysA), (uint64_t*)(ws + L.keysB)};
  uint32_t* ibuf[2] = {(uint32_t*)(ws + L.idxA), (uint32_t*)(ws + L.idxB)};
  int32_t* hist = (int32_t*)(ws + L.hist);
  int64_t* offs = (int64_t*)(ws + L.offs);
  void* scan_ws = ws + L.scan;
  const int nb = rs_num_blocks(n);
  const int64_t nh = (int64_t)k * RS_RADIX * nb;

  const uint64_t* cur_k = (const uint64_t*)keys;
  const uint32_t* cur_i = nullptr;
  int out = 0;
  bool first = true;
  // the row index packed into the key word (see rs_scatter_kernel) when both fit into 64 bits
  int idx_bits = 1;
  while (((int64_t)1 << idx_bits) < n) ++idx_bits;
  const bool pack = RS_PACK && (end_bit - begin_bit) > 0 && end_bit + idx_bits <= 64;
  const uint64_t key_mask = end_bit >= 64 ? ~0ull : (((uint64_t)1 << end_bit) - 1);
  for (int shift = begin_bit; shift < end_bit; shift += 8) {
    const int bits = (end_bit - shift) < 8 ? (end_bit - shift) : 8;
    const uint32_t mask = (1u << bits) - 1u;
    const int hshift = shift + ((pack && !first) ? idx_bits : 0);      // later passes read packed words
    hipLaunchKernelGGL(rs_histogram_kernel, dim3(nb, k), dim3(RS_THREADS), 0, s, cur_k, n, hshift, mask, nb, hist);
    PTC_CHECK_LAUNCH("rs_histogram_kernel");
    int rc = exclusive_scan_i32(hist, nh, offs, scan_ws, s);
    if (rc != PTC_OK) return rc;
#define RS_SCATTER(F, P)                                                                                                                          \
    hipLaunchKernelGGL((rs_scatter_kernel<F, P>), dim3(nb, k), dim3(RS_THREADS), 0, s, cur_k, cur_i, n, (P && !F) ? shift + idx_bits : shift, mask, nb, \
                       offs, kbuf[out], ibuf[out], idx_bits, key_mask)
    if (pack) { if (first) RS_SCATTER(true, true); else RS_SCATTER(false, true); }
    else { if (first) RS_SCATTER(true, false); else RS_SCATTER(false, false); }
#undef RS_SCATTER
    PTC_CHECK_LAUNCH("rs_scatter_kernel");
    cur_k = kbuf[out];
    cur_i = ibuf[out];
    out ^= 1;
    first = false;
  }
  {
    const int64_t total = n * k;
    int64_t grid = ptc_cdiv(total, 256);
    if (grid > 4096) grid = 4096;
    const bool packed_out = pack && !first;
    hipLaunchKernelGGL(rs_finalize_kernel, dim3((unsigned)grid), dim3(256), 0, s, packed_out ? nullptr : cur_i, packed_out ? cur_k : nullptr,
                       (((uint64_t)1 << idx_bits) - 1), idx_bits, n, k, order, inverse, cur_k, sorted_keys);
    PTC_CHECK_LAUNCH("rs_finalize_kernel");
  }
  return PTC_OK;
}
