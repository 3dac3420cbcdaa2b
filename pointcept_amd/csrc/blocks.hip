// blocks.hip -- block-local rulebooks: the gather table of a submanifold convolution re-expressed per block of
// `bm` consecutive output rows as (halo list, local table).
//
// Why (profiles/r02_a_conv_pmc_s0.json): the output-stationary convolution gathers every input row once per table
// entry that names it -- 9.3 times per voxel for a 3^3 window on indoor surfaces -- straight from L1/L2 into MFMA
// operands.  HBM traffic of that kernel is 1.03 x algorithmic, but the texture-address unit is 72 % busy: 5.6 M
// 1-KB wave loads at 16 cycles each ARE the kernel.  Rows are kept in curve order (PTC_SORT_POINTS), so the distinct
// input rows a block of 256 outputs needs (its "halo") are only ~1.5 x 256 (measured 376 +- 40 on the synthetic indoor
// scenes, max 577): stage those ONCE in LDS (one coalesced 128-byte read per row) and gather from LDS at 4x the
// bandwidth of the vector memory path.  This file builds what the kernel needs for that:
//   halo [n_blocks][hmax] int32 : the distinct input rows of the block, ASCENDING (deterministic; neighbouring output
//                                 rows then read neighbouring LDS slots)
//   hcnt [n_blocks]       int32 : how many; hmax + 1 = "does not fit" (rows in no spatial order): such blocks are
//                                 convolved through the global table by the kernel's own fallback loop
//   lnbr [kv][n]          int16 : slot of nbr[k][row] in its block's halo list, -1 = no neighbour
// One workgroup per block: LDS hash set -> compaction -> bitonic sort -> binary search per entry.  Integer work,
// bit-exact by construction: halo[block(row)][lnbr[k][row]] == nbr[k][row] wherever nbr >= 0 (tests/test_gpu_kernels.py).
#include "ptc_common.h"

#define BLK_HS 2048     // hash slots (load <= 0.5 at hmax = 1024)
#define BLK_LIST 1024   // >= hmax

__global__ void __launch_bounds__(256)
rulebook_blocks_kernel(const int32_t* __restrict__ nbr, int kv, int64_t n, int bm, int hmax, int16_t* __restrict__ lnbr,
                       int32_t* __restrict__ halo, int32_t* __restrict__ hcnt, int32_t* __restrict__ n_overflow) {
  __shared__ int keys[BLK_HS];
  __shared__ int list[BLK_LIST];
  __shared__ int cnt, cnt2, ovf;
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.x;
  const int64_t r0 = b * bm;
  const int rows = (n - r0) < bm ? (int)(n - r0) : bm;
  for (int i = tid; i < BLK_HS; i += 256) keys[i] = -1;
  for (int i = tid; i < BLK_LIST; i += 256) list[i] = 0x7fffffff;
  if (tid == 0) { cnt = 0; cnt2 = 0; ovf = 0; }
  __syncthreads();
  const int total = kv * bm;
  for (int e = tid; e < total; e += 256) {
    const int k = e / bm, r = e - k * bm;
    if (r >= rows) continue;
    const int g = nbr[(int64_t)k * n + r0 + r];
    if (g < 0) continue;
    unsigned h = ((unsigned)g * 2654435761u) >> 21;   // 11 bits
    for (int probe = 0; probe < BLK_HS; ++probe) {
      if (*(volatile int*)&ovf) break;
      const int old = atomicCAS(&keys[h], -1, g);
      if (old == -1) {
        if (atomicAdd(&cnt, 1) >= hmax) atomicExch(&ovf, 1);
        break;
      }
      if (old == g) break;
      h = (h + 1) & (BLK_HS - 1);
    }
  }
  __syncthreads();
  if (ovf) {
    if (tid == 0) {
      hcnt[b] = hmax + 1;
      atomicAdd(n_overflow, 1);
    }
    return;
  }
  for (int h = tid; h < BLK_HS; h += 256) {
    const int g = keys[h];
    if (g >= 0) list[atomicAdd(&cnt2, 1)] = g;
  }
  __syncthreads();
  const int c = cnt2;
  int P = 2;
  while (P < c) P <<= 1;
  for (int k2 = 2; k2 <= P; k2 <<= 1)
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < P; i += 256) {
        const int x = i ^ j;
        if (x > i) {
          const int a = list[i], bb = list[x];
          const bool up = (i & k2) == 0;
          if ((a > bb) == up) {
            list[i] = bb;
            list[x] = a;
          }
        }
      }
      __syncthreads();
    }
  for (int i = tid; i < c; i += 256) halo[b * hmax + i] = list[i];
  if (tid == 0) hcnt[b] = c;
  for (int e = tid; e < total; e += 256) {
    const int k = e / bm, r = e - k * bm;
    if (r >= rows) continue;
    const int g = nbr[(int64_t)k * n + r0 + r];
    int slot = -1;
    if (g >= 0) {
      int lo = 0, hi = c - 1;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (list[mid] < g) lo = mid + 1; else hi = mid;
      }
      slot = lo;   // present by construction
    }
    lnbr[(int64_t)k * n + r0 + r] = (int16_t)slot;
  }
}

extern "C" int ptc_rulebook_blocks(const int32_t* nbr, int kv, int64_t n, int bm, int hmax, int16_t* lnbr, int32_t* halo,
                                   int32_t* hcnt, int32_t* n_overflow, ptc_stream_t stream) {
  PTC_REQUIRE(n >= 0 && kv >= 1, PTC_EINVAL, "ptc_rulebook_blocks: bad sizes");
  PTC_REQUIRE(bm >= 16 && bm <= 1024 && hmax >= 1 && hmax <= BLK_LIST, PTC_EUNSUPPORTED, "ptc_rulebook_blocks: bm=%d hmax=%d (hmax <= %d)",
              bm, hmax, BLK_LIST);
  PTC_REQUIRE(n_overflow != nullptr, PTC_EINVAL, "ptc_rulebook_blocks: null counter");
  hipStream_t s = (hipStream_t)stream;
  PTC_HIP(hipMemsetAsync(n_overflow, 0, sizeof(int32_t), s));
  if (n == 0) return PTC_OK;
  PTC_REQUIRE(nbr && lnbr && halo && hcnt, PTC_EINVAL, "ptc_rulebook_blocks: null buffer");
  const int64_t nblk = ptc_cdiv(n, bm);
  hipLaunchKernelGGL(rulebook_blocks_kernel, dim3((unsigned)nblk), dim3(256), 0, s, nbr, kv, n, bm, hmax, lnbr, halo, hcnt, n_overflow);
  PTC_CHECK_LAUNCH("rulebook_blocks_kernel");
  return PTC_OK;
}
