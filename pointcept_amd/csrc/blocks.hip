// blocks.hip -- block-local rulebooks: the 3^3 gather table of a submanifold convolution re-expressed per block of 128
// consecutive output rows as (halo list, local table), the operands of the LDS-staged convolution (conv7.h).
//
// Why: the output-stationary convolution gathers every input row once per table entry that names it -- 9.3 times per voxel for a
// 3^3 window on indoor surfaces.  Rows are kept in curve order (PTC_SORT_POINTS, SpUNet's entry sort), so the DISTINCT input rows
// a block of 128 outputs names (its "halo") are only 1.66 x 128 (measured on the synthetic indoor scenes: mean 212, p99 283, max
// 352; tools/halo_stats.py): stage those ONCE in LDS and gather from LDS.  This file builds what the kernel needs for that:
//   hid  [n_blocks][hcap]              int32 : the distinct input rows of the block, ASCENDING (deterministic; neighbouring output
//                                              rows then read neighbouring LDS slots), padded with the last one to a multiple of 16
//   hcnt [n_blocks]                    int32 : how many; -1 = "does not fit" (rows in no spatial order): the kernel serves such a
//                                              block through the global table
//   tab  [2][n_blocks][28][32][4]      u16   : at [v][b][k][r][t] (r = row in its 32-row tile t) the LDS BYTE OFFSET, inside the kernel's image of the block's
//                                              rows, of piece 0 of row nbr[k][128 b + 32 t + r]: v = 0 for 128-byte rows (64
//                                              channels): slot * 128 + PTC_SWZ64(slot) * 16 (ptc_common.h), v = 1 for 64-byte rows (32
//                                              channels): slot * 64 + ((slot >> 2) & 3) * 16 -- the kernel XORs the piece it
//                                              wants into bits 4.. and adds the image base; "no neighbour" = hcap * row bytes
//                                              (an all-zero row).  Table row 27 (padding: a whole number of 1-KB DMA pieces) carries
//                                              the TAP MASKS in its first 20 bytes: four uint32, bit k of word t set when tile t
//                                              (rows 32 t .. 32 t + 31 of the block) has at least one neighbour at tap k, then
//                                              their OR; the kernel skips the MFMAs of empty (tile | block, tap) pairs -- 46 % of
//                                              the (32-row tile, tap) pairs and 33 % of the (block, tap) pairs on curve-ordered
//                                              indoor scenes (tools/halo_stats.py).  Bytes 20..51 (round 4): eight more uint32, bit k
//                                              of word s set when one of the 16 rows {32 t + 4 s + q : t, q = 0..3} -- an MFMA step
//                                              of the weight-gradient kernel, wgrad7.h -- has a neighbour at tap k.  The rest of
//                                              row 27 holds "no neighbour"
// One workgroup per block: LDS hash set -> compaction -> bitonic sort -> binary search per entry.  Integer work, bit-exact by
// construction: hid[b][tab[0][b][k][r][t] >> 7] == nbr[k][128 b + 32 t + r] wherever nbr >= 0 (tests/test_gpu_kernels.py).
#include "ptc_common.h"

#define BLK_BM 128
#define BLK_NT 8
#define BLK_KV 27
#define BLK_HS 2048     // hash slots (load <= 0.25 at hcap = 512)
#define BLK_LIST 512    // >= hcap
#define BLK_TAB_U16 (28 * 16 * BLK_NT)

__global__ void __launch_bounds__(256)
rulebook_blocks_kernel(const int32_t* __restrict__ nbr, int64_t n, int64_t n_blocks, int hcap, uint16_t* __restrict__ tab, int32_t* __restrict__ hid,
                       int32_t* __restrict__ hcnt, int32_t* __restrict__ n_overflow) {
  __shared__ int keys[BLK_HS];
  __shared__ int list[BLK_LIST];
  __shared__ __attribute__((aligned(16))) uint16_t ltab[2][BLK_TAB_U16];
  __shared__ int cnt, cnt2, ovf;
  __shared__ unsigned tmask[4], smask[8];
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.x;
  const int64_t r0 = b * BLK_BM;
  const int rows = (n - r0) < BLK_BM ? (int)(n - r0) : BLK_BM;
  for (int i = tid; i < BLK_HS; i += 256) keys[i] = -1;
  for (int i = tid; i < BLK_LIST; i += 256) list[i] = 0x7fffffff;
  for (int i = tid; i < BLK_TAB_U16; i += 256) {
    ltab[0][i] = (uint16_t)(hcap * 128);
    ltab[1][i] = (uint16_t)(hcap * 64);
  }
  if (tid == 0) { cnt = 0; cnt2 = 0; ovf = 0; }
  if (tid < 4) tmask[tid] = 0u;
  if (tid < 8) smask[tid] = 0u;
  __syncthreads();
  // the block's 27 x 128 table entries, 14 per thread, fetched ONCE with every load in flight together (the kernel is latency-bound:
  // the first version walked them one dependent load at a time, twice -- 140 us per rulebook at N = 819200, r03_c_conv_pmc_s0.json)
  constexpr int total = BLK_KV * BLK_BM, EPT = (total + 255) / 256;
  int ent[EPT];
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    const int e = tid + 256 * i, k = e / BLK_BM, r = e - k * BLK_BM;
    ent[i] = (e < total && r < rows) ? nbr[(int64_t)k * n + r0 + r] : -1;
  }
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    const int g = ent[i];
    if (g < 0) continue;
    unsigned h = ((unsigned)g * 2654435761u) >> 21;   // 11 bits
    for (int probe = 0; probe < BLK_HS; ++probe) {
      if (*(volatile int*)&ovf) break;
      const int old = atomicCAS(&keys[h], -1, g);
      if (old == -1) {
        if (atomicAdd(&cnt, 1) >= hcap) atomicExch(&ovf, 1);
        break;
      }
      if (old == g) break;
      h = (h + 1) & (BLK_HS - 1);
    }
  }
  __syncthreads();
  if (ovf) {
    if (tid == 0) {
      hcnt[b] = -1;
      atomicAdd(n_overflow, 1);
    }
    return;   // tab / hid of this block stay undefined: the kernel does not read them
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
  // the list, padded with its last row to a multiple of 16 (a DMA instruction of the kernel fetches 8 or 16 whole rows)
  const int cpad = ((c + 15) & ~15) < hcap ? ((c + 15) & ~15) : hcap;
  for (int i = tid; i < cpad; i += 256) hid[b * hcap + i] = list[i < c ? i : c - 1];
  if (tid == 0) hcnt[b] = c;
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    const int g = ent[i];
    if (g < 0) continue;
    const int e = tid + 256 * i, k = e / BLK_BM, r = e - k * BLK_BM;
    int lo = 0, hi = c - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (list[mid] < g) lo = mid + 1; else hi = mid;
    }
    const int at = (k * 32 + (r & 31)) * 4 + (r >> 5);         // [tap][row in 32-row tile][tile]; `lo` = the slot, present by construction
    ltab[0][at] = (uint16_t)(lo * 128 + PTC_SWZ64(lo) * 16);
    ltab[1][at] = (uint16_t)(lo * 64 + ((lo >> 2) & 3) * 16);
    atomicOr(&tmask[r >> 5], 1u << k);
    atomicOr(&smask[(r & 31) >> 2], 1u << k);
  }
  __syncthreads();
  if (tid < 2) {                       // the tap masks, in the padding row of both variants
    uint32_t* mw = reinterpret_cast<uint32_t*>(&ltab[tid][27 * 128]);
    mw[0] = tmask[0]; mw[1] = tmask[1]; mw[2] = tmask[2]; mw[3] = tmask[3];
    mw[4] = tmask[0] | tmask[1] | tmask[2] | tmask[3];
#pragma unroll
    for (int q = 0; q < 8; ++q) mw[5 + q] = smask[q];     // per MFMA step of the weight-gradient kernel (wgrad7.h)
  }
  __syncthreads();
  for (int v = 0; v < 2; ++v) {
    uint16_t* tout = tab + ((int64_t)v * n_blocks + b) * BLK_TAB_U16;
    for (int i = tid; i < BLK_TAB_U16 / 8; i += 256) reinterpret_cast<uint4*>(tout)[i] = reinterpret_cast<const uint4*>(ltab[v])[i];
  }
}

extern "C" size_t ptc_rulebook_blocks_tab_bytes(int64_t n) { return (size_t)2 * ptc_cdiv(n > 0 ? n : 1, BLK_BM) * BLK_TAB_U16 * sizeof(uint16_t); }

extern "C" int ptc_rulebook_blocks(const int32_t* nbr, int kv, int64_t n, int bm, int hcap, void* tab, int32_t* hid, int32_t* hcnt,
                                   int32_t* n_overflow, ptc_stream_t stream) {
  PTC_REQUIRE(n >= 0, PTC_EINVAL, "ptc_rulebook_blocks: bad sizes");
  PTC_REQUIRE(kv == BLK_KV && bm == BLK_BM, PTC_EUNSUPPORTED, "ptc_rulebook_blocks: kv=%d bm=%d (3^3 tables, 128-row blocks)", kv, bm);
  PTC_REQUIRE(hcap >= 16 && hcap < BLK_LIST && hcap % 16 == 0, PTC_EUNSUPPORTED, "ptc_rulebook_blocks: hcap=%d (multiple of 16, <= %d)", hcap,
              BLK_LIST);
  PTC_REQUIRE(n_overflow != nullptr, PTC_EINVAL, "ptc_rulebook_blocks: null counter");
  hipStream_t s = (hipStream_t)stream;
  PTC_HIP(hipMemsetAsync(n_overflow, 0, sizeof(int32_t), s));
  if (n == 0) return PTC_OK;
  PTC_REQUIRE(nbr && tab && hid && hcnt, PTC_EINVAL, "ptc_rulebook_blocks: null buffer");
  PTC_REQUIRE((uintptr_t)tab % 16 == 0, PTC_EINVAL, "ptc_rulebook_blocks: tab must be 16-byte aligned");
  const int64_t nblk = ptc_cdiv(n, BLK_BM);
  hipLaunchKernelGGL(rulebook_blocks_kernel, dim3((unsigned)nblk), dim3(256), 0, s, nbr, n, nblk, hcap, (uint16_t*)tab, hid, hcnt, n_overflow);
  PTC_CHECK_LAUNCH("rulebook_blocks_kernel");
  return PTC_OK;
}
