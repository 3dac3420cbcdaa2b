// Probe (round 3): issue rate of v_mfma_f32_16x16x32_bf16 from ONE wave per SIMD with the A operand in architectural VGPRs or in
// the accumulation half of the register file (AGPRs), 16 independent accumulators (conv7's tap loop), with and without the
// per-cell filler of that loop (2 VALU + 1 ds_read_b128 per MFMA pair).   hipcc --offload-arch=gfx950 -O3 -o probe probe_mfma_agpr.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define NK 24

template <int MODE>   // bit 0: A fragments pinned to AGPRs; bit 1: + pipelined LDS gathers (conv7's ring: the gathers of step q + 1 are
                      // issued before the MFMAs of step q); bit 2: gather addresses = random rows (else lane-linear, conflict-free)
__global__ void __launch_bounds__(256, 1) k(const s16x8* __restrict__ wsrc, const uint32_t* __restrict__ offs, float* __restrict__ out, long long* __restrict__ cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<uint32_t*>(smem)[i] = i * 2654435761u;
  __syncthreads();
  s16x8 wf[NK][2];
#pragma unroll
  for (int q = 0; q < NK; ++q)
#pragma unroll
    for (int c = 0; c < 2; ++c) wf[q][c] = wsrc[(q * 2 + c) * 64 + lane];
  if (MODE & 1) {
#pragma unroll
    for (int q = 0; q < NK; ++q)
#pragma unroll
      for (int c = 0; c < 2; ++c) asm volatile("" : "+a"(wf[q][c]));
  }
  f32x4 acc[8][2];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  s16x8 bA[8], bB[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) { bA[t] = wsrc[t * 64 + lane]; bB[t] = bA[t]; }
  // per-lane gather offsets: 8 per step, packed 2 per dword as in conv7's table (the table itself lives in LDS)
  uint32_t* tabL = reinterpret_cast<uint32_t*>(smem + 49152);
  const int r = lane & 15, g = lane >> 4;
  for (int q = 0; q < NK; ++q)
    for (int w2 = 0; w2 < 4; ++w2) {
      uint32_t lo, hi;
      if (MODE & 4) {
        const uint32_t h1 = (uint32_t)(q * 131 + w2 * 17 + r * 7919) * 2654435761u, h2 = h1 * 2246822519u + 12345u;
        const uint32_t s1 = (h1 >> 9) % 320u, s2 = (h2 >> 9) % 320u;
        lo = s1 * 128 + ((s1 >> 1) & 7) * 16; hi = s2 * 128 + ((s2 >> 1) & 7) * 16;
      } else {
        lo = (uint32_t)(((2 * w2) * 16 + r) * 128); hi = (uint32_t)(((2 * w2 + 1) * 16 + r) * 128);     // 16 consecutive rows per tile
      }
      if (g == 0 && (threadIdx.x >> 6) == 0) tabL[(q * 16 + r) * 4 + w2] = lo | (hi << 16);
    }
  __syncthreads();
  const uint32_t pxor = (uint32_t)g << 4;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE & 2) {
      auto entries = [&](int q, uint32_t (&te)[4]) {
        const uint4 v = *reinterpret_cast<const uint4*>(tabL + (q * 16 + r) * 4);
        te[0] = v.x; te[1] = v.y; te[2] = v.z; te[3] = v.w;
      };
      auto gather = [&](const uint32_t (&te)[4], s16x8 (&b)[8]) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const uint32_t off = (t & 1) ? (te[t >> 1] >> 16) : (te[t >> 1] & 0xffffu);
          b[t] = *reinterpret_cast<const s16x8*>(smem + (off ^ pxor));
        }
      };
      uint32_t teA[4], teB[4];
      entries(0, teA);
      gather(teA, bA);
      entries(1, teB);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < NK; ++q) {
        s16x8 (&bc)[8] = (q & 1) ? bB : bA;
        s16x8 (&bn)[8] = (q & 1) ? bA : bB;
        uint32_t (&tn)[4] = (q & 1) ? teA : teB;
        uint32_t (&tnn)[4] = (q & 1) ? teB : teA;
        if (q + 1 < NK) gather(tn, bn);
        if (q + 2 < NK) entries(q + 2, tnn);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[q][0], bc[t], acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[q][1], bc[t], acc[t][1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < NK; ++q)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[q][0], bA[t], acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[q][1], bA[t], acc[t][1], 0, 0, 0);
        }
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c) s += acc[t][c][0] + acc[t][c][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char* name, const s16x8* w, const uint32_t* offs, float* out, long long* cyc, int grid) {
  const int iters = 20;
  (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 65536, 0, w, offs, out, cyc, iters);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 65536, 0, w, offs, out, cyc, iters);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  long long h[8]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double n_mfma = (double)iters * NK * 16;
  printf("%-58s grid %4d: %8.1f us | s_memtime ticks per MFMA (wave 0 of WG 0) %.1f | wall ns per MFMA per wave %.2f\n", name, grid, ms * 1e3, h[0] / n_mfma, ms * 1e6 / n_mfma);
}

int main() {
  s16x8* w; uint32_t* offs; float* out; long long* cyc;
  hipMalloc(&w, 64 * 64 * 16); hipMalloc(&offs, 1024); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  hipMemset(w, 0x3c, 64 * 64 * 16); hipMemset(offs, 0x55, 1024);
  for (int grid : {1, 256}) {
    run<0>("A in VGPRs, MFMAs only", w, offs, out, cyc, grid);
    run<1>("A in AGPRs, MFMAs only", w, offs, out, cyc, grid);
    run<3>("A in AGPRs, pipelined gathers, conflict-free rows", w, offs, out, cyc, grid);
    run<7>("A in AGPRs, pipelined gathers, random rows", w, offs, out, cyc, grid);
    run<6>("A in VGPRs, pipelined gathers, random rows", w, offs, out, cyc, grid);
  }
  return 0;
}
