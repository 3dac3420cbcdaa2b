// probe_gfx950.hip -- standalone hardware probes used to ground kernel design decisions
// (results recorded in DESIGN.md):
//   1. lane <-> element mapping of ds_read_b64_tr_b16 (the LDS transposing read)
//   2. issue cost, in shader cycles per wave-instruction per SIMD, of the VALU / transcendental /
//      cross-lane / MFMA instructions the attention softmax is built from, at 1, 2 and 4 waves per SIMD.
// Build:  hipcc --offload-arch=gfx950 -O3 -o tools/probe_gfx950 tools/probe_gfx950.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

__global__ void tr_probe(short* out, int stride_elems) {
  __shared__ short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  // lane l reads 4 contiguous elements at row (l>>2), column group (l&3): a [16 rows][16 cols] block
  // with row pitch `stride_elems`
  const int l = threadIdx.x;
  const short* p = lds + (l >> 2) * stride_elems + (l & 3) * 4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  for (int i = 0; i < 4; ++i) out[l * 4 + i] = v[i];
}

// ---- throughput probes: each wave runs ITERS x 16 instances of one instruction on independent registers
#define ITERS 2000
template <int KIND>
__global__ void __launch_bounds__(1024) tp_probe(float* out, uint64_t* cycles, float seed) {
  float r[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = seed + 0.001f * (threadIdx.x + i);
  f32x16 acc16;
  f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 16; ++i) acc16[i] = 0.f;
  s16x8 fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = {8, 7, 6, 5, 4, 3, 2, 1};
  __shared__ float lds[4096];
  lds[threadIdx.x] = seed;
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (KIND == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
      if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[i]));
      if (KIND == 2) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 15]));
      if (KIND == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 15]));
      if (KIND == 4) asm volatile("v_exp_f16 %0, %0" : "+v"(r[i]));
      if (KIND == 5) asm volatile("v_pk_fma_f16 %0, %0, %0, %0" : "+v"(r[i]));
      if (KIND == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 15]));
      if (KIND == 7) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 1) & 15]), "v"(r[(i + 2) & 15]));
      if (KIND == 8) r[i] = __shfl_xor(r[i], 32, 64);
      if (KIND == 9) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(r[i]) : "v"(3));
      if (KIND == 10) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&r[(i & 7) * 2]) : "v"(*(double*)&r[((i + 1) & 7) * 2]));
      if (KIND == 11) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 15]));
    }
    if (KIND == 20) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc16, 0, 0, 0);
    }
    if (KIND == 21) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc4, 0, 0, 0);
    }
    if (KIND == 22) {  // 4 independent accumulators
      f32x4 a0 = acc4, a1 = acc4, a2 = acc4, a3 = acc4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, a3, 0, 0, 0);
      }
      acc4 = a0 + a1 + a2 + a3;
    }
    if (KIND == 23) {  // 2 independent 32x32 accumulators
      f32x16 b0 = acc16, b1 = acc16;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        b0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, b0, 0, 0, 0);
        b1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, b1, 0, 0, 0);
      }
      acc16 = b0 + b1;
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  float s = acc4[0] + acc16[0];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}


// ---- instruction-mix probe: the forward attention trip (4 MFMA 32x32x16 on two dependent pairs, NE
// v_exp_f32, NC v_cvt_pk_bf16_f32), hand-interleaved as in attn_fwd_kernel.  Gives the floor of that mix.
// PRIO: s_setprio 1 on the second-dispatched half of the workgroup's waves (MI355X_MICROARCH.md "two waves per SIMD" item 4);
// PERM: the packs are v_perm_b32 (truncating bf16 pack, full-rate VALU) instead of v_cvt_pk_bf16_f32
template <int NE, int NC, int NM, bool PRIO = false, bool PERM = false>
__global__ void __launch_bounds__(1024) mix_probe(float* out, uint64_t* cycles, float seed) {
  if (PRIO && (threadIdx.x >> 6) >= (blockDim.x >> 7)) __builtin_amdgcn_s_setprio(1);
  float r[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = seed + 0.001f * (threadIdx.x + i);
  f32x16 a0, a1, a2;
#pragma unroll
  for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; a2[i] = 0.f; }
  s16x8 fa = {0x3c00, 0x3c01, 0x3c02, 0x3c03, 0x3c04, 0x3c05, 0x3c06, 0x3c07}, fb = {0x3c08, 0x3c07, 0x3c06, 0x3c05, 0x3c04, 0x3c03, 0x3c02, 0x3c01};
  constexpr int G = NM > 0 ? NM : 4;          // instruction groups per trip
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int m = 0; m < G; ++m) {
      if (m < NM) {
        if (m % 3 == 0) a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, a0, 0, 0, 0);
        else if (m % 3 == 1) a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, a1, 0, 0, 0);
        else a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, a2, 0, 0, 0);
      }
#pragma unroll
      for (int i = (NE * m) / G; i < (NE * (m + 1)) / G; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i & 15]));
#pragma unroll
      for (int i = (NC * m) / G; i < (NC * (m + 1)) / G; ++i) {
        if (PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[i & 15]) : "v"(r[(i + 8) & 15]), "s"(0x07060302));
        else asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[i & 15]) : "v"(r[(i + 8) & 15]));
      }
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  float s = a0[0] + a1[0] + a2[0];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NE, int NC, int NM, bool PRIO = false, bool PERM = false>
static int run_mix(const char* name, float* dout, uint64_t* dcyc) {
  const int waves_per_simd[3] = {1, 2, 4};
  printf("%-34s", name);
  for (int w = 0; w < 3; ++w) {
    const int threads = 64 * 4 * waves_per_simd[w];
    const int blocks = 256;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL((mix_probe<NE, NC, NM, PRIO, PERM>), dim3(blocks), dim3(threads), 0, 0, dout, dcyc, 0.5f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((mix_probe<NE, NC, NM, PRIO, PERM>), dim3(blocks), dim3(threads), 0, 0, dout, dcyc, 0.5f);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    // ns per trip per SIMD (a "trip" = one 32x32 attention tile of one wave)
    printf(" | %dw/SIMD: %7.1f ns/trip/SIMD (%6.3f ms)", waves_per_simd[w], ms * 1e6 / (ITERS * (double)waves_per_simd[w]), ms);
  }
  printf("\n");
  return 0;
}

template <int KIND>
static int run_tp(const char* name, float* dout, uint64_t* dcyc) {
  const int waves_per_simd[3] = {1, 2, 4};
  printf("%-28s", name);
  for (int w = 0; w < 3; ++w) {
    const int threads = 64 * 4 * waves_per_simd[w];  // one workgroup per CU, waves spread over the 4 SIMDs
    const int blocks = 256;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL(tp_probe<KIND>, dim3(blocks), dim3(threads), 0, 0, dout, dcyc, 0.5f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(tp_probe<KIND>, dim3(blocks), dim3(threads), 0, 0, dout, dcyc, 0.5f);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<uint64_t> h(blocks * threads / 64);
    CK(hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost));
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= h.size();
    // per-SIMD cost of one wave-instruction = wave cycles / (instructions per wave * waves sharing the SIMD)
    const double per = mean / (ITERS * 16.0) / waves_per_simd[w];
    printf(" | %dw/SIMD: %7.2f cyc/inst/SIMD (wave %8.0f cyc, %6.3f ms)", waves_per_simd[w], per, mean, ms);
  }
  printf("\n");
  return 0;
}

int main() {
  short* dout;
  CK(hipMalloc(&dout, 64 * 4 * 2));
  const int strides[3] = {16, 32, 64};
  for (int si = 0; si < 3; ++si) {
    hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, dout, strides[si]);
    CK(hipDeviceSynchronize());
    short h[256];
    CK(hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost));
    printf("ds_read_b64_tr_b16: lane l points at row (l>>2), cols 4*(l&3).. of a block with row pitch %d elements; value = row*pitch+col\n", strides[si]);
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d:", l);
      for (int i = 0; i < 4; ++i) printf(" (r%2d,c%2d)", h[l * 4 + i] / strides[si], h[l * 4 + i] % strides[si]);
      if (l % 2 == 1) printf("\n");
    }
  }
  float* fo;
  uint64_t* cyc;
  CK(hipMalloc(&fo, 256 * 1024 * 4));
  CK(hipMalloc(&cyc, 256 * 16 * 8));
  printf("\nissue cost per wave64 instruction per SIMD (shader clock cycles; s_memtime-style counter ticks)\n");
  run_tp<1>("v_fma_f32", fo, cyc);
  run_tp<6>("v_mul_f32", fo, cyc);
  run_tp<2>("v_max_f32", fo, cyc);
  run_tp<7>("v_max3_f32", fo, cyc);
  run_tp<0>("v_exp_f32", fo, cyc);
  run_tp<4>("v_exp_f16", fo, cyc);
  run_tp<9>("v_ldexp_f32", fo, cyc);
  run_tp<3>("v_cvt_pk_bf16_f32", fo, cyc);
  run_tp<11>("v_cvt_pkrtz_f16_f32", fo, cyc);
  run_tp<5>("v_pk_fma_f16", fo, cyc);
  run_tp<10>("v_pk_mul_f32", fo, cyc);
  run_tp<8>("__shfl_xor(.,32) f32", fo, cyc);
  run_tp<20>("mfma 32x32x16 bf16 (dep)", fo, cyc);
  run_tp<23>("mfma 32x32x16 bf16 (2 acc)", fo, cyc);
  run_tp<21>("mfma 16x16x32 bf16 (dep)", fo, cyc);
  run_tp<22>("mfma 16x16x32 bf16 (4 acc)", fo, cyc);
  printf("\ninstruction-mix floors (ns per trip per SIMD; 32 cycles at 2.4 GHz = 13.3 ns)\n");
  run_mix<0, 0, 4>("4 mfma", fo, cyc);
  run_mix<16, 0, 0>("16 exp", fo, cyc);
  run_mix<16, 8, 0>("16 exp + 8 cvt", fo, cyc);
  run_mix<16, 8, 4>("4 mfma + 16 exp + 8 cvt (fwd)", fo, cyc);
  run_mix<16, 8, 4, true, false>("  same, s_setprio 1 on 2nd half", fo, cyc);
  run_mix<16, 8, 4, false, true>("  same, packs = v_perm_b32", fo, cyc);
  run_mix<16, 8, 3, true, true>("3 mfma + 16 exp + 8 perm + prio", fo, cyc);
  run_mix<8, 8, 4>("4 mfma + 8 exp + 8 cvt", fo, cyc);
  run_mix<16, 8, 3>("3 mfma + 16 exp + 8 cvt", fo, cyc);
  run_mix<0, 8, 4>("4 mfma + 8 cvt", fo, cyc);
  run_mix<0, 16, 4>("4 mfma + 16 plain", fo, cyc);
  run_mix<0, 32, 4>("4 mfma + 32 plain", fo, cyc);
  run_mix<16, 24, 5>("5 mfma + 16 exp + 24 plain (dQ)", fo, cyc);
  run_mix<16, 48, 9>("9 mfma + 16 exp + 48 plain (dKV)", fo, cyc);
  run_mix<16, 40, 11>("11 mfma + 16 exp + 40 plain (1pass)", fo, cyc);
  run_mix<8, 8, 2>("2 mfma + 8 exp + 8 cvt", fo, cyc);
  int clk = 0;
  CK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
  printf("device clock rate attribute: %d kHz\n", clk);
  return 0;
}
