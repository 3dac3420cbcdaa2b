// TEST INFRASTRUCTURE.  A host stand-in for <hip/hip_runtime.h> that lets ELEMENTWISE kernels of pointcept_amd/csrc (no LDS, no wave
// intrinsics, no MFMA) be compiled by the host clang++ and executed thread by thread on the CPU, so that their index arithmetic and
// dtype handling can be checked against the oracle / goldens without a GPU (tests/test_host_emulation_cpu.py).  The kernel source is
// compiled UNMODIFIED: `__global__` functions become plain functions, hipLaunchKernelGGL loops over (blockIdx.x, threadIdx.x) with
// thread-local index variables.  One-dimensional launches only; anything else aborts loudly.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
static thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

typedef void* hipStream_t;
typedef int hipError_t;
static const hipError_t hipSuccess = 0;
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "host emulation"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }

static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                              \
  do {                                                                                           \
    const dim3 g__ = (grid), b__ = (block);                                                      \
    if (g__.y != 1 || g__.z != 1 || b__.y != 1 || b__.z != 1 || (shmem) != 0) {                 \
      fprintf(stderr, "host emulation: only 1-D launches without LDS\n");                        \
      abort();                                                                                   \
    }                                                                                            \
    gridDim = g__; blockDim = b__;                                                               \
    for (unsigned bx__ = 0; bx__ < g__.x; ++bx__)                                                \
      for (unsigned tx__ = 0; tx__ < b__.x; ++tx__) {                                            \
        blockIdx = dim3(bx__); threadIdx = dim3(tx__);                                           \
        kernel(__VA_ARGS__);                                                                     \
      }                                                                                          \
  } while (0)
