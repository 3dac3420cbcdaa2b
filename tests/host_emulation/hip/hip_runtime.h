// TEST INFRASTRUCTURE.  A host stand-in for <hip/hip_runtime.h>: the kernel sources of pointcept_amd/csrc are compiled by the host
// clang++ and executed on the CPU, so that their index arithmetic, LDS staging, cross-lane exchanges, MFMA tilings and dtype
// handling can be checked against the oracles / goldens WITHOUT a GPU (tests/test_host_emulation_cpu.py).
//
// Execution model.  A launch runs its workgroups one after the other; inside a workgroup every thread ("lane") is a FIBER with its
// own stack (kernel locals = the lane's registers).  A lane runs until it reaches a collective:
//     __syncthreads()                                   -> waits for every live lane of the workgroup
//     __shfl*, __ballot, MFMA, ds_read_b64_tr_b16       -> exchange through a per-wave buffer between two waits for every live
//                                                          lane of the 64-lane wave
// and the scheduler resumes the others.  Collectives must therefore be reached in WAVE-UNIFORM control flow (as the MFMA and
// full-EXEC cross-lane instructions require on the hardware anyway); a lane that finishes while its wave waits in a collective is
// reported as a deadlock.  What this models: everything that is a function of program order.  What it does not: timing, memory
// ordering between waves (fences are no-ops: one lane runs at a time), the rounding ORDER inside an MFMA (fp32 sums, k ascending).
//
// The sources are compiled unmodified except for one mechanical token substitution done by the build script
// (tests/emu_backend.py): `extern __shared__` -> `extern`, every other `__shared__` -> `static` (one workgroup at a time, so a
// function-static array IS the workgroup's LDS).
//
// Matrix layouts (gfx950; the kernels' own comments and tools/probe_gfx950.hip state the same):
//   mfma_f32_16x16x4_f32    a, b scalar: A[i][k] in lane i + 16 k, B[k][j] in lane j + 16 k;   D[i][j]: lane j + 16 (i / 4), element i % 4
//   mfma_f32_16x16x32_*16   a, b 8 x 16 bit: A row i, k-group g in lane i + 16 g (g = 0..3); B column j likewise;   D as above
//   mfma_f32_32x32x16_bf16  a, b 8 x 16 bit: A row i, k-group g in lane i + 32 g (g = 0..1); B likewise;
//                           D[i][j] 16 per lane: lane j + 32 h, element r: i = 8 (r / 4) + 4 h + r % 4
//   (the contraction pairs element e of k-group g of A with element e of k-group g of B: the k order inside is immaterial)
//   ds_read_b64_tr_b16      per 16-lane group: lane l supplies the address of 4 consecutive 16-bit values; lane j receives element
//                           j % 4 of the lanes 4 r + j / 4, r = 0..3  (a [4 rows][16 columns] block read column-wise)
//   raw buffer load b128    per dword: in range iff offset + 4 <= num_records, else 0
#pragma once
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <functional>
#include <vector>

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

typedef void* hipStream_t;
typedef int hipError_t;
static const hipError_t hipSuccess = 0;
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8, hipMemcpyHostToDevice = 1 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "host emulation"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n); return *p ? hipSuccess : (hipError_t)2; }      /* conv8.h's weight scratch */
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }

// ------------------------------------------------------------------------------------------------------------------------
// fibers
// ------------------------------------------------------------------------------------------------------------------------
namespace emu {
enum { EMU_MAX_THREADS = 1024, EMU_STACK = 96 * 1024, EMU_LDS = 192 * 1024, EMU_SLOT = 160 };
struct Lane {
  void* sp;
  char* stack;
  dim3 tid;
  int wave, lane;
  int state;   // 0 runnable, 1 waiting for its wave, 2 waiting for the workgroup, 3 finished
};
struct Wave {
  alignas(16) unsigned char x[64][EMU_SLOT];
  unsigned gen;            // number of wave collectives released so far
  unsigned stamp[64];      // `gen` at which the lane last entered a collective: who takes part in the current one
};
struct State {
  Lane lanes[EMU_MAX_THREADS];
  Wave waves[EMU_MAX_THREADS / 64];
  int n_threads = 0;
  Lane* cur = nullptr;
  void* sched_sp = nullptr;
  dim3 block_idx, block_dim, grid_dim;
  const std::function<void()>* body = nullptr;
  // PTC_EMU_STATS=1: per launch, the number of lane-level MFMA / transposing-LDS-read / buffer-load executions
  unsigned long long n_mfma = 0, n_tr = 0, n_buf = 0, n_buf_oob = 0, n_shfl = 0;
};
inline State& S() { static State s; return s; }
}  // namespace emu

extern "C" void emu_switch(void** save_sp, void* load_sp);
#ifdef EMU_IMPLEMENTATION
// callee-saved registers of the SysV x86-64 ABI on the old stack, then the stack pointers are exchanged
__asm__(
    ".text\n.globl emu_switch\n.type emu_switch,@function\nemu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n");
#endif

extern unsigned char smem[];
namespace emu {
inline void yield_to_scheduler() { State& s = S(); emu_switch(&s.cur->sp, s.sched_sp); }
inline void wave_sync() { S().cur->state = 1; yield_to_scheduler(); }
inline void block_sync() { S().cur->state = 2; yield_to_scheduler(); }
inline void trampoline() {
  State& s = S();
  (*s.body)();
  s.cur->state = 3;
  yield_to_scheduler();
  abort();   // a finished lane is never resumed
}
inline void run_block() {
  State& s = S();
  const int n = s.n_threads, n_waves = (n + 63) / 64;
  for (int t = 0; t < n; ++t) {
    Lane& L = s.lanes[t];
    if (!L.stack) L.stack = (char*)aligned_alloc(64, EMU_STACK);
    L.tid = dim3((unsigned)(t % s.block_dim.x), (unsigned)((t / s.block_dim.x) % s.block_dim.y), (unsigned)(t / (s.block_dim.x * s.block_dim.y)));
    L.wave = t / 64;
    L.lane = t % 64;
    L.state = 0;
    // initial frame: six callee-saved registers, then the address emu_switch's `ret` jumps to; after that `ret` rsp % 16 == 8,
    // as at any function entry
    uintptr_t top = ((uintptr_t)L.stack + EMU_STACK) & ~(uintptr_t)15;
    void** f = (void**)(top - 16);
    f[0] = (void*)&trampoline;
    for (int r = 1; r <= 6; ++r) f[-r] = nullptr;
    L.sp = (void*)(f - 6);
  }
  for (int w = 0; w < n_waves; ++w) memset(s.waves[w].stamp, 0xff, sizeof(s.waves[w].stamp));
  int finished = 0;
  while (finished < n) {
    bool progress = false;
    for (int t = 0; t < n; ++t) {
      Lane& L = s.lanes[t];
      if (L.state != 0) continue;
      s.cur = &L;
      emu_switch(&s.sched_sp, L.sp);
      progress = true;
      if (L.state == 3) ++finished;
    }
    // release the collectives that are complete
    for (int w = 0; w < n_waves; ++w) {
      int waiting = 0, live = 0;
      for (int l = 0; l < 64 && w * 64 + l < n; ++l) {
        const int st = s.lanes[w * 64 + l].state;
        live += st != 3;
        waiting += st == 1;
      }
      // complete when every live lane of the wave is in the collective or parked at the workgroup barrier (a diverged wave: the
      // lanes that skipped the branch wait for the others at the barrier, as the hardware's reconvergence does)
      int parked = 0;
      for (int l = 0; l < 64 && w * 64 + l < n; ++l) parked += s.lanes[w * 64 + l].state == 2;
      if (waiting && waiting + parked == live) {
        for (int l = 0; l < 64 && w * 64 + l < n; ++l)
          if (s.lanes[w * 64 + l].state == 1) s.lanes[w * 64 + l].state = 0;
        ++s.waves[w].gen;
        progress = true;
      } else if (waiting && !progress) {
        bool any_runnable = false;
        for (int t = 0; t < n; ++t) any_runnable |= s.lanes[t].state == 0;
        if (!any_runnable) {
          fprintf(stderr, "host emulation: wave %d of workgroup (%u,%u,%u): %d of %d live lanes wait in a wave collective -- "
                          "divergent collective or early exit\n", w, s.block_idx.x, s.block_idx.y, s.block_idx.z, waiting, live);
          abort();
        }
      }
    }
    {
      int waiting = 0, live = 0;
      for (int t = 0; t < n; ++t) { live += s.lanes[t].state != 3; waiting += s.lanes[t].state == 2; }
      if (waiting && waiting == live) {
        for (int t = 0; t < n; ++t) if (s.lanes[t].state == 2) s.lanes[t].state = 0;
        progress = true;
      }
    }
    if (!progress && finished < n) {
      fprintf(stderr, "host emulation: workgroup (%u,%u,%u) is stuck (barrier not reached by every live lane)\n", s.block_idx.x,
              s.block_idx.y, s.block_idx.z);
      abort();
    }
  }
}
inline void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body, const char* what = "") {
  State& s = S();
  const bool stats = getenv("PTC_EMU_STATS") != nullptr;
  s.n_mfma = s.n_tr = s.n_buf = s.n_buf_oob = s.n_shfl = 0;
  const size_t n = (size_t)block.x * block.y * block.z;
  if (n == 0 || n > EMU_MAX_THREADS || shmem > EMU_LDS) { fprintf(stderr, "host emulation: launch shape not supported\n"); abort(); }
  s.n_threads = (int)n;
  s.block_dim = block;
  s.grid_dim = grid;
  s.body = &body;
  // canary behind the dynamic LDS the launch asked for: a workgroup that writes past its allocation is reported (on the hardware
  // that is a memory fault or a silent corruption of the neighbouring workgroup's LDS)
  const size_t guard = shmem + 4096 <= (size_t)EMU_LDS ? 4096 : (size_t)EMU_LDS - shmem;
  for (unsigned z = 0; z < grid.z; ++z)
    for (unsigned y = 0; y < grid.y; ++y)
      for (unsigned x = 0; x < grid.x; ++x) {
        s.block_idx = dim3(x, y, z);
        memset(smem + shmem, 0xC3, guard);
        run_block();
        for (size_t i = 0; i < guard; ++i)
          if (smem[shmem + i] != 0xC3) {
            fprintf(stderr, "host emulation: workgroup (%u,%u,%u) wrote dynamic LDS byte %zu, %zu were allocated\n", x, y, z, shmem + i, shmem);
            abort();
          }
      }
  if (stats)
    fprintf(stderr, "[emu] %-48.48s grid %u x %u x %u, %u threads, LDS %zu B | wave-level: MFMA %llu  ds_read_tr %llu  buffer loads %llu "
                    "(%.0f %% of their dwords out of range)  shuffles/ballots %llu\n", what, grid.x, grid.y, grid.z, block.x * block.y * block.z,
            shmem, s.n_mfma / 64, s.n_tr / 64, s.n_buf / 64, s.n_buf ? 25.0 * s.n_buf_oob / s.n_buf : 0.0, s.n_shfl / 64);
}
// per-lane exchange slot of the current wave
inline unsigned char* slot(int lane) { State& s = S(); return s.waves[s.cur->wave].x[lane]; }
inline int lane_id() { return S().cur->lane; }
}  // namespace emu

#define threadIdx (emu::S().cur->tid)
#define blockIdx (emu::S().block_idx)
#define blockDim (emu::S().block_dim)
#define gridDim (emu::S().grid_dim)
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  emu::launch(dim3(grid), dim3(block), (size_t)(shmem), std::function<void()>([&]() { kernel(__VA_ARGS__); }), #kernel)
#define __syncthreads() emu::block_sync()
#define __threadfence() ((void)0)
#define __threadfence_block() ((void)0)

// dynamic LDS (`extern __shared__ ... smem[]` / `lh[]` after the build script's token substitution)
#ifdef EMU_IMPLEMENTATION
alignas(64) unsigned char smem[emu::EMU_LDS + 4096];
alignas(64) unsigned int lh[emu::EMU_LDS / 4];
#endif

// ------------------------------------------------------------------------------------------------------------------------
// vector types, atomics, scalar intrinsics
// ------------------------------------------------------------------------------------------------------------------------
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(8) int2 { int x, y; };
struct alignas(8) float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
static inline int4 make_int4(int a, int b, int c, int d) { return int4{a, b, c, d}; }
static inline int2 make_int2(int a, int b) { return int2{a, b}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
// one lane runs at a time: an atomic is its plain operation
template <typename T, typename U> static inline T atomicAdd(T* p, U v) { T o = *p; *p = o + (T)v; return o; }
template <typename T, typename U> static inline T atomicMin(T* p, U v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <typename T, typename U> static inline T atomicMax(T* p, U v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <typename T, typename U> static inline T atomicOr(T* p, U v) { T o = *p; *p = o | (T)v; return o; }
template <typename T, typename U, typename V> static inline T atomicCAS(T* p, U expect, V v) { T o = *p; if (o == (T)expect) *p = (T)v; return o; }
template <typename T, typename U> static inline T atomicExch(T* p, U v) { T o = *p; *p = (T)v; return o; }

static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
#define __expf(x) expf(x)      /* (glibc declares functions of these names: macros, defined after <math.h>) */
#define __logf(x) logf(x)
// HIP's device-side min / max overloads
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline long min(long a, long b) { return a < b ? a : b; }
static inline long max(long a, long b) { return a > b ? a : b; }
static inline long long min(long long a, long long b) { return a < b ? a : b; }
static inline long long max(long long a, long long b) { return a > b ? a : b; }
static inline unsigned long min(unsigned long a, unsigned long b) { return a < b ? a : b; }
static inline unsigned long max(unsigned long a, unsigned long b) { return a > b ? a : b; }
static inline unsigned long long min(unsigned long long a, unsigned long long b) { return a < b ? a : b; }
static inline unsigned long long max(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline float __fdividef(float a, float b) { return a / b; }
static inline long long __float2ll_rn(float x) { return llrintf(x); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_wave_barrier() emu::wave_sync()   /* the source says: the wave meets here */
#define __builtin_amdgcn_sched_group_barrier(...) ((void)0)
#define __builtin_amdgcn_sched_barrier(...) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_amdgcn_s_barrier() emu::block_sync()
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)           /* one lane runs at a time: every memory operation has completed */
static inline long long clock64() { return 0; }            /* timing probes (conv7 C7_ABLATE & 64) have no meaning here */
#define __builtin_amdgcn_readfirstlane(x) (x)              /* callers pass wave-uniform values */
// global_load_lds: `sz` bytes per lane, global -> LDS at (wave-uniform base) + sz * lane.  Lands at once here; on the hardware it
// lands asynchronously (vmcnt) -- the emulation cannot see a missing wait.
static inline void emu_global_load_lds(const void* g, void* l, int sz) { memcpy((char*)l + emu::lane_id() * sz, g, (size_t)sz); }
#define __builtin_amdgcn_global_load_lds(g, l, sz, off, aux) emu_global_load_lds((const void*)(g), (void*)(l), (int)(sz))

// v_perm_b32: byte select from {a (bytes 7..4), b (bytes 3..0)}
static inline uint32_t emu_perm(uint32_t a, uint32_t b, uint32_t sel) {
  const uint64_t src = ((uint64_t)a << 32) | b;
  uint32_t out = 0;
  for (int i = 0; i < 4; ++i) {
    const unsigned s = (sel >> (8 * i)) & 0xff;
    unsigned byte;
    if (s <= 7) byte = (unsigned)(src >> (8 * s)) & 0xff;
    else if (s == 0x0c) byte = 0x00;
    else if (s >= 0x0d) byte = 0xff;
    else { const unsigned sign_of = s == 8 ? 1 : s == 9 ? 3 : s == 10 ? 5 : 7; byte = ((src >> (8 * sign_of + 7)) & 1) ? 0xff : 0x00; }
    out |= byte << (8 * i);
  }
  return out;
}
#define __builtin_amdgcn_perm(a, b, sel) emu_perm((uint32_t)(a), (uint32_t)(b), (uint32_t)(sel))

// ------------------------------------------------------------------------------------------------------------------------
// cross-lane operations (wave collectives)
// ------------------------------------------------------------------------------------------------------------------------
namespace emu {
inline Wave& my_wave() { State& s = S(); return s.waves[s.cur->wave]; }
inline unsigned enter() { Wave& w = my_wave(); w.stamp[lane_id()] = w.gen; return w.gen; }      // -> id of this collective
inline bool takes_part(int lane, unsigned id) { return my_wave().stamp[lane & 63] == id; }
template <typename T> inline T exchange(T v, int src_lane) {   // every lane publishes v, then reads lane src_lane's value
  static_assert(sizeof(T) <= EMU_SLOT, "exchange slot too small");
  memcpy(slot(lane_id()), &v, sizeof(T));
  ++S().n_shfl;
  const unsigned id = enter();
  wave_sync();
  T r = v;                                                     // a source lane outside the collective: undefined on the hardware
  if (takes_part(src_lane, id)) memcpy(&r, slot(src_lane & 63), sizeof(T));
  wave_sync();
  return r;
}
inline unsigned long long ballot(bool p) {
  unsigned char b = p ? 1 : 0;
  memcpy(slot(lane_id()), &b, 1);
  const unsigned id = enter();
  wave_sync();
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l)
    if (takes_part(l, id) && slot(l)[0]) m |= 1ull << l;
  wave_sync();
  return m;
}
}  // namespace emu
template <typename T> static inline T __shfl_xor(T v, int mask, int = 64) { return emu::exchange(v, emu::lane_id() ^ mask); }
template <typename T> static inline T __shfl(T v, int src, int = 64) { return emu::exchange(v, src); }
template <typename T> static inline T __shfl_up(T v, int d, int = 64) { const int l = emu::lane_id(); return emu::exchange(v, l >= d ? l - d : l); }
template <typename T> static inline T __shfl_down(T v, int d, int = 64) { const int l = emu::lane_id(); return emu::exchange(v, l + d < 64 ? l + d : l); }
static inline unsigned long long __ballot(int p) { return emu::ballot(p != 0); }
static inline int __any(int p) { return emu::ballot(p != 0) != 0; }
static inline int __all(int p) { return emu::ballot(p == 0) == 0; }
namespace emu {
inline int syncthreads_or(int p) {      // workgroup barrier that also ORs a predicate
  static unsigned char flags[EMU_MAX_THREADS];
  State& s = S();
  const int t = (int)(s.cur - s.lanes);
  flags[t] = p != 0;
  block_sync();
  int any = 0;
  for (int i = 0; i < s.n_threads; ++i) any |= (s.lanes[i].state != 3) && flags[i];
  block_sync();
  return any;
}
}  // namespace emu
#define __syncthreads_or(p) emu::syncthreads_or((p))
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
#define __builtin_amdgcn_ballot_w64(p) emu::ballot((p))

// ---- MFMA ------------------------------------------------------------------------------------------------------------
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
namespace emu {
inline float bf16_bits_to_float(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }
inline float f16_bits_to_float(uint16_t b) { _Float16 h; memcpy(&h, &b, 2); return (float)h; }
// M x M output tile (M = 16 or 32), G k-groups of E elements per lane; cvt: operand element -> float
template <int M, int G, int E, typename EL, typename ACC, typename CVT>
inline ACC mfma(const EL* a, const EL* b, ACC c, CVT cvt) {
  const int l = lane_id();
  unsigned char* me = slot(l);
  memcpy(me, a, sizeof(EL) * E);
  memcpy(me + 32, b, sizeof(EL) * E);
  ++S().n_mfma;
  enter();
  wave_sync();
  constexpr int NOUT = M * M / 64;
  const int j = l % M, h = l / M;
  for (int r = 0; r < NOUT; ++r) {
    const int i = (M == 32) ? 8 * (r / 4) + 4 * h + r % 4 : 4 * h + r;
    float acc = c[r];
    for (int g = 0; g < G; ++g) {
      EL av[E], bv[E];
      memcpy(av, slot(i + M * g), sizeof(EL) * E);
      memcpy(bv, slot(j + M * g) + 32, sizeof(EL) * E);
      for (int e = 0; e < E; ++e) acc += cvt(av[e]) * cvt(bv[e]);
    }
    c[r] = acc;
  }
  wave_sync();
  return c;
}
}  // namespace emu
template <typename V> static inline emu_f32x4 emu_mfma_16x16x32_bf16(V a, V b, emu_f32x4 c) {
  return emu::mfma<16, 4, 8, uint16_t>((const uint16_t*)&a, (const uint16_t*)&b, c, emu::bf16_bits_to_float);
}
template <typename V> static inline emu_f32x4 emu_mfma_16x16x32_f16(V a, V b, emu_f32x4 c) {
  return emu::mfma<16, 4, 8, uint16_t>((const uint16_t*)&a, (const uint16_t*)&b, c, emu::f16_bits_to_float);
}
static inline emu_f32x4 emu_mfma_16x16x4_f32(float a, float b, emu_f32x4 c) {
  return emu::mfma<16, 4, 1, float>(&a, &b, c, [](float x) { return x; });
}
template <typename V, typename ACC> static inline ACC emu_mfma_32x32x16_bf16(V a, V b, ACC c) {
  return emu::mfma<32, 2, 8, uint16_t>((const uint16_t*)&a, (const uint16_t*)&b, c, emu::bf16_bits_to_float);
}
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) emu_mfma_16x16x32_bf16((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z) emu_mfma_16x16x32_f16((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) emu_mfma_16x16x4_f32((a), (b), (c))
template <typename V, typename ACC> static inline ACC emu_mfma_32x32x16_f16(V a, V b, ACC c) {
  return emu::mfma<32, 2, 8, uint16_t>((const uint16_t*)&a, (const uint16_t*)&b, c, emu::f16_bits_to_float);
}
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) emu_mfma_32x32x16_bf16((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) emu_mfma_32x32x16_f16((a), (b), (c))

// ---- ds_read_b64_tr_b16 --------------------------------------------------------------------------------------------------
typedef short emu_s16x4 __attribute__((ext_vector_type(4)));
template <typename P> static inline emu_s16x4 emu_ds_read_tr16_b64(P p) {
  const int l = emu::lane_id(), grp = l & ~15, j = l & 15;
  const void* addr = (const void*)p;
  memcpy(emu::slot(l), &addr, sizeof(addr));
  ++emu::S().n_tr;
  emu::enter();
  emu::wave_sync();
  emu_s16x4 out;
  for (int r = 0; r < 4; ++r) {
    const void* src;
    memcpy(&src, emu::slot(grp + 4 * r + j / 4), sizeof(src));
    short v;
    memcpy(&v, (const char*)src + 2 * (j % 4), 2);
    out[r] = v;
  }
  emu::wave_sync();
  return out;
}
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) emu_ds_read_tr16_b64((const void*)(uintptr_t)(p))

// ---- raw buffer loads ------------------------------------------------------------------------------------------------------
struct __amdgpu_buffer_rsrc_t { const char* base; uint32_t num_records; };
typedef int emu_i32x4 __attribute__((ext_vector_type(4)));
static inline __amdgpu_buffer_rsrc_t emu_make_rsrc(void* p, int, int num, int) { return __amdgpu_buffer_rsrc_t{(const char*)p, (uint32_t)num}; }
static inline emu_i32x4 emu_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, int voff, int soff, int) {
  const uint32_t off = (uint32_t)voff + (uint32_t)soff;
  emu_i32x4 v = {0, 0, 0, 0};
  ++emu::S().n_buf;
  for (int d = 0; d < 4; ++d) {
    const uint64_t o = (uint64_t)off + 4u * d;
    if (o + 4 <= r.num_records) { int w; memcpy(&w, r.base + o, 4); v[d] = w; }
    else ++emu::S().n_buf_oob;
  }
  return v;
}
static inline int emu_raw_buffer_load_b32(__amdgpu_buffer_rsrc_t r, int voff, int soff, int) {
  const uint64_t o = (uint64_t)((uint32_t)voff + (uint32_t)soff);
  int w = 0;
  ++emu::S().n_buf;
  if (o + 4 <= r.num_records) memcpy(&w, r.base + o, 4);
  else ++emu::S().n_buf_oob;
  return w;
}
#define __builtin_amdgcn_raw_buffer_load_b32(r, v, s, a) emu_raw_buffer_load_b32((r), (v), (s), (a))
#define __builtin_amdgcn_make_buffer_rsrc(p, s, n, f) emu_make_rsrc((p), (s), (n), (f))
#define __builtin_amdgcn_raw_buffer_load_b128(r, v, s, a) emu_raw_buffer_load_b128((r), (v), (s), (a))
