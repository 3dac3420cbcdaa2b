// probe_gather.hip -- what does a 1-KB wave gather cost on gfx950's vector-memory path, by lane -> address mapping?
//   map A: lane l -> row (l & 15), 16-byte piece (l >> 4)   (the MFMA B-operand layout: a row's 64 B in lanes r, r+16, r+32, r+48)
//   map B: lane l -> row (l >> 2), piece (l & 3)           (a quad of adjacent lanes = one 64-byte segment)
//   map C: lane l -> row (l >> 3), piece (l & 7)           (8 adjacent lanes = one 128-byte row; 8 rows per instruction)
// Rows are 128 B (64 bf16 channels), indices come from a table (like the convolution's gather table), the working
// set is `window` rows (L1- / L2- / HBM-resident).  Every wave issues ITER instructions with UNR independent loads in
// flight.  Reports wave-instructions per microsecond per CU and the implied cycles per instruction at the measured clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MAP, bool BUF>
__global__ void __launch_bounds__(256) gather_kernel(const unsigned char* __restrict__ data, const int* __restrict__ table, int window_rows,
                                                     int iters, unsigned bytes, unsigned* __restrict__ sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gw = blockIdx.x * 4 + wave;
  int rsel, piece, rows_per;
  if (MAP == 0) { rsel = lane & 15; piece = lane >> 4; rows_per = 16; }
  else if (MAP == 1) { rsel = lane >> 2; piece = lane & 3; rows_per = 16; }
  else { rsel = lane >> 3; piece = lane & 7; rows_per = 8; }
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(data), 0, (int)bytes, 0x00020000);
  unsigned acc = 0;
  const int* tb = table + (size_t)gw * 64;
  for (int it = 0; it < iters; it += 4) {
    i32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = tb[((it + u) * rows_per + rsel) & 63];
      const int r2 = (row + (it + u) * 97 + gw * 13) & (window_rows - 1);     // windows are powers of two
      const unsigned off = (unsigned)r2 * 128u + (unsigned)piece * 16u;
      if (BUF) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0);
      else v[u] = *reinterpret_cast<const i32x4*>(data + off);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc ^= (unsigned)(v[u][0] + v[u][1] + v[u][2] + v[u][3]);
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int MAP, bool BUF>
static void run(const char* name, const unsigned char* data, const int* table, int window_rows, unsigned bytes, unsigned* sink) {
  const int blocks = 256 * 8, iters = 512;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL((gather_kernel<MAP, BUF>), dim3(blocks), dim3(256), 0, 0, data, table, window_rows, iters, bytes, sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((gather_kernel<MAP, BUF>), dim3(blocks), dim3(256), 0, 0, data, table, window_rows, iters, bytes, sink);
  CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
  const double instr = (double)blocks * 4 * iters;              // wave instructions
  const double per_cu_per_us = instr / 256 / (ms * 1e3);
  printf("%-34s window %8d rows (%7.1f MB): %8.3f ms  %6.2f wave-loads/us/CU = %5.1f ns each per CU  -> %6.1f cycles @2.1GHz, %7.1f GB/s aggregate\n", name,
         window_rows, window_rows * 128.0 / 1e6, ms, per_cu_per_us, 1e3 / per_cu_per_us, 2100.0 / per_cu_per_us, instr * 1024 / (ms * 1e-3) / 1e9);
}

int main() {
  const size_t rows = 1 << 20;   // 128 MB
  unsigned char* data; int* table; unsigned* sink;
  CHECK(hipMalloc(&data, rows * 128)); CHECK(hipMemset(data, 1, rows * 128));
  const int nw = 256 * 8 * 4;
  std::vector<int> t((size_t)nw * 64);
  srand(1);
  for (size_t i = 0; i < t.size(); ++i) t[i] = rand() & 0xfffff;
  CHECK(hipMalloc(&table, t.size() * 4)); CHECK(hipMemcpy(table, t.data(), t.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&sink, 4));
  const unsigned bytes = (unsigned)(rows * 128 - 1);
  for (int window : {128, 16384, 262144, 1 << 20}) {
    run<0, true>("map A (row = l&15) buffer_load", data, table, window, bytes, sink);
    run<1, true>("map B (row = l>>2) buffer_load", data, table, window, bytes, sink);
    run<2, true>("map C (row = l>>3) buffer_load", data, table, window, bytes, sink);
    run<0, false>("map A (row = l&15) global_load", data, table, window, bytes, sink);
    run<1, false>("map B (row = l>>2) global_load", data, table, window, bytes, sink);
  }
  return 0;
}
