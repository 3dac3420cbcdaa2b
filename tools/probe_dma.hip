#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const int* src, int* out, int mode) {
  __shared__ int lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = -1;
  __syncthreads();
  const int l = threadIdx.x;
  // lane l fetches the 16 bytes starting at int index 4*perm(l)
  const int p = (l * 7) & 63;
  const int* g = src + 4 * p;
  if (mode == 0) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
  else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(lds + 256), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
  int h[256]; for (int i = 0; i < 256; ++i) h[i] = i;
  int *d, *o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 4096); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, mode); hipDeviceSynchronize();
    int r[1024]; hipMemcpy(r, o, 4096, hipMemcpyDeviceToHost);
    printf("mode %d (dst base = lds+%d ints): lane l fetched ints 4p..4p+3, p=(7l)&63\n", mode, mode ? 256 : 0);
    for (int i = 0; i < 1024; i += 4) if (r[i] != -1) printf("  lds[%4d..] = %d %d %d %d  (p=%d -> lane %d)\n", i, r[i], r[i+1], r[i+2], r[i+3], r[i]/4, ((r[i]/4) * 55) & 63);
  }
  return 0;
}
