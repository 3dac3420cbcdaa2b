// wgrad7.hip -- the block-staged, accumulator-stationary weight gradient of the 3^3 submanifold convolution (wgrad7.h) as its own
// translation unit: one wave per SIMD on the whole 512-register budget (432 of them accumulators), default MFMA register form (conv7.hip's
// `-amdgpu-mfma-vgpr-form` wants every accumulator in an architectural VGPR; here most of them MUST live in the accumulation half).
#include "ptc_common.h"

#include "mma.h"
#define PTC_CONV7_IMPL          // the DMA / LDS-address helpers and the 32x32x16 MFMA wrappers of conv7.h (its kernel is not instantiated here)
#include "conv7.h"

typedef short w7_s16x4 __attribute__((ext_vector_type(4)));
// one 32x32x16 operand fragment (8 contraction slots) from two transposing reads; every lane supplies the address of ITS row's 8 bytes
template <typename F>
__device__ __forceinline__ F ld_tr_pair16(const unsigned char* p0, const unsigned char* p1) {
  const w7_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) w7_s16x4*)(p0));
  const w7_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) w7_s16x4*)(p1));
  const s16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  F out;
  __builtin_memcpy(&out, &f, sizeof(out));
  return out;
}

#define PTC_WGRAD7_IMPL
#include "wgrad7.h"

int ptc_wgrad7_launch(int dtype, const void* in, const void* dout, const uint16_t* tab, const int32_t* hid, const int32_t* hcnt,
                      const int32_t* gate, int64_t n_out, int c_in, int c_out, float* partial, hipStream_t s) {
  if (wgrad7_sliced(c_in, c_out)) {   // 96 .. 512-channel layers: (32 x 64 | 32)-channel slices of dw on the 64- / 32-channel geometry
    if (wgrad7_slice_cin(c_in) == 64) {
      if (dtype == PTC_BF16) return launch_wgrad7_i<bf16_t, 64, true>(in, dout, tab, hid, hcnt, gate, n_out, c_in, c_out, partial, s);
      return launch_wgrad7_i<f16_t, 64, true>(in, dout, tab, hid, hcnt, gate, n_out, c_in, c_out, partial, s);
    }
    if (dtype == PTC_BF16) return launch_wgrad7_i<bf16_t, 32, true>(in, dout, tab, hid, hcnt, gate, n_out, c_in, c_out, partial, s);
    return launch_wgrad7_i<f16_t, 32, true>(in, dout, tab, hid, hcnt, gate, n_out, c_in, c_out, partial, s);
  }
  if (dtype == PTC_BF16)
    return c_in == 64 ? launch_wgrad7_i<bf16_t, 64>(in, dout, tab, hid, hcnt, gate, n_out, c_in, c_out, partial, s)
                      : launch_wgrad7_i<bf16_t, 32>(in, dout, tab, hid, hcnt, gate, n_out, c_in, c_out, partial, s);
  return c_in == 64 ? launch_wgrad7_i<f16_t, 64>(in, dout, tab, hid, hcnt, gate, n_out, c_in, c_out, partial, s)
                    : launch_wgrad7_i<f16_t, 32>(in, dout, tab, hid, hcnt, gate, n_out, c_in, c_out, partial, s);
}
