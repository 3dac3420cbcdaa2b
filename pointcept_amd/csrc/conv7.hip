// conv7.hip -- the block-staged 3^3 submanifold convolution (conv7.h) as its own translation unit.
//
// Why its own file: the kernel runs ONE wave per SIMD on the whole 512-register budget, and the compiler's default for MFMA results at
// that occupancy is the accumulation half of the register file -- every block then pays 64 v_accvgpr_read (the epilogue works on
// architectural VGPRs) and 64 v_accvgpr_write (the reset).  `-mllvm -amdgpu-mfma-vgpr-form` (build.py, PER_FILE_FLAGS) keeps the
// accumulators in architectural VGPRs -- there is room: 220 of 256 -- and leaves the AGPRs to the pinned weight fragments:
// 60.4 -> 57.3 us at 32 -> 32, N = 819200 (consistently), 148 -> 142..146 us at 64 -> 64 (inside that kernel's run-to-run noise; profiles/r03_t_conv7_vgprform.txt).  The flag is a whole-translation-unit
// switch and neutral-to-unknown for the 256-register kernels of spconv.hip, so it is confined to this file.
#include "ptc_common.h"

#include "mma.h"
#define PTC_CONV7_IMPL
#include "conv7.h"

int ptc_conv7_launch(int dtype, const void* in, int64_t n_in, const void* w, const float* bias, const uint16_t* tab, const int32_t* hid,
                     const int32_t* hcnt, int64_t n_out, int c, void* out, hipStream_t s) {
  if (dtype == PTC_BF16)
    return c == 64 ? launch_conv7_i<bf16_t, 64>(in, n_in, w, bias, tab, hid, hcnt, n_out, out, s)
                   : launch_conv7_i<bf16_t, 32>(in, n_in, w, bias, tab, hid, hcnt, n_out, out, s);
  return c == 64 ? launch_conv7_i<f16_t, 64>(in, n_in, w, bias, tab, hid, hcnt, n_out, out, s)
                 : launch_conv7_i<f16_t, 32>(in, n_in, w, bias, tab, hid, hcnt, n_out, out, s);
}
