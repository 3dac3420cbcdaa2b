// rope.hip -- 3-axis rotary position embedding on point tokens, in place (LitePT / PT-v3m3 "PointROPE").
//
// Replaces libs/pointrope/kernels.cu:19-100 (pointrope_cuda_kernel + launcher) behind the operator
// pointrope.pointrope(tokens [B,N,H,D], positions [B,N,3] int64, base, F0) of libs/pointrope/pointrope.cpp:51-67
// (call sites: pointcept/models/litept/litept_v1.py:27-59, applied to q and k at :240-241).
//   D = 6 Q: one head = [u_x (Q) | v_x (Q) | u_y | v_y | u_z | v_z];  for axis a, i < Q:
//     f = pos[a] * (F0 / base^(i/Q));   u' = u cos f - v sin f;   v' = v cos f + u sin f        (fp32 math)
//   backward = the same call with -F0 (the rotation is orthogonal).
// The reference runs one block per token with D threads and a shared-memory copy of the token; here one thread owns
// one (token, axis, i) pair, computes sin / cos once and walks the H heads (the pairs of a token are adjacent lanes:
// a wave covers 64 / (D/2) whole tokens, contiguous in memory).  HBM-bound: 2 x T H D e bytes + 24 T.
#include "ptc_common.h"

template <typename T>
__global__ void __launch_bounds__(256)
rope3d_kernel(T* __restrict__ tok, const int64_t* __restrict__ pos, int64_t n_tok, int H, int D, float base, float fwd) {
  const int Q = D / 6, P = 3 * Q;                      // pairs per head
  const int64_t total = n_tok * P;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t t = e / P;
    const int pr = (int)(e - t * P), a = pr / Q, i = pr - a * Q;
    const float inv_freq = fwd / powf(base, (float)i / (float)Q);          // kernels.cu:44
    const float f = (float)pos[t * 3 + a] * inv_freq;                      // kernels.cu:54
    float sn, cs;
    sincosf(f, &sn, &cs);
    T* p = tok + t * (int64_t)H * D + a * 2 * Q + i;
    for (int h = 0; h < H; ++h, p += D) {
      const float u = ptc_to_float(p[0]), v = ptc_to_float(p[Q]);
      p[0] = ptc_from_float<T>(u * cs - v * sn);
      p[Q] = ptc_from_float<T>(v * cs + u * sn);
    }
  }
}

extern "C" int ptc_rope3d(void* tokens, int dtype, const int64_t* positions, int64_t n_tokens, int H, int D, float base, float fwd,
                          ptc_stream_t stream) {
  PTC_REQUIRE(n_tokens >= 0 && H >= 1, PTC_EINVAL, "ptc_rope3d: bad sizes");
  PTC_REQUIRE(D >= 6 && D % 6 == 0, PTC_EUNSUPPORTED, "ptc_rope3d: token dim %d must be a multiple of 6", D);   // kernels.cu:87
  PTC_REQUIRE(base > 0.f, PTC_EINVAL, "ptc_rope3d: base must be positive");
  if (n_tokens == 0) return PTC_OK;
  PTC_REQUIRE(tokens && positions, PTC_EINVAL, "ptc_rope3d: null buffer");
  const int64_t total = n_tokens * (D / 2);
  int64_t grid = ptc_cdiv(total, 256);
  if (grid > 65536) grid = 65536;
  hipStream_t s = (hipStream_t)stream;
  PTC_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(rope3d_kernel<T>, dim3((unsigned)grid), dim3(256), 0, s, (T*)tokens, positions, n_tokens, H, D,
                                                  base, fwd));
  PTC_CHECK_LAUNCH("rope3d_kernel");
  return PTC_OK;
}

// ---- PT-v3m3 `Point3DRoPE` (point_transformer_v3m3_utonia.py:43-102, applied to q and k at :274-305) ------------------------------
// Same rotation as above (a head is three chunks of D/3, each chunk = [first half | second half], rotate_half pairs element i with
// i + Q; inv_freq[i] = 1 / base^(2i / (D/3)) = 1 / base^(i/Q)), but the positions are the CONTINUOUS point coordinates (fp32, after the
// training-time shift / jitter / rescale) and the frequencies come from the module's `inv_freq` buffer (it lives in the state dict).
// The reference computes in fp32 from the qkv the Linear produced and rounds the stacked [q', k', v] to bf16 for flash-attn
// (:319-323); this kernel does the same in ONE pass over the packed rows:
//   src [n, S, H, D] of TI  ->  dst [n, S, H, D] of TO;  slabs 0 .. R-1 (q, k) rotated, slabs R .. S-1 (v) converted / copied.
//   src == dst with TI == TO: in place, the untouched slabs are skipped.
// One thread owns one (row, axis, i): sin / cos once, then the R*H heads to rotate and the (S-R)*H heads to copy.
// Backward = the same call with sign = -1 on the incoming gradient (the rotation is orthogonal).
template <typename TI, typename TO>
__global__ void __launch_bounds__(256)
rope3d_xyz_kernel(const TI* src, TO* dst /* may alias src (in place) */, const float* __restrict__ xyz, const float* __restrict__ inv_freq,
                  int64_t n_tok, int S, int R, int H, int D, float sign, int in_place) {
  const int Q = D / 6, P = 3 * Q;                      // pairs per head
  const int64_t total = n_tok * P;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t t = e / P;
    const int pr = (int)(e - t * P), a = pr / Q, i = pr - a * Q;
    const float f = xyz[t * 3 + a] * inv_freq[i];                          // utonia.py:62-67
    float sn, cs;
    sincosf(f, &sn, &cs);
    sn *= sign;
    const int64_t o = t * (int64_t)S * H * D + a * 2 * Q + i;
    const TI* p = src + o;
    TO* q = dst + o;
    const int rot = R * H, all = S * H;
    for (int h = 0; h < rot; ++h, p += D, q += D) {
      const float u = ptc_to_float(p[0]), v = ptc_to_float(p[Q]);
      float ru, rv;
      ptc_rope_pair(u, v, cs, sn, ru, rv);                                 // x cos + rotate_half(x) sin, :91-92
      q[0] = ptc_from_float<TO>(ru);
      q[Q] = ptc_from_float<TO>(rv);
    }
    if (!in_place) {
      for (int h = rot; h < all; ++h, p += D, q += D) {
        q[0] = ptc_from_float<TO>(ptc_to_float(p[0]));
        q[Q] = ptc_from_float<TO>(ptc_to_float(p[Q]));
      }
    }
  }
}

template <typename TI>
static int launch_rope3d_xyz(const void* src, void* dst, int dst_dtype, const float* xyz, const float* inv_freq, int64_t n, int S, int R,
                             int H, int D, float sign, hipStream_t s) {
  const int64_t total = n * (D / 2);
  int64_t grid = ptc_cdiv(total, 256);
  if (grid > 65536) grid = 65536;
  const int in_place = (src == dst) ? 1 : 0;
  PTC_DISPATCH_DTYPE(dst_dtype, TO, hipLaunchKernelGGL((rope3d_xyz_kernel<TI, TO>), dim3((unsigned)grid), dim3(256), 0, s, (const TI*)src,
                                                      (TO*)dst, xyz, inv_freq, n, S, R, H, D, sign, in_place));
  PTC_CHECK_LAUNCH("rope3d_xyz_kernel");
  return PTC_OK;
}

extern "C" int ptc_rope3d_xyz(const void* src, int src_dtype, void* dst, int dst_dtype, const float* xyz, const float* inv_freq,
                              int64_t n_tokens, int slabs, int rot_slabs, int H, int D, float sign, ptc_stream_t stream) {
  PTC_REQUIRE(n_tokens >= 0 && H >= 1 && slabs >= 1 && rot_slabs >= 0 && rot_slabs <= slabs, PTC_EINVAL, "ptc_rope3d_xyz: bad sizes");
  PTC_REQUIRE(D >= 6 && D % 6 == 0, PTC_EUNSUPPORTED, "ptc_rope3d_xyz: head dim %d must be a multiple of 6", D);   // utonia.py:46-48 + even chunks
  PTC_REQUIRE(sign == 1.f || sign == -1.f, PTC_EINVAL, "ptc_rope3d_xyz: sign must be +1 (forward) or -1 (gradient)");
  if (n_tokens == 0) return PTC_OK;
  PTC_REQUIRE(src && dst && xyz && inv_freq, PTC_EINVAL, "ptc_rope3d_xyz: null buffer");
  PTC_REQUIRE(src != dst || src_dtype == dst_dtype, PTC_EINVAL, "ptc_rope3d_xyz: in place needs one dtype");
  hipStream_t s = (hipStream_t)stream;
  PTC_DISPATCH_DTYPE(src_dtype, TI, return launch_rope3d_xyz<TI>(src, dst, dst_dtype, xyz, inv_freq, n_tokens, slabs, rot_slabs, H, D, sign, s));
  return PTC_OK;
}
