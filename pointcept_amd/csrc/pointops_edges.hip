// pointops_edges.hip -- the four "edge list" operator families of libs/pointops as ONE set of kernels:
//   grouping      libs/pointops/src/grouping/grouping_cuda_kernel.cu:5-27       (functions/grouping.py)
//   interpolation libs/pointops/src/interpolation/interpolation_cuda_kernel.cu:6-36 (functions/interpolation.py)
//   aggregation   libs/pointops/src/aggregation/aggregation_cuda_kernel.cu:5-45  (functions/aggregation.py)
//   subtraction   libs/pointops/src/subtraction/subtraction_cuda_kernel.cu:5-36  (functions/subtraction.py)
// The reference runs one thread per output SCALAR (index % c, index / c arithmetic per element) and scatters every gradient
// with atomicAdd, so its backward sums differ from run to run.  Here an operator is a statement about the edge list
// E = {(t, s) -> j = idx[t, s]}:
//   * forward kernels walk rows: a lane owns a 16-byte piece (4 fp32 channels) of a row, consecutive lanes consecutive pieces --
//     every gathered row leaves memory as whole 64-byte segments (rows are 12 B .. 2 KB), the edge's index is read once per row
//     piece, not once per scalar;
//   * the gradient with respect to a GATHERED operand is a segmented sum over the edges sorted by source row (ptc_edge_csr_*:
//     keys for the engine's radix sort + the CSR pointer): one lane group per source row adds its edges in ascending edge order --
//     no atomics, bit-reproducible, and a source row is written once instead of deg(j) read-modify-write round trips;
//   * idx = -1 (a neighbour slot the query could not fill, libs/pointops/functions/query.py) gathers zeros and receives nothing.
// fp32 only, like the reference kernels.  All pointers are device pointers; rows are dense ([rows, c] with c floats per row) except
// the grouped output / gradient, which may be a column window of wider rows (with_xyz = True writes [xyz | feat] into one buffer).
#include "ptc_common.h"

namespace {

constexpr int EG_THREADS = 256;

__device__ __forceinline__ float4 eg_ld4(const float* p, int c, int ch) {
  // 4 channels starting at ch of a c-channel row (c % 4 == 0: one 16-byte load; else scalar with a zero tail)
  if ((c & 3) == 0) return *reinterpret_cast<const float4*>(p + ch);
  float4 v = {0.f, 0.f, 0.f, 0.f};
  v.x = p[ch];
  if (ch + 1 < c) v.y = p[ch + 1];
  if (ch + 2 < c) v.z = p[ch + 2];
  if (ch + 3 < c) v.w = p[ch + 3];
  return v;
}
__device__ __forceinline__ void eg_st4(float* p, int c, int ch, float4 v, bool aligned) {
  if (aligned) {
    *reinterpret_cast<float4*>(p + ch) = v;
    return;
  }
  p[ch] = v.x;
  if (ch + 1 < c) p[ch + 1] = v.y;
  if (ch + 2 < c) p[ch + 2] = v.z;
  if (ch + 3 < c) p[ch + 3] = v.w;
}

// ---- per-edge outputs: grouping (mode 0: out[e] = src[j]), subtraction (mode 1: out[e] = a[t] - src[j]),
//      relative positions of grouping(with_xyz) (mode 2: out[e] = src[j] - a[t], zeros for j < 0) ------------------------------
__global__ void __launch_bounds__(EG_THREADS)
edge_rows_fwd_kernel(int mode, const float* __restrict__ src, const float* __restrict__ a, const int32_t* __restrict__ idx,
                     int64_t n_edges, int nsample, int c, int64_t n_src, float* __restrict__ out, int64_t out_stride, int out_col0) {
  const int pieces = (c + 3) >> 2;
  const int64_t total = n_edges * pieces;
  const bool aligned = ((c & 3) | (out_stride & 3) | (out_col0 & 3)) == 0;
  for (int64_t v = (int64_t)blockIdx.x * EG_THREADS + threadIdx.x; v < total; v += (int64_t)gridDim.x * EG_THREADS) {
    const int64_t e = v / pieces;
    const int ch = (int)(v - e * pieces) * 4;
    const int32_t j = idx[e];
    const bool ok = j >= 0 && j < n_src;
    float4 r = {0.f, 0.f, 0.f, 0.f};
    if (ok) r = eg_ld4(src + (int64_t)j * c, c, ch);
    if (mode != 0) {
      const float4 q = eg_ld4(a + (e / nsample) * c, c, ch);
      if (mode == 1) r = make_float4(q.x - r.x, q.y - r.y, q.z - r.z, q.w - r.w);
      else r = ok ? make_float4(r.x - q.x, r.y - q.y, r.z - q.z, r.w - q.w) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    eg_st4(out + e * out_stride + out_col0, c, ch, r, aligned);
  }
}

// ---- per-target reductions over the nsample edges of a row:
//      interpolation (mode 0): out[t, c] = sum_s w[t, s] src[idx[t, s], c]
//      aggregation   (mode 1): out[t, c] = sum_s (src[idx[t, s], c] + pos[t, s, c]) w[t, s, c % w_c]
//      row sums      (mode 2): out[t, c] = sum_s g[t, s, c]   (subtraction: gradient of input1; g = `pos`, a column window allowed;
//                              with idx != NULL the empty slots are left out: gradient of new_xyz in grouping(with_xyz))
__global__ void __launch_bounds__(EG_THREADS)
edge_reduce_fwd_kernel(int mode, const float* __restrict__ src, const float* __restrict__ pos, int64_t pos_stride, int pos_col0,
                       const float* __restrict__ w, const int32_t* __restrict__ idx, int64_t m, int nsample, int c, int w_c,
                       int64_t n_src, float* __restrict__ out) {
  const int pieces = (c + 3) >> 2;
  const int64_t total = m * pieces;
  const bool aligned = (c & 3) == 0;
  for (int64_t v = (int64_t)blockIdx.x * EG_THREADS + threadIdx.x; v < total; v += (int64_t)gridDim.x * EG_THREADS) {
    const int64_t t = v / pieces;
    const int ch = (int)(v - t * pieces) * 4;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < nsample; ++s) {
      const int64_t e = t * nsample + s;
      float4 r = {0.f, 0.f, 0.f, 0.f};
      if (mode != 2) {
        const int32_t j = idx[e];
        if (j >= 0 && j < n_src) r = eg_ld4(src + (int64_t)j * c, c, ch);
      }
      if (mode == 0) {
        const float ws = w[e];
        acc.x = fmaf(r.x, ws, acc.x); acc.y = fmaf(r.y, ws, acc.y); acc.z = fmaf(r.z, ws, acc.z); acc.w = fmaf(r.w, ws, acc.w);
      } else if (mode == 1) {
        const float4 p = eg_ld4(pos + e * c, c, ch);
        const float* wr = w + e * w_c;
        acc.x = fmaf(r.x + p.x, wr[ch % w_c], acc.x);
        if (ch + 1 < c) acc.y = fmaf(r.y + p.y, wr[(ch + 1) % w_c], acc.y);
        if (ch + 2 < c) acc.z = fmaf(r.z + p.z, wr[(ch + 2) % w_c], acc.z);
        if (ch + 3 < c) acc.w = fmaf(r.w + p.w, wr[(ch + 3) % w_c], acc.w);
      } else {
        if (idx != nullptr && idx[e] < 0) continue;      // masked row sums: the relative coordinates of an empty slot are constants
        const float* pr = pos + e * pos_stride + pos_col0;
        acc.x += pr[ch];
        if (ch + 1 < c) acc.y += pr[ch + 1];
        if (ch + 2 < c) acc.z += pr[ch + 2];
        if (ch + 3 < c) acc.w += pr[ch + 3];
      }
    }
    eg_st4(out + t * c, c, ch, acc, aligned);
  }
}

// ---- gradient of the gathered operand: grad_src[j, c] = sum over the edges e with idx[e] = j, ascending e, of coef(e, c) g[row(e), c]
//      mode 0 grouping:      coef = 1,               row(e) = e        (g = gradient of the [E, c] output, column window allowed)
//      mode 1 subtraction:   coef = -1,              row(e) = e
//      mode 2 interpolation: coef = w[e],            row(e) = e / nsample
//      mode 3 aggregation:   coef = w[e, c % w_c],   row(e) = e / nsample
__global__ void __launch_bounds__(EG_THREADS)
edge_scatter_bwd_kernel(int mode, const int64_t* __restrict__ order, const int64_t* __restrict__ indptr, const float* __restrict__ g,
                        int64_t g_stride, int g_col0, const float* __restrict__ w, int nsample, int c, int w_c, int64_t n_src,
                        float* __restrict__ grad_src) {
  const int pieces = (c + 3) >> 2;
  const int64_t total = n_src * pieces;
  const bool aligned = (c & 3) == 0;
  for (int64_t v = (int64_t)blockIdx.x * EG_THREADS + threadIdx.x; v < total; v += (int64_t)gridDim.x * EG_THREADS) {
    const int64_t j = v / pieces;
    const int ch = (int)(v - j * pieces) * 4;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int64_t i = indptr[j]; i < indptr[j + 1]; ++i) {
      const int64_t e = order[i];
      const float* gr = g + (mode >= 2 ? e / nsample : e) * g_stride + g_col0;
      float4 r;
      r.x = gr[ch];
      r.y = ch + 1 < c ? gr[ch + 1] : 0.f;
      r.z = ch + 2 < c ? gr[ch + 2] : 0.f;
      r.w = ch + 3 < c ? gr[ch + 3] : 0.f;
      if (mode == 0) {
        acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w;
      } else if (mode == 1) {
        acc.x -= r.x; acc.y -= r.y; acc.z -= r.z; acc.w -= r.w;
      } else if (mode == 2) {
        const float ws = w[e];
        acc.x = fmaf(r.x, ws, acc.x); acc.y = fmaf(r.y, ws, acc.y); acc.z = fmaf(r.z, ws, acc.z); acc.w = fmaf(r.w, ws, acc.w);
      } else {
        const float* wr = w + e * w_c;
        acc.x = fmaf(r.x, wr[ch % w_c], acc.x);
        if (ch + 1 < c) acc.y = fmaf(r.y, wr[(ch + 1) % w_c], acc.y);
        if (ch + 2 < c) acc.z = fmaf(r.z, wr[(ch + 2) % w_c], acc.z);
        if (ch + 3 < c) acc.w = fmaf(r.w, wr[(ch + 3) % w_c], acc.w);
      }
    }
    eg_st4(grad_src + j * c, c, ch, acc, aligned);
  }
}

// ---- aggregation, the two per-edge gradients (aggregation_cuda_kernel.cu:23-39 without its atomics: every output element has ONE
//      producer once the loop runs over the channels of a weight column instead of over threads):
//      grad_pos[t, s, c] = g[t, c] w[t, s, c % w_c];   grad_w[t, s, k] = sum_{c = k (mod w_c)} g[t, c] (src[idx[t, s], c] + pos[t, s, c])
__global__ void __launch_bounds__(EG_THREADS)
aggregation_edge_bwd_kernel(const float* __restrict__ src, const float* __restrict__ pos, const float* __restrict__ w,
                            const int32_t* __restrict__ idx, const float* __restrict__ g, int64_t n_edges, int nsample, int c, int w_c,
                            int64_t n_src, float* __restrict__ grad_pos, float* __restrict__ grad_w) {
  const int64_t total = n_edges * w_c;
  for (int64_t v = (int64_t)blockIdx.x * EG_THREADS + threadIdx.x; v < total; v += (int64_t)gridDim.x * EG_THREADS) {
    const int64_t e = v / w_c;
    const int k = (int)(v - e * w_c);
    const int64_t t = e / nsample;
    const int32_t j = idx[e];
    const bool ok = j >= 0 && j < n_src;
    const float wk = w[e * w_c + k];
    float acc = 0.f;
    for (int ch = k; ch < c; ch += w_c) {
      const float gv = g[t * c + ch];
      const float x = (ok ? src[(int64_t)j * c + ch] : 0.f) + pos[e * c + ch];
      grad_pos[e * c + ch] = gv * wk;
      acc = fmaf(gv, x, acc);
    }
    grad_w[e * w_c + k] = acc;
  }
}

// ---- edge CSR by source row: keys for ptc_sort_keys (absent edges sort behind every row) and the pointer array --------------------
__global__ void __launch_bounds__(EG_THREADS)
edge_keys_kernel(const int32_t* __restrict__ idx, int64_t n_edges, int64_t n_src, int64_t* __restrict__ keys) {
  for (int64_t e = (int64_t)blockIdx.x * EG_THREADS + threadIdx.x; e < n_edges; e += (int64_t)gridDim.x * EG_THREADS) {
    const int32_t j = idx[e];
    keys[e] = (j >= 0 && j < n_src) ? (int64_t)j : n_src;
  }
}
// indptr[j] = number of sorted edges with key < j, j = 0 .. n_src: position i owns the keys in (key[i-1], key[i]]
__global__ void __launch_bounds__(EG_THREADS)
edge_ptr_kernel(const int64_t* __restrict__ keys, const int64_t* __restrict__ order, int64_t n_edges, int64_t n_src,
                int64_t* __restrict__ indptr) {
  for (int64_t i = (int64_t)blockIdx.x * EG_THREADS + threadIdx.x; i <= n_edges; i += (int64_t)gridDim.x * EG_THREADS) {
    const int64_t prev = i == 0 ? -1 : keys[order[i - 1]];
    const int64_t cur = i == n_edges ? n_src : keys[order[i]];
    for (int64_t j = prev + 1; j <= cur && j <= n_src; ++j) indptr[j] = i;
  }
}

int eg_grid(int64_t work) {
  int64_t b = (work + EG_THREADS - 1) / EG_THREADS;
  if (b < 1) b = 1;
  if (b > 8192) b = 8192;   // grid-stride beyond: 32 workgroups per CU keep every memory channel busy
  return (int)b;
}


// ---- pair-list attention of libs/pointops (PTv2 grouped vector attention; src/attention/attention_cuda_kernel.cu) ---------------
// A pair m joins row ia[m] of operand a with row ib[m] of operand b; rows are [g, c] fp32.
//   pair_dot:      out[m, g]    = sum_c a[ia[m], g, c] b[ib[m], g, c] (w ? w[c] : 1)                 one lane group per (pair, group)
//   pair_segment:  A[n, g, c]   = sum over the pairs e of row n (CSR by the scatter side, ascending pair index) of
//                                  s[e, g] b[oidx[e], g, c];   out = A (w ? w[c] : 1);   prod = self[n, g, c] A   (optional)
// The reference forward scatters one product per (pair, group, channel) with atomicAdd into out[m, g] and its backward / fusion
// step scatter by row index the same way (:9-25, :27-45, :46-62, :64-82): here every output element has ONE producer that adds in a
// fixed order.  Out-of-range indices contribute zeros.
__global__ void __launch_bounds__(EG_THREADS)
pair_dot_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ w, const int32_t* __restrict__ ia,
                const int32_t* __restrict__ ib, int64_t m, int64_t n_a, int64_t n_b, int g, int c, float* __restrict__ out) {
  const int64_t total = m * g;
  for (int64_t t = (int64_t)blockIdx.x * EG_THREADS + threadIdx.x; t < total; t += (int64_t)gridDim.x * EG_THREADS) {
    const int64_t e = t / g;
    const int gi = (int)(t - e * g);
    const int32_t ja = ia[e], jb = ib[e];
    float s = 0.f;
    if (ja >= 0 && ja < n_a && jb >= 0 && jb < n_b) {
      const float* ar = a + ((int64_t)ja * g + gi) * c;
      const float* br = b + ((int64_t)jb * g + gi) * c;
      for (int ch = 0; ch < c; ch += 4) {
        const float4 x = eg_ld4(ar, c, ch), y = eg_ld4(br, c, ch);
        float4 ww = {1.f, 1.f, 1.f, 1.f};
        if (w) ww = eg_ld4(w, c, ch);
        s = fmaf(x.x * y.x, ww.x, s);
        s = fmaf(x.y * y.y, ww.y, s);
        s = fmaf(x.z * y.z, ww.z, s);
        s = fmaf(x.w * y.w, ww.w, s);
      }
    }
    out[t] = s;
  }
}

__global__ void __launch_bounds__(EG_THREADS)
pair_segment_kernel(const float* __restrict__ sc, const float* __restrict__ b, const float* __restrict__ w, const float* __restrict__ self,
                    const int64_t* __restrict__ order, const int64_t* __restrict__ indptr, const int32_t* __restrict__ oidx, int64_t n_rows,
                    int64_t n_b, int g, int c, float* __restrict__ out, float* __restrict__ prod) {
  const int pieces = (c + 3) >> 2;
  const int64_t total = n_rows * g * pieces;
  const bool aligned = (c & 3) == 0;
  for (int64_t v = (int64_t)blockIdx.x * EG_THREADS + threadIdx.x; v < total; v += (int64_t)gridDim.x * EG_THREADS) {
    const int64_t ng = v / pieces;
    const int ch = (int)(v - ng * pieces) * 4;
    const int64_t n = ng / g;
    const int gi = (int)(ng - n * g);
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int64_t p = indptr[n]; p < indptr[n + 1]; ++p) {
      const int64_t e = order[p];
      const int32_t j = oidx[e];
      if (j < 0 || j >= n_b) continue;
      const float coef = sc[e * g + gi];
      const float4 y = eg_ld4(b + ((int64_t)j * g + gi) * c, c, ch);
      acc.x = fmaf(coef, y.x, acc.x);
      acc.y = fmaf(coef, y.y, acc.y);
      acc.z = fmaf(coef, y.z, acc.z);
      acc.w = fmaf(coef, y.w, acc.w);
    }
    if (prod) {
      const float4 q = eg_ld4(self + ng * c, c, ch);
      eg_st4(prod + ng * c, c, ch, make_float4(q.x * acc.x, q.y * acc.y, q.z * acc.z, q.w * acc.w), aligned);
    }
    if (w) {
      const float4 ww = eg_ld4(w, c, ch);
      acc = make_float4(acc.x * ww.x, acc.y * ww.y, acc.z * ww.z, acc.w * ww.w);
    }
    eg_st4(out + ng * c, c, ch, acc, aligned);
  }
}

}  // namespace

extern "C" {

int ptc_edge_rows_fwd(int mode, const float* src, const float* a, const int32_t* idx, int64_t n_edges, int nsample, int c,
                      int64_t n_src, float* out, int64_t out_stride, int out_col0, ptc_stream_t stream) {
  if (mode < 0 || mode > 2 || n_edges < 0 || nsample <= 0 || c <= 0 || n_src < 0 || out_stride < out_col0 + c || out_col0 < 0)
    { ptc_set_error("ptc_edge_rows_fwd: bad shape"); return PTC_EINVAL; }
  if (n_edges == 0) return PTC_OK;
  if (!src || !idx || !out || (mode != 0 && !a)) { ptc_set_error("ptc_edge_rows_fwd: null pointer"); return PTC_EINVAL; }
  hipLaunchKernelGGL(edge_rows_fwd_kernel, dim3(eg_grid(n_edges * ((c + 3) / 4))), dim3(EG_THREADS), 0, (hipStream_t)stream, mode, src, a, idx, n_edges, nsample, c,
                                                                                                 n_src, out, out_stride, out_col0);
  PTC_CHECK_LAUNCH("ptc_edge_rows_fwd");
  return PTC_OK;
}

int ptc_edge_reduce_fwd(int mode, const float* src, const float* pos, int64_t pos_stride, int pos_col0, const float* w,
                        const int32_t* idx, int64_t m, int nsample, int c, int w_c, int64_t n_src, float* out, ptc_stream_t stream) {
  if (mode < 0 || mode > 2 || m < 0 || nsample <= 0 || c <= 0 || n_src < 0) { ptc_set_error("ptc_edge_reduce_fwd: bad shape"); return PTC_EINVAL; }
  if (mode == 1 && (w_c <= 0 || c % w_c != 0)) { ptc_set_error("ptc_edge_reduce_fwd: c must be a multiple of w_c"); return PTC_EINVAL; }
  if (mode == 2 && (pos_col0 < 0 || pos_stride < pos_col0 + c)) { ptc_set_error("ptc_edge_reduce_fwd: bad column window"); return PTC_EINVAL; }
  if (m == 0) return PTC_OK;
  if (!out || (mode != 2 && (!src || !idx || !w)) || (mode != 0 && !pos)) { ptc_set_error("ptc_edge_reduce_fwd: null pointer"); return PTC_EINVAL; }
  hipLaunchKernelGGL(edge_reduce_fwd_kernel, dim3(eg_grid(m * ((c + 3) / 4))), dim3(EG_THREADS), 0, (hipStream_t)stream, mode, src, pos, pos_stride, pos_col0, w, idx, m,
                                                                                             nsample, c, w_c > 0 ? w_c : 1, n_src, out);
  PTC_CHECK_LAUNCH("ptc_edge_reduce_fwd");
  return PTC_OK;
}

int ptc_edge_csr_keys(const int32_t* idx, int64_t n_edges, int64_t n_src, int64_t* keys, ptc_stream_t stream) {
  if (n_edges < 0 || n_src < 0) { ptc_set_error("ptc_edge_csr_keys: bad shape"); return PTC_EINVAL; }
  if (n_edges == 0) return PTC_OK;
  if (!idx || !keys) { ptc_set_error("ptc_edge_csr_keys: null pointer"); return PTC_EINVAL; }
  hipLaunchKernelGGL(edge_keys_kernel, dim3(eg_grid(n_edges)), dim3(EG_THREADS), 0, (hipStream_t)stream, idx, n_edges, n_src, keys);
  PTC_CHECK_LAUNCH("ptc_edge_csr_keys");
  return PTC_OK;
}

int ptc_edge_csr_ptr(const int64_t* keys, const int64_t* order, int64_t n_edges, int64_t n_src, int64_t* indptr, ptc_stream_t stream) {
  if (n_edges < 0 || n_src < 0) { ptc_set_error("ptc_edge_csr_ptr: bad shape"); return PTC_EINVAL; }
  if (!indptr || (n_edges > 0 && (!keys || !order))) { ptc_set_error("ptc_edge_csr_ptr: null pointer"); return PTC_EINVAL; }
  hipLaunchKernelGGL(edge_ptr_kernel, dim3(eg_grid(n_edges + 1)), dim3(EG_THREADS), 0, (hipStream_t)stream, keys, order, n_edges, n_src, indptr);
  PTC_CHECK_LAUNCH("ptc_edge_csr_ptr");
  return PTC_OK;
}

int ptc_edge_scatter_bwd(int mode, const int64_t* order, const int64_t* indptr, const float* g, int64_t g_stride, int g_col0,
                         const float* w, int nsample, int c, int w_c, int64_t n_src, float* grad_src, ptc_stream_t stream) {
  if (mode < 0 || mode > 3 || nsample <= 0 || c <= 0 || n_src < 0 || g_col0 < 0 || g_stride < g_col0 + c)
    { ptc_set_error("ptc_edge_scatter_bwd: bad shape"); return PTC_EINVAL; }
  if (mode == 3 && (w_c <= 0 || c % w_c != 0)) { ptc_set_error("ptc_edge_scatter_bwd: c must be a multiple of w_c"); return PTC_EINVAL; }
  if (n_src == 0) return PTC_OK;
  if (!order || !indptr || !g || !grad_src || (mode >= 2 && !w)) { ptc_set_error("ptc_edge_scatter_bwd: null pointer"); return PTC_EINVAL; }
  hipLaunchKernelGGL(edge_scatter_bwd_kernel, dim3(eg_grid(n_src * ((c + 3) / 4))), dim3(EG_THREADS), 0, (hipStream_t)stream, mode, order, indptr, g, g_stride, g_col0, w,
                                                                                                  nsample, c, w_c > 0 ? w_c : 1, n_src,
                                                                                                  grad_src);
  PTC_CHECK_LAUNCH("ptc_edge_scatter_bwd");
  return PTC_OK;
}

int ptc_aggregation_edge_bwd(const float* src, const float* pos, const float* w, const int32_t* idx, const float* g, int64_t m,
                             int nsample, int c, int w_c, int64_t n_src, float* grad_pos, float* grad_w, ptc_stream_t stream) {
  if (m < 0 || nsample <= 0 || c <= 0 || w_c <= 0 || c % w_c != 0 || n_src < 0) { ptc_set_error("ptc_aggregation_edge_bwd: bad shape"); return PTC_EINVAL; }
  if (m == 0) return PTC_OK;
  if (!src || !pos || !w || !idx || !g || !grad_pos || !grad_w) { ptc_set_error("ptc_aggregation_edge_bwd: null pointer"); return PTC_EINVAL; }
  hipLaunchKernelGGL(aggregation_edge_bwd_kernel, dim3(eg_grid(m * nsample * w_c)), dim3(EG_THREADS), 0, (hipStream_t)stream, src, pos, w, idx, g, m * nsample, nsample, c,
                                                                                                  w_c, n_src, grad_pos, grad_w);
  PTC_CHECK_LAUNCH("ptc_aggregation_edge_bwd");
  return PTC_OK;
}

int ptc_pair_dot_weighted(const float* a, const float* b, const float* w, const int32_t* ia, const int32_t* ib, int64_t m, int64_t n_a,
                          int64_t n_b, int g, int c, float* out, ptc_stream_t stream) {
  if (m < 0 || n_a < 0 || n_b < 0 || g <= 0 || c <= 0) { ptc_set_error("ptc_pair_dot_weighted: bad shape"); return PTC_EINVAL; }
  if (m == 0) return PTC_OK;
  if (!a || !b || !ia || !ib || !out) { ptc_set_error("ptc_pair_dot_weighted: null pointer"); return PTC_EINVAL; }
  hipLaunchKernelGGL(pair_dot_kernel, dim3(eg_grid(m * g)), dim3(EG_THREADS), 0, (hipStream_t)stream, a, b, w, ia, ib, m, n_a, n_b, g, c, out);
  PTC_CHECK_LAUNCH("ptc_pair_dot_weighted");
  return PTC_OK;
}

int ptc_pair_segment_sum(const float* s, const float* b, const float* w, const float* self, const int64_t* order, const int64_t* indptr,
                         const int32_t* oidx, int64_t n_rows, int64_t n_b, int g, int c, float* out, float* prod, ptc_stream_t stream) {
  if (n_rows < 0 || n_b < 0 || g <= 0 || c <= 0) { ptc_set_error("ptc_pair_segment_sum: bad shape"); return PTC_EINVAL; }
  if (n_rows == 0) return PTC_OK;
  if (!indptr || !out || (prod && !self)) { ptc_set_error("ptc_pair_segment_sum: null pointer"); return PTC_EINVAL; }
  hipLaunchKernelGGL(pair_segment_kernel, dim3(eg_grid(n_rows * g * ((c + 3) / 4))), dim3(EG_THREADS), 0, (hipStream_t)stream, s, b, w, self, order, indptr,
                     oidx, n_rows, n_b, g, c, out, prod);
  PTC_CHECK_LAUNCH("ptc_pair_segment_sum");
  return PTC_OK;
}

}  // extern "C"
