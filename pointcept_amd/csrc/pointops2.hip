// pointops2.hip -- the pair-list attention operators of libs/pointops2 (Stratified Transformer), fp32.
//
// The six families libs/pointops2 shares with libs/pointops (kNN, FPS, grouping, interpolation, subtraction, aggregation)
// are served by pointops.hip; this file adds what only pointops2 has (reference: libs/pointops2/functions/pointops.py):
//   attention_step1 / _v2            :93-258   out[m,h]   = sum_c q[i0[m],h,c] k[i1[m],h,c]
//   dot_prod_with_idx / _v2 / _v3    :407-755  out[m,h]   = sum_c q[i0[m],h,c] Tq(m,h,c) (+ sum_c k[i1[m],h,c] Tk(m,h,c) in v2 / v3)
//   attention_step2 / _v2            :261-404  out[n,h,c] = sum_{m: i0[m] = n} attn[m,h] v[i1[m],h,c]
//   attention_step2_with_rel_pos_value / _v2 :758-961   ... attn[m,h] (v[i1[m],h,c] + Tv(m,h,c))
// with T(m,h,c) = sum_{a<3} table[rel_idx[m,a], h, c, a]   (table [L, h, d, 3], rel_idx [M, 3]; kernels:
// libs/pointops2/src/{attention,attention_v2,rpe,rpe_v2}/*.cu).  The v1 / v2 / v3 forms of the reference differ in how the
// pair list is handed over (a query index per pair, or CSR offsets of the pairs of each query + n_max) and in their thread
// mapping, not in the arithmetic; here ONE pair-dot and ONE pair-aggregate operator take both forms:
//   i0      [M] int32    query of every pair (always given)
//   offsets [Nq+1] int32 CSR of the pairs by query, or NULL.  With offsets the sums over a query's pairs (aggregate forward,
//                        dQ) run as fixed-order segment loops (bit-reproducible); without, and for everything that scatters
//                        by key / table entry (dK, dV, d table), float atomics are used exactly where the reference uses
//                        atomicAdd.
// All of it is HBM / L2 gather traffic with ~1 flop per byte: one thread per (pair or query, head, channel group), no LDS.
#include "ptc_common.h"

#define P2_THREADS 256

__device__ __forceinline__ float p2_table(const float* __restrict__ t, const int32_t* __restrict__ rel, int64_t m, int h, int H,
                                          int d, int c) {
  const int64_t C3 = (int64_t)H * d * 3, off = ((int64_t)h * d + c) * 3;
  return t[rel[m * 3] * C3 + off] + t[rel[m * 3 + 1] * C3 + off + 1] + t[rel[m * 3 + 2] * C3 + off + 2];
}

// out[m,h] = [k] q.k + [tq] q.Tq + [tk] k.Tk                       one thread per (pair, head)
__global__ void __launch_bounds__(P2_THREADS)
p2_pair_dot_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const int32_t* __restrict__ i0,
                       const int32_t* __restrict__ i1, const float* __restrict__ tq, const float* __restrict__ tk,
                       const int32_t* __restrict__ rel, int qk, int64_t M, int H, int d, float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * P2_THREADS + threadIdx.x;
  if (t >= M * H) return;
  const int64_t m = t / H;
  const int h = (int)(t - m * H);
  const float* qr = q + ((int64_t)i0[m] * H + h) * d;
  const float* kr = k ? k + ((int64_t)i1[m] * H + h) * d : nullptr;
  float s = 0.f;
  for (int c = 0; c < d; ++c) {
    const float qv = qr[c];
    if (qk) s = fmaf(qv, kr[c], s);
    if (tq) s = fmaf(qv, p2_table(tq, rel, m, h, H, d, c), s);
    if (tk) s = fmaf(kr[c], p2_table(tk, rel, m, h, H, d, c), s);
  }
  out[t] = s;
}

// backward of the pair dot.  dq: segment loop per (query, head, channel) when offsets are given, else atomics per pair;
// dk / d tables: atomics per (pair, head, channel)
__global__ void __launch_bounds__(P2_THREADS)
p2_pair_dot_bwd_pairs_kernel(const float* __restrict__ g, const float* __restrict__ q, const float* __restrict__ k,
                             const int32_t* __restrict__ i0, const int32_t* __restrict__ i1, const float* __restrict__ tq,
                             const float* __restrict__ tk, const int32_t* __restrict__ rel, int qk, int dq_here, int64_t M, int H,
                             int d, float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dtq,
                             float* __restrict__ dtk) {
  const int64_t t = (int64_t)blockIdx.x * P2_THREADS + threadIdx.x;
  if (t >= M * H * d) return;
  const int c = (int)(t % d);
  const int64_t mh = t / d;
  const int h = (int)(mh % H);
  const int64_t m = mh / H;
  const float gv = g[mh];
  if (gv == 0.f) return;
  const int64_t qo = ((int64_t)i0[m] * H + h) * d + c;
  const int64_t ko = k ? ((int64_t)i1[m] * H + h) * d + c : 0;
  const int64_t C3 = (int64_t)H * d * 3, off = ((int64_t)h * d + c) * 3;
  if (dq_here) {
    float a = 0.f;
    if (qk) a += k[ko];
    if (tq) a += p2_table(tq, rel, m, h, H, d, c);
    atomicAdd(dq + qo, gv * a);
  }
  if (dk && (qk || tk)) {
    float a = 0.f;
    if (qk) a += q[qo];
    if (tk) a += p2_table(tk, rel, m, h, H, d, c);
    atomicAdd(dk + ko, gv * a);
  }
  if (dtq) {
    const float v = gv * q[qo];
#pragma unroll
    for (int a = 0; a < 3; ++a) atomicAdd(dtq + rel[m * 3 + a] * C3 + off + a, v);
  }
  if (dtk) {
    const float v = gv * k[ko];
#pragma unroll
    for (int a = 0; a < 3; ++a) atomicAdd(dtk + rel[m * 3 + a] * C3 + off + a, v);
  }
}
__global__ void __launch_bounds__(P2_THREADS)
p2_pair_dot_bwd_dq_seg_kernel(const float* __restrict__ g, const float* __restrict__ k, const int32_t* __restrict__ offsets,
                              const int32_t* __restrict__ i1, const float* __restrict__ tq, const int32_t* __restrict__ rel, int qk,
                              int64_t Nq, int H, int d, float* __restrict__ dq) {
  const int64_t t = (int64_t)blockIdx.x * P2_THREADS + threadIdx.x;
  if (t >= Nq * H * d) return;
  const int c = (int)(t % d);
  const int64_t nh = t / d;
  const int h = (int)(nh % H);
  const int64_t n = nh / H;
  float s = 0.f;
  for (int64_t m = offsets[n]; m < offsets[n + 1]; ++m) {
    float a = 0.f;
    if (qk) a += k[((int64_t)i1[m] * H + h) * d + c];
    if (tq) a += p2_table(tq, rel, m, h, H, d, c);
    s = fmaf(g[m * H + h], a, s);
  }
  dq[t] = s;
}

// out[n,h,c] = sum_{pairs of n} attn[m,h] (v[i1[m],h,c] + [tv] Tv(m,h,c))
__global__ void __launch_bounds__(P2_THREADS)
p2_pair_agg_fwd_seg_kernel(const float* __restrict__ attn, const float* __restrict__ v, const int32_t* __restrict__ offsets,
                           const int32_t* __restrict__ i1, const float* __restrict__ tv, const int32_t* __restrict__ rel, int64_t Nq,
                           int H, int d, float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * P2_THREADS + threadIdx.x;
  if (t >= Nq * H * d) return;
  const int c = (int)(t % d);
  const int64_t nh = t / d;
  const int h = (int)(nh % H);
  const int64_t n = nh / H;
  float s = 0.f;
  for (int64_t m = offsets[n]; m < offsets[n + 1]; ++m) {
    float a = v[((int64_t)i1[m] * H + h) * d + c];
    if (tv) a += p2_table(tv, rel, m, h, H, d, c);
    s = fmaf(attn[m * H + h], a, s);
  }
  out[t] = s;
}
__global__ void __launch_bounds__(P2_THREADS)
p2_pair_agg_fwd_atomic_kernel(const float* __restrict__ attn, const float* __restrict__ v, const int32_t* __restrict__ i0,
                              const int32_t* __restrict__ i1, const float* __restrict__ tv, const int32_t* __restrict__ rel, int64_t M,
                              int H, int d, float* __restrict__ out /* zeroed */) {
  const int64_t t = (int64_t)blockIdx.x * P2_THREADS + threadIdx.x;
  if (t >= M * H * d) return;
  const int c = (int)(t % d);
  const int64_t mh = t / d;
  const int h = (int)(mh % H);
  const int64_t m = mh / H;
  float a = v[((int64_t)i1[m] * H + h) * d + c];
  if (tv) a += p2_table(tv, rel, m, h, H, d, c);
  atomicAdd(out + ((int64_t)i0[m] * H + h) * d + c, attn[mh] * a);
}
// backward of the aggregate: d attn per (pair, head) (plain sum), dv / d table atomics per (pair, head, channel)
__global__ void __launch_bounds__(P2_THREADS)
p2_pair_agg_bwd_attn_kernel(const float* __restrict__ go, const float* __restrict__ v, const int32_t* __restrict__ i0,
                            const int32_t* __restrict__ i1, const float* __restrict__ tv, const int32_t* __restrict__ rel, int64_t M,
                            int H, int d, float* __restrict__ dattn) {
  const int64_t t = (int64_t)blockIdx.x * P2_THREADS + threadIdx.x;
  if (t >= M * H) return;
  const int64_t m = t / H;
  const int h = (int)(t - m * H);
  const float* gr = go + ((int64_t)i0[m] * H + h) * d;
  const float* vr = v + ((int64_t)i1[m] * H + h) * d;
  float s = 0.f;
  for (int c = 0; c < d; ++c) {
    float a = vr[c];
    if (tv) a += p2_table(tv, rel, m, h, H, d, c);
    s = fmaf(gr[c], a, s);
  }
  dattn[t] = s;
}
__global__ void __launch_bounds__(P2_THREADS)
p2_pair_agg_bwd_scatter_kernel(const float* __restrict__ go, const float* __restrict__ attn, const int32_t* __restrict__ i0,
                               const int32_t* __restrict__ i1, const int32_t* __restrict__ rel, int64_t M, int H, int d,
                               float* __restrict__ dv, float* __restrict__ dtv) {
  const int64_t t = (int64_t)blockIdx.x * P2_THREADS + threadIdx.x;
  if (t >= M * H * d) return;
  const int c = (int)(t % d);
  const int64_t mh = t / d;
  const int h = (int)(mh % H);
  const int64_t m = mh / H;
  const float val = attn[mh] * go[((int64_t)i0[m] * H + h) * d + c];
  if (val == 0.f) return;
  if (dv) atomicAdd(dv + ((int64_t)i1[m] * H + h) * d + c, val);
  if (dtv) {
    const int64_t C3 = (int64_t)H * d * 3, off = ((int64_t)h * d + c) * 3;
#pragma unroll
    for (int a = 0; a < 3; ++a) atomicAdd(dtv + rel[m * 3 + a] * C3 + off + a, val);
  }
}

static unsigned p2_grid(int64_t n) { return (unsigned)ptc_cdiv(n > 0 ? n : 1, P2_THREADS); }
// one thread per work item on a 1-D grid: the item count must fit 2^31 - 1 workgroups
static bool p2_fits(int64_t a, int64_t b, int64_t c = 1) {
  if (a == 0 || b == 0 || c == 0) return true;
  const int64_t lim = (int64_t)0x7fffffff * P2_THREADS;
  return a <= lim / b && a * b <= lim / c;
}

extern "C" int ptc_pair_dot_fwd(const float* q, const float* k, const int32_t* i0, const int32_t* i1, const float* table_q,
                                const float* table_k, const int32_t* rel_idx, int with_qk, int64_t M, int H, int d, float* out,
                                ptc_stream_t stream) {
  PTC_REQUIRE(M >= 0 && H >= 1 && d >= 1, PTC_EINVAL, "ptc_pair_dot_fwd: bad sizes");
  PTC_REQUIRE(p2_fits(M, H), PTC_EUNSUPPORTED, "ptc_pair_dot_fwd: too many pairs");
  if (M == 0) return PTC_OK;
  PTC_REQUIRE(q && i0 && out, PTC_EINVAL, "ptc_pair_dot_fwd: null buffer");
  PTC_REQUIRE(!(with_qk || table_k) || (k && i1), PTC_EINVAL, "ptc_pair_dot_fwd: the q.k / k.table terms need k and i1");
  PTC_REQUIRE(!(table_q || table_k) || rel_idx, PTC_EINVAL, "ptc_pair_dot_fwd: tables need rel_idx");
  hipLaunchKernelGGL(p2_pair_dot_fwd_kernel, dim3(p2_grid(M * H)), dim3(P2_THREADS), 0, (hipStream_t)stream, q, k, i0, i1, table_q,
                     table_k, rel_idx, with_qk, M, H, d, out);
  PTC_CHECK_LAUNCH("p2_pair_dot_fwd_kernel");
  return PTC_OK;
}

extern "C" int ptc_pair_dot_bwd(const float* grad_out, const float* q, const float* k, const int32_t* i0, const int32_t* offsets,
                                const int32_t* i1, const float* table_q, const float* table_k, const int32_t* rel_idx, int with_qk,
                                int64_t M, int64_t Nq, int64_t Nk, int64_t L, int H, int d, float* dq, float* dk, float* dtable_q,
                                float* dtable_k, ptc_stream_t stream) {
  PTC_REQUIRE(M >= 0 && Nq >= 0 && Nk >= 0 && L >= 0 && H >= 1 && d >= 1, PTC_EINVAL, "ptc_pair_dot_bwd: bad sizes");
  PTC_REQUIRE(p2_fits(M, H, d) && p2_fits(Nq, H, d), PTC_EUNSUPPORTED, "ptc_pair_dot_bwd: too many work items for one launch");
  hipStream_t s = (hipStream_t)stream;
  const size_t row = (size_t)H * d * sizeof(float);
  const bool seg = offsets != nullptr;
  if (dq && !seg) PTC_HIP(hipMemsetAsync(dq, 0, (size_t)Nq * row, s));
  if (dk) PTC_HIP(hipMemsetAsync(dk, 0, (size_t)Nk * row, s));
  if (dtable_q) PTC_HIP(hipMemsetAsync(dtable_q, 0, (size_t)L * row * 3, s));
  if (dtable_k) PTC_HIP(hipMemsetAsync(dtable_k, 0, (size_t)L * row * 3, s));
  if (M == 0) {
    if (dq && seg && Nq > 0) PTC_HIP(hipMemsetAsync(dq, 0, (size_t)Nq * row, s));
    return PTC_OK;
  }
  PTC_REQUIRE(grad_out && q && i0, PTC_EINVAL, "ptc_pair_dot_bwd: null buffer");
  PTC_REQUIRE(!(with_qk || table_k || dk || dtable_k) || (k && i1), PTC_EINVAL, "ptc_pair_dot_bwd: k / i1 missing");
  PTC_REQUIRE(!(table_q || table_k || dtable_q || dtable_k) || rel_idx, PTC_EINVAL, "ptc_pair_dot_bwd: rel_idx missing");
  PTC_REQUIRE(!dtable_q || table_q, PTC_EINVAL, "ptc_pair_dot_bwd: d table_q without table_q");
  PTC_REQUIRE(!dtable_k || table_k, PTC_EINVAL, "ptc_pair_dot_bwd: d table_k without table_k");
  if (dq && seg) {
    hipLaunchKernelGGL(p2_pair_dot_bwd_dq_seg_kernel, dim3(p2_grid(Nq * H * d)), dim3(P2_THREADS), 0, s, grad_out, k, offsets, i1,
                       table_q, rel_idx, with_qk, Nq, H, d, dq);
    PTC_CHECK_LAUNCH("p2_pair_dot_bwd_dq_seg_kernel");
  }
  if ((dq && !seg) || dk || dtable_q || dtable_k) {
    hipLaunchKernelGGL(p2_pair_dot_bwd_pairs_kernel, dim3(p2_grid(M * H * d)), dim3(P2_THREADS), 0, s, grad_out, q, k, i0, i1, table_q,
                       table_k, rel_idx, with_qk, (dq && !seg) ? 1 : 0, M, H, d, dq, dk, dtable_q, dtable_k);
    PTC_CHECK_LAUNCH("p2_pair_dot_bwd_pairs_kernel");
  }
  return PTC_OK;
}

extern "C" int ptc_pair_aggregate_fwd(const float* attn, const float* v, const int32_t* i0, const int32_t* offsets, const int32_t* i1,
                                      const float* table_v, const int32_t* rel_idx, int64_t M, int64_t Nq, int H, int d, float* out,
                                      ptc_stream_t stream) {
  PTC_REQUIRE(M >= 0 && Nq >= 0 && H >= 1 && d >= 1, PTC_EINVAL, "ptc_pair_aggregate_fwd: bad sizes");
  PTC_REQUIRE(p2_fits(M, H, d) && p2_fits(Nq, H, d), PTC_EUNSUPPORTED, "ptc_pair_aggregate_fwd: too many work items for one launch");
  hipStream_t s = (hipStream_t)stream;
  if (Nq == 0) return PTC_OK;
  PTC_REQUIRE(out, PTC_EINVAL, "ptc_pair_aggregate_fwd: null buffer");
  if (M == 0 || !offsets) PTC_HIP(hipMemsetAsync(out, 0, (size_t)Nq * H * d * sizeof(float), s));
  if (M == 0) return PTC_OK;
  PTC_REQUIRE(attn && v && i1 && (offsets || i0), PTC_EINVAL, "ptc_pair_aggregate_fwd: null buffer");
  PTC_REQUIRE(!table_v || rel_idx, PTC_EINVAL, "ptc_pair_aggregate_fwd: table needs rel_idx");
  if (offsets) {
    hipLaunchKernelGGL(p2_pair_agg_fwd_seg_kernel, dim3(p2_grid(Nq * H * d)), dim3(P2_THREADS), 0, s, attn, v, offsets, i1, table_v,
                       rel_idx, Nq, H, d, out);
    PTC_CHECK_LAUNCH("p2_pair_agg_fwd_seg_kernel");
  } else {
    hipLaunchKernelGGL(p2_pair_agg_fwd_atomic_kernel, dim3(p2_grid(M * H * d)), dim3(P2_THREADS), 0, s, attn, v, i0, i1, table_v,
                       rel_idx, M, H, d, out);
    PTC_CHECK_LAUNCH("p2_pair_agg_fwd_atomic_kernel");
  }
  return PTC_OK;
}

extern "C" int ptc_pair_aggregate_bwd(const float* grad_out, const float* attn, const float* v, const int32_t* i0, const int32_t* i1,
                                      const float* table_v, const int32_t* rel_idx, int64_t M, int64_t Nv, int64_t L, int H, int d,
                                      float* dattn, float* dv, float* dtable_v, ptc_stream_t stream) {
  PTC_REQUIRE(M >= 0 && Nv >= 0 && L >= 0 && H >= 1 && d >= 1, PTC_EINVAL, "ptc_pair_aggregate_bwd: bad sizes");
  PTC_REQUIRE(p2_fits(M, H, d), PTC_EUNSUPPORTED, "ptc_pair_aggregate_bwd: too many work items for one launch");
  hipStream_t s = (hipStream_t)stream;
  const size_t row = (size_t)H * d * sizeof(float);
  if (dv) PTC_HIP(hipMemsetAsync(dv, 0, (size_t)Nv * row, s));
  if (dtable_v) PTC_HIP(hipMemsetAsync(dtable_v, 0, (size_t)L * row * 3, s));
  if (M == 0) return PTC_OK;
  PTC_REQUIRE(grad_out && attn && v && i0 && i1, PTC_EINVAL, "ptc_pair_aggregate_bwd: null buffer");
  PTC_REQUIRE(!(table_v || dtable_v) || rel_idx, PTC_EINVAL, "ptc_pair_aggregate_bwd: rel_idx missing");
  PTC_REQUIRE(!dtable_v || table_v, PTC_EINVAL, "ptc_pair_aggregate_bwd: d table without table");
  if (dattn) {
    hipLaunchKernelGGL(p2_pair_agg_bwd_attn_kernel, dim3(p2_grid(M * H)), dim3(P2_THREADS), 0, s, grad_out, v, i0, i1, table_v, rel_idx,
                       M, H, d, dattn);
    PTC_CHECK_LAUNCH("p2_pair_agg_bwd_attn_kernel");
  }
  if (dv || dtable_v) {
    hipLaunchKernelGGL(p2_pair_agg_bwd_scatter_kernel, dim3(p2_grid(M * H * d)), dim3(P2_THREADS), 0, s, grad_out, attn, i0, i1,
                       rel_idx, M, H, d, dv, dtable_v);
    PTC_CHECK_LAUNCH("p2_pair_agg_bwd_scatter_kernel");
  }
  return PTC_OK;
}
