// spconv_internal.h -- C++-side entry points of spconv.hip for the other translation units of the library (block_exec.hip).
// Not part of the C-ABI (include/ptcore.h): the Block executor uses them to batch the split-K reductions of a Block's weight gradients.
#pragma once
#include "ptc_common.h"

// what is left to do for one weight gradient whose partial sums were written by its kernel: dw[i] = sum_p partial[p][i] (and the bias
// segment).  splits == 0: nothing (the kernel wrote dw itself, or the call took a path that reduces at once).
struct PtcWgradJob {
  const float* partial;
  int splits;
  int64_t count;
  float* dw;
  const float* bias_partial;
  int64_t c_out;
  float* dbias;
  // device-side choice between two producers (ptc_spconv_wgrad_blk): *gate != 0 -> sum alt_partial[0 .. alt_splits) instead
  const int32_t* gate = nullptr;
  const float* alt_partial = nullptr;
  int alt_splits = 0;
};

// ptc_spconv_wgrad with the reduction left to the caller: same arguments, same partials, same workspace layout; `job` describes the
// reduction still owed.  The workspace must stay untouched until ptc_wgrad_reduce_jobs has been enqueued.
int ptc_spconv_wgrad_deferred(const void* in, int64_t n_in, const void* dout, const int32_t* nbr, int64_t n_out, int kv, int c_in, int c_out,
                              int dtype, float* dw, float* dbias, void* workspace, size_t workspace_bytes, ptc_stream_t stream,
                              PtcWgradJob* job);
// ptc_spconv_wgrad_blk (block-staged weight gradient of a 3^3 submanifold convolution, wgrad7.h) with the reduction left to the caller
int ptc_spconv_wgrad_blk_deferred(const void* in, int64_t n_in, const void* dout, const int32_t* nbr, const void* tab, const int32_t* hid,
                                  const int32_t* hcnt, const int32_t* n_overflow, int bm, int hcap, int64_t n_out, int kv, int c_in, int c_out,
                                  int dtype, float* dw, void* workspace, size_t workspace_bytes, ptc_stream_t stream, PtcWgradJob* job);
// Several weight gradients enqueued together: those that share the (4, 4, 1) kernel instance of wgrad2 (the Linear layers of a Block from
// 64 channels up) run as ONE launch, the others as their own; every reduction is left to the caller (jobs[i] for calls[i]).
struct PtcWgradCall {
  const void* in; int64_t n_in; const void* dout; const int32_t* nbr; int64_t n_out; int kv, c_in, c_out, dtype;
  float* dw; float* dbias; void* workspace; size_t workspace_bytes;
};
int ptc_spconv_wgrad_group(const PtcWgradCall* calls, int n, PtcWgradJob* jobs, ptc_stream_t stream);
// ONE launch for up to PTC_WGRAD_JOBS_MAX reductions (bit-identical to the separate launches: same per-output summation order)
#define PTC_WGRAD_JOBS_MAX 8
int ptc_wgrad_reduce_jobs(const PtcWgradJob* jobs, int n, ptc_stream_t stream);

// ptc_mlp_bwd (mlp.hip) with the reduction of its per-workgroup partials left to the caller: jobs for (dw1, db1) and (dw2, db2)
int ptc_mlp_bwd_deferred(const void* dm, const void* x, int64_t n, int c, int dtype, const void* w1, const float* b1, const void* w2t, void* dx,
                         float* dw1, float* db1, float* dw2, float* db2, void* workspace, size_t workspace_bytes, ptc_stream_t stream,
                         PtcWgradJob* job_fc1, PtcWgradJob* job_fc2);
