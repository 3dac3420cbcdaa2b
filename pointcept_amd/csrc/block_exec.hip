// block_exec.hip -- one C-ABI call per PT-v3m1 Block and direction: the ~12 (forward) / ~28 (backward) kernel launches of a Block
// (ptv3m1:318-338: x += LN(Linear(SubMConv(x))); x += DropPath(Attn(LN(x))); x += DropPath(MLP(LN(x)))) enqueued from C.
//
// Why: the step enqueued ~1900 launches from Python through ~220 autograd Functions -- ~33 us of
// interpreter + ctypes + allocator work per Function (profiles/r02_y_ddp_single_rank.txt: a 2 x 20000-voxel step cost 37 ms, the
// 8 x 102400-voxel one 32 ms of host time): the floor under any kernel work, and what 8 ranks sharing one host contend for.  The 22
// Blocks are 2/3 of those launches.  Here a Block is ONE autograd Function (functional._BlockFn) whose forward / backward make ONE
// call into this file with a table of pointers (activations saved for the backward live in slabs the caller allocated with two
// torch.empty calls); the arithmetic is the SAME sequence of the library's own entry points the Python-composed path issues
// (ptc_spconv_fwd[_blk], ptc_linear_fwd_ex, ptc_add_norm_*, ptc_attn_varlen_*, ptc_spconv_wgrad, ptc_column_sum) with the same
// operands: bit-identical results (tests/test_host_emulation_cpu.py, tests/test_gpu_model.py).
//
// No device code of its own; compiled with the kernels so that the host emulation (tests/host_emulation) builds it too.
#include "ptc_common.h"
#ifndef PTC_BLK_JOINT_EPILOGUE
#define PTC_BLK_JOINT_EPILOGUE 1     // 0: timing A/B only (`python -m pointcept_amd.build --variant d_PTC_BLK_JOINT_EPILOGUE_0`)
#endif
#include "spconv_internal.h"

#define RUN(call)               \
  do {                          \
    const int rc__ = (call);    \
    if (rc__ != PTC_OK) return rc__; \
  } while (0)

static inline const void* P(const void* const* p, int i) { return p[i]; }
template <typename T> static inline T* M(void* const* p, int i) { return (T*)p[i]; }

extern "C" int ptc_ptv3_block_abi(void) { return PTC_BLK_ABI; }

// Round 6: the MLP of a Block (steps 8 + 9 / 9' + 8') on csrc/mlp.hip's one kernel per direction where it exists (C = 32 | 64: the
// 819200- and ~200000-row stages).  Then H / ACT / M of the forward's slab and DH of the backward's are never touched (the caller need not
// allocate them: functional._blk_plan asks this function), the backward recomputes h.  PTC_BLK_MLP_FUSED=0: the split kernels (A/B).
extern "C" int ptc_ptv3_block_mlp_fused(int c, int dtype) {
  static const int off = [] { const char* e = getenv("PTC_BLK_MLP_FUSED"); return e && e[0] == '0'; }();
  return !off && ptc_mlp_supported(c, dtype);
}

// workspace of the backward: [scratch shared by the calls that finish inside themselves | one region per weight gradient: their
// partial sums wait there for the ONE reduction launch at the end of the Block's backward]
struct BlkWs { size_t common; size_t wg[6]; size_t off[6]; size_t total; };
static BlkWs blk_ws(int64_t n, int64_t n_pad, int c, int heads) {
  const int hid = 4 * c;
  BlkWs w;
  w.common = 256;
  auto up = [&](size_t v) { if (v > w.common) w.common = v; };
  up(ptc_add_norm_bwd_workspace_bytes(n, c));
  up(ptc_attn_varlen_bwd_workspace_bytes(n_pad, heads));
  up(ptc_batch_norm_workspace_bytes(n, c));
  w.common = ptc_align_up(w.common, 256);
  w.wg[0] = ptc_spconv_wgrad_workspace_bytes(n, 1, hid, c);        // fc2 (fused MLP: the partials of fc1 AND fc2, below)
  if (ptc_mlp_supported(c, PTC_BF16) && ptc_mlp_bwd_workspace_bytes(n, c) > w.wg[0]) w.wg[0] = ptc_mlp_bwd_workspace_bytes(n, c);
  w.wg[1] = ptc_spconv_wgrad_workspace_bytes(n, 1, c, hid);        // fc1
  w.wg[2] = ptc_spconv_wgrad_workspace_bytes(n, 1, c, c);          // proj
  w.wg[3] = ptc_spconv_wgrad_workspace_bytes(n_pad, 1, c, 3 * c);  // qkv
  w.wg[4] = ptc_spconv_wgrad_workspace_bytes(n, 1, c, c);          // the Linear of the positional encoding
  w.wg[5] = ptc_spconv_wgrad_blk_workspace_bytes(n, 27, c, c);     // the convolution (block-staged weight gradient where it applies)
  size_t o = w.common;
  for (int i = 0; i < 6; ++i) { w.off[i] = o; o += ptc_align_up(w.wg[i], 256); }
  w.total = o;
  return w;
}

extern "C" size_t ptc_ptv3_block_workspace_bytes(int64_t n, int64_t n_pad, int c, int heads) { return blk_ws(n, n_pad, c, heads).total; }

// ---- forward ------------------------------------------------------------------------------------------------------------------
extern "C" int ptc_ptv3_block_fwd(const int64_t* iv, const float* fv, const void* const* in, void* const* out, ptc_stream_t s) {
  PTC_REQUIRE(iv && fv && in && out, PTC_EINVAL, "ptc_ptv3_block_fwd: null argument table");
  PTC_REQUIRE(iv[PTC_BLK_I_ABI] == PTC_BLK_ABI, PTC_EINVAL, "ptc_ptv3_block_fwd: argument tables of another ABI revision");
  const int64_t n = iv[PTC_BLK_I_N], np = iv[PTC_BLK_I_NPAD], n_seq = iv[PTC_BLK_I_NSEQ];
  const int c = (int)iv[PTC_BLK_I_C], H = (int)iv[PTC_BLK_I_HEADS], dt = (int)iv[PTC_BLK_I_DTYPE], hid = 4 * c;
  const int a_dt = (int)iv[PTC_BLK_I_A_DTYPE], patch = (int)iv[PTC_BLK_I_PATCH];
  // f16 (round 4): the reference's fp16-autocast recipe -- every GEMM, joint and convolution on f16 operands; the window attention
  // keeps its bf16 arithmetic and does the call site's casts (ptv3m1:209,215) in its load / store paths (attention.hip, F16 I/O)
  PTC_REQUIRE(dt == PTC_BF16 || dt == PTC_F16, PTC_EUNSUPPORTED, "ptc_ptv3_block_fwd: 16-bit GEMM operands only");
  // (c <= 256: linear2's range; wider Blocks -- stage 4 of PT-v3m1, 512 channels -- when the MLP GEMMs with their GELU epilogues exist at that
  //  width: gemm3.h)
  PTC_REQUIRE(c % 16 == 0 && H * 16 == c && (c <= 256 || (c <= 512 && ptc_linear_supported_ex(c, hid, dt))), PTC_EUNSUPPORTED,
              "ptc_ptv3_block_fwd: c=%d heads=%d (head_dim 16, c <= 256, or <= 512 with the wide GEMM kernels)", c, H);
  if (n == 0) return PTC_OK;
  const int32_t* nbr = (const int32_t*)P(in, PTC_BLK_P_NBR);
  // 1. positional encoding: 3^3 submanifold convolution (ptv3m1:278-284) ...
  RUN(ptc_spconv_fwd_blk(P(in, PTC_BLK_P_XC), n, P(in, PTC_BLK_P_W_CONV), (const float*)P(in, PTC_BLK_P_B_CONV), nbr, P(in, PTC_BLK_P_BLK_TAB),
                         (const int32_t*)P(in, PTC_BLK_P_BLK_HID), (const int32_t*)P(in, PTC_BLK_P_BLK_HCNT), (int)iv[PTC_BLK_I_BLK_BM],
                         (int)iv[PTC_BLK_I_BLK_HCAP], n, 27, c, c, dt, out[PTC_BLK_O_CONV], s));
  // 2 + 3 in ONE launch where the shape allows (fwd2_joint.h, normA form): the Linear (ptv3m1:285), x1 = x0 + LN_cpe(lin), y1 = norm1(x1);
  // `lin` is still written (LN_cpe's backward reads it) but not read back
  if (PTC_BLK_JOINT_EPILOGUE && ptc_linear_joint_supported(c, c, dt)) {
    RUN(ptc_linear_norm_joint_fwd(out[PTC_BLK_O_CONV], n, P(in, PTC_BLK_P_W_LIN), (const float*)P(in, PTC_BLK_P_B_LIN), n, c, c, dt,
                                  (const float*)P(in, PTC_BLK_P_G_CPE), (const float*)P(in, PTC_BLK_P_BE_CPE), fv[PTC_BLK_F_EPS_CPE], P(in, PTC_BLK_P_X0), a_dt,
                                  (const float*)P(in, PTC_BLK_P_G_N1), (const float*)P(in, PTC_BLK_P_BE_N1), fv[PTC_BLK_F_EPS_N1], 1, out[PTC_BLK_O_LIN],
                                  M<float>(out, PTC_BLK_O_X1), out[PTC_BLK_O_Y1], M<float>(out, PTC_BLK_O_ST_CPE), M<float>(out, PTC_BLK_O_ST_N1), s));
  } else {
  // 2. ... its Linear (ptv3m1:285)
  RUN(ptc_spconv_fwd(out[PTC_BLK_O_CONV], n, P(in, PTC_BLK_P_W_LIN), (const float*)P(in, PTC_BLK_P_B_LIN), nullptr, n, 1, c, c, dt, out[PTC_BLK_O_LIN], s));
  // 3. x1 = x0 + LN_cpe(lin);  y1 = norm1(x1)
  RUN(ptc_add_norm_fwd(out[PTC_BLK_O_LIN], dt, P(in, PTC_BLK_P_X0), a_dt, nullptr, n, c, (const float*)P(in, PTC_BLK_P_G_CPE),
                       (const float*)P(in, PTC_BLK_P_BE_CPE), fv[PTC_BLK_F_EPS_CPE], 1, (const float*)P(in, PTC_BLK_P_G_N1),
                       (const float*)P(in, PTC_BLK_P_BE_N1), fv[PTC_BLK_F_EPS_N1], 1, M<float>(out, PTC_BLK_O_X1), out[PTC_BLK_O_Y1], dt,
                       M<float>(out, PTC_BLK_O_ST_CPE), M<float>(out, PTC_BLK_O_ST_N1), s));
  }
  // 4. qkv = Linear(y1)[order[pad]]: the serialization gather rides in the GEMM's row table (ptv3m1:184-188)
  RUN(ptc_spconv_fwd(out[PTC_BLK_O_Y1], n, P(in, PTC_BLK_P_W_QKV), (const float*)P(in, PTC_BLK_P_B_QKV), (const int32_t*)P(in, PTC_BLK_P_T_QKV_FWD), np, 1, c,
                     3 * c, dt, out[PTC_BLK_O_QKV], s));
  // 5. window attention (ptv3m1:208-214)
  RUN(ptc_attn_varlen_fwd(out[PTC_BLK_O_QKV], (const int32_t*)P(in, PTC_BLK_P_CU), n_seq, np, H, patch, fv[PTC_BLK_F_SCALE], dt, out[PTC_BLK_O_ATT],
                          M<float>(out, PTC_BLK_O_LSE), s));
  // 6. proj(att[inverse]) (ptv3m1:216-219)
  // 6 + 7 in ONE launch where the shape allows (round 4, fwd2_joint.h): the joint x2 = x1 + droppath(a), y2 = norm2(x2) runs in proj's
  // epilogue, `a` never reaches memory (the backward of this joint does not read it: its branch operand is not normalised)
  if (PTC_BLK_JOINT_EPILOGUE && ptc_linear_joint_supported(c, c, dt)) {
    RUN(ptc_linear_joint_fwd(out[PTC_BLK_O_ATT], np, P(in, PTC_BLK_P_W_PROJ), (const float*)P(in, PTC_BLK_P_B_PROJ), (const int32_t*)P(in, PTC_BLK_P_T_PROJ_FWD),
                             n, c, c, dt, M<float>(out, PTC_BLK_O_X1), (const float*)P(in, PTC_BLK_P_RS1), (const float*)P(in, PTC_BLK_P_G_N2),
                             (const float*)P(in, PTC_BLK_P_BE_N2), fv[PTC_BLK_F_EPS_N2], 1, M<float>(out, PTC_BLK_O_X2), out[PTC_BLK_O_Y2],
                             M<float>(out, PTC_BLK_O_ST_N2), s));
  } else {
    RUN(ptc_spconv_fwd(out[PTC_BLK_O_ATT], np, P(in, PTC_BLK_P_W_PROJ), (const float*)P(in, PTC_BLK_P_B_PROJ), (const int32_t*)P(in, PTC_BLK_P_T_PROJ_FWD), n, 1,
                       c, c, dt, out[PTC_BLK_O_A], s));
    // 7. x2 = x1 + droppath(a);  y2 = norm2(x2)
    RUN(ptc_add_norm_fwd(out[PTC_BLK_O_A], dt, out[PTC_BLK_O_X1], PTC_F32, (const float*)P(in, PTC_BLK_P_RS1), n, c, nullptr, nullptr, 0.f, 0,
                         (const float*)P(in, PTC_BLK_P_G_N2), (const float*)P(in, PTC_BLK_P_BE_N2), fv[PTC_BLK_F_EPS_N2], 1, M<float>(out, PTC_BLK_O_X2),
                         out[PTC_BLK_O_Y2], dt, nullptr, M<float>(out, PTC_BLK_O_ST_N2), s));
  }
  // 8 + 9 in ONE launch (mlp.hip): fc1 -> GELU -> fc2 -> x3 = x2 + droppath(m), xb3 = cast(x3); the hidden tensor stays on the CU
  if (ptc_ptv3_block_mlp_fused(c, dt))
    return ptc_mlp_fwd(out[PTC_BLK_O_Y2], n, c, dt, P(in, PTC_BLK_P_W_FC1), (const float*)P(in, PTC_BLK_P_B_FC1), P(in, PTC_BLK_P_W_FC2),
                       (const float*)P(in, PTC_BLK_P_B_FC2), M<float>(out, PTC_BLK_O_X2), (const float*)P(in, PTC_BLK_P_RS2), M<float>(out, PTC_BLK_O_X3),
                       out[PTC_BLK_O_XB3], s);
  // 8. MLP: (h, act) = fc1 + GELU in one kernel, then fc2 (ptv3m1:225-248)
  RUN(ptc_linear_fwd_ex(out[PTC_BLK_O_Y2], n, P(in, PTC_BLK_P_W_FC1), (const float*)P(in, PTC_BLK_P_B_FC1), c, hid, dt, 1, nullptr, out[PTC_BLK_O_H],
                        out[PTC_BLK_O_ACT], s));
  if (PTC_BLK_JOINT_EPILOGUE && ptc_linear_joint_supported(hid, c, dt)) {       // fc2 with joint 9 in its epilogue (4 c <= 256)
    RUN(ptc_linear_joint_fwd(out[PTC_BLK_O_ACT], n, P(in, PTC_BLK_P_W_FC2), (const float*)P(in, PTC_BLK_P_B_FC2), nullptr, n, hid, c, dt,
                             M<float>(out, PTC_BLK_O_X2), (const float*)P(in, PTC_BLK_P_RS2), nullptr, nullptr, 0.f, 0, M<float>(out, PTC_BLK_O_X3),
                             out[PTC_BLK_O_XB3], nullptr, s));
  } else {
    RUN(ptc_spconv_fwd(out[PTC_BLK_O_ACT], n, P(in, PTC_BLK_P_W_FC2), (const float*)P(in, PTC_BLK_P_B_FC2), nullptr, n, 1, hid, c, dt, out[PTC_BLK_O_M], s));
    // 9. x3 = x2 + droppath(m);  xb3 = cast(x3): the operand of the next convolution / Linear
    RUN(ptc_add_norm_fwd(out[PTC_BLK_O_M], dt, out[PTC_BLK_O_X2], PTC_F32, (const float*)P(in, PTC_BLK_P_RS2), n, c, nullptr, nullptr, 0.f, 0, nullptr,
                         nullptr, 0.f, 0, M<float>(out, PTC_BLK_O_X3), out[PTC_BLK_O_XB3], dt, nullptr, nullptr, s));
  }
  return PTC_OK;
}

// ---- backward -----------------------------------------------------------------------------------------------------------------
// `in`: the forward's inputs and saved outputs under the same indices plus the incoming gradients and the transposed weight
// layouts; `out`: gradients (PTC_BLK_G_*) and scratch activations gradients (PTC_BLK_S_*), all caller-allocated.
extern "C" int ptc_ptv3_block_bwd(const int64_t* iv, const float* fv, const void* const* in, const void* const* sv, void* const* g,
                                  void* workspace, size_t workspace_bytes, ptc_stream_t s) {
  PTC_REQUIRE(iv && fv && in && sv && g && workspace, PTC_EINVAL, "ptc_ptv3_block_bwd: null argument table");
  PTC_REQUIRE(iv[PTC_BLK_I_ABI] == PTC_BLK_ABI, PTC_EINVAL, "ptc_ptv3_block_bwd: argument tables of another ABI revision");
  const int64_t n = iv[PTC_BLK_I_N], np = iv[PTC_BLK_I_NPAD], n_seq = iv[PTC_BLK_I_NSEQ];
  const int c = (int)iv[PTC_BLK_I_C], H = (int)iv[PTC_BLK_I_HEADS], dt = (int)iv[PTC_BLK_I_DTYPE], hid = 4 * c;
  const int a_dt = (int)iv[PTC_BLK_I_A_DTYPE], patch = (int)iv[PTC_BLK_I_PATCH];
  PTC_REQUIRE(workspace_bytes >= ptc_ptv3_block_workspace_bytes(n, np, c, H), PTC_EWORKSPACE, "ptc_ptv3_block_bwd: workspace too small");
  if (n == 0) return PTC_OK;
  const int32_t* nbr = (const int32_t*)P(in, PTC_BLK_P_NBR);
  const float* rs1 = (const float*)P(in, PTC_BLK_P_RS1);
  const float* rs2 = (const float*)P(in, PTC_BLK_P_RS2);
  const BlkWs W = blk_ws(n, np, c, H);
  void* ws = workspace;                  // scratch of the calls that finish inside themselves
  const size_t wb = W.common;
  PtcWgradJob jobs[6];
  PtcWgradCall calls[5];                 // the five Linear weight gradients: recorded where their operands become final, enqueued together at the end
  auto wgws = [&](int i) { return (void*)((char*)workspace + W.off[i]); };
  // 9'. x3 = x2 + rs2 * m, xb3 = cast(x3):  dx2 = dz3 + dyb3;  dm = rs2 * dx2
  RUN(ptc_add_norm_bwd((const float*)P(in, PTC_BLK_P_DZ3), P(in, PTC_BLK_P_DYB3), dt, (const float*)P(sv, PTC_BLK_O_X3), P(sv, PTC_BLK_O_M), dt, rs2, n, c,
                       nullptr, nullptr, 0, nullptr, nullptr, 0, g[PTC_BLK_S_DX2], PTC_F32, g[PTC_BLK_S_DM], nullptr, nullptr, nullptr, nullptr, ws, wb, s));
  // 8'. MLP: fc2 weight / bias gradients, dh = (dm W2) * GELU'(h), fc1 weight / bias gradients, dy2 = dh W1
  const bool mlp_fused = ptc_ptv3_block_mlp_fused(c, dt) != 0;
  if (mlp_fused) {    // ONE launch (mlp.hip: h recomputed, dh never written); its partial sums wait in the fc2 region for the reduction below
    RUN(ptc_mlp_bwd_deferred(g[PTC_BLK_S_DM], P(sv, PTC_BLK_O_Y2), n, c, dt, P(in, PTC_BLK_P_W_FC1), (const float*)P(in, PTC_BLK_P_B_FC1),
                             P(in, PTC_BLK_P_WT_FC2), g[PTC_BLK_S_DY2], M<float>(g, PTC_BLK_G_W_FC1), M<float>(g, PTC_BLK_G_B_FC1),
                             M<float>(g, PTC_BLK_G_W_FC2), M<float>(g, PTC_BLK_G_B_FC2), wgws(0), W.wg[0], s, &jobs[1], &jobs[0]));
  } else {
  calls[0] = PtcWgradCall{P(sv, PTC_BLK_O_ACT), n, g[PTC_BLK_S_DM], nullptr, n, 1, hid, c, dt, M<float>(g, PTC_BLK_G_W_FC2), M<float>(g, PTC_BLK_G_B_FC2), wgws(0), W.wg[0]};
  RUN(ptc_linear_fwd_ex(g[PTC_BLK_S_DM], n, P(in, PTC_BLK_P_WT_FC2), nullptr, c, hid, dt, 2, P(sv, PTC_BLK_O_H), g[PTC_BLK_S_DH], nullptr, s));
  calls[1] = PtcWgradCall{P(sv, PTC_BLK_O_Y2), n, g[PTC_BLK_S_DH], nullptr, n, 1, c, hid, dt, M<float>(g, PTC_BLK_G_W_FC1), M<float>(g, PTC_BLK_G_B_FC1), wgws(1), W.wg[1]};
  RUN(ptc_spconv_fwd(g[PTC_BLK_S_DH], n, P(in, PTC_BLK_P_WT_FC1), nullptr, nullptr, n, 1, hid, c, dt, g[PTC_BLK_S_DY2], s));
  }
  // 7'. x2 = x1 + rs1 * a, y2 = norm2(x2):  dx1 = dx2 + LN'(dy2);  da = rs1 * dx1
  RUN(ptc_add_norm_bwd(M<float>(g, PTC_BLK_S_DX2), g[PTC_BLK_S_DY2], dt, (const float*)P(sv, PTC_BLK_O_X2), P(sv, PTC_BLK_O_A), dt, rs1, n, c, nullptr, nullptr, 0,
                       (const float*)P(in, PTC_BLK_P_G_N2), (const float*)P(sv, PTC_BLK_O_ST_N2), 1, g[PTC_BLK_S_DX1], PTC_F32, g[PTC_BLK_S_DA], nullptr, nullptr,
                       M<float>(g, PTC_BLK_G_G_N2), M<float>(g, PTC_BLK_G_BE_N2), ws, wb, s));
  // 6'. proj: weight / bias gradients over the inverse table, datt through the table of the padded slots
  calls[2] = PtcWgradCall{P(sv, PTC_BLK_O_ATT), np, g[PTC_BLK_S_DA], (const int32_t*)P(in, PTC_BLK_P_T_PROJ_FWD), n, 1, c, c, dt, M<float>(g, PTC_BLK_G_W_PROJ),
                          M<float>(g, PTC_BLK_G_B_PROJ), wgws(2), W.wg[2]};
  RUN(ptc_spconv_fwd(g[PTC_BLK_S_DA], n, P(in, PTC_BLK_P_WT_PROJ), nullptr, (const int32_t*)P(in, PTC_BLK_P_T_PROJ_BWD), np, 1, c, c, dt, g[PTC_BLK_S_DATT], s));
  // 5'. attention
  RUN(ptc_attn_varlen_bwd(P(sv, PTC_BLK_O_QKV), P(sv, PTC_BLK_O_ATT), g[PTC_BLK_S_DATT], (const float*)P(sv, PTC_BLK_O_LSE), (const int32_t*)P(in, PTC_BLK_P_CU), n_seq,
                          np, H, patch, fv[PTC_BLK_F_SCALE], dt, g[PTC_BLK_S_DQKV], ws, wb, s));
  // 4'. qkv: weight / bias gradients over the gather table, dy1 through the two-slot table (a point sits in <= 2 padded slots)
  calls[3] = PtcWgradCall{P(sv, PTC_BLK_O_Y1), n, g[PTC_BLK_S_DQKV], (const int32_t*)P(in, PTC_BLK_P_T_QKV_FWD), np, 1, c, 3 * c, dt, M<float>(g, PTC_BLK_G_W_QKV),
                          M<float>(g, PTC_BLK_G_B_QKV), wgws(3), W.wg[3]};
  RUN(ptc_spconv_fwd(g[PTC_BLK_S_DQKV], np, P(in, PTC_BLK_P_WT_QKV), nullptr, (const int32_t*)P(in, PTC_BLK_P_T_QKV_BWD), n, 2, 3 * c, c, dt, g[PTC_BLK_S_DY1], s));
  // 3'. x1 = x0 + LN_cpe(lin), y1 = norm1(x1):  dx0 = dx1 + LN_n1'(dy1);  dlin = LN_cpe'(dx0)
  RUN(ptc_add_norm_bwd(M<float>(g, PTC_BLK_S_DX1), g[PTC_BLK_S_DY1], dt, (const float*)P(sv, PTC_BLK_O_X1), P(sv, PTC_BLK_O_LIN), dt, nullptr, n, c,
                       (const float*)P(in, PTC_BLK_P_G_CPE), (const float*)P(sv, PTC_BLK_O_ST_CPE), 1, (const float*)P(in, PTC_BLK_P_G_N1),
                       (const float*)P(sv, PTC_BLK_O_ST_N1), 1, g[PTC_BLK_G_X0], a_dt, g[PTC_BLK_S_DLIN], M<float>(g, PTC_BLK_G_G_CPE), M<float>(g, PTC_BLK_G_BE_CPE),
                       M<float>(g, PTC_BLK_G_G_N1), M<float>(g, PTC_BLK_G_BE_N1), ws, wb, s));
  // 2'. the Linear of the positional encoding
  calls[4] = PtcWgradCall{P(sv, PTC_BLK_O_CONV), n, g[PTC_BLK_S_DLIN], nullptr, n, 1, c, c, dt, M<float>(g, PTC_BLK_G_W_LIN), M<float>(g, PTC_BLK_G_B_LIN), wgws(4), W.wg[4]};
  RUN(ptc_spconv_fwd(g[PTC_BLK_S_DLIN], n, P(in, PTC_BLK_P_WT_LIN), nullptr, nullptr, n, 1, c, c, dt, g[PTC_BLK_S_DCONV], s));
  // 1'. the convolution: weight gradient, bias gradient, input gradient over the same table with mirrored weights
  RUN(ptc_spconv_wgrad_blk_deferred(P(in, PTC_BLK_P_XC), n, g[PTC_BLK_S_DCONV], nbr, P(in, PTC_BLK_P_BLK_TAB), (const int32_t*)P(in, PTC_BLK_P_BLK_HID),
                                    (const int32_t*)P(in, PTC_BLK_P_BLK_HCNT), (const int32_t*)P(in, PTC_BLK_P_BLK_NOVF), (int)iv[PTC_BLK_I_BLK_BM],
                                    (int)iv[PTC_BLK_I_BLK_HCAP], n, 27, c, c, dt, M<float>(g, PTC_BLK_G_W_CONV), wgws(5), W.wg[5], s, &jobs[5]));
  RUN(ptc_column_sum(g[PTC_BLK_S_DCONV], n, c, dt, M<float>(g, PTC_BLK_G_B_CONV), ws, wb, s));
  RUN(ptc_spconv_fwd_blk(g[PTC_BLK_S_DCONV], n, P(in, PTC_BLK_P_WT_CONV), nullptr, nbr, P(in, PTC_BLK_P_BLK_TAB), (const int32_t*)P(in, PTC_BLK_P_BLK_HID),
                         (const int32_t*)P(in, PTC_BLK_P_BLK_HCNT), (int)iv[PTC_BLK_I_BLK_BM], (int)iv[PTC_BLK_I_BLK_HCAP], n, 27, c, c, dt, g[PTC_BLK_G_XC], s));
  // the five Linear weight gradients in one grouped launch (their operands -- saved activations and the scratch gradients above -- are
  // all still in place), then the split-K reductions of all six weight gradients in one launch
  if (mlp_fused) RUN(ptc_spconv_wgrad_group(calls + 2, 3, jobs + 2, s));     // (jobs[0], jobs[1]: the MLP kernel's own partials)
  else RUN(ptc_spconv_wgrad_group(calls, 5, jobs, s));
  RUN(ptc_wgrad_reduce_jobs(jobs, 6, s));
  return PTC_OK;
}
