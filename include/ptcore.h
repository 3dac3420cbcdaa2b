/*
 * ptcore.h -- C-ABI of libptcore.so, the MI355X (gfx950) engine behind Pointcept's
 * SparseUNet voxel-convolution and PTv3 serialized-attention hot paths.
 *
 * Boundary contract (level B4):
 *   - extern "C", plain pointers + sizes, no torch / C++ types.
 *   - OWNERSHIP: the caller allocates every input, output and workspace buffer (device
 *     memory unless a parameter is documented as HOST) and passes raw pointers.  The
 *     library never allocates or frees device memory and keeps no mutable global state.
 *   - ERRORS: every entry point returns PTC_OK (0) or a negative PTC_E* code; it never
 *     throws across the ABI and never calls exit().  ptc_last_error() returns a
 *     thread-local message for the last failing call on this thread.
 *   - STREAMS: all work is enqueued on the hipStream_t passed as `stream` (the Python
 *     host passes torch.cuda.current_stream().cuda_stream).  No implicit synchronisation,
 *     no use of the legacy default stream (contrast the reference's in-repo ops:
 *     libs/pointops/src/knn_query/knn_query_cuda_kernel.cu:111 launches on stream 0).
 *   - DTYPES: feature tensors carry a ptc_dtype tag; accumulation is always fp32.
 *     Index tensors are int64 where the reference's Python API exposes int64
 *     (serialized_order / inverse / pad / unpad / cluster), int32 for rulebook tables.
 *
 * Each entry point cites the reference interface it replaces (file:line under
 * /root/reference).  Third-party operators that the reference calls but does not vendor
 * (spconv, flash_attn, torch_scatter) are cited by their call sites.
 */
#ifndef PTCORE_H
#define PTCORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ptc_stream_t; /* hipStream_t */

enum ptc_status {
  PTC_OK = 0,
  PTC_EINVAL = -1,       /* bad argument (null pointer, negative size, bad enum) */
  PTC_EUNSUPPORTED = -2, /* shape / dtype combination not implemented */
  PTC_EHIP = -3,         /* a HIP runtime call failed (see ptc_last_error) */
  PTC_EWORKSPACE = -4    /* workspace too small */
};

enum ptc_dtype { PTC_F32 = 0, PTC_F16 = 1, PTC_BF16 = 2 };

/* serialization orders, pointcept/models/utils/serialization/default.py:8-24 */
enum ptc_order { PTC_ORDER_Z = 0, PTC_ORDER_Z_TRANS = 1, PTC_ORDER_HILBERT = 2, PTC_ORDER_HILBERT_TRANS = 3 };

/* segment reductions, torch_scatter.segment_csr as called at
 * pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py:416-421 */
enum ptc_reduce { PTC_REDUCE_SUM = 0, PTC_REDUCE_MEAN = 1, PTC_REDUCE_MAX = 2, PTC_REDUCE_MIN = 3 };

const char* ptc_version(void);
const char* ptc_last_error(void);

/* ------------------------------------------------------------------------------------------
 * A. Serialization keys.
 * Replaces serialization.encode() for all requested orders in ONE pass:
 *   pointcept/models/utils/serialization/default.py:8-24 (dispatch, x/y swap, batch prefix)
 *   pointcept/models/utils/serialization/z_order.py:66-101 (Morton LUT interleave)
 *   pointcept/models/utils/serialization/hilbert.py:91-192 (Skilling transform + Gray decode)
 * as called from Point.serialization, pointcept/models/utils/structure.py:89-92.
 *   grid_coord : [n,3] int64 (coord_is_i64=1) or int32 (0), non-negative, < 2^depth
 *   batch      : [n] int64 or NULL (no batch prefix)
 *   orders     : HOST array of k ptc_order values
 *   code_out   : [k,n] int64,  code = batch << 3*depth | key
 * depth in [1,16]  (structure.py:82 asserts depth <= 16).
 * ------------------------------------------------------------------------------------------ */
int ptc_serialize_encode(const void* grid_coord, int coord_is_i64, const int64_t* batch, int64_t n,
                         int depth, const int* orders, int k, int64_t* code_out, ptc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * B. Key sort -> order / inverse.
 * Replaces torch.argsort(code) + the inverse-permutation scatter of
 *   pointcept/models/utils/structure.py:93-100 and
 *   pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py:399-406.
 * Stable LSD radix sort (8-bit digits) of k independent rows of n int64 keys, restricted
 * to bits [begin_bit, end_bit).  Canonical tie order = ascending original index
 * (SURVEY Appendix A.3).  inverse may be NULL.  keys are not modified.
 *   order   : [k,n] int64, keys[r][order[r][i]] ascending in i
 *   inverse : [k,n] int64, inverse[r][order[r][i]] = i
 * ------------------------------------------------------------------------------------------ */
size_t ptc_sort_keys_workspace_bytes(int64_t n, int k);
int ptc_sort_keys(const int64_t* keys, int64_t n, int k, int begin_bit, int end_bit, int64_t* order,
                  int64_t* inverse, void* workspace, size_t workspace_bytes, ptc_stream_t stream);

/* Exclusive prefix sum of n int32 values into int64 (building block, exported for tests). */
size_t ptc_exclusive_scan_workspace_bytes(int64_t n);
int ptc_exclusive_scan_i32(const int32_t* in, int64_t n, int64_t* out, void* workspace,
                           size_t workspace_bytes, ptc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * C. Patch padding maps.
 * Replaces SerializedAttention.get_padding_and_inverse,
 *   pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py:114-170
 * (python loop with host syncs) by one launch.
 *   offset      : [B] int64 device, cumulative scene sizes
 *   patch       : K
 *   n_pad       : total padded length N' (host computes it from a host copy of offset)
 *   n           : offset[B-1]
 *   n_seq       : number of sequences (host computed)
 *   pad         : [n_pad] int64   padded slot -> sorted rank
 *   unpad       : [n]     int64   sorted rank -> padded slot
 *   cu_seqlens  : [n_seq+1] int32 sequence starts + n_pad
 *   dup         : [n] int64 or NULL: second padded slot holding sorted rank r, or -1
 *                 (engine-side extra: makes the backward of the padded gather a pure gather)
 * ------------------------------------------------------------------------------------------ */
int ptc_patch_pad_maps(const int64_t* offset, int B, int patch, int64_t n, int64_t n_pad,
                       int64_t n_seq, int64_t* pad, int64_t* unpad, int32_t* cu_seqlens,
                       int64_t* dup, ptc_stream_t stream);

/* C2. int32 gather tables of ONE serialization order for the gather-fused qkv / proj GEMMs (the index algebra of
 * SerializedAttention.forward, ptv3m1:184-188,216, and of its backward) in one launch:
 *   t_qkv_fwd [n_pad] = order[pad[s]];  t_qkv_bwd [2][n] = (unpad[inverse[p]], dup[inverse[p]]);
 *   t_proj_fwd [n] = unpad[inverse[p]];  t_proj_bwd [n_pad] = point of slot s if s is its primary slot, else -1.
 * order / inverse: one row of serialized_order / serialized_inverse; pad / unpad / dup from ptc_patch_pad_maps. */
int ptc_attn_tables(const int64_t* order, const int64_t* inverse, const int64_t* pad, const int64_t* unpad,
                    const int64_t* dup, int64_t n, int64_t n_pad, int32_t* t_qkv_fwd, int32_t* t_qkv_bwd,
                    int32_t* t_proj_fwd, int32_t* t_proj_bwd, ptc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * D. Serialized pooling maps.
 * Replaces the index arithmetic of SerializedPooling.forward,
 *   pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py:383-396
 * (torch.unique(sorted, inverse, counts) + torch.sort(cluster) + cumsum) using the already
 * sorted row 0 of the parent codes.
 *   code0   : [n] int64  parent serialized_code[0]
 *   order0  : [n] int64  parent serialized_order[0]
 *   shift   : 3 * pooling_depth
 * Phase 1 (ptc_pool_maps_count): writes cluster[n] (= pooling_inverse, ascending
 * code0>>shift numbering) and *n_cluster_out (device int64).  Host reads n_cluster.
 * Phase 2 (ptc_pool_maps_fill): idx_ptr[n_cluster+1] int64, head[n_cluster] int64.
 * `indices` of the reference (points sorted by cluster) IS order0 (same cluster order;
 * member order inside a cluster is immaterial for max/mean/sum/min).
 * ------------------------------------------------------------------------------------------ */
size_t ptc_pool_maps_workspace_bytes(int64_t n);
int ptc_pool_maps_count(const int64_t* code0, const int64_t* order0, int64_t n, int shift,
                        int64_t* cluster, int64_t* n_cluster_out, void* workspace,
                        size_t workspace_bytes, ptc_stream_t stream);
int ptc_pool_maps_fill(const int64_t* order0, const int64_t* cluster, int64_t n, int64_t n_cluster,
                       int64_t* idx_ptr, int64_t* head, ptc_stream_t stream);
/* code_out[r][c] = code_in[r][head[c]] >> shift, r < k   (ptv3m1:383,398) */
int ptc_pool_child_codes(const int64_t* code_in, int64_t n, int k, const int64_t* head,
                         int64_t n_cluster, int shift, int64_t* code_out, ptc_stream_t stream);

/* Cluster counts of every pooling level from the stage-0 codes, in one pass: counts[l][b] = number of distinct values of
 * code0 >> shifts[l] among the points of scene b (b = code >> batch_shift), i.e. the child point counts that
 * `torch.unique(code >> depth*3)` (ptv3m1:384-390) will produce at level l.  Lets the host size every level of the
 * hierarchy with ONE device->host copy instead of two per SerializedPooling.  code0 [n] = any row of serialized_code,
 * order0 [n] = its sorting permutation; n_levels <= 8; counts [n_levels][n_batch] int64 (zeroed here). */
int ptc_pool_level_counts(const int64_t* code0, const int64_t* order0, int64_t n, int batch_shift, int n_batch,
                          const int* shifts, int n_levels, int64_t* counts, ptc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * E. Row gathers / segmented reductions (feature traffic around the attention and pooling).
 *   ptc_gather_rows      : out[i,:] = src[idx[i],:] (+ src[idx2[i],:] if idx2 && idx2[i]>=0);
 *                          idx[i] < 0 writes zeros.  Replaces feat[order] / feat[inverse] /
 *                          feat[pooling_inverse] (ptv3m1:188,216,478) and their backward.
 *   ptc_segment_csr_fwd  : out[s,:] = reduce_{r in [indptr[s],indptr[s+1])} src[perm ? perm[r] : r,:]
 *                          arg_out (int32 [n_seg,c], may be NULL; only for MAX/MIN) = winning SOURCE
 *                          row (first arg-max in segment order); -1 for an empty segment.
 *                          Replaces torch_scatter.segment_csr(src[indices], idx_ptr, reduce)
 *                          (ptv3m1:416-421) with the gather fused.
 *   ptc_segment_csr_bwd  : grad_src[perm[r],:] from grad_out (SUM/MEAN: broadcast (/count);
 *                          MAX/MIN: grad on the arg row, zero on the other member rows).  Every
 *                          member row of every segment is written; rows outside all segments are
 *                          left untouched (caller zero-fills if perm is not a full cover).
 * ------------------------------------------------------------------------------------------ */
int ptc_gather_rows(const void* src, int64_t n_src, const int64_t* idx, const int64_t* idx2,
                    int64_t n_out, int c, int dtype, void* out, ptc_stream_t stream);
/* out[i] = addend[i] + src[idx[i]] (idx[i] < 0: addend[i]) in one pass, one rounding in the feature dtype -- SerializedUnpooling's
 * `parent.feat + point.feat[inverse]` (pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py:478).  c a multiple of a 16-byte lane. */
int ptc_gather_rows_add(const void* src, int64_t n_src, const int64_t* idx, const void* addend, int64_t n_out, int c, int dtype,
                        void* out, ptc_stream_t stream);
int ptc_segment_csr_fwd(const void* src, const int64_t* perm, const int64_t* indptr, int64_t n_seg,
                        int c, int dtype, int reduce, void* out, int32_t* arg_out,
                        ptc_stream_t stream);
int ptc_segment_csr_bwd(const void* grad_out, const int64_t* perm, const int64_t* indptr,
                        const int32_t* arg, int64_t n_seg, int64_t n_src, int c, int dtype,
                        int reduce, void* grad_src, ptc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * F. Rulebook (kernel maps) for sparse convolution.  Replaces what spconv builds inside
 * SubMConv3d / SparseConv3d / SparseInverseConv3d on the first use of an indice_key
 * (call sites: ptv3m1:278-284,499-506; spconv_unet_v1m1_base.py:43-68,114-121,137-144,173-179).
 * Canonical form: gather tables nbr[kv][n_out] int32, -1 = no input.
 *   indices : [n,4] int32 (batch,x,y,z)  (SparseConvTensor.indices, structure.py:139-143)
 * Voxel table (opaque, caller-allocated, 64-byte aligned, ptc_hash_table_bytes(n) bytes): open addressing
 * over 64-byte BUCKETS of 2x2x2 voxels { block key, 8 row indices }; row = LOWEST row index with that
 * coordinate (duplicate voxels after Mix3D: lowest index wins).
 * ------------------------------------------------------------------------------------------ */
int64_t ptc_hash_table_size(int64_t n);   /* number of buckets */
size_t ptc_hash_table_bytes(int64_t n);
int ptc_hash_build(const int32_t* indices, int64_t n, void* table, size_t table_bytes, ptc_stream_t stream);
/* SubM: nbr[k][i] = row at coord_i + delta_k, k = ((d0+r)*ks + (d1+r))*ks + (d2+r), r = ks/2,
 * (d0,d1,d2) applied to (x,y,z) = indices columns 1..3 (cross-correlation convention). */
int ptc_rulebook_subm(const int32_t* indices, int64_t n, int ksize, const void* table, size_t table_bytes,
                      int32_t* nbr, ptc_stream_t stream);
/* Strided k=2,s=2 (SparseConv3d at spconv_unet_v1m1_base.py:137-144).
 * Phase 1: out_of_in[n_in] int32 = coarse row of each fine row, numbering = ascending
 *          (batch, Morton code of (x>>1, y>>1, z>>1): bit i of x at 3i+2, y at 3i+1, z at 3i) -- consecutive coarse
 *          rows are spatial neighbours (round 6; spconv's own numbering is hash-insertion order);
 *          *n_out_dev (device int64) = number of coarse rows.
 *          coord_bits: every (coord>>1) < 2^coord_bits; batch_bits: every batch < 2^batch_bits
 *          (host derives both from spatial_shape / batch_size; they only bound the sort passes).
 * Phase 2: out_indices[n_out,4] int32, nbr_down[8][n_out] int32 (gather table of the down conv,
 *          k = (x&1)*4 + (y&1)*2 + (z&1)), nbr_up[8][n_in] int32 (gather table of the inverse conv). */
size_t ptc_rulebook_down_workspace_bytes(int64_t n_in);
int ptc_rulebook_down_count(const int32_t* indices, int64_t n_in, int coord_bits, int batch_bits,
                            int32_t* out_of_in, int64_t* n_out_dev, void* workspace,
                            size_t workspace_bytes, ptc_stream_t stream);
int ptc_rulebook_down_fill(const int32_t* indices, int64_t n_in, const int32_t* out_of_in,
                           int64_t n_out, int32_t* out_indices, int32_t* nbr_down, int32_t* nbr_up,
                           ptc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * G. Sparse convolution compute: out[o,:] = bias + sum_k W_k . in[nbr[k][o],:]
 * One implicit-GEMM kernel (MFMA, gathered operand, W_k slices staged through LDS) serves
 * forward, dgrad and the inverse convolution -- they differ only in the table and in how the
 * host lays out the (tiny) weight tensor:
 *   weight : [c_out][kv][c_in] row-major = the spconv-2.x parameter layout
 *            [C_out,k0,k1,k2,C_in] flattened.
 *   forward : table nbr, weight as stored.
 *   dgrad   : SubM: the SAME table with weight' = W.permute(ci,k,co).flip(k) (mirror, A.6);
 *             down conv: table nbr_up, weight' = W.permute(ci,k,co); inverse conv: nbr_down.
 *   bias    : fp32 [c_out] or NULL.   in/out: [n, c] of `dtype`.
 *   c_in % 8 == 0 and c_out % 16 == 0 (the host zero-pads the 6-channel stem input).
 *   PTC_F32 uses the exact-f32 MFMA (16x16x4), PTC_F16/PTC_BF16 the 16x16x32 MFMA; fp32 accumulate.
 * wgrad: dw[c_out][kv][c_in] (fp32) = sum_o dout[o,:]^T (x) in[nbr[k][o],:]
 *        dbias[c_out] (fp32, may be NULL) = sum_o dout[o,:]  (fused: one extra MFMA against ones)
 * nbr == NULL (only with kv == 1) is the IDENTITY table: the same kernels then are the dense
 * row-wise GEMMs of the nn.Linear layers on the path (ptv3m1:97-98,240-244,286,366,463-464;
 * pointcept/models/default.py:52): out = in W^T + b, dX = dY W, dW = dY^T X, db = colsum(dY),
 * tall-skinny shapes (N ~ 8e5 rows, 32..512 channels) that are HBM-bound.
 * ------------------------------------------------------------------------------------------ */
int ptc_spconv_fwd(const void* in, int64_t n_in, const void* weight, const float* bias,
                   const int32_t* nbr, int64_t n_out, int kv, int c_in, int c_out, int dtype,
                   void* out, ptc_stream_t stream);
/* Block-local form of a submanifold 3^3 gather table (csrc/blocks.hip) and the convolution that uses it (csrc/conv7.h).
 * Replaces nothing new in the reference: it is a faster way to run the SAME SubMConv3d call sites as ptc_spconv_fwd
 * (ptv3m1:278-284; spconv_unet_v1m1_base.py:49-68) when rows are in a spatial (curve) order.
 *   ptc_rulebook_blocks: for every block of bm = 128 consecutive output rows of nbr [27][n]
 *       hid  [n_blocks][hcap] int32 = the distinct input rows named by nbr[:, block], ascending, padded with the last one to
 *                                     a multiple of 16 entries
 *       hcnt [n_blocks]       int32 = how many (-1 = more than hcap: block served through the global table)
 *       tab  [2][n_blocks][28][32][4] uint16 (ptc_rulebook_blocks_tab_bytes(n) bytes, 16-byte aligned): at [v][b][k][r][t]
 *                                     the byte offset, inside the convolution kernel's LDS image of the block's rows, of the
 *                                     first 16 bytes of row nbr[k][128 b + 32 t + r] -- v = 0: 128-byte rows (64 channels),
 *                                     slot * 128 + ((((slot >> 1) & 1) << 2) | ((slot >> 2) & 3)) * 16; v = 1: 64-byte rows (32 channels), slot * 64 +
 *                                     ((slot >> 2) & 3) * 16, slot = position in the block's list; none = hcap * row bytes;
 *                                     row k = 27: the tap masks in its first 52 bytes -- four uint32, bit k of word t set when
 *                                     tile t (rows 32 t .. 32 t + 31) has a neighbour at tap k, then their OR, then eight
 *                                     uint32, bit k of word s set when one of the rows {32 t + 4 s + q : t, q = 0..3} has one
 *                                     (the 16-row MFMA steps of ptc_spconv_wgrad_blk) -- the rest "none"
 *       *n_overflow (device int32)  = number of blocks with hcnt = -1; ptc_spconv_wgrad_blk reads it ON THE DEVICE to choose its
 *                                     kernel (no host copy is ever made)
 *     hcap a multiple of 16, < 512.
 *   ptc_spconv_fwd_blk: same result as ptc_spconv_fwd(in, ..., nbr, ...) up to the fp32 rounding of a different summation
 *       order, with the input rows of a block staged once in LDS and the weights held in registers.  16-bit dtypes,
 *       kv = 27, c_in = c_out in {32, 64}, bm = 128, hcap = 416, n_in = n_out >= 4096; any other shape is forwarded to
 *       ptc_spconv_fwd.
 *   ptc_spconv_wgrad_blk (round 4): dw of the same convolution (= ptc_spconv_wgrad(in, dout, nbr) without the bias gradient, up to
 *       the fp32 rounding of a different summation order; bit-reproducible) with the whole gradient held in MFMA accumulators
 *       of persistent workgroups and both operands built from the block's LDS images by transposing reads (csrc/wgrad7.h).
 *       Same shape range as ptc_spconv_fwd_blk; other shapes are forwarded to ptc_spconv_wgrad.  When *n_overflow != 0 (a block
 *       whose halo did not fit) the call is served by the global-gather kernel instead -- decided on the device, both kernels are
 *       always enqueued.  workspace >= ptc_spconv_wgrad_blk_workspace_bytes.
 *       Replaces spconv's weight-gradient kernel behind SubMConv3d (ptv3m1:278-284, spunet:49-68) under autograd. */
size_t ptc_rulebook_blocks_tab_bytes(int64_t n);
int ptc_rulebook_blocks(const int32_t* nbr, int kv, int64_t n, int bm, int hcap, void* tab, int32_t* hid,
                        int32_t* hcnt, int32_t* n_overflow, ptc_stream_t stream);
int ptc_spconv_fwd_blk(const void* in, int64_t n_in, const void* weight, const float* bias, const int32_t* nbr,
                       const void* tab, const int32_t* hid, const int32_t* hcnt, int bm, int hcap, int64_t n_out,
                       int kv, int c_in, int c_out, int dtype, void* out, ptc_stream_t stream);
size_t ptc_spconv_wgrad_blk_workspace_bytes(int64_t n_out, int kv, int c_in, int c_out);
int ptc_spconv_wgrad_blk(const void* in, int64_t n_in, const void* dout, const int32_t* nbr, const void* tab, const int32_t* hid,
                         const int32_t* hcnt, const int32_t* n_overflow, int bm, int hcap, int64_t n_out, int kv, int c_in,
                         int c_out, int dtype, float* dw, void* workspace, size_t workspace_bytes, ptc_stream_t stream);
/* Dense row-wise GEMM out = in W^T + b with an MLP epilogue fused (PTv3 MLP, ptv3m1:225-248: fc1 -> GELU -> fc2):
 *   epilogue 1 : out = h (the pre-activation, saved for the backward), aux_out = GELU(h)         [fc1 forward]
 *   epilogue 2 : out = (in W^T) * GELU'(aux_in), aux_in = h [n, c_out]                          [fc2 input gradient]
 * weight [c_out][c_in] in `dtype` (bf16 / f16), c_in <= 256 or a multiple of 64 from 128 up (ptc_linear_supported_ex); GELU = erf form, fp32. */
int ptc_linear_supported_ex(int c_in, int c_out, int dtype);
int ptc_linear_fwd_ex(const void* in, int64_t n, const void* weight, const float* bias, int c_in, int c_out, int dtype,
                      int epilogue, const void* aux_in, void* out, void* aux_out, ptc_stream_t stream);
/* A Linear with the residual joint of Block.forward (ptv3m1:318-338) in its EPILOGUE (round 4): the two joints whose branch operand is a
 * bare Linear output -- `proj` of SerializedAttention (ptv3m1:219) and `fc2` of the MLP (:246) --
 *     z = a + row_scale * (in W^T + b),   y = LN_B(z) (normB = 1; statB [2][n_out] = mean / rstd) or the cast of z (normB = 0),
 * in place of ptc_spconv_fwd(nbr = NULL | kv = 1 table) followed by ptc_add_norm_fwd: the 16-bit GEMM output never reaches memory.  The
 * arithmetic of the joint is ptc_add_norm_fwd's (same lane mapping and statement order): bit-identical results.  16-bit features,
 * c_out in {32, 64, 128}, c_in in {32, 64, 128, 256}; a / z fp32 [n_out, c_out]; y (may be NULL) in the feature dtype; nbr as in
 * ptc_spconv_fwd for kv = 1 (the inverse serialization table of `proj`) or NULL. */
/* ptc_linear_norm_joint_fwd: the joint of the positional encoding, x1 = x0 + LN_cpe(Linear(conv)), y1 = norm1(x1) (ptv3m1:318-323): the branch
 * operand is normalised first (statA [2][n_out]); the Linear's own output is written too (u_out, feature dtype): the backward of LN_cpe
 * reads it.  Same kernel, same bit-identity with ptc_spconv_fwd + ptc_add_norm_fwd(normA = 1). */
int ptc_linear_joint_supported(int c_in, int c_out, int dtype);
int ptc_linear_norm_joint_fwd(const void* in, int64_t n_in, const void* weight, const float* bias, int64_t n_out, int c_in, int c_out, int dtype,
                              const float* gA, const float* bA, float epsA, const void* a, int a_dtype, const float* gB, const float* bB, float epsB,
                              int normB, void* u_out, float* z, void* y, float* statA, float* statB, ptc_stream_t stream);
int ptc_linear_joint_fwd(const void* in, int64_t n_in, const void* weight, const float* bias, const int32_t* nbr, int64_t n_out, int c_in,
                         int c_out, int dtype, const float* a, const float* row_scale, const float* gB, const float* bB, float epsB,
                         int normB, float* z, void* y, float* statB, ptc_stream_t stream);

/* The whole MLP of a PT-v3m1 Block in ONE kernel per direction (round 6, csrc/mlp.hip), C = 32 | 64, hidden = 4 C, 16-bit features:
 *   ptc_mlp_fwd : m = GELU(x W1^T + b1) W2^T + b2 (pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py:225-248, drop = 0), and --
 *                 with a != NULL -- the residual joint behind it (:334-337): z = a + row_scale * m (fp32), y = cast(z) (feature dtype, may
 *                 be NULL); with a == NULL: y = m.  The hidden tensor never reaches memory (per 64-channel chunk it goes from fc1's
 *                 accumulators through GELU into fc2's operand registers).  Bit-identical to ptc_linear_fwd_ex(epilogue 1) followed by
 *                 ptc_linear_joint_fwd / ptc_spconv_fwd.  w1 [4C][C], w2 [C][4C] in `dtype`; b1 / b2 fp32 or NULL; x [n][C].
 *   ptc_mlp_bwd : from dm = d loss / d m [n][C] (feature dtype) and the forward's x: dx = ((dm W2) * GELU'(h)) W1 [n][C], dw1 [4C][C], db1 [4C],
 *                 dw2 [C][4C], db2 [C] (fp32; db1 / db2 may be NULL), h = x W1^T + b1 RECOMPUTED per tile (the forward's bits).  w2t = W2^T
 *                 [4C][C] in `dtype`.  Deterministic: per-workgroup partial sums in `workspace` (ptc_mlp_bwd_workspace_bytes), summed in a fixed
 *                 order.  dx equals the split kernels' bits; the weight gradients differ from theirs in summation order only. */
int ptc_mlp_supported(int c, int dtype);
int ptc_mlp_fwd(const void* x, int64_t n, int c, int dtype, const void* w1, const float* b1, const void* w2, const float* b2,
                const float* a, const float* row_scale, float* z, void* y, ptc_stream_t stream);
size_t ptc_mlp_bwd_workspace_bytes(int64_t n, int c);
int ptc_mlp_bwd(const void* dm, const void* x, int64_t n, int c, int dtype, const void* w1, const float* b1, const void* w2t, void* dx,
                float* dw1, float* db1, float* dw2, float* db2, void* workspace, size_t workspace_bytes, ptc_stream_t stream);

size_t ptc_spconv_wgrad_workspace_bytes(int64_t n_out, int kv, int c_in, int c_out);
int ptc_spconv_wgrad(const void* in, int64_t n_in, const void* dout, const int32_t* nbr,
                     int64_t n_out, int kv, int c_in, int c_out, int dtype, float* dw, float* dbias,
                     void* workspace, size_t workspace_bytes, ptc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * G1a. Evaluation tail of the segmentation step in one pass.
 * Replaces pred = seg_logits.max(1)[1]; pred = pred[inverse]; intersection_and_union_gpu(pred, segment, K, ignore)
 * (pointcept/engines/hooks/evaluator.py:139-152, pointcept/utils/misc.py:57-69: three torch.histc).
 *   logits [n_rows, c] of `dtype`, row pitch `row_stride` elements (or NULL and pred [m] int64 given instead);
 *   inverse [m] int64 (evaluated point -> logit row) or NULL (identity); target [m] int64;
 *   hist3k [3][k] int64: intersection, area_output (ignored targets excluded), area_target;  union = [1] + [2] - [0].
 *   arg-max ties -> lowest class.  k <= 1024.
 * ------------------------------------------------------------------------------------------ */
int ptc_seg_eval_hist(const void* logits, int dtype, int64_t row_stride, int c, const int64_t* pred, const int64_t* inverse,
                      const int64_t* target, int64_t m, int64_t n_rows, int k, int64_t ignore_index, int64_t* hist3k,
                      ptc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * G1b. 3-axis rotary embedding on point tokens, IN PLACE (LitePT / PT-v3m3 "PointROPE").
 * Replaces libs/pointrope/kernels.cu:19-100 behind pointrope.pointrope(tokens, positions, base, F0)
 * (libs/pointrope/pointrope.cpp:51-67; call sites pointcept/models/litept/litept_v1.py:27-59,240-241).
 *   tokens [n_tokens, H, D] of `dtype` (fp32 / f16 / bf16), D % 6 == 0, Q = D/6, head = [u_x|v_x|u_y|v_y|u_z|v_z] (Q each);
 *   positions [n_tokens, 3] int64;  f = pos[a] * (fwd / base^(i/Q));  (u, v) <- (u cos f - v sin f, v cos f + u sin f).
 *   The backward of the operator is the same call with fwd = -F0.
 * ------------------------------------------------------------------------------------------ */
int ptc_rope3d(void* tokens, int dtype, const int64_t* positions, int64_t n_tokens, int H, int D, float base, float fwd,
               ptc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * G1c. PT-v3m3 `Point3DRoPE` on the packed qkv rows (pointcept/models/point_transformer_v3/point_transformer_v3m3_utonia.py:43-102,
 * call site :274-323: q, k = rope(q, k, coord[order]); flash-attn reads stack([q', k', v]).to(bf16)).
 *   src [n_tokens, slabs, H, D] of src_dtype -> dst (same shape) of dst_dtype: the first `rot_slabs` slabs (q, k) are rotated, the
 *   rest (v) converted / copied; src == dst (one dtype) rotates in place.  xyz [n_tokens, 3] fp32 = the continuous coordinates of
 *   the rows (after the training-time shift / jitter / rescale of :276-300); inv_freq [D/6] fp32 = the module's buffer (:53-56).
 *   Per head: three chunks of D/3, element i of a chunk's first half pairs with element i of its second half (rotate_half, :75-77):
 *   (u, v) <- (u cos f - sign v sin f, v cos f + sign u sin f),  f = xyz[a] * inv_freq[i];  fp32 math.
 *   sign = +1 forward, -1 for the gradient (the rotation is orthogonal).  D % 6 == 0.
 * ------------------------------------------------------------------------------------------ */
int ptc_rope3d_xyz(const void* src, int src_dtype, void* dst, int dst_dtype, const float* xyz, const float* inv_freq,
                   int64_t n_tokens, int slabs, int rot_slabs, int H, int D, float sign, ptc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * G2. LayerNorm over channels of [n, c] features (nn.LayerNorm inside every PTv3 Block:
 * ptv3m1:286 cpe.2, :289 norm1, :305 norm2).  ptc_layer_norm_supported(c): 1 = c in {32,64,128,256,512} (c / 8 lanes per row; the
 * widths the fused joints of G3 also take), 2 = any other even c <= 1024 (one wave per row: the widths of PT-v3m2 / m3 / LitePT,
 * configs/sonata :45, configs/utonia :21, litept_v1.py:601), 0 = unsupported (odd or wider).
 *   fwd: y = (x-mean)*rstd*gamma + beta, statistics in fp32; y dtype may differ from x
 *        (fp32 out under autocast, or bf16 out when the consumer is a bf16 GEMM);
 *        mean[n], rstd[n] fp32 are saved for the backward.
 *   bwd: dx (x's dtype), dgamma[c], dbeta[c] fp32 (either may be NULL).
 * ------------------------------------------------------------------------------------------ */
int ptc_layer_norm_supported(int c);
int ptc_layer_norm_fwd(const void* x, int64_t n, int c, int in_dtype, const float* gamma,
                       const float* beta, float eps, void* y, int out_dtype, float* mean, float* rstd,
                       ptc_stream_t stream);
size_t ptc_layer_norm_bwd_workspace_bytes(int64_t n, int c);
int ptc_layer_norm_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean,
                       const float* rstd, const float* gamma, int64_t n, int c, void* dx,
                       float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                       ptc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * G3. Fused residual joint of the PTv3 Block (ptv3m1:318-338; `x + LN(cpe)` then `norm1`,
 * `x + drop_path(attn)` then `norm2`, `x + drop_path(mlp)` then the cast feeding the next conv):
 *     z = a + row_scale * f(u),  f = LayerNorm_A (normA != 0) or identity      [n,c] fp32
 *     y = LayerNorm_B(z) (normB != 0) or y = z, stored as y_dtype; y may be NULL
 *   u: [n,c] bf16 or f32 (branch output), a: [n,c] fp32 (residual stream) or bf16 (first block of a stage: the
 *   pooling / unpooling output; da is then written as bf16 too, a_dtype / da_dtype), row_scale: [n] fp32 or
 *   NULL (DropPath keep-mask / keep_prob, per POINT: timm DropPath on [N,C]).
 *   statA / statB: [2][n] fp32 (mean, rstd) saved for the backward when the norm is present.
 * Backward: da = dz_in + LN_B'(dy) (or + dy);  du = row_scale * LN_A'(da) (or row_scale * da);
 *   dz_in (fp32) and dy may each be NULL (not both); affine gradients fp32 [c], any may be NULL.
 * ------------------------------------------------------------------------------------------ */
int ptc_add_norm_fwd(const void* u, int u_dtype, const void* a, int a_dtype, const float* row_scale, int64_t n, int c,
                     const float* gA, const float* bA, float epsA, int normA, const float* gB,
                     const float* bB, float epsB, int normB, float* z, void* y, int y_dtype,
                     float* statA, float* statB, ptc_stream_t stream);
size_t ptc_add_norm_bwd_workspace_bytes(int64_t n, int c);
int ptc_add_norm_bwd(const float* dz_in, const void* dy, int dy_dtype, const float* z, const void* u,
                     int u_dtype, const float* row_scale, int64_t n, int c, const float* gA,
                     const float* statA, int normA, const float* gB, const float* statB, int normB,
                     void* da, int da_dtype, void* du, float* dgA, float* dbA, float* dgB, float* dbB,
                     void* workspace, size_t workspace_bytes, ptc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * H. Serialized (variable-length, fixed-window) attention, head_dim 16.
 * Replaces flash_attn.flash_attn_varlen_qkvpacked_func as called at ptv3m1:208-214:
 *   qkv        : [total, 3, H, 16] bf16 (or f16), packed
 *   cu_seqlens : [n_seq+1] int32
 *   out        : [total, H, 16] same dtype ; lse : [H, total] fp32 (natural-log sum-exp)
 * non-causal, no dropout, softmax_scale given.  max_seqlen <= 1024.
 * Backward recomputes P from (q,k,lse):  dqkv [total,3,H,16].
 * ------------------------------------------------------------------------------------------ */
int ptc_attn_varlen_fwd(const void* qkv, const int32_t* cu_seqlens, int64_t n_seq, int64_t total,
                        int H, int max_seqlen, float softmax_scale, int dtype, void* out,
                        float* lse, ptc_stream_t stream);
/* workspace of the backward: delta[H,total] fp32 */
size_t ptc_attn_varlen_bwd_workspace_bytes(int64_t total, int H);
int ptc_attn_varlen_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                        const int32_t* cu_seqlens, int64_t n_seq, int64_t total, int H,
                        int max_seqlen, float softmax_scale, int dtype, void* dqkv,
                        void* workspace, size_t workspace_bytes, ptc_stream_t stream);

/* The same operator WITH attention dropout (`dropout_p = self.attn_drop if self.training else 0`, ptv3m1:212; flash-attn's semantics:
 * softmax over all keys, then every probability dropped with probability dropout_p in [0, 1) and the survivors scaled by
 * 1 / (1 - dropout_p); lse is that of the undropped scores).  The mask is a pure function of (seed, sequence, head, query, key) --
 * a 32-bit integer hash compared with dropout_p 2^32 (csrc/attention_drop.h: ad_unit_key / ad_keep; restated in oracle/ops.py) --
 * regenerated by the backward from the same seed; it is NOT flash-attn's Philox stream.  head_dim 16, bf16 or f16 tensors. */
int ptc_attn_varlen_dropout_fwd(const void* qkv, const int32_t* cu_seqlens, int64_t n_seq, int64_t total, int H, int max_seqlen,
                                float softmax_scale, int dtype, float dropout_p, uint64_t seed, void* out, float* lse,
                                ptc_stream_t stream);
int ptc_attn_varlen_dropout_bwd(const void* qkv, const void* out, const void* dout, const float* lse, const int32_t* cu_seqlens,
                                int64_t n_seq, int64_t total, int H, int max_seqlen, float softmax_scale, int dtype,
                                float dropout_p, uint64_t seed, void* dqkv, void* workspace, size_t workspace_bytes,
                                ptc_stream_t stream);

/* Same operator for head_dim 17..64 (PT-v3m3 / LitePT: head_dim 18, flash_attn_varlen_qkvpacked_func as called at
 * pointcept/models/point_transformer_v3/point_transformer_v3m3_utonia.py:353-359 and pointcept/models/litept/litept_v1.py:244-256):
 *   qkv [total, 3, H, head_dim] bf16 packed, out [total, H, head_dim], lse [H, total] fp32, dqkv like qkv.
 * The window's operands live in LDS, which bounds max_seqlen: head_dim <= 32: 1024, <= 48: 672, <= 64: 512
 * (ptc_attn_varlen_hd_supported returns 1 / 0; the calls return PTC_EUNSUPPORTED outside that range).
 * The backward's workspace is ptc_attn_varlen_bwd_workspace_bytes(total, H). */
int ptc_attn_varlen_hd_supported(int head_dim, int max_seqlen);
int ptc_attn_varlen_hd_fwd(const void* qkv, const int32_t* cu_seqlens, int64_t n_seq, int64_t total,
                           int H, int head_dim, int max_seqlen, float softmax_scale, int dtype,
                           void* out, float* lse, ptc_stream_t stream);
int ptc_attn_varlen_hd_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                           const int32_t* cu_seqlens, int64_t n_seq, int64_t total, int H,
                           int head_dim, int max_seqlen, float softmax_scale, int dtype, void* dqkv,
                           void* workspace, size_t workspace_bytes, ptc_stream_t stream);
/* ... with the 3-D rotary embedding of q and k FUSED into the attention prologue / epilogue (round 4), head_dim 18 = every
 * PT-v3m3 / LitePT configuration: qkv holds the UN-rotated rows the qkv Linear wrote; the kernels rotate K while staging it and q when
 * loading it -- pair (x[6 a + i], x[6 a + 3 + i]) by the angle xyz[t][a] * inv_freq[i], fp32, rounded to the operand dtype: Point3DRoPE of
 * point_transformer_v3m3_utonia.py:58-101,303-323 / libs/pointrope/kernels.cu:19-75 with integer positions -- and apply the inverse
 * rotation to dq / dk in the backward's epilogue, so dqkv is the gradient of the un-rotated rows.  Replaces the separate pass
 * ptc_rope3d_xyz (one read + one write of q and k per direction).  xyz [total, 3] fp32 = positions of the padded, serialized rows,
 * inv_freq [3] fp32.  ptc_attn_varlen_hd_rope_supported: 1 for head_dim 18 with a window the LDS-resident form holds. */
int ptc_attn_varlen_hd_rope_supported(int head_dim, int max_seqlen);
int ptc_attn_varlen_hd_rope_fwd(const void* qkv, const int32_t* cu_seqlens, const float* xyz, const float* inv_freq, int64_t n_seq,
                                int64_t total, int H, int head_dim, int max_seqlen, float softmax_scale, int dtype, void* out, float* lse,
                                ptc_stream_t stream);
int ptc_attn_varlen_hd_rope_bwd(const void* qkv, const void* out, const void* dout, const float* lse, const int32_t* cu_seqlens,
                                const float* xyz, const float* inv_freq, int64_t n_seq, int64_t total, int H, int head_dim, int max_seqlen,
                                float softmax_scale, int dtype, void* dqkv, void* workspace, size_t workspace_bytes, ptc_stream_t stream);


/* Window attention with PTv3's relative position bias (the non-flash branch with enable_rpe=True,
 * pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py:29-48,104-112,190-206), head_dim 16:
 *   softmax(scale q k^T + sum_a rpe_table[a R + clamp(gc_i[a] - gc_j[a], -B, B) + B][h]) v     (i = query, j = key)
 *   grid_coord : [total, 3] int32, rows in the SAME (serialized, padded) order as qkv; values in [0, 2^16)
 *   rpe_table  : [3 R, H] fp32, R = 2 pos_bnd + 1 (RPE.rpe_table);  d_rpe_table: same shape, overwritten
 * The bias is evaluated per pair in the tile loop; nothing of size L^2 is materialised.  d_rpe_table is accumulated in
 * 2^-24 fixed point with 64-bit integer atomics (order-independent: bit-reproducible, unlike the reference's float
 * atomicAdd) and converted at the end; workspace = ptc_attn_rpe_bwd_workspace_bytes. */
int ptc_attn_rpe_fwd(const void* qkv, const int32_t* cu_seqlens, const int32_t* grid_coord,
                     const float* rpe_table, int pos_bnd, int64_t n_seq, int64_t total, int H,
                     int max_seqlen, float softmax_scale, int dtype, void* out, float* lse,
                     ptc_stream_t stream);
size_t ptc_attn_rpe_bwd_workspace_bytes(int64_t total, int H, int pos_bnd);
int ptc_attn_rpe_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                     const int32_t* cu_seqlens, const int32_t* grid_coord, const float* rpe_table,
                     int pos_bnd, int64_t n_seq, int64_t total, int H, int max_seqlen,
                     float softmax_scale, int dtype, void* dqkv, float* d_rpe_table, void* workspace,
                     size_t workspace_bytes, ptc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * I. Ends of the step.
 * ptc_coord_max: out3[a] = max_i grid_coord[i][a] (0 for n == 0).  Replaces the reductions behind
 *   `int(self.grid_coord.max() + 1).bit_length()` (pointcept/models/utils/structure.py:74) and
 *   `torch.max(self.grid_coord, dim=0).values` (:136-138).  grid_coord [n,3] int64 / int32, >= 0.
 * ptc_cross_entropy_*: nn.CrossEntropyLoss(ignore_index) of pointcept/models/losses/misc.py as called at
 *   pointcept/models/default.py:78-84 on seg_logits [n, c] (`logits` may be a strided view: row_stride
 *   in elements; dtype = ptc_dtype of the logits).
 *   fwd: lse[n] fp32; partial[2*b], partial[2*b+1] = (sum of -log p[target], number of counted rows) of
 *        workgroup b, b < ptc_cross_entropy_partials(n).  loss = sum(partial[::2]) / sum(partial[1::2]).
 *   bwd: dlogits[i][j] = scale[0] * (softmax(logits[i])[j] - [j == target[i]]) for counted rows, else 0;
 *        `scale` is a DEVICE scalar (= grad_loss / count).
 * ------------------------------------------------------------------------------------------ */
int ptc_coord_max(const void* grid_coord, int coord_is_i64, int64_t n, int64_t* out3, ptc_stream_t stream);
int64_t ptc_cross_entropy_partials(int64_t n);
int ptc_cross_entropy_fwd(const void* logits, int64_t row_stride, const int64_t* target, int64_t n, int c, int dtype,
                          int64_t ignore_index, float* lse, float* partial, ptc_stream_t stream);
int ptc_cross_entropy_bwd(const void* logits, int64_t row_stride, const int64_t* target, const float* lse,
                          const float* scale, int64_t n, int c, int dtype, int64_t ignore_index, void* dlogits,
                          int64_t drow_stride, ptc_stream_t stream);

/* Lovasz-Softmax loss, multiclass, classes = "present", whole batch (per_image = False), with its gradient.
 * Replaces LovaszLoss(mode="multiclass", ignore_index, loss_weight) of pointcept/models/losses/lovasz.py:209-260
 * (_lovasz_softmax_flat :118-146, _lovasz_grad :22-33, _flatten_probas :149-166), the second criterion of
 * configs/scannet/semseg-pt-v3m1-0-base.py:49-52, called from pointcept/models/default.py:78-84.
 *   logits [n, c] (row_stride in elements, dtype = ptc_dtype), target [n] int64 (ignore_index / out of range = not
 *   counted), c <= 64.  loss[0] (fp32, DEVICE) = mean over the classes present in the counted labels of
 *   <sorted errors, Jaccard steps>; dlogits [n, c] fp32 contiguous = d loss / d logits (0 for uncounted rows).
 *   One segmented radix sort of the [c, n] error matrix; reductions in a fixed order (bit-reproducible).
 *   Ties between equal errors are ordered by ascending point index (the loss does not depend on it). */
size_t ptc_lovasz_softmax_workspace_bytes(int64_t n, int c);
int ptc_lovasz_softmax(const void* logits, int64_t row_stride, const int64_t* target, int64_t n, int c, int dtype,
                       int64_t ignore_index, float* loss, float* dlogits, void* workspace, size_t workspace_bytes,
                       ptc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K. Voxelisation front end (GridSample), for point clouds that already live on the device.
 * Replaces the numpy prologue of GridSample.__call__, pointcept/datasets/transform.py:867-875 with
 * fnv_hash_vec (:997-1011):  grid = floor(coord / grid_size) (float64 division, as numpy promotes it),
 * min_coord3 = grid.min(0), grid_coord = grid - min_coord3 (written in place, int64 [n,3]),
 * key[n] = FNV64 of the three shifted coordinates (bit pattern of the reference's uint64, stored as int64).
 * argsort / unique / inverse / count of the keys: ptc_sort_keys (bits [0,64)) + ptc_pool_maps_count(shift 0) / _fill.
 * ------------------------------------------------------------------------------------------ */
int ptc_voxel_keys(const float* coord, int64_t n, double grid_size, int64_t* grid_coord, int64_t* min_coord3,
                   int64_t* key, ptc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * L. pointops subset (offset-batched point sets; offsets int32 cumulative ends as the reference passes `offset.int()`).
 * ptc_knn_query: knn_query_cuda of libs/pointops/src/knn_query/knn_query_cuda_kernel.cu:60-108 behind
 *   libs/pointops/functions/query.py:7-26 -- for each of the m query points (new_xyz, scene given by new_offset) the
 *   nsample (<= 128) nearest points of the same scene in xyz: idx [m, nsample] int32 (row index into xyz, -1 = scene
 *   has fewer points), dist [m, nsample] fp32 = sqrt(squared distance) (1e5 for empty slots), ascending; equal
 *   distances ordered by ascending index.
 * ptc_farthest_point_sampling: farthest_point_sampling_cuda of src/sampling/sampling_cuda_kernel.cu:15-122 behind
 *   functions/sampling.py:7-24 -- idx [new_offset[b-1]] int32; scene s contributes new_offset[s]-new_offset[s-1]
 *   picks, the first one its first point; tmp [n] fp32 scratch (initialised here).  Equal distances: lowest index.
 * ------------------------------------------------------------------------------------------ */
int ptc_knn_query(const float* xyz, const int32_t* offset, const float* new_xyz, const int32_t* new_offset, int b,
                  int64_t n, int64_t m, int nsample, int32_t* idx, float* dist, ptc_stream_t stream);
int ptc_farthest_point_sampling(const float* xyz, const int32_t* offset, const int32_t* new_offset, int b, int64_t n,
                                float* tmp, int32_t* idx, ptc_stream_t stream);
/* Ball query and random ball query (libs/pointops/src/ball_query/ball_query_cuda_kernel.cu:59-123,
 * src/random_ball_query/random_ball_query_cuda_kernel.cu:58-108; wrappers libs/pointops/functions/query.py:29-113).
 * In range: d2 <= 1e-5 or min_radius^2 <= d2 < max_radius^2, inside the query's own scene.
 *   order == NULL : all in-range points (first 2048 in index order) sorted by (distance, index); <= nsample -> all + padding
 *                   (idx -1, dist2 1e10), else the nsample ranks int(i * float(count) / nsample).
 *   order != NULL : int32 [n], a permutation of every scene's point indices: the first nsample in-range points in that order.
 * Outputs idx [m, nsample] int32, dist2 [m, nsample] fp32 (SQUARED distances; the python wrapper takes the root). */
int ptc_ball_query(const float* xyz, const int32_t* offset, const float* new_xyz, const int32_t* new_offset, const int32_t* order,
                   int b, int64_t n, int64_t m, int nsample, float min_radius, float max_radius, int32_t* idx, float* dist2,
                   ptc_stream_t stream);
/* Edge-list operators (round 4): grouping, interpolation, aggregation, subtraction of libs/pointops as operations on the edge list
 * E = {(t, s) -> j = idx[t * nsample + s]} (idx int32 [m, nsample], -1 = empty slot: gathers zeros, receives no gradient).  fp32 rows,
 * like the reference kernels.  Every gradient of a GATHERED operand is a segmented sum over the edges sorted by source row (fixed
 * order, no atomics) where the reference scatters with atomicAdd.
 *   ptc_edge_rows_fwd    per-edge rows into out[e * out_stride + out_col0 .. + c):
 *        mode 0  src[j]          grouping_forward_cuda, libs/pointops/src/grouping/grouping_cuda_kernel.cu:5-15 (functions/grouping.py:8-25)
 *        mode 1  a[t] - src[j]   subtraction_forward_cuda, src/subtraction/subtraction_cuda_kernel.cu:5-17 (functions/subtraction.py:8-22)
 *        mode 2  src[j] - a[t]   (0 for j < 0) the relative coordinates of grouping(with_xyz=True), functions/grouping.py:44-68
 *   ptc_edge_reduce_fwd  per-target sums over the nsample edges of a row, out [m, c]:
 *        mode 0  sum_s w[t,s] src[j]                               interpolation_forward_cuda, src/interpolation/interpolation_cuda_kernel.cu:6-20
 *        mode 1  sum_s (src[j][c] + pos[t,s,c]) w[t,s,c % w_c]     aggregation_forward_cuda, src/aggregation/aggregation_cuda_kernel.cu:5-21
 *        mode 2  sum_s pos[(t,s) * pos_stride + pos_col0 + c]      gradient of subtraction's input1 (subtraction_cuda_kernel.cu:19-33);
 *                idx != NULL: empty slots (idx < 0) are left out of the sum
 *   ptc_edge_csr_keys / ptc_edge_csr_ptr  the edge CSR by source row: keys [E] int64 (j, or n_src for empty slots) for ptc_sort_keys over
 *        bits [0, bit_length(n_src)), then indptr [n_src + 1] int64 from the sorted order (indptr[j] = edges with key < j)
 *   ptc_edge_scatter_bwd grad_src[j] = sum over the edges of source row j, ascending edge index, of coef(e) g[row(e)]:
 *        mode 0 grouping (coef 1, row e), 1 subtraction (coef -1, row e), 2 interpolation (coef w[e], row e / nsample),
 *        mode 3 aggregation (coef w[e, c % w_c], row e / nsample); g rows may be a column window (g_stride, g_col0) of wider rows
 *        -- grouping_backward_cuda / interpolation_backward_cuda / aggregation_backward_cuda / subtraction_backward_cuda without atomics
 *   ptc_aggregation_edge_bwd  grad_position [m, nsample, c] and grad_weight [m, nsample, w_c] of aggregation
 *        (aggregation_cuda_kernel.cu:23-39; one producer per element) */
int ptc_edge_rows_fwd(int mode, const float* src, const float* a, const int32_t* idx, int64_t n_edges, int nsample, int c,
                      int64_t n_src, float* out, int64_t out_stride, int out_col0, ptc_stream_t stream);
int ptc_edge_reduce_fwd(int mode, const float* src, const float* pos, int64_t pos_stride, int pos_col0, const float* w,
                        const int32_t* idx, int64_t m, int nsample, int c, int w_c, int64_t n_src, float* out, ptc_stream_t stream);
int ptc_edge_csr_keys(const int32_t* idx, int64_t n_edges, int64_t n_src, int64_t* keys, ptc_stream_t stream);
int ptc_edge_csr_ptr(const int64_t* keys, const int64_t* order, int64_t n_edges, int64_t n_src, int64_t* indptr, ptc_stream_t stream);
int ptc_edge_scatter_bwd(int mode, const int64_t* order, const int64_t* indptr, const float* g, int64_t g_stride, int g_col0,
                         const float* w, int nsample, int c, int w_c, int64_t n_src, float* grad_src, ptc_stream_t stream);
int ptc_aggregation_edge_bwd(const float* src, const float* pos, const float* w, const int32_t* idx, const float* g, int64_t m,
                             int nsample, int c, int w_c, int64_t n_src, float* grad_pos, float* grad_w, ptc_stream_t stream);
/* Pair-list attention of libs/pointops (round 5; PTv2's grouped vector attention): a pair m joins row ia[m] of a [n_a, g, c] with row
 * ib[m] of b [n_b, g, c] (fp32; indices outside their range contribute zeros).
 *   ptc_pair_dot_weighted  out[m, g] = sum_c a[ia[m], g, c] b[ib[m], g, c] (w ? w[c] : 1)
 *        = attention_relation_step_forward_cuda (libs/pointops/src/attention/attention_cuda_kernel.cu:9-25; functions/attention.py:11-62)
 *          with (a, b, w) = (query, key, weight), and d weight of the fusion step (:64-82) with (a, b, w) = (grad_output, value, NULL)
 *   ptc_pair_segment_sum   A[n, g, c] = sum over the pairs e of row n -- order / indptr: the CSR of the pairs by that row index
 *        (ptc_edge_csr_*), ascending pair index -- of s[e, g] b[oidx[e], g, c];  out = A (w ? w[c] : 1);  prod = self[n, g, c] A if prod
 *        = attention_fusion_step_forward_cuda (:46-62) with (s, b) = (weight, value), CSR by index_target, oidx = index_refer;
 *          d query / d key of the relation step (:27-45) with (s, b, w) = (grad_output, key | query, weight) -- prod summed over its rows
 *          is d weight --; d value of the fusion step with (s, b) = (weight, grad_output), CSR by index_refer, oidx = index_target.
 *        One producer per output element, fixed order: no atomicAdd (the reference's sums differ from run to run). */
int ptc_pair_dot_weighted(const float* a, const float* b, const float* w, const int32_t* ia, const int32_t* ib, int64_t m, int64_t n_a,
                          int64_t n_b, int g, int c, float* out, ptc_stream_t stream);
int ptc_pair_segment_sum(const float* s, const float* b, const float* w, const float* self, const int64_t* order, const int64_t* indptr,
                         const int32_t* oidx, int64_t n_rows, int64_t n_b, int g, int c, float* out, float* prod, ptc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * J. BatchNorm1d over the rows of [n, c] features with the following activation fused:
 *    y = act((x - mean) * rstd * gamma + beta),  act in {0 none, 1 GELU (erf), 2 ReLU}.
 * Replaces `nn.BatchNorm1d(eps=1e-3, momentum=0.01)` + `nn.GELU()` of PTv3's Embedding / SerializedPooling /
 * SerializedUnpooling (ptv3m1:485-515,371-444,447-482; norm built at :581) and BatchNorm1d + ReLU of
 * SpUNet (spconv_unet_v1m1_base.py:49-68,110-121,137-146,173-181).  Per-GPU statistics (sync_bn = False).
 *   training != 0 : batch statistics (biased variance for normalisation), running statistics updated as
 *                   r = (1 - momentum) * r + momentum * stat (unbiased variance), both may be NULL;
 *   training == 0 : running statistics are used.
 *   save_mean / save_rstd [c] fp32 feed the backward, which RECOMPUTES the pre-activation from x:
 *   dx (x's dtype), dgamma / dbeta [c] fp32 (may be NULL).  c % 8 == 0 (16-bit) or c % 4 == 0 (fp32),
 *   c <= 2048 / 1024 (ptc_batch_norm_supported).  Reductions run in a fixed order: bit-reproducible.
 * ------------------------------------------------------------------------------------------ */
int ptc_batch_norm_supported(int c, int dtype);
size_t ptc_batch_norm_workspace_bytes(int64_t n, int c);
int ptc_batch_norm_act_fwd(const void* x, int64_t n, int c, int dtype, const float* gamma, const float* beta, float eps,
                           float momentum, int training, float* running_mean, float* running_var, int act, void* y,
                           int y_dtype, float* save_mean, float* save_rstd, void* workspace, size_t workspace_bytes,
                           ptc_stream_t stream);
/* out[c] (fp32) = sum_i x[i][c]: bias gradients (`grad.sum(0)`), one read of x, fixed summation order.
 * Same shape limits and workspace size as the BatchNorm entry points. */
int ptc_column_sum(const void* x, int64_t n, int c, int dtype, float* out, void* workspace, size_t workspace_bytes,
                   ptc_stream_t stream);
int ptc_batch_norm_act_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* gamma, const float* beta,
                           const float* save_mean, const float* save_rstd, int64_t n, int c, int training, int act,
                           void* dx, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                           ptc_stream_t stream);
/* ptc_batch_norm_add_act_{fwd,bwd}: y = act(BN(x) + res) -- the tail of the reference's residual block (spconv_unet_v1m1_base.py:79-83:
 * `out = self.bn2(out); out = out.replace_feature(out.features + self.proj(residual).features); out = self.relu(out)`), three elementwise
 * passes there, the BatchNorm's own apply pass here.  res / dres have x's dtype; statistics are those of x alone. */
int ptc_batch_norm_add_act_fwd(const void* x, const void* res, int64_t n, int c, int dtype, const float* gamma, const float* beta, float eps,
                               float momentum, int training, float* running_mean, float* running_var, int act, void* y, int y_dtype,
                               float* save_mean, float* save_rstd, void* workspace, size_t workspace_bytes, ptc_stream_t stream);
int ptc_batch_norm_add_act_bwd(const void* dy, int dy_dtype, const void* x, const void* res, int x_dtype, const float* gamma, const float* beta,
                               const float* save_mean, const float* save_rstd, int64_t n, int c, int training, int act, void* dx, void* dres,
                               float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, ptc_stream_t stream);

/* Backward-pass weight layouts of every layer in one launch (16-bit elements):
 *   dst[ci][j][co] = src[co][m(j)][ci]; desc [n][6] int64 = { src, dst, c_out, taps_src, c_in, taps_dst | mode << 32 } (device),
 *   mode 0: m(j) = taps_src - 1 - j (the mirrored-weight dgrad of a submanifold convolution, `functional._SparseConv`; plain
 *   transpose at taps = 1: nn.Linear), 1: m(j) = 0 (one matrix repeated taps_dst times), 2: m(j) = j;
 *   prefix [n+1] int64 = first output element of every entry, total = prefix[n]. */
int ptc_weight_layouts(const int64_t* desc, const int64_t* prefix, int n, int64_t total, ptc_stream_t stream);

/* fp32 master weights -> 16-bit shadows of every stale weight in one launch (what torch._foreach_copy_ does in ~29 multi-tensor
 * launches after each optimizer step; functional._CastCache):  desc [n][3] int64 = { src (fp32), dst (dst_dtype), n_elements } (device),
 * prefix [n+1] int64 = first UNIT of 8 elements of every entry (entries rounded up to whole units), total_units = prefix[n];
 * dst_dtype = PTC_BF16 | PTC_F16; round-to-nearest-even like Tensor.to(). */
int ptc_cast_many(const int64_t* desc, const int64_t* prefix, int n, int64_t total_units, int dst_dtype, ptc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * M. libs/pointops2: the pair-list attention operators of Stratified Transformer (fp32), reference wrappers
 *    libs/pointops2/functions/pointops.py:93-961, kernels libs/pointops2/src/{attention,attention_v2,rpe,rpe_v2}/.
 *    (kNN / FPS / grouping / interpolation / subtraction / aggregation of pointops2 = sections L and E.)
 *    Pairs m = 0..M-1: query i0[m], key i1[m]; offsets [Nq+1] = CSR of the pairs by query or NULL (then i0 may be unsorted);
 *    T(m,h,c) = sum_a table[rel_idx[m,a], h, c, a], table [L, H, d, 3], rel_idx [M, 3] int32.
 *  ptc_pair_dot_fwd      : out[m,h] = [with_qk] q[i0].k[i1] + [table_q] q[i0].Tq + [table_k] k[i1].Tk
 *        = attention_step1(_v2) (with_qk, no tables), dot_prod_with_idx (table_q only), dot_prod_with_idx_v2 / _v3 (both tables)
 *  ptc_pair_dot_bwd      : dq (segment loops with offsets, else atomics), dk, d tables (atomics; any may be NULL)
 *  ptc_pair_aggregate_fwd: out[n,h,c] = sum_{pairs of n} attn[m,h] (v[i1[m],h,c] + [table_v] Tv)
 *        = attention_step2(_v2), attention_step2_with_rel_pos_value(_v2)
 *  ptc_pair_aggregate_bwd: dattn [M,H], dv [Nv,H,d], d table (atomics)
 * ------------------------------------------------------------------------------------------ */
int ptc_pair_dot_fwd(const float* q, const float* k, const int32_t* i0, const int32_t* i1, const float* table_q,
                     const float* table_k, const int32_t* rel_idx, int with_qk, int64_t M, int H, int d, float* out,
                     ptc_stream_t stream);
int ptc_pair_dot_bwd(const float* grad_out, const float* q, const float* k, const int32_t* i0, const int32_t* offsets,
                     const int32_t* i1, const float* table_q, const float* table_k, const int32_t* rel_idx, int with_qk,
                     int64_t M, int64_t Nq, int64_t Nk, int64_t L, int H, int d, float* dq, float* dk, float* dtable_q,
                     float* dtable_k, ptc_stream_t stream);
int ptc_pair_aggregate_fwd(const float* attn, const float* v, const int32_t* i0, const int32_t* offsets, const int32_t* i1,
                           const float* table_v, const int32_t* rel_idx, int64_t M, int64_t Nq, int H, int d, float* out,
                           ptc_stream_t stream);
int ptc_pair_aggregate_bwd(const float* grad_out, const float* attn, const float* v, const int32_t* i0, const int32_t* i1,
                           const float* table_v, const int32_t* rel_idx, int64_t M, int64_t Nv, int64_t L, int H, int d,
                           float* dattn, float* dv, float* dtable_v, ptc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * N. One call per PT-v3m1 Block and direction (csrc/block_exec.hip): the kernel sequence of Block.forward (ptv3m1:318-338:
 *    x += LN(Linear(SubMConv(x))); x += DropPath(Attn(LN(x))); x += DropPath(MLP(LN(x)))) and of its backward, enqueued from C
 *    instead of from ~16 Python autograd Functions (the host-side launch floor).  Same entry points, same operands, same
 *    order as the Python-composed path: bit-identical.  bf16 GEMM operands, head_dim 16, c <= 256 (MLP hidden 4 c), pre-norm.
 *    Arguments are index tables (the enums below; a Python caller reads the names from this header):
 *      iv  int64 [PTC_BLK_I_COUNT]  sizes and dtype tags         fv  float [PTC_BLK_F_COUNT]  softmax scale, LayerNorm eps
 *      in  [PTC_BLK_P_COUNT]        inputs: activations, 16-bit weight shadows ([c_out][taps][c_in]), fp32 biases / LayerNorm
 *                                   parameters, tables, DropPath row scales (NULL = none); for the backward also the incoming
 *                                   gradients (either may be NULL) and the transposed weight layouts ([c_in][taps'][c_out],
 *                                   ptc_weight_layouts: conv mirrored, qkv repeated twice)
 *      out [PTC_BLK_O_COUNT]        forward outputs = what the backward reads back (`sv`): caller-allocated
 *      g   [PTC_BLK_GS_COUNT]       backward outputs: gradients (G_*; fp32 parameters' gradients, G_X0 in the dtype of x0, G_XC
 *                                   16-bit) and scratch (S_*), caller-allocated; workspace >= ptc_ptv3_block_workspace_bytes
 * ------------------------------------------------------------------------------------------ */
#define PTC_BLK_ABI 2
enum { PTC_BLK_I_ABI, PTC_BLK_I_N, PTC_BLK_I_NPAD, PTC_BLK_I_NSEQ, PTC_BLK_I_C, PTC_BLK_I_HEADS, PTC_BLK_I_DTYPE, PTC_BLK_I_A_DTYPE,
       PTC_BLK_I_PATCH, PTC_BLK_I_BLK_BM, PTC_BLK_I_BLK_HCAP, PTC_BLK_I_COUNT };
enum { PTC_BLK_F_SCALE, PTC_BLK_F_EPS_CPE, PTC_BLK_F_EPS_N1, PTC_BLK_F_EPS_N2, PTC_BLK_F_COUNT };
enum { PTC_BLK_P_X0, PTC_BLK_P_XC, PTC_BLK_P_NBR, PTC_BLK_P_BLK_TAB, PTC_BLK_P_BLK_HID, PTC_BLK_P_BLK_HCNT, PTC_BLK_P_BLK_NOVF, PTC_BLK_P_T_QKV_FWD,
       PTC_BLK_P_T_QKV_BWD, PTC_BLK_P_T_PROJ_FWD, PTC_BLK_P_T_PROJ_BWD, PTC_BLK_P_CU, PTC_BLK_P_RS1, PTC_BLK_P_RS2,
       PTC_BLK_P_W_CONV, PTC_BLK_P_B_CONV, PTC_BLK_P_W_LIN, PTC_BLK_P_B_LIN, PTC_BLK_P_G_CPE, PTC_BLK_P_BE_CPE, PTC_BLK_P_G_N1,
       PTC_BLK_P_BE_N1, PTC_BLK_P_W_QKV, PTC_BLK_P_B_QKV, PTC_BLK_P_W_PROJ, PTC_BLK_P_B_PROJ, PTC_BLK_P_G_N2, PTC_BLK_P_BE_N2,
       PTC_BLK_P_W_FC1, PTC_BLK_P_B_FC1, PTC_BLK_P_W_FC2, PTC_BLK_P_B_FC2,
       PTC_BLK_P_DZ3, PTC_BLK_P_DYB3, PTC_BLK_P_WT_CONV, PTC_BLK_P_WT_LIN, PTC_BLK_P_WT_QKV, PTC_BLK_P_WT_PROJ, PTC_BLK_P_WT_FC1,
       PTC_BLK_P_WT_FC2, PTC_BLK_P_COUNT };
enum { PTC_BLK_O_CONV, PTC_BLK_O_LIN, PTC_BLK_O_X1, PTC_BLK_O_Y1, PTC_BLK_O_ST_CPE, PTC_BLK_O_ST_N1, PTC_BLK_O_QKV, PTC_BLK_O_ATT,
       PTC_BLK_O_LSE, PTC_BLK_O_A, PTC_BLK_O_X2, PTC_BLK_O_Y2, PTC_BLK_O_ST_N2, PTC_BLK_O_H, PTC_BLK_O_ACT, PTC_BLK_O_M, PTC_BLK_O_X3,
       PTC_BLK_O_XB3, PTC_BLK_O_COUNT };
enum { PTC_BLK_G_X0, PTC_BLK_G_XC, PTC_BLK_G_W_CONV, PTC_BLK_G_B_CONV, PTC_BLK_G_W_LIN, PTC_BLK_G_B_LIN, PTC_BLK_G_G_CPE,
       PTC_BLK_G_BE_CPE, PTC_BLK_G_G_N1, PTC_BLK_G_BE_N1, PTC_BLK_G_W_QKV, PTC_BLK_G_B_QKV, PTC_BLK_G_W_PROJ, PTC_BLK_G_B_PROJ,
       PTC_BLK_G_G_N2, PTC_BLK_G_BE_N2, PTC_BLK_G_W_FC1, PTC_BLK_G_B_FC1, PTC_BLK_G_W_FC2, PTC_BLK_G_B_FC2,
       PTC_BLK_S_DX2, PTC_BLK_S_DM, PTC_BLK_S_DH, PTC_BLK_S_DY2, PTC_BLK_S_DX1, PTC_BLK_S_DA, PTC_BLK_S_DATT, PTC_BLK_S_DQKV,
       PTC_BLK_S_DY1, PTC_BLK_S_DLIN, PTC_BLK_S_DCONV, PTC_BLK_GS_COUNT };
int ptc_ptv3_block_abi(void);
/* 1: the executor runs the Block's MLP on ptc_mlp_fwd / ptc_mlp_bwd for this (c, dtype) -- then O_H, O_ACT, O_M and S_DH are never
 * touched and need not be allocated (round 6; PTC_BLK_MLP_FUSED=0 in the environment keeps the split kernels) */
int ptc_ptv3_block_mlp_fused(int c, int dtype);
size_t ptc_ptv3_block_workspace_bytes(int64_t n, int64_t n_pad, int c, int heads);
int ptc_ptv3_block_fwd(const int64_t* iv, const float* fv, const void* const* in, void* const* out, ptc_stream_t stream);
int ptc_ptv3_block_bwd(const int64_t* iv, const float* fv, const void* const* in, const void* const* sv, void* const* g,
                       void* workspace, size_t workspace_bytes, ptc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PTCORE_H */
