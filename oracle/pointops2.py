"""TEST INFRASTRUCTURE (only tests/ may import this).  CPU restatement in plain torch of the pair-list attention operators of
libs/pointops2 (Stratified Transformer), written from the torch formulations the reference's own operator tests compare their CUDA
kernels against:
    libs/pointops2/functions/test_attention_op_step1.py      (attn_flat = (query[index_0] * key[index_1]).sum(-1))
    libs/pointops2/functions/test_attention_op_step2.py:31-33 (x = scatter_sum(attn.unsqueeze(-1) * value[index_1], index_0, dim_size=N))
    libs/pointops2/functions/test_relative_pos_encoding_op_step1.py:32-36
        (rel_pos_encoding = table_x[rel_x] + table_y[rel_y] + table_z[rel_z];  output = (query[index] * rel_pos_encoding).sum(-1))
    libs/pointops2/functions/test_relative_pos_encoding_op_step1_v3.py:62-66  (v3 = dot_prod(q, ...) + dot_prod(k, ...))
    libs/pointops2/functions/test_relative_pos_encoding_op_step2.py
        (output = scatter_sum(attn.unsqueeze(-1) * (value[index_1] + rel_pos_encoding), index_0, dim_size=N))
and from the kernels' index arithmetic (table [L, h, hdim, 3], libs/pointops2/src/rpe_v2/relative_pos_encoding_cuda_kernel_v2.cu:248-282).
Differentiable (autograd gives the gradients the CUDA backward kernels implement); run in fp64 for a tight reference."""
import torch


def rel_pos_encoding(table: torch.Tensor, rel_idx: torch.Tensor) -> torch.Tensor:
    """T[m, h, c] = sum_a table[rel_idx[m, a], h, c, a]"""
    r = rel_idx.long()
    return table[r[:, 0], :, :, 0] + table[r[:, 1], :, :, 1] + table[r[:, 2], :, :, 2]


def attention_step1(q, k, index0, index1):
    return (q[index0.long()] * k[index1.long()]).sum(-1)


def dot_prod_with_idx(q, index, table, rel_idx):
    return (q[index.long()] * rel_pos_encoding(table, rel_idx)).sum(-1)


def dot_prod_with_idx_v3(q, index_q, k, index_k, table_q, table_k, rel_idx):
    return dot_prod_with_idx(q, index_q, table_q, rel_idx) + dot_prod_with_idx(k, index_k, table_k, rel_idx)


def attention_step2(attn, v, index0, index1, n_q, table=None, rel_idx=None):
    val = v[index1.long()]
    if table is not None:
        val = val + rel_pos_encoding(table, rel_idx)
    contrib = attn.unsqueeze(-1) * val
    out = torch.zeros((n_q,) + tuple(v.shape[1:]), dtype=contrib.dtype)
    return out.index_add(0, index0.long(), contrib)


def offsets_of(index0_sorted: torch.Tensor, n: int) -> torch.Tensor:
    counts = torch.bincount(index0_sorted.long(), minlength=n)
    return torch.cat([counts.new_zeros(1), torch.cumsum(counts, 0)]).to(torch.int32)
