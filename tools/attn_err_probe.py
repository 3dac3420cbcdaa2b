import sys, numpy as np, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from oracle import ops as oops
from pointcept_amd import ops
cuda = torch.device('cuda:0')
for lens, H in [([48, 48, 17], 2), ([1, 2, 31, 32, 33, 65], 3), ([1024, 700, 33], 4), ([1024] * 3, 4)]:
    g = torch.Generator().manual_seed(sum(lens) + H)
    T = sum(lens)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    qkv = (torch.randn(T, 3, H, 16, generator=g) * 1.5).to(torch.bfloat16)
    scale = 16 ** -0.5
    out, lse = ops.attn_varlen_fwd(qkv.to(cuda), cu.to(cuda), max(lens), scale)
    dout = torch.randn(T, H, 16, generator=g).to(torch.bfloat16)
    dqkv = ops.attn_varlen_bwd(qkv.to(cuda), out, dout.to(cuda), lse, cu.to(cuda), max(lens), scale).float().cpu()
    res = {}
    for name, fn in (("fp32", lambda q: oops.attention_varlen(q, cu, scale)), ("rounding", lambda q: oops.attention_varlen_kernel_rounding(q, cu, scale))):
        q32 = qkv.float().requires_grad_(True)
        ref = fn(q32)
        ref.backward(dout.float())
        gr = q32.grad
        res[name] = (float((dqkv - gr).norm() / gr.norm()), float((dqkv - gr).abs().max() / gr.abs().max()),
                     float((out.float().cpu() - ref.detach()).norm() / ref.detach().norm()))
    print(lens[:3], H, {k: tuple(f"{x:.2e}" for x in v) for k, v in res.items()})
