#!/usr/bin/env python
"""Forward attention launch plans: time attn_varlen_fwd at L = 1024, head_dim 16 for unit counts around whole rounds of the 512
workgroup slots, under PTC_AT_PLAN settings (each in a child process: the plan is read per call, the cache per process).
usage: python tools/attn_tail_probe.py            (parent: runs the sweep)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch
    from pointcept_amd import ops
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_ops import timeit

    dev = torch.device("cuda:0")
    out = []
    for n_seq, H in ((768, 4), (800, 4), (832, 4), (896, 4), (400, 4), (800, 2)):
        L = 1024
        g = torch.Generator(device=dev).manual_seed(1)
        qkv = (torch.randn(n_seq * L, 3, H, 16, device=dev, generator=g) * 1.0).to(torch.bfloat16)
        cu = torch.arange(0, (n_seq + 1) * L, L, device=dev, dtype=torch.int32)
        t = timeit(lambda: ops.attn_varlen_fwd(qkv, cu, L, 0.25), iters=20)
        out.append(f"{n_seq * H:5d} units {t * 1e6:7.1f} us")
    print(" | ".join(out))


if __name__ == "__main__":
    if os.environ.get("AT_CHILD"):
        child()
    else:
        plans = sys.argv[1:] or ["", "0", "384,4", "384,2", "320,2", "256,2", "0,2"]
        for p in plans:
            env = dict(os.environ, AT_CHILD="1")
            env.pop("PTC_AT_PLAN", None)
            if p:
                env["PTC_AT_PLAN"] = p
            r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True)
            print(f"PTC_AT_PLAN={p or '(default)':10s} {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}")
