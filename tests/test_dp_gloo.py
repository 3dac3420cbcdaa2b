"""N > 1 path on CPU: world_size 2, gloo backend, rendezvous on 127.0.0.1.  Covers the host logic
bench.py uses for data parallelism (pointcept_amd/dp.py): process-group init from the torchrun
environment, disjoint scene sharding, DDP gradient averaging with per-rank BatchNorm statistics,
barrier + max-over-ranks timing.  (The HIP ops themselves cannot run on CPU by design.)
"""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from pointcept_amd import dp

    dev = torch.device("cpu")
    dp.init_distributed(backend="gloo")
    torch.manual_seed(0)  # identical initial weights on every rank
    model = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.BatchNorm1d(16), torch.nn.GELU(), torch.nn.Linear(16, 5))
    ddp = dp.wrap_ddp(model, dev)
    seeds = dp.scene_seeds(rank, 3)
    g = torch.Generator().manual_seed(seeds[0])
    x = torch.randn(64 + 8 * rank, 6, generator=g)  # ragged per-rank batch
    y = torch.randint(0, 5, (x.shape[0],), generator=g)
    loss = torch.nn.functional.cross_entropy(ddp(x), y)
    loss.backward()
    dp.barrier(dev)
    t = dp.max_over_ranks(1.0 + rank, dev)
    total = dp.sum_over_ranks(float(x.shape[0]), dev)
    grads = [p.grad.numpy().copy() for p in model.parameters()]
    # local (un-averaged) gradient of the same loss on a fresh copy
    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.BatchNorm1d(16), torch.nn.GELU(), torch.nn.Linear(16, 5))
    torch.nn.functional.cross_entropy(ref(x), y).backward()
    q.put((rank, seeds, t, total, grads, [p.grad.numpy().copy() for p in ref.parameters()],
           model[1].running_mean.numpy().copy()))  # numpy: plain pickles, no shared-memory handles
    torch.distributed.destroy_process_group()


def test_ddp_gloo_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, t0, n0, g0, l0, bn0), (r1, s1, t1, n1, g1, l1, bn1) = res
    assert set(s0).isdisjoint(s1)                      # scenes shard, no overlap
    assert t0 == t1 == 2.0                             # max over ranks
    assert n0 == n1 == 64 + 72                         # sum over ranks
    import numpy as np

    for a, b, la, lb in zip(g0, g1, l0, l1):
        assert np.allclose(a, b, atol=1e-7)            # all-reduced gradients are identical on both ranks
        assert np.allclose(a, (la + lb) / 2, atol=1e-6)  # ... and equal the mean of the local gradients
    assert not np.allclose(bn0, bn1)                   # broadcast_buffers=False: BN statistics stay per rank
