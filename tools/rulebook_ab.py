import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pointcept_amd import ops, synthetic
dev = torch.device("cuda:0")
for sizes in ((3000, 1200), (102400,), (20000, 500, 7000)):
    b = synthetic.to_torch(synthetic.collate([synthetic.indoor_scene(31 + i, n) for i, n in enumerate(sizes)]), dev)
    off = b["offset"]
    bt = torch.repeat_interleave(torch.arange(off.numel(), device=dev), torch.diff(off, prepend=off.new_zeros(1)))
    for srt in (False, True):
        gc = b["grid_coord"]
        if srt:
            code = ops.serialize_encode(gc, bt, 16, ("hilbert",))
            o, _ = ops.sort_keys(code, 0, 51)
            gc2, bt2 = gc[o[0]], bt[o[0]]
        else:
            gc2, bt2 = gc, bt
        ind = torch.cat([bt2[:, None].int(), gc2.int()], 1).contiguous()
        tab = ops.HashTable(ind)
        for k in (1, 3, 5):
            os.environ["PTC_RULEBOOK_V1"] = "1"
            a = ops.rulebook_subm(ind, k, tab)
            os.environ["PTC_RULEBOOK_V1"] = "0"
            c = ops.rulebook_subm(ind, k, tab)
            print(sizes, "sorted" if srt else "raw", k, "equal" if torch.equal(a, c) else f"DIFF {(a != c).sum().item()} of {a.numel()}")
