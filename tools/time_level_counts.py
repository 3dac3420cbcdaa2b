import sys, torch
sys.path.insert(0, '.')
from pointcept_amd import ops, synthetic
from pointcept_amd.structure import Point
b = synthetic.to_torch(synthetic.indoor_batch(8, 102400), "cuda")
p = Point(b); p.serialization(order=("z","z-trans","hilbert","hilbert-trans"))
d = p.serialized_depth
def f(): return ops.pool_level_counts(p.serialized_code[0], p.serialized_order[0], 3*d, 8, [3,6,9,12])
for _ in range(3): f()
a,e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); a.record()
for _ in range(20): f()
e.record(); torch.cuda.synchronize(); print("pool_level_counts us", a.elapsed_time(e)/20*1e3, f().tolist())
