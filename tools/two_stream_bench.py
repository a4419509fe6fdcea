"""Do two small-grid convs on two streams overlap on the GPU?  (CHARM runs its mean / scale networks that way.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dc_vic_amd import ops
dev = torch.device("cuda:0")
def mk(seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((32, 224, 16, 16), generator=g).to(dev)
    w = (torch.randn((128, 224, 5, 5), generator=g) * 0.01).to(dev)
    plan = ops.ConvPlan(w, torch.zeros(128, device=dev), "conv", pad=(2, 2))
    return plan, x, plan(x)
(p1, x1, o1), (p2, x2, o2) = mk(1), mk(2)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
def run(two, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        if two:
            s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s1): p1(x1, out=o1)
            with torch.cuda.stream(s2): p2(x2, out=o2)
            torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
        else:
            p1(x1, out=o1); p2(x2, out=o2)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for kb in (os.environ.get("DCVIC_ASYNC_STAGE_KB", "40"),):
    print(f"stage_kb={kb}: sequential pair {run(False):.3f} ms, two streams {run(True):.3f} ms")
