"""GroupNorm in context (diagnostic): the same 32x128x256x256 launch (a) on reused buffers, (b) on freshly allocated outputs,
(c) right after a Winograd conv wrote its input."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dc_vic_amd import ops

dev = "cuda:0"
N, C, H, W = 32, 128, 256, 256
x = torch.randn((N, C, H, W), device=dev)
g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
w = (torch.randn((C, C, 3, 3)) * (C * 9) ** -0.5).to(dev)
plan = ops.ConvPlan(w, None, "conv", pad=(1, 1)); plan.wino = "force"


def timed(fn, reps=6):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        pre = fn.pre() if hasattr(fn, "pre") else None
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sum(ts) / len(ts)

y = torch.empty_like(x)
print("reused buffers      : %.3f ms" % timed(lambda: ops.groupnorm(x, g, b, 32, 1e-6, ops.ACT_SWISH, out=y)))
print("fresh output tensor : %.3f ms" % timed(lambda: ops.groupnorm(x, g, b, 32, 1e-6, ops.ACT_SWISH)))
print("in place            : %.3f ms" % timed(lambda: ops.groupnorm(y, g, b, 32, 1e-6, ops.ACT_NONE, out=y)))
h = plan(x)
def after_conv():
    ops.groupnorm(h, g, b, 32, 1e-6, ops.ACT_SWISH, out=y)
def conv_then():
    plan(x, out=h)
for name, fn in (("conv alone", conv_then),):
    print("%-20s: %.3f ms" % (name, timed(fn)))
def both():
    plan(x, out=h); ops.groupnorm(h, g, b, 32, 1e-6, ops.ACT_SWISH, out=y)
print("conv + groupnorm    : %.3f ms" % timed(both))
