import os, sys, ctypes
sys.path.insert(0, os.getcwd())
import torch
from dc_vic_amd import ops
from dc_vic_amd._lib import lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x = torch.randn((32, 224, 16, 16), generator=g).to(dev); w = (torch.randn((128, 224, 5, 5), generator=g) * 0.01).to(dev)
plan = ops.ConvPlan(w, torch.zeros(128, device=dev), "conv", pad=(2, 2))
for _ in range(3): out = plan(x)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8)()
lib().dcvic_debug_prof(buf)
v = list(buf)
print("variant", lib().dcvic_conv_last_variant(), "total(before epilogue)", v[3], "stages", v[4])
print("per stage: issue %.0f stage-prologue %.0f compute %.0f barrier %.0f cycles" % (v[0] / v[4], v[5] / v[4], v[1] / v[4], v[2] / v[4]))
