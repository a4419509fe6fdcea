"""Runs only the VQ search kernel (2^22 vectors, 256x4 codebook) a few times -- a target for rocprofv3 --pmc."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dc_vic_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
z = torch.randn((64, 4, 256, 256), generator=g).to(dev)
if len(sys.argv) > 1 and sys.argv[1] == "zeros":
    z.zero_()
cb = ((torch.rand((256, 4), generator=g) * 2 - 1) / 256).to(dev)
for _ in range(5):
    ops.vq_argmin(z, cb, want_zq=False, want_feat=False)
torch.cuda.synchronize()
