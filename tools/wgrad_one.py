"""one wgrad shape, a few launches (for rocprofv3 --pmc): python tools/wgrad_one.py N M Cx H W"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dc_vic_amd.train import kernels as K
N, M, Cx, H, W = (int(v) for v in sys.argv[1:6])
G = torch.randn((N, M, H, W), device="cuda:0"); X = torch.randn((N, Cx, H, W), device="cuda:0")
dW = torch.zeros((M, Cx, 3, 3), device="cuda:0")
for _ in range(3):
    K.conv_wgrad(G, X, dW, 3, 3, 1, 1)
torch.cuda.synchronize()
print("ok")
