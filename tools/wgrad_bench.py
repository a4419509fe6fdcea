"""conv weight-gradient micro-benchmark (diagnostic): ms and TFLOP/s of dcvic_conv_wgrad_f32 on the trained 3x3 layers' shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dc_vic_amd.train import kernels as K

for (N, M, Cx, H, W) in ((8, 256, 256, 64, 64), (8, 256, 256, 128, 128), (8, 256, 448, 128, 128), (8, 512, 704, 32, 32), (8, 128, 128, 256, 256), (8, 96, 96, 128, 128)):
    G = torch.randn((N, M, H, W), device="cuda:0"); X = torch.randn((N, Cx, H, W), device="cuda:0")
    dW = torch.zeros((M, Cx, 3, 3), device="cuda:0")
    K.conv_wgrad(G, X, dW, 3, 3, 1, 1); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        K.conv_wgrad(G, X, dW, 3, 3, 1, 1)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"wgrad N{N} {Cx}->{M} @{H}x{W}: {ms:.3f} ms  {2 * N * H * W * M * Cx * 9 / ms * 1e-9:.1f} TFLOP/s", flush=True)
