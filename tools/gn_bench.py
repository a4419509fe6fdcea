"""GroupNorm(+swish) micro-benchmark at the VQGAN shapes (diagnostic): ms and effective TB/s for 2 reads + 1 write."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dc_vic_amd import ops

for (N, C, H, W) in ((32, 128, 256, 256), (32, 256, 128, 128), (32, 256, 64, 64), (32, 512, 32, 32), (32, 704, 32, 32)):
    x = torch.randn((N, C, H, W), device="cuda:0")
    g = torch.ones(C, device="cuda:0"); b = torch.zeros(C, device="cuda:0")
    y = ops.groupnorm(x, g, b, 32, 1e-6, ops.ACT_SWISH)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.groupnorm(x, g, b, 32, 1e-6, ops.ACT_SWISH, out=y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"groupnorm {N}x{C}x{H}x{W}: {ms:.3f} ms  {3 * x.numel() * 4 / ms * 1e-9:.2f} TB/s (3 passes)", flush=True)
