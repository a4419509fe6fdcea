#!/usr/bin/env python3
"""Fused attention kernel micro-benchmark: TFLOP/s of the two products (4 * HW^2 * C flop per image) at the bench shape
(N=32, C=512, 32x32 tokens) and the Kodak shape (N=8, 64x96 tokens), against the materialised-score path."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dc_vic_amd import ops  # noqa: E402
from dc_vic_amd.vqgan import AttnBlock  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    dev = torch.device("cuda:0")
    for N, C, H, W in ((32, 512, 32, 32), (8, 512, 64, 96), (1, 512, 32, 32)):
        qkv = torch.randn(N, 3 * C, H, W, device=dev)
        HW = H * W
        flop = 4.0 * N * HW * HW * C
        t = timeit(lambda: ops.attn_fused(qkv, C))
        line = f"N={N} C={C} HW={HW}: fused {t * 1e3:.3f} ms = {flop / t / 1e12:.1f} TFLOP/s"
        for nw in (2, 4):
            t2 = timeit(lambda: ops.attn_fused(qkv, C, force_nw=nw))
            line += f" | nw={nw} {t2 * 1e3:.3f} ms"
        if N * HW * HW * 4 < 8e9:
            tu = timeit(lambda: AttnBlock._attn_unfused(qkv, N, C, H, W), iters=5)
            line += f" | bgemm+softmax+bgemm {tu * 1e3:.3f} ms = {flop / tu / 1e12:.1f} TFLOP/s"
        print(line, flush=True)


if __name__ == "__main__":
    main()
