#!/usr/bin/env python3
"""Where a conv3x3_dma_kernel workgroup spends its cycles (diagnostic build, tools/build_stamps.sh):
DCVIC_LIB_PATH=tools/libdcvic_stamps.so python tools/conv_stamps.py [Cin Cout H W N]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dc_vic_amd import ops  # noqa: E402
from dc_vic_amd._lib import lib  # noqa: E402


def main():
    Cin, Cout, H, W, N = [int(v) for v in (sys.argv[1:6] if len(sys.argv) >= 6 else (128, 128, 256, 256, 32))]
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    x = torch.randn((N, Cin, H, W), generator=g).to(dev)
    w = (torch.randn((Cout, Cin, 3, 3), generator=g) * (Cin * 9) ** -0.5).to(dev)
    plan = ops.ConvPlan(w, torch.zeros(Cout, device=dev), "conv", pad=(1, 1))
    out = plan(x)
    nblocks = N * (H // 8) * (W // 32) * ((Cout + 127) // 128)
    buf = torch.zeros(nblocks * 8 * 8, dtype=torch.int64, device=dev)
    assert lib().dcvic_debug_set_stamp_buffer(C.c_void_p(buf.data_ptr())) == 0
    for _ in range(3):
        plan(x, out=out)
    torch.cuda.synchronize()
    b = buf.cpu().numpy().reshape(nblocks, 8, 8).astype(np.float64)
    issue, mfma, bar, t0, t1, ns = (b[..., i] for i in range(6))
    life = t1 - t0
    print(f"{nblocks} workgroups x 8 waves, {int(ns.max())} stages; per-wave means (cycles):")
    print(f"  K loop total {life.mean():10.0f}   per stage: issue {issue.mean() / ns.mean():7.0f}  mfma loop {mfma.mean() / ns.mean():7.0f}  barrier {bar.mean() / ns.mean():7.0f}"
          f"  (sum {(issue + mfma + bar).mean() / ns.mean():7.0f})")
    print(f"  MFMA floor per stage and wave alone: {72 * 64} cycles; x4 waves per SIMD = {4 * 72 * 64}")
    # kernel span and concurrency
    span = t1.max() - t0.min()
    print(f"  kernel span {span:.0f} cycles; mean workgroup K-loop life {life[:, 0].mean():.0f}; workgroups per CU-slot = {nblocks / 512:.1f}")
    per_wave_spread = (mfma.max(axis=1) - mfma.min(axis=1)).mean() / ns.mean()
    print(f"  spread of the MFMA-loop time across the 8 waves of a workgroup, per stage: {per_wave_spread:.0f} cycles")


if __name__ == "__main__":
    main()
