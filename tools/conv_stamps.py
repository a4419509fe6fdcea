#!/usr/bin/env python3
"""Timeline of conv3x3_dma_kernel workgroups per CU (diagnostic build, tools/build_stamps.sh):
DCVIC_LIB_PATH=tools/libdcvic_stamps.so python tools/conv_stamps.py [Cin Cout H W N]
Per workgroup: entry -> K-loop start (prologue), K loop, loop end -> exit (epilogue); per CU: how the two resident
workgroups overlap and how long a slot stays empty between two workgroups."""
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
    b = buf.cpu().numpy().reshape(nblocks, 8, 8)
    entry, l0, l1, end = (b[..., i].astype(np.float64) for i in range(4))
    hw, xcc = b[:, 0, 4], b[:, 0, 5] & 0xF
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 0x7
    key = xcc * 1000 + se * 100 + sh * 10 + cu
    t_in, t_l0, t_l1, t_out = entry.min(1), l0.min(1), l1.max(1), end.max(1)
    print(f"{nblocks} workgroups on {len(np.unique(key))} CUs; stages {int(b[0, 0, 6])}")
    print(f"per workgroup (cycles of s_memtime): prologue {np.mean(t_l0 - t_in):8.0f}  K loop {np.mean(t_l1 - t_l0):9.0f}  epilogue {np.mean(t_out - t_l1):8.0f}"
          f"  life {np.mean(t_out - t_in):9.0f}")
    lead, trail = l1[:, :4].max(1) - l0[:, :4].min(1), l1[:, 4:].max(1) - l0[:, 4:].min(1)
    print(f"  K loop of the leading / trailing wave group: {lead.mean():.0f} / {trail.mean():.0f}")
    busy2, busy1, busy0, gaps, spans = [], [], [], [], []
    for k in np.unique(key):
        m = np.nonzero(key == k)[0]
        ev = sorted([(t_in[i], +1) for i in m] + [(t_out[i], -1) for i in m])
        span = ev[-1][0] - ev[0][0]
        cnt, last, acc = 0, ev[0][0], [0.0, 0.0, 0.0, 0.0]
        for t, d in ev:
            acc[min(cnt, 3)] += t - last
            cnt += d; last = t
        busy0.append(acc[0] / span); busy1.append(acc[1] / span); busy2.append((acc[2] + acc[3]) / span); spans.append(span)
        # K-loop concurrency: fraction of the span with 2 / 1 / 0 workgroups inside their K loop
    kl2, kl1, kl0 = [], [], []
    for k in np.unique(key):
        m = np.nonzero(key == k)[0]
        ev = sorted([(t_l0[i], +1) for i in m] + [(t_l1[i], -1) for i in m])
        t_first, t_last = min(t_in[m]), max(t_out[m])
        cnt, last, acc = 0, t_first, [0.0, 0.0, 0.0, 0.0]
        for t, d in ev:
            acc[min(cnt, 3)] += t - last
            cnt += d; last = t
        acc[0] += t_last - last
        sp = t_last - t_first
        kl0.append(acc[0] / sp); kl1.append(acc[1] / sp); kl2.append((acc[2] + acc[3]) / sp)
    print(f"per CU: span {np.mean(spans):.0f} cycles; resident workgroups 2 / 1 / 0: {np.mean(busy2):.3f} / {np.mean(busy1):.3f} / {np.mean(busy0):.3f}")
    print(f"per CU: workgroups INSIDE their K loop   2 / 1 / 0: {np.mean(kl2):.3f} / {np.mean(kl1):.3f} / {np.mean(kl0):.3f}")
    floor = int(b[0, 0, 6]) * 72 * 64 * 2           # MFMA cycles of one workgroup on its SIMDs (2 waves per SIMD)
    print(f"MFMA floor per workgroup {floor}; x workgroups per CU {len(key) / len(np.unique(key)):.1f} = {floor * len(key) / len(np.unique(key)):.0f} of span {np.mean(spans):.0f}"
          f" -> pipe busy {floor * len(key) / len(np.unique(key)) / np.mean(spans):.3f}")


if __name__ == "__main__":
    main()
