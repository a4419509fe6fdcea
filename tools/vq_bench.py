#!/usr/bin/env python3
"""VQ search-only throughput under SURVEY 8(d)'s byte definition (24 B per latent vector: 16 B read + 8 B int64 index; 40 B
with z_q) on the saturating synthetic (2^22 vectors), per kernel variant and codebook size."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dc_vic_amd import ops  # noqa: E402


def timeit(fn, iters=10):
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
    g = torch.Generator().manual_seed(0)
    z = torch.randn((64, 4, 256, 256), generator=g).to(dev)      # 2^22 vectors
    M = 64 * 256 * 256
    for n_e in (256, 1024, 16384):
        cb = ((torch.rand((n_e, 4), generator=g) * 2 - 1) / n_e).to(dev)
        zz = z if n_e <= 1024 else z[:4].contiguous()
        m = M if n_e <= 1024 else M // 16
        for var in (("scalar", "lds", "shard") if n_e <= 1024 else ("shard",)):
            os.environ["DCVIC_VQ_KERNEL"] = var
            t = timeit(lambda: ops.vq_argmin(zz, cb, want_zq=False))
            t2 = timeit(lambda: ops.vq_argmin(zz, cb, want_zq=True))
            pairs = m * n_e
            print(f"n_e={n_e:6d} {var:6s}: index only {t * 1e6:9.1f} us = {24 * m / t / 1e12:.3f} TB/s ({24 * m / t / 8e12 * 100:.1f} % of 8 TB/s) | "
                  f"index + z_q {t2 * 1e6:9.1f} us = {40 * m / t2 / 1e12:.3f} TB/s | {pairs / t / 1e12:.2f} T pairs/s", flush=True)
    os.environ.pop("DCVIC_VQ_KERNEL", None)


if __name__ == "__main__":
    main()
