#!/usr/bin/env python3
"""samples/s of the stage-3 training step (BASELINE config 5: 256x256 crops, batch 8 per GPU, G + D step) -- a SEPARATE metric,
never mixed into bench.py's headline.  python tools/train_bench.py [--batch 8] [--steps 5] [--warmup 2]; under
torch.distributed.run it is data-parallel (weak scaling) and reports the aggregate."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--batch", type=int, default=8)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=2)
    a = p.parse_args()
    rank, world, lr_ = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(lr_)
    dev = f"cuda:{lr_}"
    dist = None
    if world > 1:
        import torch.distributed as dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        dist = dist_
    from dc_vic_amd import BaseConfig, build_comp_model
    from dc_vic_amd.synth import load_synth_weights
    from dc_vic_amd.train import DualBetaCondGanDistortionVqCodeTrainer, DualBetaCondTamingNLayerDiscriminator
    m = build_comp_model(BaseConfig.fromfile(os.path.join(ROOT, "config", "dc_vic_synthetic.yaml"), {"device": dev}))
    load_synth_weights(m, 1234)
    torch.manual_seed(0)
    D = DualBetaCondTamingNLayerDiscriminator(input_nc=11, n_layers=3, ndf=64, norm_type="none", max_beta_1=3.0, max_beta_2=3.5).to(dev)
    tr = DualBetaCondGanDistortionVqCodeTrainer(m, D, dist=dist, seed=rank)
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.rand((a.batch, 3, 256, 256), generator=g) * 2 - 1
    for _ in range(a.warmup):
        tr.optimize_parameters(0, {"real_images": x})
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        log = tr.optimize_parameters(i, {"real_images": x})
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"metric": "training samples/sec, stage-3 G+D step @256x256", "value": world * a.batch * a.steps / dt, "unit": "samples/s",
                          "n_gpus": world, "steps": a.steps, "ms_per_step": 1e3 * dt / a.steps, "batch_per_gpu": a.batch, "dtype": "f32",
                          "data": "synthetic", "lpips": "included, synthetic weights (parity unpinned)", "last_log": log}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
