#!/usr/bin/env python3
"""Stage-3 GAN training of DC-VIC on MI355X -- counterpart of the reference's scripts/train.py:16-27 +
src/trainer/base_trainer.py:130-152 (train_loop) for the `DualBetaCondGanDistortionVqCodeTrainer` of config/exp1_stage3.yaml.

    python scripts/train.py CONFIG [--model_path CKPT | --synthetic_weights] [--dataset_root DIR | --synthetic_data]
                            [--batch_size 8] [--total_iter N] [--save_dir DIR] [--save_step N] [--seed S] [-d cuda:0]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/train.py ...

One process per GPU; every rank draws its own crops (sampler sharded by rank), gradients are averaged with bucketed RCCL
all-reduces of the flat gradient buffers (the reference is single-GPU, README.md:64-65).  Data: PNG / JPG files under
--dataset_root -> RandomCrop 256 (reflect pad if smaller) -> horizontal flip -> [-1, 1] (src/dataset/data_transform.py:19-45),
or seeded synthetic tensors.  Checkpoints use the reference's file format: `comp_model_iterXXXXXXX.pth.tar` =
{'iter', 'comp_model': state_dict}, `discriminator_iter...` = {'iter', 'discriminator': state_dict} (model_saver.py:39-46).
The optimizer / loss settings are the YAML's `optim` / `loss` sections when present (config/exp1_stage1_3.yaml:43-79), else
their stage-3 values (g_scheduler and d_scheduler milestones are read separately).  LPIPS: pass --lpips_path (a torch.save'd state
dict of lpips.LPIPS(net='alex'), loaded with weights_only=True); its weights cannot be fetched offline, so without it a positive
perceptual weight is an ERROR unless --allow_synthetic_lpips opts into synthetic AlexNet weights (benchmarks / plumbing only).
--save_step also writes training_state_iter*.pth.tar (Adam moments, step counts, scheduler epochs, beta-sampler RNG; the
reference saves optimizer + scheduler state too, base_trainer.py:178-214) and --resume continues from it.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from glob import glob

import numpy as np
import torch

sys.path.append(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dc_vic_amd import BaseConfig, build_comp_model  # noqa: E402
from dc_vic_amd.parallel import launched_by_a_launcher, pin_rank_cpus, self_launch  # noqa: E402
from dc_vic_amd.train import DualBetaCondGanDistortionVqCodeTrainer, DualBetaCondTamingNLayerDiscriminator  # noqa: E402


def _get(d, *keys, default=None):
    for k in keys:
        try:
            d = d[k]
        except (KeyError, TypeError):
            return default
    return d


class CropDataset:
    """OpenImages-style folder -> random 256 crops, flips, [-1, 1]; sharded by rank, reshuffled every epoch."""

    def __init__(self, root: str, size: int, rank: int, world: int, seed: int):
        self.paths = sorted(p for ext in ("png", "jpg", "jpeg") for p in glob(os.path.join(root, f"*.{ext}")))
        assert self.paths, f'dataset_root "{root}" holds no image'
        self.size, self.rank, self.world = size, rank, world
        self.rng = np.random.RandomState(seed * 1000 + rank)
        self.order, self.pos = [], 0

    def _next_path(self) -> str:
        if self.pos >= len(self.order):
            perm = np.random.RandomState(len(self.order) + 17).permutation(len(self.paths))     # same permutation on every rank
            self.order = [int(i) for i in perm[self.rank::self.world]] or [int(perm[0])]
            self.pos = 0
        p = self.paths[self.order[self.pos]]
        self.pos += 1
        return p

    def batch(self, n: int) -> torch.Tensor:
        from PIL import Image
        out = torch.empty((n, 3, self.size, self.size), dtype=torch.float32)
        for i in range(n):
            a = np.asarray(Image.open(self._next_path()).convert("RGB"), dtype=np.uint8)
            H, W = a.shape[:2]
            if H < self.size or W < self.size:        # RandomCrop(pad_if_needed, padding_mode='reflect')
                a = np.pad(a, ((0, max(0, self.size - H)), (0, max(0, self.size - W)), (0, 0)), mode="reflect")
                H, W = a.shape[:2]
            y0, x0 = self.rng.randint(0, H - self.size + 1), self.rng.randint(0, W - self.size + 1)
            c = a[y0:y0 + self.size, x0:x0 + self.size]
            if self.rng.rand() < 0.5:
                c = c[:, ::-1]
            out[i] = (torch.from_numpy(c.copy()).permute(2, 0, 1).float().div(255.0) - 0.5) / 0.5
        return out


def main():
    p = argparse.ArgumentParser()
    p.add_argument("config_path", type=str)
    p.add_argument("--model_path", type=str, default=None)
    p.add_argument("--synthetic_weights", action="store_true")
    p.add_argument("--dataset_root", type=str, default=None)
    p.add_argument("--synthetic_data", action="store_true")
    p.add_argument("--batch_size", type=int, default=8, help="per GPU (the reference trains with 6, BASELINE config 5 asks 8)")
    p.add_argument("--total_iter", type=int, default=None)
    p.add_argument("--save_dir", type=str, default=None)
    p.add_argument("--save_step", type=int, default=0)
    p.add_argument("--log_step", type=int, default=10)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("-d", "--device", type=str, default="cuda:0")
    p.add_argument("--lpips_path", type=str, default=None, help="state dict of lpips.LPIPS(net='alex') (torch.save'd; loaded with weights_only=True)")
    p.add_argument("--allow_synthetic_lpips", action="store_true", help="train the perceptual term against SYNTHETIC AlexNet / head weights (plumbing / benchmarks only)")
    p.add_argument("--resume", type=str, default=None, help="training_state_iterXXXXXXX.pth.tar written by --save_step (loads the comp_model / discriminator files beside it)")
    p.add_argument("--gpus", type=int, default=0, help="data-parallel over N GPUs of this node: without a launcher this process starts the N ranks itself")
    a = p.parse_args()
    if a.gpus > 1 and not launched_by_a_launcher():
        sys.exit(self_launch(a.gpus))                     # parent: never touches the GPU
    if a.gpus > 0 and int(os.environ.get("WORLD_SIZE", "1")) != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}")

    rank, world, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    pin_rank_cpus()
    dist = None
    device = a.device
    if world > 1:
        import torch.distributed as dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        device = f"cuda:{local_rank}"
        torch.cuda.set_device(local_rank)
        dist_.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
        dist = dist_
    opt = BaseConfig.fromfile(a.config_path, {"device": device, "is_train": True})
    ck = opt["subnet"]["vq_model"].get("ckpt_path")
    if ck and not os.path.exists(ck):
        opt["subnet"]["vq_model"]["ckpt_path"] = None
    model = build_comp_model(opt)
    if a.synthetic_weights:
        from dc_vic_amd.synth import load_synth_weights
        load_synth_weights(model, 1234)
    else:
        model.load_learned_weight(ckpt_path=a.model_path)
    dopt = dict(_get(opt, "discriminator", default=None) or dict(type="DualBetaCondTamingNLayerDiscriminator", input_nc=11, n_layers=3, ndf=64,
                                                                  norm_type="none", max_beta_1=3.0, max_beta_2=3.5, L=10, cond_ch=8,
                                                                  use_pi=False, include_x=True))
    dopt.pop("type", None)
    torch.manual_seed(a.seed)                       # identical D initialisation on every rank
    D = DualBetaCondTamingNLayerDiscriminator(**dopt).to(device)
    lw = {}
    for name, key in (("distortion", "distortion_loss"), ("perceptual", "perceptual_loss"), ("gan", "gan_loss"),
                      ("code_distortion", "code_distortion_loss"), ("code_ce", "code_ce_loss")):
        v = _get(opt, "loss", key, "loss_weight")
        if v is not None:
            lw[name] = float(v)
    from dc_vic_amd.train.trainer import DEFAULT_LOSS
    w_perc = lw.get("perceptual", DEFAULT_LOSS["perceptual"])
    lpips_state = None
    if a.lpips_path:
        lpips_state = torch.load(a.lpips_path, map_location="cpu", weights_only=True)
    elif w_perc > 0:
        msg = (f"perceptual_loss has weight {w_perc} but no --lpips_path was given: the `lpips` AlexNet / head weights cannot be fetched "
               "offline, so the term would be computed with SYNTHETIC weights (not LPIPS).")
        if not a.allow_synthetic_lpips:
            raise SystemExit(msg + "  Pass --lpips_path STATE_DICT, set loss.perceptual_loss.loss_weight: 0, or opt in with --allow_synthetic_lpips.")
        if rank == 0:
            print("[train] WARNING: " + msg, file=sys.stderr, flush=True)
    trainer = DualBetaCondGanDistortionVqCodeTrainer(
        model, D, lr_g=float(_get(opt, "optim", "g_optimizer", "lr", default=1e-4)), lr_d=float(_get(opt, "optim", "d_optimizer", "lr", default=1e-4)),
        milestones=list(_get(opt, "optim", "g_scheduler", "milestones", default=[300000])), gamma=float(_get(opt, "optim", "g_scheduler", "gamma", default=0.1)),
        clip_max_norm=_get(opt, "optim", "clip_max_norm", default=1.0), loss_weights=lw,
        sample_beta_batch=bool(_get(opt, "trainer", "sample_beta_batch", default=True)), dist=dist, seed=a.seed * 100 + rank,
        d_milestones=_get(opt, "optim", "d_scheduler", "milestones", default=None), d_gamma=_get(opt, "optim", "d_scheduler", "gamma", default=None),
        lpips_state=lpips_state)
    start_iter = 0
    if a.resume:
        # base_trainer.py:178-214: comp_model / discriminator / training_state files of one iteration
        st = torch.load(a.resume, map_location="cpu", weights_only=False)        # our own file (numpy RNG state inside)
        d_ = os.path.dirname(a.resume)
        it_tag = os.path.basename(a.resume).replace("training_state_", "")
        model.load_state_dict(torch.load(os.path.join(d_, "comp_model_" + it_tag), map_location="cpu", weights_only=True)["comp_model"])
        D.load_state_dict(torch.load(os.path.join(d_, "discriminator_" + it_tag), map_location="cpu", weights_only=True)["discriminator"])
        trainer.resync_parameters()
        start_iter = trainer.load_training_state(st)
    total_iter = a.total_iter or int(_get(opt, "total_iter", default=500000))
    data = None if a.synthetic_data else CropDataset(a.dataset_root, 256, rank, world, a.seed)
    gen = torch.Generator().manual_seed(a.seed * 7919 + rank)
    if a.save_dir and rank == 0:
        os.makedirs(a.save_dir, exist_ok=True)
    if data is None:
        for _ in range(start_iter):          # resume: the synthetic stream continues where the interrupted run stopped
            torch.rand((a.batch_size, 3, 256, 256), generator=gen)
    t0 = time.perf_counter()
    for it in range(start_iter + 1, total_iter + 1):
        x = (torch.rand((a.batch_size, 3, 256, 256), generator=gen) * 2 - 1) if data is None else data.batch(a.batch_size)
        log = trainer.optimize_parameters(it, {"real_images": x})
        if rank == 0 and (it % a.log_step == 0 or it == 1 or it == total_iter):
            dt = time.perf_counter() - t0
            msg = "skipped (loss anomaly)" if log is None else " ".join(f"{k} {v:.5g}" for k, v in log.items())
            print(f"iter {it:7d} | {world * a.batch_size * (it - start_iter) / dt:7.2f} samples/s | {msg}", flush=True)
        if a.save_dir and a.save_step and it % a.save_step == 0 and rank == 0:
            torch.save({"iter": it, "comp_model": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}},
                       os.path.join(a.save_dir, f"comp_model_iter{it:07d}.pth.tar"))
            torch.save({"iter": it, "discriminator": {k: v.detach().cpu().clone() for k, v in D.state_dict().items()}},
                       os.path.join(a.save_dir, f"discriminator_iter{it:07d}.pth.tar"))
            torch.save(trainer.training_state(it), os.path.join(a.save_dir, f"training_state_iter{it:07d}.pth.tar"))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
