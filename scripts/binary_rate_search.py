#!/usr/bin/env python3
"""Binary search of beta_rate for a target bitrate at fixed beta_vq -- the caller of the batched rate-estimation
path, same flags, loop and csv output as the reference's scripts/binary_rate_search.py:25-196:
  positional config_path, --model_path --save_dir --dataset_root --beta_vq ... --target_rate ... --max_beta_rate
  --error_delta --batch_size -d/--device.
The dataset is a directory of <name>.png (+ optional <name>.npy pre-computed VQ tokens, as written by
build_openimage_val_dataset.py; without the .npy the tokens are computed by the VQGAN encoder on the fly).
Each probe runs vq_encode -> comp_encode -> estimate_entropy -> get_rate_summary_dict on the GPU.
`--synthetic_weights` replaces --model_path by the deterministic synthetic weights.
"""
from __future__ import annotations

import argparse
import os
import sys
from glob import glob
from itertools import product

import numpy as np
import pandas as pd
import torch

sys.path.append(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dc_vic_amd import BaseConfig, build_comp_model  # noqa: E402

MEMO_DICT = {}
MAX_RUN_CNT = 10


def arg_parse() -> dict:
    p = argparse.ArgumentParser()
    p.add_argument("config_path", type=str)
    p.add_argument("--model_path", type=str)
    p.add_argument("--save_dir", type=str)
    p.add_argument("--dataset_root", type=str)
    p.add_argument("--beta_vq", type=float, nargs="+")
    p.add_argument("--target_rate", type=float, nargs="+")
    p.add_argument("--max_beta_rate", type=float)
    p.add_argument("--error_delta", type=float, default=0.001)
    p.add_argument("--batch_size", type=int, default=1)
    p.add_argument("-d", "--device", type=str, default="cuda:0")
    p.add_argument("--synthetic_weights", action="store_true")
    return vars(p.parse_args())


def load_dataset(root: str):
    from PIL import Image
    items = []
    for path in sorted(glob(os.path.join(root, "*.png"))):
        img = np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)
        x = (torch.from_numpy(img.copy()).permute(2, 0, 1).float().div(255.0) - 0.5) / 0.5
        npy = path.replace(".png", ".npy")
        idx = torch.from_numpy(np.load(npy).astype(np.int32)).long() if os.path.exists(npy) else None
        items.append((x, idx))
    return items


def batches(items, bs):
    """consecutive items of equal shape share a batch (the reference's DataLoader needs equal shapes too)"""
    i = 0
    while i < len(items):
        j = i + 1
        while j < len(items) and j - i < bs and items[j][0].shape == items[i][0].shape and (items[j][1] is None) == (items[i][1] is None):
            j += 1
        x = torch.stack([it[0] for it in items[i:j]])
        idx = torch.stack([it[1] for it in items[i:j]]) if items[i][1] is not None else None
        yield x, idx
        i = j


@torch.no_grad()
def run_one_search(model, items, bs, beta_rate: float, beta_vq: float) -> float:
    bpp_list = []
    for x, idx in batches(items, bs):
        N, _, H, W = x.shape
        pi = model.data_preprocess(vq_indices=idx, real_images=x, beta_rate=beta_rate, beta_vq=beta_vq, is_train=False)
        gt_vq_latent, gt_vq_indices = model.vq_encode(pi["real_images"], pi["vq_indices"])
        y = model.comp_encode(real_images=pi["real_images"], gt_vq_latent=gt_vq_latent, gt_vq_indices=gt_vq_indices,
                              enc_kwargs=dict(beta_1=beta_rate, beta_2=beta_vq))
        entropy_dict = model.estimate_entropy(y, is_train=False)
        bpp_list.append(float(model.get_rate_summary_dict(entropy_dict, num_pixel=N * H * W)["bpp"]))
    return float(np.mean(bpp_list))


def memo_dict_key(beta_vq: float, beta_rate: float) -> str:
    return f"{beta_vq:.4f}-{beta_rate:.4f}".replace(".", "_")


def run(opt, model, items, target_rate, beta_vq):
    data_list = []
    lo, hi = 0.0, opt["max_beta_rate"]
    run_cnt = 0
    while True:
        run_cnt += 1
        beta_rate = round((lo + hi) / 2.0, 3)
        key = memo_dict_key(beta_vq, beta_rate)
        if key not in MEMO_DICT:
            MEMO_DICT[key] = run_one_search(model, items, opt["batch_size"], beta_rate, beta_vq)
        avg_bpp = MEMO_DICT[key]
        diff = abs(avg_bpp - target_rate)
        data_list.append({"run_cnt": run_cnt, "beta_vq": beta_vq, "beta_rate": beta_rate, "avg_bpp": avg_bpp, "diff": diff})
        print(f"run_cnt {run_cnt:2} | beta_rate {beta_rate} avg_bpp {avg_bpp:.5f} diff {diff:.5f}", flush=True)
        if diff <= opt["error_delta"]:
            break
        elif avg_bpp > target_rate:      # beta_rate is too small
            lo = beta_rate
        else:                            # beta_rate is too large
            hi = beta_rate
        if run_cnt >= MAX_RUN_CNT:
            break
    return pd.json_normalize(data_list).sort_values("diff").reset_index(drop=True)


def main() -> None:
    args = arg_parse()
    opt = BaseConfig.fromfile(args["config_path"], {k: v for k, v in args.items() if k not in ("synthetic_weights",)})
    ck = opt["subnet"]["vq_model"].get("ckpt_path")
    if ck and not os.path.exists(ck):
        opt["subnet"]["vq_model"]["ckpt_path"] = None
    os.makedirs(args["save_dir"], exist_ok=True)
    model = build_comp_model(opt)
    if args["synthetic_weights"]:
        from dc_vic_amd.synth import load_synth_weights
        load_synth_weights(model, 1234)
    else:
        model.load_learned_weight(ckpt_path=args["model_path"])
    items = load_dataset(args["dataset_root"])
    assert items, f'dataset_root "{args["dataset_root"]}" holds no png'
    for beta_vq, target_rate in product(args["beta_vq"], args["target_rate"]):
        df = run(args, model, items, target_rate, beta_vq)
        df.to_csv(os.path.join(args["save_dir"], f"result_beta_vq_{beta_vq:.2f}_target_rate_{target_rate:.3f}.csv"))


if __name__ == "__main__":
    main()
