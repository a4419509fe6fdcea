#!/usr/bin/env python3
"""Selection of (beta_rate, beta_vq) per target bitrate -- the second caller of the batched rate / reconstruction path
(SURVEY 8f-4).  Same flags, loop and output files as the reference's scripts/beta_selection.py:30-246:

    positional config_path, --model_path --search_dir --save_dir --dataset_root --beta_vq ... --target_rate ...
    --alpha --batch_size --keep_recon -d/--device

For every target rate and every beta_vq it reads the best probe of `binary_rate_search.py`
(`<search_dir>/result_beta_vq_X_target_rate_Y.csv`, skipped when its bpp error exceeds 0.001, :175-183), reconstructs the
dataset with run_model(is_train=False) at that (beta_rate, beta_vq) (:119-155), writes the PNGs (truncating uint8, as
img_utils.imwrite), `_rate_summary.csv` and `_avg_bitrate.json` per setting, scores `alpha * PSNR - FID` (Eq. 13, :205) and
writes `target_rate_*/result.csv` (sorted by score) and `beta_selection_results.csv`.

FID needs the pytorch_fid Inception weights, which cannot be fetched offline: when they are unavailable the `fid` column is
NaN and the score falls back to `alpha * PSNR` (a message says so); `--fid_csv` lets an external FID tool supply
`beta_vq,target_rate,fid` rows instead.  `--synthetic_weights` replaces --model_path by the deterministic synthetic weights.
Per-image bpp comes from the kernel-side per-image bit counts (`bits_per_image`), not from a second pass over the
likelihood maps (calc_batch_bpp :96-104 computes the same quantity).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import shutil
import sys
from glob import glob

import numpy as np
import pandas as pd
import torch

sys.path.append(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.append(os.path.dirname(os.path.abspath(__file__)))
from binary_rate_search import batches, load_dataset  # noqa: E402
from calc_metrics import average_psnr  # noqa: E402
from dc_vic_amd import BaseConfig, build_comp_model, ops  # noqa: E402
from dc_vic_amd.io_pipeline import AsyncWriter, encode_png_u8  # noqa: E402

SEARCH_ERROR_THRESHOLD = 0.001


def arg_parse() -> dict:
    p = argparse.ArgumentParser()
    p.add_argument("config_path", type=str)
    p.add_argument("--model_path", type=str)
    p.add_argument("--search_dir", type=str)
    p.add_argument("--save_dir", type=str)
    p.add_argument("--dataset_root", type=str)
    p.add_argument("--beta_vq", type=float, nargs="+")
    p.add_argument("--target_rate", type=float, nargs="+")
    p.add_argument("--alpha", type=float, default=2.0, help="score = alpha * PSNR - FID (Eq. 13 of the paper)")
    p.add_argument("--batch_size", type=int, default=1)
    p.add_argument("--keep_recon", action="store_true")
    p.add_argument("-d", "--device", type=str, default="cuda:0")
    p.add_argument("--synthetic_weights", action="store_true")
    p.add_argument("--fid_csv", type=str, default=None, help="optional csv (beta_vq,target_rate,fid) from an external FID tool")
    return vars(p.parse_args())


@torch.no_grad()
def save_reconstructions(model, items, names, bs, save_dir, beta_vq, beta_rate) -> float:
    """beta_selection.py:119-155: run_model on every batch, write <name>.png + the per-image rate table."""
    rows = []
    writer = AsyncWriter()
    k = 0
    try:
        for x, idx in batches(items, bs):
            N, _, H, W = x.shape
            out = model.run_model(real_images=x, vq_indices=idx, beta_vq=beta_vq, beta_rate=beta_rate, is_train=False)
            bits = out["bits_per_image"].double().cpu().numpy()
            _, u8 = ops.crop_clamp(out["fake_images"], H, W, want_u8=True)      # truncating uint8, img_utils.py:19-44
            u8 = u8.cpu().numpy()
            for i in range(N):
                name = names[k]; k += 1
                writer.submit(encode_png_u8, os.path.join(save_dir, name), u8[i].copy())
                rows.append({"img_name": name.split(".")[0], "num_pixel": H * W, "total_bit": float(bits[i]), "bitrate": float(bits[i]) / (H * W)})
    finally:
        writer.close()
    df = pd.json_normalize(rows)
    df.to_csv(os.path.join(save_dir, "_rate_summary.csv"))
    avg_bpp = float(df["bitrate"].mean())
    with open(os.path.join(save_dir, "_avg_bitrate.json"), "w") as f:
        json.dump({"avg_bpp": avg_bpp}, f)
    return avg_bpp


def try_fid(real_paths, fake_paths, device):
    """calc_metrics.py:220-320 (HiFiC-style patch FID).  Needs pytorch_fid + its Inception weights."""
    try:
        import pytorch_fid  # noqa: F401
    except ImportError:
        return None
    return None            # the weights are a remote download (pt_inception-2015-12-05): unavailable offline


def main() -> None:
    a = arg_parse()
    opt = BaseConfig.fromfile(a["config_path"], {k: v for k, v in a.items() if k not in ("synthetic_weights", "fid_csv")})
    ck = opt["subnet"]["vq_model"].get("ckpt_path")
    if ck and not os.path.exists(ck):
        opt["subnet"]["vq_model"]["ckpt_path"] = None
    os.makedirs(a["save_dir"], exist_ok=True)
    assert os.path.exists(a["dataset_root"]), f'dataset_root "{a["dataset_root"]}" does not exist.'
    items = load_dataset(a["dataset_root"])
    names = [os.path.basename(p) for p in sorted(glob(os.path.join(a["dataset_root"], "*.png")))]
    assert items, f'dataset_root "{a["dataset_root"]}" holds no png'
    model = build_comp_model(opt)
    if a["synthetic_weights"]:
        from dc_vic_amd.synth import load_synth_weights
        load_synth_weights(model, 1234)
    else:
        model.load_learned_weight(ckpt_path=a["model_path"])
    ext_fid = {}
    if a["fid_csv"]:
        for _, r in pd.read_csv(a["fid_csv"]).iterrows():
            ext_fid[(round(float(r["beta_vq"]), 4), round(float(r["target_rate"]), 4))] = float(r["fid"])

    selection = []
    warned = False
    for target_rate in a["target_rate"]:
        data = []
        save_dir = os.path.join(a["save_dir"], f"target_rate_{target_rate}")
        os.makedirs(save_dir, exist_ok=True)
        for beta_vq in a["beta_vq"]:
            csv = os.path.join(a["search_dir"], f"result_beta_vq_{beta_vq:.2f}_target_rate_{target_rate:.3f}.csv")
            best = pd.read_csv(csv).sort_values(by="diff").iloc[0]
            if best["diff"] > SEARCH_ERROR_THRESHOLD:
                print(f'[beta_selection] bpp difference is larger than threshold: {best["diff"]} > {SEARCH_ERROR_THRESHOLD}. Skip.', flush=True)
                continue
            beta_rate = float(best["beta_rate"])
            recon_dir = os.path.join(save_dir, f"beta_vq_{beta_vq:.2f}")
            os.makedirs(recon_dir, exist_ok=True)
            avg_bpp = save_reconstructions(model, items, names, a["batch_size"], recon_dir, beta_vq, beta_rate)
            fake = sorted(glob(os.path.join(recon_dir, "*.png")))
            real = sorted(glob(os.path.join(a["dataset_root"], "*.png")))
            psnr = average_psnr(real, fake)
            fid = ext_fid.get((round(beta_vq, 4), round(target_rate, 4)))
            if fid is None:
                fid = try_fid(real, fake, a["device"])
            if fid is None:
                if not warned:
                    print("[beta_selection] FID unavailable offline (Inception weights): score = alpha * PSNR; pass --fid_csv to add it", file=sys.stderr)
                    warned = True
                fid, score = float("nan"), a["alpha"] * psnr
            else:
                score = a["alpha"] * psnr - fid
            data.append({"beta_vq": beta_vq, "beta_rate": beta_rate, "bpp": avg_bpp, "psnr": psnr, "fid": fid, "score": score})
            if not a["keep_recon"]:
                shutil.rmtree(recon_dir)
        if not data:
            continue
        df = pd.json_normalize(data).sort_values(by="score", ascending=False)
        df.to_csv(os.path.join(save_dir, "result.csv"))
        b = df.iloc[0]
        print(f'target_rate: {target_rate}, selected beta_vq: {b["beta_vq"]}, selected beta_rate: {b["beta_rate"]}', flush=True)
        selection.append({"target_rate": target_rate, "selected_beta_vq": b["beta_vq"], "selected_beta_rate": b["beta_rate"]})
    pd.json_normalize(selection).to_csv(os.path.join(a["save_dir"], "beta_selection_results.csv"), index=False)


if __name__ == "__main__":
    main()
