#!/usr/bin/env python3
"""Distortion / rate summary of a decoded directory -- the offline-computable part of the reference's
scripts/calc_metrics.py (121-171, 322-365).

Same flags (--real_dir --fake_dir -d/--device) and the same output file `<fake_dir>/_metrics.json`:
  * "bpp"  : read from `<fake_dir>/_avg_bitrate.json`, which scripts/compress.py wrote (calc_metrics.py:322-327);
  * "PSNR" : image-averaged PSNR over the sorted, name-matched *.png pairs, on RGB float32 in [0, 255]:
             20 log10(255) - 10 log10(mean squared error)  (calc_metrics.py:121-171), threads over images.
FID, LPIPS and DISTS (calc_metrics.py:174-320) need pretrained Inception / AlexNet / DISTS weights that the reference
downloads at first use; there is no network here, so they are skipped with a message (and absent from the json) unless
the packages AND their weights are importable.  The HiFiC FID patch cropper is provided (`crop_hific_fid_patches`,
calc_metrics.py:307-320) because the patch sets it produces are what an external FID tool consumes.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor
from glob import glob
from typing import List, Tuple

import numpy as np


def get_real_fake_path_list(real_dir: str, fake_dir: str) -> Tuple[List[str], List[str]]:
    assert os.path.exists(real_dir), real_dir
    assert os.path.exists(fake_dir), fake_dir
    real = sorted(glob(os.path.join(real_dir, "*.png")))
    fake = sorted(glob(os.path.join(fake_dir, "*.png")))
    assert len(real) == len(fake), f"{len(real)} real vs {len(fake)} decoded images"
    for r, f in zip(real, fake):
        assert os.path.basename(r) == os.path.basename(f), (r, f)
    return real, fake


def read_img(path: str) -> np.ndarray:
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.float32)


def image_psnr(real_path: str, fake_path: str) -> Tuple[float, float, int]:
    a, b = read_img(real_path), read_img(fake_path)
    assert a.shape == b.shape, (real_path, a.shape, b.shape)
    sq = float(np.sum(np.square(b - a)))
    mse = sq / a.size
    return 20.0 * np.log10(255.0) - 10.0 * np.log10(mse), sq, a.size


def average_psnr(real_paths: List[str], fake_paths: List[str], workers: int = 8) -> float:
    """Mean of the per-image PSNRs (the reference reports the image average, not the pixel average)."""
    with ThreadPoolExecutor(max_workers=workers) as ex:
        vals = list(ex.map(lambda rf: image_psnr(*rf)[0], zip(real_paths, fake_paths)))
    return float(np.mean(vals))


def crop_hific_fid_patches(img: np.ndarray, patch_size: int) -> np.ndarray:
    """All non-overlapping p x p blocks of the image, plus those of the image shifted by p/2 in both directions."""
    p = patch_size
    H, W = img.shape[:2]

    def blocks(a):
        h, w = a.shape[0] // p * p, a.shape[1] // p * p
        a = a[:h, :w]
        return a.reshape(h // p, p, w // p, p, 3).transpose(0, 2, 1, 3, 4).reshape(-1, p, p, 3)

    o = p // 2
    return np.concatenate([blocks(img), blocks(img[o:, o:])], axis=0)


def retrieve_bitrate(fake_dir: str) -> float:
    path = os.path.join(fake_dir, "_avg_bitrate.json")
    assert os.path.exists(path), f"{path} missing: run scripts/compress.py --decompress into this directory first"
    with open(path) as f:
        return json.load(f)["avg_bpp"]


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--real_dir", type=str, required=True)
    ap.add_argument("--fake_dir", type=str, required=True)
    ap.add_argument("-d", "--device", type=str, default="cuda:0")     # kept for flag compatibility; PSNR runs on the host
    a = ap.parse_args(argv)
    out = {"bpp": retrieve_bitrate(a.fake_dir)}
    real, fake = get_real_fake_path_list(a.real_dir, a.fake_dir)
    out["PSNR"] = average_psnr(real, fake)
    print(f"{len(real)} images: PSNR: {out['PSNR']:.4}")
    for name in ("FID", "LPIPS", "DISTS"):
        print(f"[calc_metrics] {name} skipped: its pretrained network weights cannot be fetched offline", file=sys.stderr)
    with open(os.path.join(a.fake_dir, "_metrics.json"), "w") as f:
        json.dump(out, f, indent=4)
    print(f"Results: {a.fake_dir}")
    for k, v in out.items():
        print(f"{k:>7}: {v:.4f}")
    return out


if __name__ == "__main__":
    main()
