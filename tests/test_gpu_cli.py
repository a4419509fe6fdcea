"""scripts/compress.py on a real MI355X: same flags and output files as the reference CLI (compress.py:85-144),
here on synthetic PNGs (the reference's demo_images do not travel to the GPU box) with synthetic weights."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cli_compress_decompress(tmp_path):
    from PIL import Image
    img_dir, save_dir = tmp_path / "imgs", tmp_path / "out"
    img_dir.mkdir()
    rng = np.random.default_rng(0)
    shapes = {"b_kodak_like.png": (128, 192), "a_small.png": (70, 100), "c_same.png": (128, 192)}
    for name, (h, w) in shapes.items():
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([(xx * 255 // w), (yy * 255 // h), ((xx + yy) * 255 // (h + w))], -1).astype(np.float64)
        im = np.clip(base + rng.normal(0, 12, (h, w, 3)), 0, 255).astype(np.uint8)
        Image.fromarray(im).save(img_dir / name)
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "compress.py"), "--config_path", os.path.join(ROOT, "config", "dc_vic_synthetic.yaml"),
           "--model_path", "unused.pth.tar", "--img_dir", str(img_dir), "--save_dir", str(save_dir), "-q", "2", "--decompress",
           "-d", "cuda:0", "--synthetic_weights", "--batch_size", "2"]
    subprocess.check_call(cmd, cwd=ROOT)
    import pandas as pd
    df = pd.read_csv(save_dir / "_bitrates.csv", index_col=0)
    assert list(df.columns) == ["img_name", "header_bit", "z_bit", "y_bit", "real_bit", "real_bpp", "pred_z_bit", "pred_y_bit",
                                "pred_bit", "pred_bpp", "num_pixel"]
    assert list(df["img_name"]) == sorted(shapes)                        # sorted glob order, like the reference
    avg = json.load(open(save_dir / "_avg_bitrate.json"))["avg_bpp"]
    assert abs(avg - df["real_bpp"].mean()) < 1e-12
    for name, (h, w) in shapes.items():
        b = (save_dir / name.replace(".png", ".bin")).read_bytes()
        row = df[df["img_name"] == name].iloc[0]
        assert row["real_bit"] == 8 * len(b) and row["num_pixel"] == h * w and row["header_bit"] == 48
        assert abs(row["real_bpp"] - 8 * len(b) / h / w) < 1e-12
        assert row["real_bit"] == row["header_bit"] + row["z_bit"] + row["y_bit"] + 3 * 32   # three uint32 length prefixes
        # container: header carries H, W (uint16 LE) and the quality index
        assert int.from_bytes(b[0:4], "little") == 6
        assert int.from_bytes(b[4:6], "little") == h and int.from_bytes(b[6:8], "little") == w and b[9] == 2
        rec = np.asarray(Image.open(save_dir / name))
        assert rec.shape == (h, w, 3) and rec.dtype == np.uint8
    # the two equal-shaped images went through one batch; decoding each .bin alone gives the same PNG bytes
    from dc_vic_amd import BaseConfig, build_comp_model
    from dc_vic_amd.codec_utils import load_byte_strings
    from dc_vic_amd.synth import load_synth_weights
    opt = BaseConfig.fromfile(os.path.join(ROOT, "config", "dc_vic_synthetic.yaml"), {"device": "cuda:0"})
    m = build_comp_model(opt); load_synth_weights(m, 1234); m.codec_setup()
    _, _, _, u8 = m.decompress_batch([load_byte_strings(str(save_dir / "c_same.bin"))], want_u8=True)
    assert np.array_equal(u8[0].cpu().numpy(), np.asarray(Image.open(save_dir / "c_same.png")))


def test_binary_rate_search_script(tmp_path):
    """Caller of the batched rate path (reference scripts/binary_rate_search.py): the probed bpp is monotone in
    beta_rate's bisection and the csv has the reference's columns."""
    from PIL import Image
    root, out = tmp_path / "data", tmp_path / "search"
    root.mkdir()
    rng = np.random.default_rng(1)
    for i in range(3):
        Image.fromarray(rng.integers(0, 256, (128, 128, 3), dtype=np.uint8)).save(root / f"{i}.png")
    np.save(root / "1.npy", rng.integers(0, 256, (16, 16)).astype(np.int64))      # one item with pre-computed VQ tokens
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "binary_rate_search.py"), os.path.join(ROOT, "config", "dc_vic_synthetic.yaml"),
           "--save_dir", str(out), "--dataset_root", str(root), "--beta_vq", "3.0", "--target_rate", "0.2", "--max_beta_rate", "3.0",
           "--error_delta", "0.0005", "--batch_size", "2", "--synthetic_weights"]
    subprocess.check_call(cmd, cwd=ROOT)
    import pandas as pd
    df = pd.read_csv(out / "result_beta_vq_3.00_target_rate_0.200.csv", index_col=0)
    assert list(df.columns) == ["run_cnt", "beta_vq", "beta_rate", "avg_bpp", "diff"]
    assert 1 <= len(df) <= 10 and (df["diff"].values[:-1] <= df["diff"].values[1:]).all()      # sorted by diff
    assert (df["avg_bpp"] > 0).all() and (df["beta_rate"] >= 0).all() and (df["beta_rate"] <= 3.0).all()
