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
           "-d", "cuda", "--synthetic_weights", "--batch_size", "2"]
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


def _cli(img_dir, save_dir, q, extra=()):
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "compress.py"), "--config_path", os.path.join(ROOT, "config", "dc_vic_synthetic.yaml"),
           "--model_path", "unused.pth.tar", "--img_dir", str(img_dir), "--save_dir", str(save_dir), "-q", str(q), "--decompress",
           "-d", "cuda:0", "--synthetic_weights", *extra]
    subprocess.check_call(cmd, cwd=ROOT)


def test_cli_config1_demo_images_vs_oracle(tmp_path, oracle, oracle_compress):
    """BASELINE config 1 (README.md:48-61): the reference's three demo_images (768x512 Kodak PNGs, data fixtures) through
    the CLI at -q 0 (on the GPU: the product has no CPU path).  Every .bin must equal the oracle's container byte for
    byte, so real_bpp / avg_bpp are the oracle's exactly and pred_bpp agrees to 4 decimals.  An image whose bitstream differs is
    NOT waved through on its rate: the CLI's bytes must equal the in-process model.compress() of the same file, and that
    compress is itemised by parity_util (first differing integer decision = a bounded, capped fp32 near-tie)."""
    import pandas as pd
    from conftest import GOLDEN, demo_image
    from oracle.dcvic_oracle import pack_strings, postprocess, to_uint8_rgb
    from PIL import Image
    save = tmp_path / "out"
    _cli(os.path.join(GOLDEN, "demo_images"), save, 0)
    df = pd.read_csv(save / "_bitrates.csv", index_col=0)
    names = ["kodim03.png", "kodim15.png", "kodim23.png"]
    assert list(df["img_name"]) == names
    real_o, identical, model = [], 0, None
    for name in names:
        ro = oracle_compress(("demo", name), demo_image(name), 0)
        blob_o = pack_strings(ro["string_list"])
        blob_g = (save / name.replace(".png", ".bin")).read_bytes()
        row = df[df["img_name"] == name].iloc[0]
        real_o.append(8 * len(blob_o) / (512 * 768))
        if blob_g == blob_o:
            identical += 1
            assert row["real_bpp"] == real_o[-1]
            assert abs(row["pred_bpp"] - (ro["pred_y_bpp"] + ro["pred_z_bpp"])) < 5e-5
        else:
            print(f"[config1] {name}: free-running bitstream differs from the oracle's -- itemising")
            from parity_util import Report, encode_parity, free_running_compress
            if model is None:
                from dc_vic_amd import BaseConfig, build_comp_model
                from dc_vic_amd.synth import load_synth_weights
                model = build_comp_model(BaseConfig.fromfile(os.path.join(ROOT, "config", "dc_vic_synthetic.yaml"), {"device": "cuda:0"}))
                load_synth_weights(model, 1234); model.codec_setup()
            rep = Report(f"cli_config1_{name[:-4]}_q0")
            try:
                encode_parity(model, ro, demo_image(name), 0, rep)
                rg = free_running_compress(model, ro, demo_image(name), 0, rep)     # asserts: first flip is a bounded, capped near-tie
            finally:
                rep.dump()
            assert pack_strings(rg["string_list"]) == blob_g, "the CLI's bytes are not those of model.compress() on the same file"
            assert rep.n_flips() >= 1
        assert Image.open(save / name).size == (768, 512)
    avg = json.load(open(save / "_avg_bitrate.json"))["avg_bpp"]
    assert abs(avg - float(df["real_bpp"].mean())) < 1e-12
    if identical == 3:
        assert abs(avg - float(np.mean(real_o))) < 1e-12
    # decoded PNG of the first image vs the oracle's decode of the same stream (truncating uint8, img_utils.py:19-44)
    ro = oracle_compress(("demo", names[0]), demo_image(names[0]), 0)
    if (save / "kodim03.bin").read_bytes() == pack_strings(ro["string_list"]):
        img_o, _, _, _ = oracle.decompress(ro["string_list"])
        png_o = to_uint8_rgb(img_o).astype(np.int32)
        png_g = np.asarray(Image.open(save / "kodim03.png")).astype(np.int32)
        d = np.abs(png_g - png_o)
        assert d.max() <= 1 and (d > 0).mean() < 0.02, (d.max(), (d > 0).mean())     # 1e-3 on [-1,1] = 0.13 grey levels


def test_cli_config4_mixed_folder(tmp_path):
    """BASELINE config 4 shape: a folder of different sizes including a tiled (> 1024 px) image, shape-bucketed batches
    (--batch_size 2).  Each .bin equals the in-process one-image compress() of the same file (bucket / batch logic
    changes nothing), decodes alone to the written PNG, and the csv is sorted and complete."""
    import pandas as pd
    import torch
    from PIL import Image
    img_dir, save = tmp_path / "imgs", tmp_path / "out"
    img_dir.mkdir()
    rng = np.random.default_rng(4)
    shapes = {"a.png": (256, 256), "b_big.png": (1088, 576), "c.png": (256, 256), "d_ragged.png": (200, 333), "e.png": (256, 256)}
    for name, (h, w) in shapes.items():
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([(xx * 255 // w), (yy * 255 // h), ((xx + yy) * 255 // (h + w))], -1).astype(np.float64)
        Image.fromarray(np.clip(base + rng.normal(0, 10, (h, w, 3)), 0, 255).astype(np.uint8)).save(img_dir / name)
    _cli(img_dir, save, 3, ("--batch_size", "2"))
    df = pd.read_csv(save / "_bitrates.csv", index_col=0)
    assert list(df["img_name"]) == sorted(shapes)
    from dc_vic_amd import BaseConfig, build_comp_model
    from dc_vic_amd.codec_utils import load_byte_strings
    from dc_vic_amd.synth import load_synth_weights
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    opt = BaseConfig.fromfile(os.path.join(ROOT, "config", "dc_vic_synthetic.yaml"), {"device": "cuda:0"})
    m = build_comp_model(opt); load_synth_weights(m, 1234); m.codec_setup()
    for name, (h, w) in shapes.items():
        a = np.asarray(Image.open(img_dir / name).convert("RGB"), dtype=np.uint8)
        x = ((torch.from_numpy(a.copy()).permute(2, 0, 1).float().div(255.0) - 0.5) / 0.5).unsqueeze(0)
        r = m.compress(x, 3)
        sl = load_byte_strings(str(save / name.replace(".png", ".bin")))
        assert [bytes(s) for s in sl] == [bytes(s) for s in r["string_list"]], name
        _, _, _, u8 = m.decompress_batch([sl], want_u8=True)
        assert np.array_equal(u8[0].cpu().numpy(), np.asarray(Image.open(save / name))), name
        row = df[df["img_name"] == name].iloc[0]
        assert row["num_pixel"] == h * w


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
    # beta_selection.py on top of search results (reference scripts/beta_selection.py:158-246).  With synthetic weights the
    # rate is not monotone in beta_rate, so the bisection above need not converge: the selection step is fed two search
    # tables in the search script's format whose best rows are within the 0.001 threshold, plus one that is not (skipped)
    sdir = tmp_path / "search_tables"
    sdir.mkdir()
    picks = {3.0: 1.5, 2.0: 0.75}
    for bv, br in picks.items():
        pd.DataFrame([{"run_cnt": 1, "beta_vq": bv, "beta_rate": br, "avg_bpp": 0.2004, "diff": 0.0004},
                      {"run_cnt": 2, "beta_vq": bv, "beta_rate": 2.9, "avg_bpp": 0.3, "diff": 0.1}]).to_csv(
            sdir / f"result_beta_vq_{bv:.2f}_target_rate_0.200.csv")
    pd.DataFrame([{"run_cnt": 1, "beta_vq": 1.0, "beta_rate": 1.0, "avg_bpp": 0.25, "diff": 0.05}]).to_csv(sdir / "result_beta_vq_1.00_target_rate_0.200.csv")
    sel = tmp_path / "selection"
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "beta_selection.py"), os.path.join(ROOT, "config", "dc_vic_synthetic.yaml"),
           "--search_dir", str(sdir), "--save_dir", str(sel), "--dataset_root", str(root), "--beta_vq", "3.0", "2.0", "1.0", "--target_rate", "0.2",
           "--batch_size", "2", "--keep_recon", "--synthetic_weights"]
    subprocess.check_call(cmd, cwd=ROOT)
    res = pd.read_csv(sel / "target_rate_0.2" / "result.csv", index_col=0)
    assert list(res.columns) == ["beta_vq", "beta_rate", "bpp", "psnr", "fid", "score"]
    assert sorted(res["beta_vq"]) == [2.0, 3.0]                       # beta_vq 1.0 exceeded the search-error threshold
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import binary_rate_search as brs
    from dc_vic_amd import BaseConfig, build_comp_model
    from dc_vic_amd.synth import load_synth_weights
    m = build_comp_model(BaseConfig.fromfile(os.path.join(ROOT, "config", "dc_vic_synthetic.yaml"), {"device": "cuda:0"}))
    load_synth_weights(m, 1234)
    items = brs.load_dataset(str(root))
    assert (res["score"].values[:-1] >= res["score"].values[1:]).all()        # sorted by score, best first
    for _, r in res.iterrows():
        assert r["beta_rate"] == picks[r["beta_vq"]]
        probe = brs.run_one_search(m, items, 2, float(r["beta_rate"]), float(r["beta_vq"]))
        assert abs(r["bpp"] - probe) < 1e-6        # run_model's per-image bits == the search probe's rate (same kernels)
        assert np.isfinite(r["psnr"]) and 5.0 < r["psnr"] < 60.0 and abs(r["score"] - 2.0 * r["psnr"]) < 1e-9
        rd = sel / "target_rate_0.2" / f"beta_vq_{r['beta_vq']:.2f}"
        assert sorted(os.listdir(rd)) == ["0.png", "1.png", "2.png", "_avg_bitrate.json", "_rate_summary.csv"]
        rs = pd.read_csv(rd / "_rate_summary.csv", index_col=0)
        assert list(rs.columns) == ["img_name", "num_pixel", "total_bit", "bitrate"] and (rs["num_pixel"] == 128 * 128).all()
    top = pd.read_csv(sel / "beta_selection_results.csv")
    assert list(top.columns) == ["target_rate", "selected_beta_vq", "selected_beta_rate"] and len(top) == 1
    assert top.iloc[0]["selected_beta_vq"] == res.iloc[0]["beta_vq"]


def test_train_cli_synthetic(tmp_path):
    """scripts/train.py (reference scripts/train.py:16-27 + train_loop): stage-3 G + D iterations on synthetic crops, the log
    line per iteration, checkpoints in the reference's file format ({'iter', 'comp_model'} / {'iter', 'discriminator'}) that load
    back through load_learned_weight, the lpips state-dict loader, the refusal to train against synthetic LPIPS weights
    silently, and --resume: a run resumed from the iteration-2 training state reproduces the uninterrupted run's
    iteration-4 weights BIT FOR BIT (Adam moments, step counts, scheduler epochs, beta-sampler RNG, data stream)."""
    import torch
    out, out2 = tmp_path / "ckpt", tmp_path / "ckpt2"
    base = [sys.executable, os.path.join(ROOT, "scripts", "train.py"), os.path.join(ROOT, "config", "dc_vic_synthetic.yaml"), "--synthetic_weights",
            "--synthetic_data", "--batch_size", "2", "--log_step", "1"]
    res = subprocess.run(base + ["--total_iter", "1"], cwd=ROOT, capture_output=True, text=True)
    assert res.returncode != 0 and "SYNTHETIC" in res.stderr and "--lpips_path" in res.stderr, res.stderr[-2000:]
    from dc_vic_amd.train.lpips import LPIPSAlex
    lp = tmp_path / "lpips_alex.pth"
    torch.save({k: v.clone() for k, v in LPIPSAlex(seed=0).state_dict().items()}, lp)      # same key names as lpips.LPIPS(net='alex')
    res = subprocess.run(base + ["--total_iter", "4", "--save_dir", str(out), "--save_step", "2", "--lpips_path", str(lp)], cwd=ROOT, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("iter")]
    assert len(lines) == 4 and all("samples/s" in ln and "distortion" in ln and "perceptual" in ln and "d_total" in ln for ln in lines), res.stdout
    ck = torch.load(out / "comp_model_iter0000004.pth.tar", map_location="cpu", weights_only=True)
    assert ck["iter"] == 4 and "decoder.conv1.weight" in ck["comp_model"] and "vq_model.decoder.conv_in.weight" in ck["comp_model"]
    dk = torch.load(out / "discriminator_iter0000004.pth.tar", map_location="cpu", weights_only=True)
    assert sorted(k for k in dk["discriminator"] if k.startswith("main.")) == [f"main.{i}.{p}" for i in (0, 11, 2, 5, 8) for p in ("bias", "weight")]
    res2 = subprocess.run(base + ["--total_iter", "4", "--save_dir", str(out2), "--save_step", "4", "--lpips_path", str(lp),
                                  "--resume", str(out / "training_state_iter0000002.pth.tar")], cwd=ROOT, capture_output=True, text=True)
    assert res2.returncode == 0, res2.stderr[-2000:]
    lines2 = [ln for ln in res2.stdout.splitlines() if ln.startswith("iter")]
    assert len(lines2) == 2 and lines2[0].split("|")[0].split() == ["iter", "3"]
    assert [ln.split("|")[2] for ln in lines2] == [ln.split("|")[2] for ln in lines[2:]], (lines2, lines[2:])       # same losses, digit for digit
    ck2 = torch.load(out2 / "comp_model_iter0000004.pth.tar", map_location="cpu", weights_only=True)
    dk2 = torch.load(out2 / "discriminator_iter0000004.pth.tar", map_location="cpu", weights_only=True)
    assert all(torch.equal(ck["comp_model"][k], ck2["comp_model"][k]) for k in ck["comp_model"])
    assert all(torch.equal(dk["discriminator"][k], dk2["discriminator"][k]) for k in dk["discriminator"])
    from dc_vic_amd import BaseConfig, build_comp_model
    m = build_comp_model(BaseConfig.fromfile(os.path.join(ROOT, "config", "dc_vic_synthetic.yaml"), {"device": "cuda:0"}))
    m.load_learned_weight(str(out / "comp_model_iter0000004.pth.tar"))
    from dc_vic_amd.synth import full_synth_state_dict
    sd = full_synth_state_dict(1234)
    assert torch.equal(m.state_dict()["encoder.conv1.weight"].cpu(), sd["encoder.conv1.weight"])            # frozen: untouched
    assert not torch.equal(m.state_dict()["decoder.conv1.weight"].cpu(), sd["decoder.conv1.weight"])       # trained
