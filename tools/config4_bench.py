"""SURVEY 8(d) config 4, one GPU: a folder of variable-resolution PNGs (sizes drawn from a fixed CLIC-like list up to 2048 px,
so both the whole-image branch and the 512 / 256 tiling branch run) through scripts/compress.py --decompress, shape-bucketed.

    python tools/config4_bench.py [--images 107] [--batch_size 8] [--gpus 1] > profiles/rN_config4_variable_res.json

The images are synthetic (smooth gradients + texture, seeded): no dataset travels to the GPU box.  Reported: wall clock of the whole
CLI run (PNG decode, compress, .bin write / read, decompress, PNG encode, csv), the same minus the model set-up measured on an empty
folder, images/s and megapixels/s on the second figure.
"""
import argparse, json, os, shutil, subprocess, sys, tempfile, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (H, W, weight): the CLIC professional / mobile validation mix is mostly 2048 px on the long side
SIZES = [(1365, 2048, 30), (2048, 1365, 12), (1152, 2048, 8), (1536, 2048, 8), (1024, 1536, 8), (768, 1024, 8), (512, 768, 10),
         (768, 512, 6), (1080, 1920, 6), (720, 1280, 4)]


def synth_png(path, H, W, rng):
    from PIL import Image
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    img = np.empty((H, W, 3), np.float32)
    for c in range(3):
        fx, fy, ph = rng.uniform(0.002, 0.02), rng.uniform(0.002, 0.02), rng.uniform(0, 6.28)
        img[..., c] = 127 + 90 * np.sin(fx * xx + fy * yy + ph) + 20 * np.sin(0.11 * xx * (c + 1)) * np.cos(0.07 * yy)
    img += rng.normal(0, 4, img.shape).astype(np.float32)
    Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(path, compress_level=1)


def run_cli(img_dir, save_dir, a):
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "compress.py"), "--config_path",
           os.path.join(ROOT, "config", "dc_vic_synthetic.yaml"), "--synthetic_weights", "--img_dir", img_dir, "--save_dir", save_dir, "-q", "0", "-d", "cuda", "--decompress", "--batch_size", str(a.batch_size)]
    if a.gpus > 1:
        cmd += ["--gpus", str(a.gpus)]
    t0 = time.time()
    subprocess.run(cmd, check=True, stdout=sys.stderr, env=dict(os.environ, DCVIC_CLI_TIMING="1") if a.timing else None)
    return time.time() - t0


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--images", type=int, default=107)
    p.add_argument("--batch_size", type=int, default=8)
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--timing", action="store_true", help="per-phase seconds of the CLI on stderr (adds a device synchronise per phase)")
    a = p.parse_args()
    rng = np.random.default_rng(4)
    w = np.array([s[2] for s in SIZES], np.float64)
    pick = rng.choice(len(SIZES), size=a.images, p=w / w.sum())
    tmp = tempfile.mkdtemp(prefix="dcvic_cfg4_")
    try:
        src, out, empty, out0 = (os.path.join(tmp, d) for d in ("src", "out", "empty", "out0"))
        for d in (src, out, empty, out0):
            os.makedirs(d)
        t0 = time.time()
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(8) as ex:
            list(ex.map(lambda k: synth_png(os.path.join(src, f"img{k:04d}.png"), SIZES[pick[k]][0], SIZES[pick[k]][1],
                                             np.random.default_rng(100 + k)), range(a.images)))
        t_gen = time.time() - t0
        t_setup = run_cli(empty, out0, a)
        t_all = run_cli(src, out, a)
        mpx = sum(SIZES[k][0] * SIZES[k][1] for k in pick) / 1e6
        import pandas as pd
        df = pd.read_csv(os.path.join(out, "_bitrates.csv"))
        assert len(df) == a.images and all(os.path.exists(os.path.join(out, f"img{k:04d}.png")) for k in range(a.images))
        tiled = int(sum(max(SIZES[k][0], SIZES[k][1]) > 1024 for k in pick))
        work = max(t_all - t_setup, 1e-9)
        print(json.dumps({"config": "SURVEY 8(d) config 4 (synthetic CLIC-like sizes), scripts/compress.py --decompress -q 0",
                          "images": a.images, "tiled_images": tiled, "megapixels": round(mpx, 1), "batch_size": a.batch_size,
                          "gpus": a.gpus, "wall_s": round(t_all, 2), "setup_s_empty_folder": round(t_setup, 2),
                          "images_per_s": round(a.images / work, 2), "megapixels_per_s": round(mpx / work, 2),
                          "avg_bpp": float(df["real_bpp"].mean()), "png_synthesis_s": round(t_gen, 1),
                          "shapes": sorted({f"{SIZES[k][0]}x{SIZES[k][1]}" for k in pick})}, indent=1))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
