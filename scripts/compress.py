#!/usr/bin/env python3
"""Encode / decode CLI -- drop-in for the reference's scripts/compress.py (85-144).

Same flags (--config_path --model_path --img_dir --save_dir -q/--quality --decompress -d/--device),
same inputs (sorted *.png of --img_dir, RGB, scaled to [-1, 1] as ToTensor + Normalize(.5, .5) do,
compress.py:52-70) and same outputs: <name>.bin (codec_utils container), <name>.png when
--decompress (truncating uint8 as img_utils.imwrite), _bitrates.csv (columns of compress.py:117-129,
written through pandas like the reference) and _avg_bitrate.json {"avg_bpp": mean(real_bpp)}.

MI355X additions (not in the reference, which is single-GPU, README.md:64-65):
  * --batch_size B codes B same-sized images per call (one rANS stream per image); PNG decode / encode run on
    --io_workers threads with pinned-memory staging, the next batch is decoded while the GPU codes the current one;
  * `--gpus N` (self-launching: N child ranks, one per GPU) or `python -m torch.distributed.run --nproc-per-node N`: the image list is sharded
    across the N GPUs (longest-processing-time-first on padded pixel count), every rank writes its own
    .bin/.png files and the per-image rows are all-gathered over RCCL so rank 0 writes the same csv /
    json a single-GPU run writes.
  * --synthetic_weights skips --model_path and loads the deterministic synthetic weights (no checkpoint
    can be fetched offline); the YAML's vq_model.ckpt_path is ignored when it does not exist.
"""
from __future__ import annotations

import json
import os
import sys
from glob import glob

import numpy as np
import torch

sys.path.append(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from dc_vic_amd import BaseConfig, build_comp_model  # noqa: E402
from dc_vic_amd.codec_utils import load_byte_strings, save_byte_strings  # noqa: E402
from dc_vic_amd.io_pipeline import AsyncWriter, BatchPrefetcher, encode_png_u8  # noqa: E402
from dc_vic_amd.options import compress_arg_parser  # noqa: E402
from dc_vic_amd.parallel import gather_rate_table, launched_by_a_launcher, pin_rank_cpus, self_launch, shard_indices  # noqa: E402

COLUMNS = ["img_name", "header_bit", "z_bit", "y_bit", "real_bit", "real_bpp", "pred_z_bit", "pred_y_bit", "pred_bit",
           "pred_bpp", "num_pixel"]


def load_png(path: str) -> torch.Tensor:
    from PIL import Image
    img = np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)
    x = torch.from_numpy(img.copy()).permute(2, 0, 1).float().div(255.0)       # ToTensor
    return ((x - 0.5) / 0.5).unsqueeze(0)                                      # Normalize(.5, .5)


def write_png(path: str, u8_hwc: np.ndarray) -> None:
    from PIL import Image
    Image.fromarray(u8_hwc, mode="RGB").save(path)


def main():
    p = compress_arg_parser()
    p.add_argument("--batch_size", type=int, default=1)
    p.add_argument("--synthetic_weights", action="store_true")
    p.add_argument("--io_workers", type=int, default=0, help="PNG decode / encode threads (0: min(8, this rank's share of the cores))")
    p.add_argument("--gpus", type=int, default=0, help="shard the folder over N GPUs of this node: without a launcher this process starts "
                                                        "the N ranks itself (0: as launched -- WORLD_SIZE ranks under torch.distributed.run, else 1)")
    args = p.parse_args()
    if args.gpus > 1 and not launched_by_a_launcher():
        sys.exit(self_launch(args.gpus, need_gpus=args.device.startswith("cuda")))      # parent: never touches the GPU
    if args.gpus > 0 and int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    device = args.device
    pin_rank_cpus()                      # N ranks share the host: each takes its slice of the cores (rANS + PNG threads)
    if world > 1:
        import torch.distributed as dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if device.startswith("cuda"):
            device = f"cuda:{local_rank}"
            torch.cuda.set_device(local_rank)
            dist_.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
        else:
            dist_.init_process_group("gloo", rank=rank, world_size=world)
        dist = dist_

    overrides = {k: v for k, v in vars(args).items() if k not in ("batch_size", "synthetic_weights", "io_workers", "gpus")}
    overrides["device"] = device
    if device == "cuda":
        device = overrides["device"] = f"cuda:{torch.cuda.current_device()}"      # the reference's `-d cuda`
    if device.startswith("cuda"):
        torch.cuda.set_device(torch.device(device))      # `-d cuda:1` makes cuda:1 the current device (kernels launch there)
    overrides["is_train"] = False
    opt = BaseConfig.fromfile(args.config_path, overrides)
    ck = opt["subnet"]["vq_model"].get("ckpt_path")
    if ck and not os.path.exists(ck):
        print(f"[compress] VQGAN checkpoint {ck} not found: skipping (weights must come from --model_path)", file=sys.stderr)
        opt["subnet"]["vq_model"]["ckpt_path"] = None
    os.makedirs(args.save_dir, exist_ok=True)

    img_path_list = sorted(glob(os.path.join(args.img_dir, "*.png")))
    model = build_comp_model(opt)
    if args.synthetic_weights:
        from dc_vic_amd.synth import load_synth_weights
        load_synth_weights(model, 1234)
    else:
        model.load_learned_weight(ckpt_path=args.model_path)
    model.codec_setup()

    # shard by padded pixel count (images are independent units)
    sizes = []
    from PIL import Image
    for pth in img_path_list:
        with Image.open(pth) as im:
            w, h = im.size
        sizes.append((h, w))
    costs = [int(np.ceil(h / 64) * 64) * int(np.ceil(w / 64) * 64) for h, w in sizes]
    mine = shard_indices(len(img_path_list), rank, world, costs if world > 1 else None)

    rows = {}
    # bucket the local images by shape so that equal shapes share a batch
    buckets = {}
    for i in mine:
        buckets.setdefault(sizes[i], []).append(i)
    chunks = []
    for (H, W), idxs in buckets.items():
        for s in range(0, len(idxs), max(1, args.batch_size)):
            chunks.append(idxs[s:s + max(1, args.batch_size)])
    writer = AsyncWriter(args.io_workers or None)
    loader = BatchPrefetcher([[img_path_list[i] for i in c] for c in chunks], device, workers=args.io_workers or None)
    timing = {"wait_png_decode": 0.0, "compress": 0.0, "bin_io_csv_rows": 0.0, "decompress": 0.0, "d2h_submit_png": 0.0, "drain_png_encode": 0.0}
    import time
    def lap(key, t0):
        if device.startswith("cuda") and os.environ.get("DCVIC_CLI_TIMING"):
            torch.cuda.synchronize()
        timing[key] += time.time() - t0
        return time.time()
    try:
        it = iter(loader)
        for chunk in chunks:
            t = time.time()
            _, x = next(it)
            t = lap("wait_png_decode", t)
            H, W = sizes[chunk[0]]
            out = model.compress_batch(x, args.quality)
            t = lap("compress", t)
            bins = []
            for j, i in enumerate(chunk):
                name = os.path.basename(img_path_list[i])
                sl = out["string_lists"][j]
                bin_path = os.path.join(args.save_dir, name.replace(".png", ".bin"))
                save_byte_strings(bin_path, sl)
                actual_byte = os.path.getsize(bin_path)
                bins.append(bin_path)
                rows[i] = [len(sl[0]) * 8, len(sl[1]) * 8, len(sl[2]) * 8, actual_byte * 8, actual_byte * 8 / H / W,
                           float(out["pred_z_bit"][j]), float(out["pred_y_bit"][j]),
                           float(out["pred_z_bit"][j] + out["pred_y_bit"][j]),
                           float(out["pred_z_bpp"][j] + out["pred_y_bpp"][j]), H * W]
            if args.decompress:
                loaded = [load_byte_strings(b) for b in bins]
                t = lap("bin_io_csv_rows", t)
                _, _, _, u8 = model.decompress_batch(loaded, want_u8=True)
                t = lap("decompress", t)
                u8 = u8.cpu().numpy()
                for j, i in enumerate(chunk):
                    writer.submit(encode_png_u8, os.path.join(args.save_dir, os.path.basename(img_path_list[i])), u8[j].copy())
                t = lap("d2h_submit_png", t)
    finally:
        t = time.time()
        writer.close()
        timing["drain_png_encode"] = time.time() - t
    if os.environ.get("DCVIC_CLI_TIMING"):
        print(f"[compress] rank {rank} seconds: " + json.dumps({k: round(v, 2) for k, v in timing.items()}), file=sys.stderr)

    # gather the per-image rows (RCCL all_gather of a small fp64 table) and write the summary on rank 0
    local = np.array([[float(i)] + rows[i] for i in sorted(rows)], dtype=np.float64).reshape(-1, 11)
    table = gather_rate_table(local, dist, torch.device(device) if device.startswith("cuda") else None)
    if rank == 0:
        import pandas as pd
        table = table[np.argsort(table[:, 0])]
        recs = []
        for r in table:
            i = int(r[0])
            rec = {"img_name": os.path.basename(img_path_list[i]), "header_bit": int(r[1]), "z_bit": int(r[2]), "y_bit": int(r[3]),
                   "real_bit": int(r[4]), "real_bpp": r[5], "pred_z_bit": r[6], "pred_y_bit": r[7], "pred_bit": r[8],
                   "pred_bpp": r[9], "num_pixel": int(r[10])}
            recs.append(rec)
        df = pd.json_normalize(recs) if recs else pd.DataFrame(columns=COLUMNS)
        df.to_csv(os.path.join(args.save_dir, "_bitrates.csv"))
        with open(os.path.join(args.save_dir, "_avg_bitrate.json"), "w") as f:
            json.dump({"avg_bpp": float(df["real_bpp"].mean()) if len(df) else float("nan")}, f)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
