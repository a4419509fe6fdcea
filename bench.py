#!/usr/bin/env python3
"""bench.py -- images/s of DC-VIC encode+decode at 256x256, q=0, on N x MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of B=32 synthetic 256x256 images that are already
resident in HBM: compress_batch (VQGAN encoder -> VQ search -> ELIC encoder -> hyper-encoder ->
hyper-decoder + CHARM -> symbols -> host rANS -> 3 byte strings per image) followed by decompress_batch
(host rANS decode interleaved with the GPU-resident CHARM -> ELIC decoder features -> Swin estimator ->
argmax LUT -> SFT-fused VQGAN decoder -> crop/clamp), i.e. real bytes are produced and consumed.
Images shard across GPUs (weak scaling: B per GPU); RCCL only all-gathers the per-image rate table.
Weights: deterministic synthetic (dc_vic_amd.synth) -- no checkpoint can be fetched offline.

Output: ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant
kernel (HIP-event timed inside the timed region) and `cpu_baseline` (the CPU oracle on the host cores).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
PEAK_HBM_TBPS = 8.0              # MI355X_MICROARCH.md: HBM3E peak (achievable ~5 - 6 TB/s)
GFLOP_PER_IMAGE = 1009.6         # one 256x256 image, compress + decompress


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--batch", type=int, default=32)
    p.add_argument("--quality", type=int, default=0)
    p.add_argument("--height", type=int, default=256, help="image height (default 256: the BASELINE metric; 512 x 768 = Kodak-shaped)")
    p.add_argument("--width", type=int, default=256)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-kernel-events", action="store_true", help="skip the per-launch HIP events (roofline becomes null)")
    p.add_argument("--event-steps", type=int, default=2, help="eager steps with per-launch HIP events behind the timed region (roofline)")
    return p.parse_args()


def cpu_baseline(sd, quality: int):
    """The oracle (CPU restatement of the reference's path) on this host's cores, bounded sample."""
    from oracle.dcvic_oracle import Oracle
    try:
        cores = len(os.sched_getaffinity(0))      # the box's CPU share, not the host's core count
    except AttributeError:
        cores = os.cpu_count() or 1
    share = cores
    cap = int(os.environ.get("DCVIC_CPU_BASELINE_THREADS", "16"))     # torch-CPU convs stop scaling past ~16 threads on these maps
    cores = max(1, min(cores, cap))
    torch.set_num_threads(cores)
    orc = Oracle(sd)
    g = torch.Generator().manual_seed(99)
    x = torch.rand((1, 3, 256, 256), generator=g) * 2 - 1
    r = orc.compress(x, quality)           # warm-up (first-call allocations)
    orc.decompress(r["string_list"])
    n = int(os.environ.get("DCVIC_CPU_BASELINE_SAMPLES", "8"))
    times = []
    for i in range(n):
        t0 = time.perf_counter()
        r = orc.compress(x, quality)
        orc.decompress(r["string_list"])
        times.append(time.perf_counter() - t0)
    dt = sum(times)
    return {"value": n / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "node_cores": os.cpu_count(), "process_cpu_share": share, "thread_cap": cap,
            "best_sample_images_per_s": 1.0 / min(times), "worst_sample_images_per_s": 1.0 / max(times),
            "sample": f"{n} x (compress+decompress) of one 256x256 image, q={quality}, torch {torch.__version__} fp32 CPU oracle, "
                      f"{cores} threads (cap {cap}; this process may use {share} of the node's {os.cpu_count()} cores), after 1 warm-up"}


def main():
    a = parse()
    from dc_vic_amd.parallel import launched_by_a_launcher, self_launch
    if not launched_by_a_launcher():
        # plain `python bench.py --gpus N`: this process becomes the launcher of N ranks (one per GPU) BEFORE anything touches
        # the GPU; fewer than N devices is an error, never a 1-rank run reported as the answer to --gpus N
        if a.gpus > 1:
            sys.exit(self_launch(a.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from dc_vic_amd.parallel import pin_rank_cpus
    cpus = pin_rank_cpus()            # the ranks of a node share its cores: each takes its slice (rANS threads)
    dist = None
    saved_stdout = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ or os.environ.get("DCVIC_FORCE_DIST") == "1":
        # RCCL prints a version banner on STDOUT when its first communicator comes up; the contract is ONE JSON line there.  Point fd 1
        # at stderr until the result line is printed (every rank: only rank 0 ever writes to stdout).
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        # launched by torch.distributed.run: take the collective path even at world size 1 (same code as N > 1)
        import torch.distributed as dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        dist = dist_

    from dc_vic_amd import BaseConfig, build_comp_model, ops
    from dc_vic_amd.parallel import gather_rate_table
    from dc_vic_amd.synth import load_synth_weights

    opt = BaseConfig.fromfile(os.path.join(ROOT, "config", "dc_vic_synthetic.yaml"), {"device": str(dev)})
    model = build_comp_model(opt)
    sd = load_synth_weights(model, 1234)
    model.codec_setup()

    B = a.batch
    g = torch.Generator().manual_seed(1000 + rank)            # every rank codes its own shard
    IH, IW = a.height, a.width
    x = (torch.rand((B, 3, IH, IW), generator=g) * 2 - 1).to(dev)

    gather_s = [0.0]

    def step():
        r = model.compress_batch(x, a.quality)
        imgs, _, _ = model.decompress_batch(r["string_lists"])
        real_bits = np.array([8.0 * sum(len(s) for s in sl) + 32 * 3 for sl in r["string_lists"]])   # + 3 uint32 length prefixes
        table = np.stack([real_bits, r["pred_y_bit"] + r["pred_z_bit"]], axis=1)
        if dist is not None:
            torch.cuda.synchronize(dev)                   # (so that the figure below is the collective, not the tail of the decoder kernels)
        tg = time.perf_counter()
        table = gather_rate_table(table, dist, dev)       # the only collective: a few KB over RCCL
        gather_s[0] += time.perf_counter() - tg
        return imgs, table

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    # model setup (untimed, before the W warm-up steps): lazy weight packing on the first pass, hipGraph capture of the encoder / decoder
    # networks on the second (comp_model._GraphCache captures a shape the second time it sees it)
    t_s = time.perf_counter()
    for _ in range(2):
        step()
    sync()
    log(f"model set up (weight packs, hipGraph capture) in {time.perf_counter() - t_s:.2f}s")
    t_w = time.perf_counter()
    for _ in range(a.warmup):
        step()
    sync()
    log(f"{a.warmup} warm-up step(s) in {time.perf_counter() - t_w:.2f}s")
    gather_s[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(a.steps):
        imgs, table = step()
    torch.cuda.synchronize(dev)
    dt_local = time.perf_counter() - t0       # this rank's own time (before the barrier): shows load imbalance
    sync()
    dt = time.perf_counter() - t0
    log(f"{a.steps} timed step(s) in {dt:.2f}s")
    # Per-launch HIP events for the roofline.  The timed steps replay the two network segments as hipGraphs, which cannot carry
    # per-launch events, so the SAME step is run `--event-steps` more times right behind the timed region with the events on (eager
    # launches of the same kernels with the same arguments, on the stream they run on); rocprofv3 of this command reports the same
    # average durations (profiles/).  dt_ev = wall time of those steps, the denominator of the per-kernel time shares.
    ev, dt_ev, gather_keep = None, None, gather_s[0]
    if not a.no_kernel_events:
        ops.kernel_events_start()
        t_e = time.perf_counter()
        for _ in range(max(1, a.event_steps)):
            step()
        sync()
        dt_ev = time.perf_counter() - t_e
        ev = ops.kernel_events_stop()
        gather_s[0] = gather_keep
    if ev and rank == 0 and os.environ.get("DCVIC_BENCH_DETAIL"):
        print(ops.shape_stats_report(), file=sys.stderr, flush=True)

    # per-stage wall clock, from ONE extra instrumented step outside the timed region (a device sync at
    # every stage boundary, so the stages add up to a little more than a pipelined step)
    stage_ms = None
    if not a.no_kernel_events:          # every rank runs the extra step (it contains the gather); rank 0 reports
        from dc_vic_amd import comp_model as cm
        marks = []

        def hook(name):
            torch.cuda.synchronize(dev)
            marks.append((name, time.perf_counter()))
        cm.STAGE_HOOK = hook
        try:
            step()
        finally:
            cm.STAGE_HOOK = None
        stage_ms = {}
        for (n0, t0_), (n1, t1_) in zip(marks[:-1], marks[1:]):
            if n1 != "begin":
                stage_ms[n1] = stage_ms.get(n1, 0.0) + 1e3 * (t1_ - t0_)
        stage_ms["sum"] = sum(stage_ms.values())
    sync()

    per_rank = [[1e3 * dt_local / a.steps, 1e3 * gather_s[0] / a.steps, float(len(cpus))]]
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        mine = torch.tensor(per_rank[0], dtype=torch.float64, device=dev)
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        per_rank = [p_.tolist() for p_ in parts]
    for r_, (ms_, g_, c_) in enumerate(per_rank):
        log(f"rank {r_}: {ms_:.1f} ms/step (own clock), rate-table all_gather {g_:.2f} ms/step, {int(c_)} host cores")

    if rank == 0:
        n_img = world * B * a.steps
        value = n_img / dt
        avg_bpp = float(table[:, 0].mean() / (IH * IW))
        out = {
            "metric": f"images/sec encode+decode @{IH}x{IW} q={a.quality}", "value": value, "unit": "images/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"dc_vic_patchgan.yaml architecture, synthetic random {IH}x{IW}, batch={B}/GPU, q={a.quality}, "
                                   "compress_batch+decompress_batch through real rANS bytes, synthetic weights",
                       "batch_per_gpu": B, "quality": a.quality, "image": f"{IH}x{IW}", "parallelism": f"dp{world} (images sharded, RCCL all_gather of the rate table)"},
            "avg_bpp": avg_bpp,
            "per_rank_ms_per_step": [p_[0] for p_ in per_rank], "per_rank_gather_ms_per_step": [p_[1] for p_ in per_rank],
            "per_rank_host_cores": [int(p_[2]) for p_ in per_rank],
            "stage_ms_per_step": stage_ms,
            "avg_pred_bpp": float(table[:, 1].mean() / (IH * IW)),
            # ALGORITHMIC work rate (1 009.6 GF per 256^2 image at direct-convolution cost) per GPU: a throughput in
            # TFLOP/s of useful work, NOT a utilisation (Winograd skips multiplies); the utilisation is
            # end_to_end_executed_conv_mfma_frac_of_f32_peak below
            "end_to_end_algorithmic_tflops_per_gpu": value / world * GFLOP_PER_IMAGE * (IH * IW) / 65536.0 / 1e3,
        }
        if ev:
            k = max(ev.values(), key=lambda d: d["time_s"])
            # HBM traffic per launch of that kernel: PMC counters need rocprofv3 (separate --pmc passes, see
            # tools/pmc_summary.py); the committed summary of the same command is reported when present
            traffic, traffic_src = None, None
            for pmf in ("r3_pmc_traffic.json", "r2_pmc_traffic.json", "r2_direct_pmc_traffic.json", "r1_pmc_traffic.json"):
                try:
                    with open(os.path.join(ROOT, "profiles", pmf)) as f:
                        pm = json.load(f)
                    if k["kernel"] in pm:
                        traffic = pm[k["kernel"]]["hbm_bytes_per_launch"]
                        traffic_src = f"profiles/{pmf} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py; 2*FETCH+WRITE, bytes per launch)"
                        break
                except (OSError, ValueError):
                    pass
            # flop accounting: `achieved` / `frac` are the MFMA flops the kernel EXECUTES per second (a matrix-pipe utilisation,
            # <= 1): for a direct convolution that is the algorithmic 2*MAC count of SURVEY 8(d); Winograd F(2x2,3x3) issues 16/36
            # of it (F(4x4,3x3) 36/144, the upsample form 9/36).  `algorithmic_tflops` keeps the direct-convolution figure, i.e.
            # the rate at which the layer's work gets done (> peak when the algorithm skips multiplies).
            alg = k["flops"] / k["time_s"] / 1e12
            exe = k["exec_flops"] / k["time_s"] / 1e12
            tot_exec = sum(d["exec_flops"] for d in ev.values())
            out["end_to_end_executed_conv_mfma_frac_of_f32_peak"] = tot_exec / max(1, a.event_steps) * a.steps / dt / 1e12 / PEAK_F32_MFMA_TFLOPS
            out["roofline"] = {"bound": "mfma", "kernel": k["kernel"], "achieved": exe,
                               "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": exe / PEAK_F32_MFMA_TFLOPS,
                               "algorithmic_tflops": alg, "executed_per_algorithmic_flop": k["exec_flops"] / k["flops"],
                               "flop_accounting": "achieved / frac = MFMA flops EXECUTED per second / fp32 matrix peak (pipe utilisation); "
                                                  "algorithmic_tflops = SURVEY 8d direct-convolution flops (2*N*H*W*Cout*Cin*9 per launch) per second",
                               "traffic": traffic, "traffic_source": traffic_src, "launches": k["launches"], "avg_launch_us": 1e6 * k["time_s"] / k["launches"],
                               "gflop_per_launch": k["flops"] / k["launches"] / 1e9,
                               "executed_gflop_per_launch": k["exec_flops"] / k["launches"] / 1e9,
                               "share_of_step_time": k["time_s"] / dt_ev,
                               "measured_over": f"{max(1, a.event_steps)} eager step(s) with per-launch HIP events right behind the timed region "
                                                "(the timed steps replay hipGraphs, which cannot carry per-launch events); same kernels, same arguments",
                               # per kernel: executed MFMA rate, and the rate of its ALGORITHMIC bytes (input read once, output written
                               # once, weights; residual reads not counted) against HBM -- `bound` names the roof it sits closer to
                               # (the 96 / 192-channel 1x1 layers and the 3-channel layers are HBM-bound: < 20 flop per byte)
                               "all_conv_kernels": {n: {"tflops": d["exec_flops"] / d["time_s"] / 1e12,
                                                        "algorithmic_tflops": d["flops"] / d["time_s"] / 1e12, "launches": d["launches"],
                                                        "mfma_frac": d["exec_flops"] / d["time_s"] / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                                        "algorithmic_hbm_tbps": d["bytes"] / d["time_s"] / 1e12,
                                                        "hbm_frac": d["bytes"] / d["time_s"] / 1e12 / PEAK_HBM_TBPS,
                                                        "bound": "hbm" if d["bytes"] / PEAK_HBM_TBPS > d["exec_flops"] / PEAK_F32_MFMA_TFLOPS else "mfma",
                                                        "time_share": d["time_s"] / dt_ev} for n, d in ev.items()}}
        else:
            out["roofline"] = None
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd, a.quality)
        else:
            out["cpu_baseline"] = None
        if saved_stdout is not None:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)                      # the real stdout back for the one result line
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
