"""N>1 path on CPU: image sharding + the all_gather of the per-image rate table, world_size 2 over gloo."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from dc_vic_amd.parallel import gather_rate_table, rank_cpu_slice, shard_indices


def test_shard_indices_partition_and_balance():
    costs = [512 * 768] * 5 + [2048 * 1344] * 2 + [64 * 64] * 9
    for world in (1, 2, 3, 8):
        parts = [shard_indices(len(costs), r, world, costs) for r in range(world)]
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(len(costs)))                       # every image exactly once
        loads = [sum(costs[i] for i in p) for p in parts]
        assert max(loads) - min(loads) <= max(costs)                  # LPT bound
    assert shard_indices(10, 1, 4) == [1, 5, 9]                        # uniform cost -> round robin
    assert shard_indices(3, 0, 1) == [0, 1, 2]


def _worker(rank, world, port, out_dir):
    import json
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_WORLD_SIZE"] = str(world)
    os.environ["LOCAL_RANK"] = str(rank)
    os.environ.pop("DCVIC_HOST_THREADS", None)
    from dc_vic_amd.entropy import host_threads
    from dc_vic_amd.io_pipeline import default_workers
    from dc_vic_amd.parallel import host_core_budget, pin_rank_cpus
    before = sorted(os.sched_getaffinity(0))
    budget = dict(n_before=len(before), threads_unpinned=host_threads(), io_unpinned=default_workers(), budget_unpinned=host_core_budget())
    cpus = pin_rank_cpus()
    budget.update(cpus=cpus, threads_pinned=host_threads(), io_pinned=default_workers())
    with open(os.path.join(out_dir, f"budget{rank}.json"), "w") as f:
        json.dump(budget, f)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_items = 7
    mine = shard_indices(n_items, rank, world, [10, 1, 1, 1, 9, 1, 1])
    table = np.array([[i, 100.0 + i, 0.5 * i] for i in mine], dtype=np.float64).reshape(-1, 3)
    full = gather_rate_table(table, dist)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), full)
    dist.barrier()
    dist.destroy_process_group()


def test_gather_rate_table_world2(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    assert np.array_equal(a, b)                                        # every rank sees the same table
    assert sorted(a[:, 0].tolist()) == list(range(7))                  # ragged shards (4 + 3 rows) gathered without loss
    a = a[np.argsort(a[:, 0])]
    assert np.allclose(a[:, 1], 100.0 + np.arange(7)) and np.allclose(a[:, 2], 0.5 * np.arange(7))
    # single process: identity
    t = np.arange(6, dtype=np.float64).reshape(3, 2)
    assert np.array_equal(gather_rate_table(t, None), t)
    # host-core budget (the ranks of a node share its cores): each rank takes affinity // LOCAL_WORLD_SIZE
    # rANS / PNG threads, pinned or not, and the pinned CPU sets of the two ranks are disjoint
    import json
    b0, b1 = (json.load(open(tmp_path / f"budget{r}.json")) for r in (0, 1))
    n = b0["n_before"]
    share = max(1, n // 2)
    for b in (b0, b1):
        assert b["budget_unpinned"] == share
        assert b["threads_unpinned"] == b["threads_pinned"] == min(32, share)
        assert b["io_unpinned"] == b["io_pinned"] == min(8, share)
        assert len(b["cpus"]) == share
    if n >= 2:
        assert not set(b0["cpus"]) & set(b1["cpus"])
    assert b0["threads_pinned"] + b1["threads_pinned"] <= max(n, 2)


def test_rank_cpu_slices_cover_and_do_not_overlap():
    cpus = list(range(3, 3 + 128))
    for lw in (1, 2, 4, 8):
        sl = [rank_cpu_slice(r, lw, cpus) for r in range(lw)]
        flat = [c for s_ in sl for c in s_]
        assert len(flat) == len(set(flat)) == 128 and all(len(s_) == 128 // lw for s_ in sl)
    assert rank_cpu_slice(5, 8, [0, 1]) == [1]        # more ranks than cores: wrap, never empty


def _ar_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dc_vic_amd.train.trainer import allreduce_mean_
    g = torch.Generator().manual_seed(rank)
    flat = torch.randn(100003, generator=g)
    nb = allreduce_mean_(flat, dist, bucket_bytes=64 * 1024)          # 7 buckets, the last one ragged
    torch.save({"flat": flat, "buckets": nb}, os.path.join(out_dir, f"ar{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_gradient_allreduce_world2(tmp_path):
    """Data-parallel training: the flat gradient buffer is averaged over the ranks with bucketed
    all-reduces; every rank ends with the same mean, whatever the bucket boundaries."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_ar_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "ar0.pt"), torch.load(tmp_path / "ar1.pt")
    assert a["buckets"] == b["buckets"] == 7
    assert torch.equal(a["flat"], b["flat"])
    ref = 0.5 * (torch.randn(100003, generator=torch.Generator().manual_seed(0)) + torch.randn(100003, generator=torch.Generator().manual_seed(1)))
    assert torch.allclose(a["flat"], ref, rtol=0, atol=1e-7)
    from dc_vic_amd.train.trainer import MultiStepLR, allreduce_mean_
    t = torch.ones(5)
    assert allreduce_mean_(t, None) == 0 and torch.equal(t, torch.ones(5))         # single process: untouched
    sch = MultiStepLR(1e-4, [3, 5], 0.1)
    lrs = []
    for _ in range(7):
        lrs.append(sch.lr()); sch.step()
    assert np.allclose(lrs, [1e-4] * 3 + [1e-5] * 2 + [1e-6] * 2)


# ------------------------------------------------------------------------------------------- `--gpus N` self-launch
def _run(cmd, timeout=300):
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    return subprocess.run([sys.executable] + cmd, env=env, capture_output=True, text=True, timeout=timeout)


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_starts_n_ranks_gloo():
    """`prog --gpus 2` with no launcher in the environment must itself become two ranks: the spawner sets
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, rank 0 prints the one result line with n_gpus = 2, exit code 0."""
    import json
    r = _run([os.path.join(ROOT, "tests", "_spawn_probe.py"), "--gpus", "2"])
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out == {"n_gpus": 2, "rows": [float(i) for i in range(7)], "self_launched": True, "local_world": "2"}
    r1 = _run([os.path.join(ROOT, "tests", "_spawn_probe.py"), "--gpus", "1"])         # N = 1: no children, plain run
    assert r1.returncode == 0 and json.loads(r1.stdout.strip())["self_launched"] is False


def test_self_launch_propagates_a_rank_failure():
    """One rank dying must end the job with its exit code instead of leaving the others in a collective forever."""
    r = _run([os.path.join(ROOT, "tests", "_spawn_probe.py"), "--gpus", "2", "--fail_rank", "1"], timeout=120)
    assert r.returncode == 7


def test_gpus_n_without_devices_fails_loudly():
    """With fewer HIP devices than --gpus asks for, bench.py / compress.py / train.py exit non-zero and say why -- never a
    1-rank run reported as the answer (and never a CPU fallback).  Runs wherever fewer than 2 devices are visible."""
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a node with < 2 HIP devices")
    for cmd in ([os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                [os.path.join(ROOT, "scripts", "compress.py"), "--config_path", os.path.join(ROOT, "config", "dc_vic_synthetic.yaml"),
                 "--img_dir", ROOT, "--save_dir", "/tmp/_dcvic_never", "--gpus", "2", "--synthetic_weights", "-q", "0"],
                [os.path.join(ROOT, "scripts", "train.py"), os.path.join(ROOT, "config", "dc_vic_synthetic.yaml"), "--gpus", "2", "--synthetic_weights",
                 "--synthetic_data"]):
        r = _run(cmd)
        assert r.returncode != 0
        assert "--gpus 2 requested but this node exposes" in r.stderr, r.stderr
        assert not any(ln.startswith("{") for ln in r.stdout.splitlines())
    # a launcher that disagrees with --gpus is an error too
    import subprocess
    import sys
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def _anomaly_worker(rank, world, port, out_dir):
    import json
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dc_vic_amd.train.trainer import loss_anomaly
    res = [loss_anomaly(float("inf") if rank == 1 else 3.0, dist),      # one rank sees an inf loss
           loss_anomaly(2.0e4 if rank == 0 else 1.0, dist),             # one rank above the 1e4 threshold
           loss_anomaly(5.0 + rank, dist)]                               # healthy everywhere
    with open(os.path.join(out_dir, f"anom{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


def test_loss_anomaly_skip_is_collective_world2(tmp_path):
    """the reference's loss-anomaly skip (base_trainer.py:235-245) decided per rank would leave the healthy ranks
    blocked in the gradient all-reduce.  The decision is one MAX all-reduce: every rank skips or none does."""
    import json
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_anomaly_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = (json.load(open(tmp_path / f"anom{r}.json")) for r in (0, 1))
    assert a == b == [True, True, False]
    from dc_vic_amd.train.trainer import loss_anomaly
    assert loss_anomaly(float("nan"), None) and loss_anomaly(1.0e5, None) and not loss_anomaly(12.0, None)
