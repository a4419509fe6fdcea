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
        assert b["threads_unpinned"] == b["threads_pinned"] == min(16, share)
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
