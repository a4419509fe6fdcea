"""Multi-GPU data parallelism of the codec: images are independent units, so ranks never exchange
data on the data path; the only collective is an all_gather of the small per-image result table
(SURVEY sec.8e).  One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on
ROCm, "gloo" on CPU for tests).  The reference is single-process (README.md:64-65) -- this is new.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import numpy as np
import torch


# ------------------------------------------------------------------------------------------- host-core budget
# N ranks of one node share the host's cores: the rANS coder threads and the PNG workers of every rank
# together must not exceed them, so each rank gets affinity // LOCAL_WORLD_SIZE cores and may pin itself to its slice.
def _affinity() -> List[int]:
    try:
        return sorted(os.sched_getaffinity(0))
    except AttributeError:
        return list(range(os.cpu_count() or 1))


def local_world_size() -> int:
    return max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")) or 1))


def host_core_budget() -> int:
    """Cores this rank may use: its share of the process affinity (already the share after pin_rank_cpus())."""
    n = len(_affinity())
    if os.environ.get("DCVIC_CPUS_PINNED") == "1":
        return max(1, n)
    return max(1, n // local_world_size())


def rank_cpu_slice(local_rank: int, local_world: int, cpus: Optional[Sequence[int]] = None) -> List[int]:
    """Contiguous slice of the affinity list for a local rank (disjoint across ranks, all cores used when divisible)."""
    cpus = list(_affinity() if cpus is None else cpus)
    k = max(1, len(cpus) // max(1, local_world))
    lo = (local_rank * k) % len(cpus)
    return cpus[lo:lo + k] or cpus[:1]


def pin_rank_cpus() -> List[int]:
    """Pin this process (and the threads it creates later) to its rank's core slice.  No-op for a single local rank or
    when DCVIC_PIN_CPUS=0.  Returns the CPU list in effect."""
    lw = local_world_size()
    if lw <= 1 or os.environ.get("DCVIC_PIN_CPUS", "1") == "0" or os.environ.get("DCVIC_CPUS_PINNED") == "1":
        return _affinity()
    sl = rank_cpu_slice(int(os.environ.get("LOCAL_RANK", "0")), lw)
    try:
        os.sched_setaffinity(0, sl)
        os.environ["DCVIC_CPUS_PINNED"] = "1"
    except (AttributeError, OSError):
        pass
    return _affinity()


def shard_indices(n_items: int, rank: int, world: int, costs: Optional[Sequence[float]] = None) -> List[int]:
    """Indices of the items rank `rank` codes.  Uniform cost -> round robin; otherwise longest-
    processing-time-first on `costs` (e.g. padded pixel count of variable-resolution images), which is
    deterministic and identical on every rank."""
    if world <= 1:
        return list(range(n_items))
    if costs is None:
        return list(range(rank, n_items, world))
    order = sorted(range(n_items), key=lambda i: (-float(costs[i]), i))
    load = [0.0] * world
    mine: List[int] = []
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        load[r] += float(costs[i])
        if r == rank:
            mine.append(i)
    return sorted(mine)


def gather_rate_table(table: np.ndarray, dist, device=None) -> np.ndarray:
    """all_gather of a per-image fp64 table [n_local, k] -> [sum n_local, k] in rank order.
    Ranks may hold different row counts (ragged shards): rows are padded to the maximum and trimmed."""
    table = np.ascontiguousarray(table, dtype=np.float64)
    if dist is None or not dist.is_initialized():
        return table
    world = dist.get_world_size()
    dev = device if device is not None else torch.device("cpu")
    n = torch.tensor([table.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    k = table.shape[1]
    pad = torch.zeros((max(counts), k), dtype=torch.float64, device=dev)
    pad[: table.shape[0]] = torch.from_numpy(table).to(dev)
    parts = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return np.concatenate([p[:c].cpu().numpy() for p, c in zip(parts, counts)], axis=0)


# ------------------------------------------------------------------------------------------- self-launch
def _free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def launched_by_a_launcher() -> bool:
    """True when a launcher (torch.distributed.run, or self_launch below) already set up this process as one rank."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def self_launch(n_ranks: int, argv: Optional[Sequence[str]] = None, need_gpus: bool = True) -> int:
    """`prog --gpus N` typed without a launcher: become the launcher.  Starts N child processes of the same command line,
    one per GPU (RANK = LOCAL_RANK = i, WORLD_SIZE = LOCAL_WORLD_SIZE = N, MASTER_ADDR = 127.0.0.1, a free MASTER_PORT), waits
    for them and returns the worst exit code (the caller passes it to sys.exit).  The parent never initialises the GPU
    (`torch.cuda.device_count()` does not, on this image) -- the children are ordinary child processes, not an exec.  Too few
    devices is an error, never a silent 1-rank run.  If one rank fails the others are terminated (exact PIDs)."""
    import subprocess
    import sys
    import time
    if n_ranks < 1:
        raise SystemExit(f"--gpus {n_ranks}: need at least one rank")
    if need_gpus:
        have = torch.cuda.device_count()
        if have < n_ranks:
            raise SystemExit(f"--gpus {n_ranks} requested but this node exposes {have} HIP device(s): refusing to run fewer ranks "
                             "than asked (the path has no CPU fallback)")
    argv = list(sys.argv if argv is None else argv)
    port = _free_port()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DCVIC_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable] + argv, env=env))
    rc = 0
    try:
        alive = list(procs)
        while alive:
            for p_ in list(alive):
                c = p_.poll()
                if c is None:
                    continue
                alive.remove(p_)
                if c != 0 and rc == 0:
                    rc = c if c > 0 else 128 - c
                    for q in alive:                     # one rank failed: the others would wait in a collective forever
                        q.terminate()
            if alive:
                time.sleep(0.05)
    except BaseException:
        for q in procs:
            if q.poll() is None:
                q.kill()
        raise
    return rc
