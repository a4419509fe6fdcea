"""Multi-GPU data parallelism of the codec: images are independent units, so ranks never exchange
data on the data path; the only collective is an all_gather of the small per-image result table
(SURVEY sec.8e).  One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on
ROCm, "gloo" on CPU for tests).  The reference is single-process (README.md:64-65) -- this is new.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch


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
