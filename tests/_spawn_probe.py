"""Helper of test_parallel_gloo.py (not a test): a `prog --gpus N` style program that self-launches its ranks through
dc_vic_amd.parallel.self_launch, then -- as a rank -- joins a gloo group, gathers a ragged rate table and (rank 0) prints ONE line."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402

from dc_vic_amd.parallel import gather_rate_table, launched_by_a_launcher, self_launch, shard_indices  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--fail_rank", type=int, default=-1)
    p.add_argument("--need_gpus", action="store_true")
    a = p.parse_args()
    if a.gpus > 1 and not launched_by_a_launcher():
        sys.exit(self_launch(a.gpus, need_gpus=a.need_gpus))
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if rank == a.fail_rank:
        raise SystemExit(7)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist_.init_process_group("gloo", rank=rank, world_size=world)
        dist = dist_
    mine = shard_indices(7, rank, world, [10, 1, 1, 1, 9, 1, 1] if world > 1 else None)
    table = gather_rate_table(np.array([[i, 100.0 + i] for i in mine], dtype=np.float64).reshape(-1, 2), dist)
    if rank == 0:
        print(json.dumps({"n_gpus": world, "rows": sorted(table[:, 0].tolist()), "self_launched": os.environ.get("DCVIC_SELF_LAUNCHED") == "1",
                          "local_world": os.environ.get("LOCAL_WORLD_SIZE")}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
