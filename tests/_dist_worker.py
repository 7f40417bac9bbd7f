"""Worker for tests/test_sharding_cpu.py (launched with torch.distributed.run, gloo, CPU)."""
import json
import os
import sys
import time

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wrp_amd  # noqa: E402
from wrp_amd import sharding  # noqa: E402


def main():
    out_path = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    E, S, G = 2, 7, 4
    plan = sharding.volume_plan(E, S, rank, world)
    # stand-in for the GPU: a value that identifies (elevation, sector, gate)
    local = {(e, s): (np.arange(G * 2, dtype=np.float32).reshape(G, 2) + 100 * e + 10 * s) for e, s in plan}
    dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))                      # rank 1 is the slow one
    elapsed = sharding.max_over_ranks(dist, time.perf_counter() - t0)
    table = sharding.gather_results(dist, local, E, S, G, rank, world)
    owners = [None] * world
    dist.all_gather_object(owners, plan)
    if rank == 0:
        json.dump({"elapsed": elapsed, "owners": owners, "table": table.tolist(), "world": world}, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
