"""Worker for tests/test_sharding_cpu.py::test_bench_control_plane: bench.py's barrier + MAX over the launcher's TCP store."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ctl = bench.RankControl(rank, world)
    ctl.barrier()
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))                      # the last rank is the slow one
    elapsed = ctl.max(time.perf_counter() - t0)
    ctl.barrier()
    if rank == 0:
        print('{"elapsed": %.4f, "world": %d}' % (elapsed, world), flush=True)      # the ONE line on stdout


if __name__ == "__main__":
    main()
