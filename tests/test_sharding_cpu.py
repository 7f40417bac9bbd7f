"""The N > 1 path on CPU: two gloo ranks shard a volume scan by sector index, no data-path
collective; whole-job time is the MAX over ranks; rank 0 can assemble the result table."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_is_exact():
    import wrp_amd
    from wrp_amd import sharding
    for world in (1, 2, 3, 8):
        seen = []
        for r in range(world):
            mine = sharding.sectors_for_rank(360, r, world)
            assert all(sharding.owner_of(s, world) == r for s in mine)
            seen += mine
        assert sorted(seen) == list(range(360))
        sizes = [len(sharding.sectors_for_rank(360, r, world)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1
    assert sharding.volume_plan(2, 5, 1, 2) == [(0, 1), (0, 3), (1, 1), (1, 3)]
    assert sharding.sectors_for_rank(3, 2, 8) == [2] and sharding.sectors_for_rank(3, 5, 8) == []   # ragged


def test_two_rank_gloo_run(tmp_path):
    out = tmp_path / "r0.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", GLOO_SOCKET_IFNAME="lo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29641",
                        os.path.join(ROOT, "tests", "_dist_worker.py"), str(out)],
                       capture_output=True, text=True, timeout=180, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.load(open(out))
    assert d["world"] == 2
    assert d["elapsed"] >= 0.09                                  # the slow rank's 0.1 s, not rank 0's 0.05 s
    flat = sorted(tuple(x) for part in d["owners"] for x in part)
    assert flat == sorted((e, s) for e in range(2) for s in range(7))
    table = np.array(d["table"], dtype=np.float32)
    assert not np.isnan(table).any()
    for e in range(2):
        for s in range(7):
            assert table[e, s, 0, 0] == 100 * e + 10 * s


def test_bench_control_plane_is_silent_and_takes_the_max():
    """bench.py --gpus N synchronises its ranks through torch.distributed's TCP store (RankControl) -- no process group, so
    nothing but rank 0's JSON line reaches stdout (gloo prints its connections there) -- and reports the slowest rank."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
                        "--master-addr", "127.0.0.1", "--master-port", "29643", os.path.join(ROOT, "tests", "_ctl_worker.py")],
                       capture_output=True, text=True, timeout=180, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [x for x in r.stdout.splitlines() if x.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["world"] == 3 and d["elapsed"] >= 0.14          # rank 2's 0.15 s
