#!/usr/bin/env python3
"""bench.py -- sectors/s and achieved HBM GB/s of the fused per-sector chain on MI355X.

A "step" is one pass of the hot path over one elevation sweep of synthetic sectors
(BASELINE.json configs[2]'s sweep size, 360 sectors of the in/00iq.altb shape C=2, m=1024,
n=512, fp32 complex) that is ALREADY RESIDENT IN HBM when the timed region starts.  Every
sector of the sweep has its own 8 MiB of device memory (2.95 GiB per sweep, >> the 256 MiB
Infinity Cache), so re-reads cannot be served on-die.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--sectors S] [--no-cpu-baseline]

N > 1 is launched by the driver with torch.distributed.run, one rank per GPU; sectors are
sharded by rank with NO data-path collective (weak scaling: every GPU owns a full sweep);
torch.distributed only provides the barrier and the MAX over ranks of the elapsed time.

Prints ONE JSON line on rank 0 (see DESIGN.md §6 for the field definitions).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(m, n, budget_s=12.0):
    """Time the oracle's fp32 port (OpenMP) on this host: a bounded sample of the same workload."""
    import numpy as np
    from oracle import oracle as O
    cores = min(os.cpu_count() or 1, 16)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    O.build()
    iq = O.synthetic_sector(0, m, n)
    coef = O.hamming_coef(m, n, np.float32)
    O.sector(iq[0], iq[1], dtype=np.float32, coef=coef)            # warm-up
    t0 = time.perf_counter()
    O.sector(iq[0], iq[1], dtype=np.float32, coef=coef)
    one = time.perf_counter() - t0
    cnt = max(3, min(400, int(budget_s / max(one, 1e-4))))
    t0 = time.perf_counter()
    for _ in range(cnt):
        O.sector(iq[0], iq[1], dtype=np.float32, coef=coef)
    dt = time.perf_counter() - t0
    return {"value": round(cnt / dt, 2), "unit": "sectors/s", "cores": cores, "kind": "port",
            "sample": f"{cnt} sectors of the same shape (oracle/radar_oracle.c fp32, OpenMP {cores} threads, "
                      f"{dt:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # a step is 1.25 ms: the defaults keep the GPU busy for ~0.3 s so that its clocks have settled
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--sectors", type=int, default=360, help="sectors per step per GPU (one elevation sweep)")
    ap.add_argument("--max-batch", type=int, default=int(os.environ.get("WRP_MAX_BATCH", "0")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch   # first: libwrp.so then binds to the HIP runtime torch already loaded
    import wrp_amd
    from oracle import oracle as O

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    ngpu = torch.cuda.device_count()
    dev_index = local_rank % max(ngpu, 1)     # one rank per GPU; wraps only when rehearsed on a smaller box
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    ctl_dev = dev                              # device of the control-plane tensors (barrier / MAX)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # control plane only (barrier + MAX of the elapsed time): RCCL when every rank has its own
        # GPU, gloo otherwise (RCCL refuses two ranks on one device)
        if ngpu >= world:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            dist.init_process_group("gloo", rank=rank, world_size=world)
            ctl_dev = torch.device("cpu")

    m, n, C = 1024, 512, 2
    S = args.sectors
    cfg = dict(n_slots=1, n_sectors=1, n_elevations=1)
    if args.max_batch > 0:
        cfg["max_batch"] = args.max_batch
    if os.environ.get("WRP_FLAGS"):          # A/B measurements only (e.g. 0x100 = fused launch)
        cfg["flags"] = int(os.environ["WRP_FLAGS"], 0)
    eng = wrp_amd.Engine(device=dev_index, **cfg)

    # synthetic sweep: a pool of 8 distinct sectors (SURVEY §8d generator), replicated on the
    # device into S distinct 8 MiB blocks; sector index = rank*S + k so ranks see different data
    pool = np.stack([O.synthetic_sector((rank * S + k) % 4096, m, n, C) for k in range(8)])
    d_pool = torch.from_numpy(pool.view(np.float32).reshape(8, -1)).to(dev)
    d_iq = d_pool[torch.arange(S, device=dev) % 8].contiguous()       # [S][C*m*n*2] fp32
    d_out = torch.empty((S, m // 2, 2), dtype=torch.float32, device=dev)
    del d_pool
    torch.cuda.synchronize()

    def step():
        eng.process_batch_device(d_iq.data_ptr(), S, d_out.data_ptr())

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # correctness spot check on what was just computed (first sector of the sweep)
    got = d_out[0].cpu().numpy()
    want = O.sector(pool[0][0], pool[0][1], dtype=np.float64)
    ok = bool(np.isneginf(got[0, 0]) and np.max(np.abs(got[1:] - want[1:])) < 1e-3)

    # roofline of the fused chain: HIP events on the engine's own stream around the same launches
    # (the GPU has idled during the host-side spot check: bring the clocks back up first, untimed)
    iters = max(3, min(args.steps, 20))
    for _ in range(max(args.warmup, 10)):
        step()
    ms_total, ms_range, ms_dopp = eng.time_batch_device(d_iq.data_ptr(), S, d_out.data_ptr(), iters, per_kernel=True)
    algo = eng.algorithmic_bytes
    t_sector = ms_total * 1e-3 / (iters * S)
    achieved = algo / t_sector / 1e9
    max_batch = eng.lib.wrp_get_config  # noqa: F841  (config is echoed below)
    c2 = wrp_amd.WrpConfig()
    eng.lib.wrp_get_config(eng.handle, c2)
    launches = -(-S // c2.max_batch)
    # HBM traffic per launch pair from the committed rocprofv3 PMC run of this same configuration
    # (FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 correction applied; profiles/r01/traffic.json)
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r01", "traffic.json")))
        if tj["sectors_per_launch"] == c2.max_batch and c2.flags == 0:
            traffic = round(tj["bytes_per_launch_pair"])
    except Exception:
        pass
    roofline = {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
        "algorithmic_bytes_per_launch_pair": algo * min(S, c2.max_batch),
        "kernel": "range_pass_1024 + doppler_pass_512 (one launch pair per chunk)",
        "algorithmic_bytes_per_sector": algo, "sectors_per_launch": c2.max_batch,
        "avg_launch_pair_us": round(ms_total * 1e3 / (iters * launches), 2),
        "range_pass_us_per_sector": round(ms_range * 1e3 / (iters * S), 3),
        "doppler_pass_us_per_sector": round(ms_dopp * 1e3 / (iters * S), 3),
    }

    total_sectors = world * S * args.steps
    line = {
        "metric": "sectors/sec + achieved HBM GB/s on in/00iq.altb shape, 1/2/4/8 GPUs",
        "value": round(total_sectors / elapsed, 1), "unit": "sectors/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed * 1e3 / args.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"00iq.altb shape (C=2, m=1024, n=512, fp32 complex), {S}-sector elevation "
                               f"sweep per GPU per step, device-resident", "sectors_per_step_per_gpu": S,
                   "parallelism": f"sector-sharded x{world}, no collective"},
        "achieved_hbm_GBps": round(world * achieved, 1),
        "spot_check_vs_oracle": ok,
        "roofline": roofline,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(m, n)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
