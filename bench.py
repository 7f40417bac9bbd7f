#!/usr/bin/env python3
"""bench.py -- sectors/s and achieved HBM GB/s of the fused per-sector chain on MI355X.

A "step" is one pass of the hot path over one elevation sweep of synthetic sectors
(BASELINE.json configs[2]'s sweep size, 360 sectors of the in/00iq.altb shape C=2, m=1024,
n=512, fp32 complex) that is ALREADY RESIDENT IN HBM when the timed region starts.  Every
sector of the sweep has its own 8 MiB of device memory (2.95 GiB per sweep, >> the 256 MiB
Infinity Cache), so re-reads cannot be served on-die.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--sectors S] [--no-cpu-baseline] [--no-end-to-end]

N > 1 is launched by the driver with torch.distributed.run, one rank per GPU; sectors are
sharded by rank with NO data-path collective (weak scaling: every GPU owns a full sweep);
the barrier and the MAX over ranks of the elapsed time go through torch.distributed's TCP store
(control plane only: no process group, no RCCL).

Prints ONE JSON line on rank 0 (DESIGN.md 6 defines the fields):
  value        device-resident sectors/s (PCIe not included) -- BASELINE.json's kernel-only figure
  roofline     of the dominant launch, from HIP events on the engine's own stream in this run;
               `traffic` from the rocprofv3 PMC run of the SAME library sources (fingerprint checked)
  wire_format_input / shape_b   the same sweep in the wire format and BASELINE configs[4]'s 2048 x 128
               shape through its own fused launch, each with its own algorithmic bytes and roofline fraction
  end_to_end   sectors/s through the 4-slot cascade: pinned H2D of the wire-format sector + decode +
               chain + D2H per sector, all ranks at once -- BASELINE.json's end-to-end figure
  cpu_baseline the oracle's fp32 port on this host's cores (all usable cores and one thread)
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PROFILE_ROUND = "r05"
SETTLE_S = 0.3     # untimed launches before the W warm-up steps: the clocks reach their working point (see settle())


def host_cpus():
    """Usable logical CPUs, physical cores among them and the model string (/proc/cpuinfo)."""
    usable = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    model, cores, cur = "unknown", set(), {}
    try:
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = [x.strip() for x in line.split(":", 1)]
                cur[k] = v
            elif cur:
                if int(cur.get("processor", -1)) in usable:
                    cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                    model = cur.get("model name", model)
                cur = {}
    except OSError:
        pass
    return len(usable), max(len(cores), 1), model


def cpu_baseline_child(threads, budget_s, m=1024, n=512):
    """Runs in a FRESH process whose OMP_* environment was set before anything was imported: three bounded samples."""
    import numpy as np
    from oracle import oracle as O
    O.build()
    iq = O.synthetic_sector(0, m, n)
    coef = O.hamming_coef(m, n, np.float32)
    O.sector(iq[0], iq[1], dtype=np.float32, coef=coef)            # warm-up (threads created, pages touched)
    t0 = time.perf_counter()
    O.sector(iq[0], iq[1], dtype=np.float32, coef=coef)
    one = time.perf_counter() - t0
    cnt = max(3, min(5000, int(budget_s / 3 / max(one, 1e-4))))
    rates = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(cnt):
            O.sector(iq[0], iq[1], dtype=np.float32, coef=coef)
        rates.append(cnt / (time.perf_counter() - t0))
    print(json.dumps({"sectors": cnt, "rates": rates, "affinity": len(os.sched_getaffinity(0))}))


def cgroup_cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else round(int(q) / int(p), 2)
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / p, 2)
    except Exception:
        return None


def cpu_baseline(m=1024, n=512):
    """the oracle's fp32 port on this host: 1 thread, 16 threads (the CPU share of a one-GPU box) and all
    PHYSICAL cores this process may use; each figure is the median of three bounded samples taken in a fresh process
    with OMP_PLACES=cores OMP_PROC_BIND=spread.  What the process was actually allowed (affinity mask, cgroup CPU quota)
    is part of the record: a quota of 16 CPUs under a 256-CPU mask is what makes 'all cores' slower than 16 threads."""
    import statistics
    logical, physical, model = host_cpus()
    quota = cgroup_cpu_quota()

    def run(t, budget):
        env = dict(os.environ, OMP_NUM_THREADS=str(t), OMP_PLACES="cores", OMP_PROC_BIND="spread" if t > 1 else "close")
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", str(t), str(budget), str(m), str(n)],
                             env=env, capture_output=True, text=True, timeout=300)
        r = json.loads(out.stdout.strip().splitlines()[-1])
        return round(statistics.median(r["rates"]), 2), r
    t16 = min(logical, 16)
    runs = {}
    for t, budget in ((1, 8.0), (t16, 8.0), (physical, 8.0)):
        if t not in runs:
            runs[t] = run(t, budget)
    best_t = max(runs, key=lambda t: runs[t][0])
    return {"value": runs[best_t][0], "unit": "sectors/s", "cores": best_t, "kind": "port",
            "one_thread": runs[1][0], "threads_16": runs[t16][0], "all_physical_cores": runs[physical][0],
            "physical_cores": physical, "logical_cpus_in_affinity_mask": logical, "cgroup_cpu_quota": quota, "cpu_model": model,
            "samples_sectors_per_s": {str(t): [round(x, 2) for x in r["rates"]] for t, (_, r) in runs.items()},
            "sample": "median of 3 samples of %s sectors of the same shape per thread count (1, %d, %d = physical cores in the mask), "
                      "OMP_PLACES=cores OMP_PROC_BIND=spread, each thread count in a fresh process; `value` is the best of them; "
                      "oracle/radar_oracle.c fp32 port" % ("/".join(str(r["sectors"]) for _, r in runs.values()), t16, physical)}


class RankControl:
    """Barrier and MAX over the ranks of one node through torch.distributed's TCP store (the launcher's own store under
    torch.distributed.run, else one that rank 0 hosts on MASTER_ADDR:MASTER_PORT)."""

    def __init__(self, rank, world):
        from datetime import timedelta
        from torch.distributed import TCPStore
        host = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(os.environ.get("MASTER_PORT", "29533"))
        agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "False").lower() == "true"
        self.store = TCPStore(host, port, world, is_master=(rank == 0 and not agent), timeout=timedelta(seconds=900),
                              wait_for_workers=False)
        self.rank, self.world, self.n = rank, world, 0

    def barrier(self):
        self.n += 1
        key = "wrp/bench/%d" % self.n
        if self.store.add(key, 1) == self.world:
            self.store.set(key + "/go", "1")
        self.store.wait([key + "/go"])

    def max(self, x):
        self.n += 1
        key = "wrp/bench/%d" % self.n
        self.store.set("%s/%d" % (key, self.rank), repr(float(x)))
        self.barrier()
        return max(float(self.store.get("%s/%d" % (key, r)).decode()) for r in range(self.world))


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes with torch.distributed.run BEFORE this
    process has touched a GPU, relay rank 0's JSON line and exit with the launcher's code."""
    import socket
    import torch
    have = torch.cuda.device_count()            # counts devices without initialising the GPU
    if have < n and not os.environ.get("WRP_BENCH_OVERSUBSCRIBE"):      # (rehearsal of the N-rank plumbing on a smaller box)
        sys.exit("bench.py: --gpus %d asked for, %d GPU(s) visible" % (n, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    if len(sys.argv) >= 6 and sys.argv[1] == "--cpu-baseline-child":
        return cpu_baseline_child(int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # a step is ~1 ms: the defaults keep the GPU busy for ~0.3 s so that its clocks have settled
    # (with --steps 20 the same binary reads about 5 % lower: DESIGN.md 6)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--sectors", type=int, default=360, help="sectors per step per GPU (one elevation sweep)")
    ap.add_argument("--max-batch", type=int, default=int(os.environ.get("WRP_MAX_BATCH", "0")))
    ap.add_argument("--shape", choices=["A", "B"], default="A",
                    help="A: the 00iq.altb shape 1024 x 512 (BASELINE metric); B: configs[4]'s 2048 range gates x 128 pulses")
    ap.add_argument("--settle", type=float, default=SETTLE_S,
                    help="seconds of untimed launches before the warm-up steps (profiling passes shorten it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the single-sector latency and wire-format sections (profiling passes)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%s" % (args.gpus, os.environ["WORLD_SIZE"]))

    import numpy as np
    import torch   # first: libwrp.so then binds to the HIP runtime torch already loaded
    import wrp_amd
    from oracle import oracle as O

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    ctl = None
    ngpu = torch.cuda.device_count()
    if ngpu < world and not os.environ.get("WRP_BENCH_OVERSUBSCRIBE"):
        sys.exit("bench.py: %d ranks but %d GPU(s) visible (WRP_BENCH_OVERSUBSCRIBE=1 rehearses on a smaller box)" % (world, ngpu))
    dev_index = local_rank % max(ngpu, 1)     # one rank per GPU; wraps only when rehearsed on a smaller box
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    if world > 1:
        # control plane only (barrier + MAX of the elapsed time) over torch.distributed's TCP store: the data path has no
        # collective, RCCL is never initialised (north_star), and no process group is created (gloo announces its
        # connections on STDOUT, in front of the one JSON line this program owes its caller)
        ctl = RankControl(rank, world)

    m, n, C = (1024, 512, 2) if args.shape == "A" else (2048, 128, 2)
    S = args.sectors
    SLOTS = 4
    cfg = dict(n_slots=SLOTS, n_sectors=S, n_elevations=1, m=m, n=n)
    if args.max_batch > 0:
        cfg["max_batch"] = args.max_batch
    if os.environ.get("WRP_FLAGS"):          # A/B measurements only (e.g. 0x800 = two kernels)
        cfg["flags"] = int(os.environ["WRP_FLAGS"], 0)
    eng = wrp_amd.Engine(device=dev_index, **cfg)

    # synthetic sweep: a pool of 8 distinct sectors (SURVEY 8d generator), replicated on the
    # device into S distinct 8 MiB blocks; sector index = rank*S + k so ranks see different data
    pool = np.stack([O.synthetic_sector((rank * S + k) % 4096, m, n, C) for k in range(8)])
    d_pool = torch.from_numpy(pool.view(np.float32).reshape(8, -1)).to(dev)
    d_iq = d_pool[torch.arange(S, device=dev) % 8].contiguous()       # [S][C*m*n*2] fp32
    d_out = torch.empty((S, m // 2, 2), dtype=torch.float32, device=dev)
    del d_pool
    torch.cuda.synchronize()

    def step():
        eng.process_batch_device(d_iq.data_ptr(), S, d_out.data_ptr())

    def profile_json(name):
        """profiles/rNN/<name> when it was measured on the library sources that run here (fingerprint), else (None, why)."""
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", PROFILE_ROUND, name)))
        except Exception:
            return None, "no profiles/%s/%s for this build" % (PROFILE_ROUND, name)
        if tj.get("fingerprint") != wrp_amd.source_fingerprint():
            return None, "profiles/%s/%s was measured on other sources (fingerprint differs)" % (PROFILE_ROUND, name)
        return tj, tj.get("source", "")

    def traffic_of(name, per_launch, fused_launch=True):
        tj, note = profile_json(name)
        if tj is None:
            return None, note
        if tj.get("sectors_per_launch") != per_launch or tj.get("fused") != fused_launch:
            return None, "profiles/%s/%s was measured on another configuration" % (PROFILE_ROUND, name)
        return round(tj["bytes_per_launch"]), note

    def wire_pool(sectors, wb):
        """sectors [K][2][m][n] complex (integer valued) -> [K][m*n*wb] bytes: hhI hhQ vvI vvQ (vhI vhQ), big-endian int16"""
        k, _, mm, nn = sectors.shape
        w = np.zeros((k, mm * nn, wb // 2), dtype=">i2")
        for c in range(2):
            w[:, :, 2 * c] = sectors[:, c].real.reshape(k, -1)
            w[:, :, 2 * c + 1] = sectors[:, c].imag.reshape(k, -1)
        return np.frombuffer(w.tobytes(), np.uint8).reshape(k, -1)

    def raw_sweep(e, sectors, wb, d_ref, traffic_name, what):
        """the sweep in a wire format through e's raw batch entry: untimed settle, wall clock over <= 100 launches"""
        mm, nn = sectors.shape[2:]
        d_wp = torch.from_numpy(wire_pool(sectors, wb)).to(dev)
        d_raw = d_wp[torch.arange(S, device=dev) % len(sectors)].contiguous()      # [S][m*n*wb] bytes, distinct blocks
        d_o = torch.empty_like(d_ref)
        del d_wp
        t_end = time.perf_counter() + max(args.settle, 0.05)     # the GPU has idled through the host's conversion above
        while time.perf_counter() < t_end:
            for _ in range(8):
                e.process_batch_raw_device(d_raw.data_ptr(), S, d_o.data_ptr())
            e.check()
        steps = max(10, min(args.steps, 100))
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            e.process_batch_raw_device(d_raw.data_ptr(), S, d_o.data_ptr())
        e.check()
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        algo_w = mm * nn * wb + (mm // 2) * 8
        tr, note = traffic_of(traffic_name, S)
        r = {"value": round(world * S * steps / dt, 1), "unit": "sectors/s", "steps": steps, "bytes_per_sample": wb,
             "algorithmic_bytes_per_sector": algo_w, "achieved": round(algo_w * S * steps / dt / 1e9, 1), "peak": HBM_PEAK_GBS,
             "frac": round(algo_w * S * steps / dt / 1e9 / HBM_PEAK_GBS, 4), "traffic": tr, "traffic_note": note,
             "bit_identical_to_planar": bool(torch.equal(d_o, d_ref)), "fused_fallbacks": e.fused_fallbacks, "what": what}
        del d_raw, d_o
        return r

    def barrier():
        torch.cuda.synchronize()
        if ctl is not None:
            ctl.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        return x if ctl is None else ctl.max(x)

    def settle(seconds=None):
        seconds = args.settle if seconds is None else seconds
        # The GPU has idled while the host prepared the input (or checked results): its clocks are down and the first
        # launches run slower.  Untimed launches for a fixed wall time bring it to its working point, whatever W is
        # (with --steps 20 --warmup 5 the same binary otherwise reads 4-5 % lower than with the defaults).
        t = time.perf_counter()
        while time.perf_counter() - t < seconds:
            for _ in range(8):
                step()
            torch.cuda.synchronize()

    settle()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    eng.check()                               # a fused launch that gave up would be reported here

    # correctness spot check on what was just computed (first and last sector of the sweep)
    ok = True
    for k in (0, S - 1):
        got = d_out[k].cpu().numpy()
        want = O.sector(pool[k % 8][0], pool[k % 8][1], dtype=np.float64)
        ok = ok and bool(np.isneginf(got[0, 0]) and np.max(np.abs(got[1:] - want[1:])) < 1e-3)

    # roofline of the dominant launch: HIP events on the engine's own stream around the same launches
    # (the GPU has idled during the host-side spot check: bring the clocks back up first, untimed)
    iters = max(3, min(args.steps, 20))
    settle()
    for _ in range(max(args.warmup, 10)):
        step()
    ms_total, ms_range, ms_dopp = eng.time_batch_device(d_iq.data_ptr(), S, d_out.data_ptr(), iters, per_kernel=True)
    algo = eng.algorithmic_bytes
    t_sector = ms_total * 1e-3 / (iters * S)
    achieved = algo / t_sector / 1e9
    c2 = wrp_amd.WrpConfig()
    eng.lib.wrp_get_config(eng.handle, c2)
    fused = (c2.flags & (wrp_amd.FLAG_TWO_KERNELS | wrp_amd.FLAG_GENERIC_KERNELS)) == 0 and S >= wrp_amd.FUSED_MIN_SECTORS
    per_launch = S if fused else min(S, c2.max_batch)
    launches = 1 if fused else -(-S // c2.max_batch)
    kernel = (("fused_chain_1024x512" if args.shape == "A" else "fused_chain_2048x128") +
              " (one persistent launch per sweep: tile + row workgroups, intermediate in the XCDs' L2)"
              if fused else "range_pass_1024_persistent + doppler_pass_512 (one launch pair per chunk)" if args.shape == "A"
              else "range_pass_2048 + doppler_pass_128 (one launch pair per chunk)")
    # HBM traffic per launch from the rocprofv3 PMC run of the SAME library sources (FETCH_SIZE / WRITE_SIZE in
    # separate passes, gfx950 correction applied; tools/profile_pmc.sh -> tools/make_traffic.py).  The file
    # records a fingerprint of csrc/, include/ and the compiler flags: anything else reads null.
    traffic, traffic_note = traffic_of("traffic.json" if args.shape == "A" else "traffic_%s.json" % args.shape, per_launch, fused)
    # what the launch keeps busy besides HBM: from the PMC passes of the same sources (tools/make_busy.py) --
    # valu_busy = SQ_INSTS_VALU x 2 cycles / (SIMDs x kernel cycles), lds_busy = SQ_LDS_IDX_ACTIVE / (CUs x kernel cycles) --
    # and the launch with every request of its input dropped by the descriptor (a timing build: tools/make_floor.sh)
    busy, _ = profile_json("busy.json" if args.shape == "A" else "busy_%s.json" % args.shape)
    floor, _ = profile_json("floor.json" if args.shape == "A" else "floor_%s.json" % args.shape)
    roofline = {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_note": traffic_note,
        "kernel": kernel, "algorithmic_bytes_per_sector": algo, "sectors_per_launch": per_launch,
        "algorithmic_bytes_per_launch": algo * per_launch,
        "avg_launch_us": round(ms_total * 1e3 / (iters * launches), 2),
        "valu_busy": busy and busy.get("valu_busy"), "lds_busy": busy and busy.get("lds_busy"),
        "no_input_us_per_sector": floor and floor.get("no_input_us_per_sector"),
        "l2_hit_input_us_per_sector": floor and floor.get("l2_hit_input_us_per_sector"),
    }
    if not fused:
        roofline["range_pass_us_per_sector"] = round(ms_range * 1e3 / (iters * S), 3)
        roofline["doppler_pass_us_per_sector"] = round(ms_dopp * 1e3 / (iters * S), 3)

    # BASELINE configs[1]: ONE sector, one stream -- pinned slot -> wrp_submit -> wrp_wait (H2D of the 8 MiB fp32 block, the
    # range pass and the Doppler pass on 1 sector, D2H of 4 KiB); and the same sector already on the device (kernels only)
    import statistics
    eng.slot_array(0)[:] = pool[0]
    lat, lat_dev = [], []
    for k in range(0 if args.no_extras else 210):
        t0 = time.perf_counter()
        eng.submit(0, 0, 0)
        eng.wait(0)
        lat.append((time.perf_counter() - t0) * 1e6)
    for k in range(0 if args.no_extras else 210):
        t0 = time.perf_counter()
        eng.process_batch_device(d_iq.data_ptr(), 1, d_out.data_ptr())
        eng.check()
        lat_dev.append((time.perf_counter() - t0) * 1e6)
    single = None if args.no_extras else {"submit_wait_pinned_us": round(statistics.median(lat[10:]), 1), "device_resident_us": round(statistics.median(lat_dev[10:]), 1),
              "samples": 200, "what": "median wall time of one wrp_submit -> wrp_wait from a pinned slot (PCIe included) and of one "
                                      "wrp_process_batch_device(1 sector) -> wrp_check on device-resident input; two-kernel path"}

    # The same sweep in the WIRE format, device-resident: the tile workgroups of the fused launch read the
    # 12-byte samples themselves.  A roofline entry of its own -- the headline stays on the fp32 definition of SURVEY 8d.
    wire = None
    if not args.no_extras or os.environ.get("WRP_BENCH_WIRE"):
        want_wb = int(os.environ.get("WRP_BENCH_WIRE", "0") or 0)       # profiling passes: 12 or 8 only
        tag = "" if args.shape == "A" else args.shape
        mib = m * n / 2**20
        if want_wb in (0, 1, 12):
            wire = raw_sweep(eng, pool, 12, d_out, "traffic_%sW.json" % tag,
                             "wrp_process_batch_raw_device: 12 B/sample big-endian int16 read by the tile workgroups (%g MiB/sector), wall clock" % (12 * mib))
        if want_wb in (0, 8):
            with wrp_amd.Engine(device=dev_index, n_slots=1, n_sectors=1, n_elevations=1, m=m, n=n, flags=wrp_amd.FLAG_WIRE_8) as e8:
                w8 = raw_sweep(e8, pool, 8, d_out, "traffic_%sW8.json" % tag,
                               "WRP_FLAG_WIRE_8: 8 B/sample (hh, vv; VH dropped by the feeder) read by the tile workgroups (%g MiB/sector), wall clock" % (8 * mib))
            if wire is None:
                wire = w8
            else:
                wire["wire8"] = w8

    # BASELINE configs[4] (2048 range gates x 128 pulses) through its own fused launch, on this GPU, in the default line: a
    # second engine, its own sweep (360 x 4 MiB, distinct blocks), wall clock + HIP events + spot check against the oracle.
    shape_b = None
    if args.shape == "A" and not args.no_extras:
        mb, nb = 2048, 128
        with wrp_amd.Engine(device=dev_index, n_slots=1, n_sectors=1, n_elevations=1, m=mb, n=nb) as eb:
            pool_b = np.stack([O.synthetic_sector((rank * S + k) % 4096, mb, nb, C) for k in range(8)])
            d_pb = torch.from_numpy(pool_b.view(np.float32).reshape(8, -1)).to(dev)
            d_iq_b = d_pb[torch.arange(S, device=dev) % 8].contiguous()
            d_out_b = torch.empty((S, mb // 2, 2), dtype=torch.float32, device=dev)
            del d_pb
            t_end = time.perf_counter() + args.settle
            while time.perf_counter() < t_end:
                for _ in range(8):
                    eb.process_batch_device(d_iq_b.data_ptr(), S, d_out_b.data_ptr())
                eb.check()
            bsteps = max(10, min(args.steps, 100))
            barrier()
            t0 = time.perf_counter()
            for _ in range(bsteps):
                eb.process_batch_device(d_iq_b.data_ptr(), S, d_out_b.data_ptr())
            eb.check()
            barrier()
            bdt = max_over_ranks(time.perf_counter() - t0)
            b_ok = True
            for k in (0, S - 1):
                got = d_out_b[k].cpu().numpy()
                want = O.sector(pool_b[k % 8][0], pool_b[k % 8][1], dtype=np.float64)
                b_ok = b_ok and bool(np.isneginf(got[0, 0]) and np.max(np.abs(got[1:] - want[1:])) < 1e-3)
            t_end = time.perf_counter() + args.settle      # the GPU has idled during the spot check: clocks back up, untimed
            while time.perf_counter() < t_end:
                for _ in range(8):
                    eb.process_batch_device(d_iq_b.data_ptr(), S, d_out_b.data_ptr())
                eb.check()
            b_iters = max(3, min(args.steps, 20))
            b_ms = eb.time_batch_device(d_iq_b.data_ptr(), S, d_out_b.data_ptr(), b_iters)[0]
            balgo = eb.algorithmic_bytes
            b_ach = balgo * S * b_iters / (b_ms * 1e-3) / 1e9
            b_traffic, b_note = traffic_of("traffic_B.json", S)
            shape_b = {"value": round(world * S * bsteps / bdt, 1), "unit": "sectors/s", "steps": bsteps,
                       "workload": f"BASELINE configs[4]: C=2, m=2048 range gates, n=128 pulses, fp32 complex, {S} sectors per launch, device-resident",
                       "kernel": "fused_chain_2048x128", "fused_fallbacks": eb.fused_fallbacks,
                       "algorithmic_bytes_per_sector": balgo, "algorithmic_bytes_per_launch": balgo * S,
                       "avg_launch_us": round(b_ms * 1e3 / b_iters, 2), "achieved": round(b_ach, 1), "peak": HBM_PEAK_GBS,
                       "frac": round(b_ach / HBM_PEAK_GBS, 4), "traffic": b_traffic, "traffic_note": b_note,
                       "frac_wall_clock": round(balgo * S * bsteps / bdt / 1e9 / HBM_PEAK_GBS, 4), "spot_check_vs_oracle": b_ok,
                       "what": "`value`: wall clock over `steps` launches; `achieved` / `frac`: HIP events on the engine's stream over "
                               "%d launches, as the headline's roofline" % b_iters}
            # ... and the same sweep in the wire formats, read by that launch's tile workgroups (3 / 2 MiB per sector instead of 4)
            shape_b["wire_format_input"] = raw_sweep(eb, pool_b, 12, d_out_b, "traffic_BW.json",
                "wrp_process_batch_raw_device on the 2048 x 128 engine: 12 B/sample read by the tile workgroups (3 MiB/sector), wall clock")
            with wrp_amd.Engine(device=dev_index, n_slots=1, n_sectors=1, n_elevations=1, m=mb, n=nb, flags=wrp_amd.FLAG_WIRE_8) as eb8:
                shape_b["wire_format_input"]["wire8"] = raw_sweep(eb8, pool_b, 8, d_out_b, "traffic_BW8.json",
                    "WRP_FLAG_WIRE_8 on the 2048 x 128 engine: 8 B/sample (2 MiB/sector: a member's bytes are the planar form's, both channels' members read the same), wall clock")
            del d_iq_b, d_out_b

    # end to end: every sector crosses PCIe.  Wire-format sector (12 B/sample, big-endian int16, 6 MiB) in
    # the slot's pinned buffer -> H2D -> decode -> chain -> D2H of 4 KiB, 4 slots cascading, all ranks at once.
    end_to_end = None
    if not args.no_end_to_end:
        want = O.sector(pool[0][0], pool[0][1], dtype=np.float64)
        rpv2 = os.path.join(ROOT, "weather-radar-processing_amd", "host", "rpv2")

        def cascade(e, wb):
            """the sweep through e's 4-slot cascade from pinned wire-format slots; the second repetition is the one reported"""
            raw = wire_pool(pool[:1], wb)[0]
            for s_ in range(SLOTS):
                e.raw_slot_array(s_)[:] = raw
            dt = 0.0
            for rep in range(2):
                barrier()
                t0 = time.perf_counter()
                for k in range(S):
                    s_ = k % SLOTS
                    if k >= SLOTS:
                        e.wait(s_)
                    e.submit_raw(s_, k, 0)
                for s_ in range(min(SLOTS, S)):
                    e.wait(s_)
                barrier()
                dt = max_over_ranks(time.perf_counter() - t0)
            ok_ = bool(np.max(np.abs(e.result(S - 1, 0)[1:] - want[1:])) < 1e-3)
            return {"value": round(world * S / dt, 1), "unit": "sectors/s", "bytes_per_sample": wb,
                    "h2d_GBps_per_gpu": round(S * m * n * wb / dt / 1e9, 1), "spot_check_vs_oracle": ok_}

        def host_fill(wb):
            """the same WITH the host's share: the C++ feeder (host/rpv2, one thread per GPU as rpv2.cu:665-683) brings every
            sector from a pageable buffer of 12-byte samples into its pinned slot before it submits it -- a memcpy, or (wire8)
            the copy that drops VH -- bound to the GPU's NUMA node, with 1 and with 4 threads sharing the copy"""
            import re
            if args.shape != "A" or not os.path.exists(rpv2):
                return None
            r = {"unit": "sectors/s", "what": "steady state: the rate past the run's first 128 sectors (first use of kernels and pinned buffers)"}
            for T in (1, 2, 4):
                barrier()
                out = subprocess.run([rpv2, str(SLOTS), "--device", str(dev_index), "--in", "synthetic:copy:%d" % T, "--bind-numa",
                                      "--out", "none", "--scan", "%d,1" % S, "--sectors", str(3 * S)] + (["--wire8"] if wb == 8 else []),
                                     capture_output=True, text=True, timeout=300)
                mm_ = re.search(r"steady state .*: ([0-9.]+) sectors/s", out.stderr) or re.search(r"\(([0-9.]+) sectors/s end to end", out.stderr)
                rate = float(mm_.group(1)) if mm_ else 0.0
                slowest = max_over_ranks(1.0 / rate if rate > 0 else float("inf"))          # every rank takes part
                r["fill_threads_%d" % T] = round(world / slowest, 1)                        # the slowest rank's rate x ranks
                r["numa_bound"] = "NUMA-bound" in out.stderr
            barrier()
            return r

        # every sector crosses PCIe: wire-format sector in the slot's pinned buffer -> H2D -> decode -> chain -> D2H of 4 KiB,
        # 4 slots cascading, all ranks at once.  `value`: 8 bytes per sample (WRP_FLAG_WIRE_8: the feeder has dropped VH, which
        # no output reads); `wire12`: the reference's 12-byte sample as it arrives
        e12 = cascade(eng, 12)
        with wrp_amd.Engine(device=dev_index, n_slots=SLOTS, n_sectors=S, n_elevations=1, m=m, n=n, flags=wrp_amd.FLAG_WIRE_8) as e8:
            e8r = cascade(e8, 8)
        e12["with_host_fill"] = host_fill(12)
        end_to_end = dict(e8r, slots=SLOTS, sectors_per_gpu=S, with_host_fill=host_fill(8), wire12=e12,
                          ingest=f"WRP_FLAG_WIRE_8: 8 B/sample big-endian int16, hh + vv ({m * n * 8 / 2**20:g} MiB/sector), decoded on the GPU; "
                                 f"wire12: the 12-byte sample with VH ({m * n * 12 / 2**20:g} MiB/sector)",
                          includes="pinned H2D + decode + range/Doppler kernels + D2H per sector; host refill of the pinned slots not "
                                   "included (with_host_fill: rpv2 --in synthetic:copy:T [--wire8] --bind-numa --out none -- pageable -> pinned "
                                   "copy by T host thread(s) + everything else, all ranks at once)")

    # batches on a CALLER's stream (INTEGRATION.md's production form): behind every fused launch the gated two-kernel repeat is
    # queued (include/wrp.h: stream order alone); its cost when the launch has succeeded -- the usual case -- is what this
    # sweep shows against the headline
    caller = None
    if not args.no_extras:
        st = torch.cuda.Stream(device=dev)
        caller = {"unit": "sectors/s", "what": "the headline sweep, and the 12-byte wire-format sweep, on a caller's stream: fused launch + its "
                  "gated repeat (decode / range / Doppler grids that return at once) per batch; wall clock over <= 100 batches"}
        d_raw = torch.from_numpy(wire_pool(pool, 12)).to(dev)[torch.arange(S, device=dev) % 8].contiguous()
        d_o = torch.empty_like(d_out)
        for name, fn, src in (("planar", eng.process_batch_device, d_iq), ("wire12", eng.process_batch_raw_device, d_raw)):
            t_end = time.perf_counter() + max(args.settle, 0.05)     # the GPU has idled through the end-to-end runs: clocks back up, untimed
            while time.perf_counter() < t_end:
                for _ in range(8):
                    fn(src.data_ptr(), S, d_o.data_ptr(), stream=st.cuda_stream)
                st.synchronize()
            steps = max(10, min(args.steps, 100))
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn(src.data_ptr(), S, d_o.data_ptr(), stream=st.cuda_stream)
            st.synchronize()
            eng.check()
            barrier()
            caller[name] = round(world * S * steps / max_over_ranks(time.perf_counter() - t0), 1)
            caller[name + "_bit_identical"] = bool(torch.equal(d_o, d_out))
        caller["fused_fallbacks"] = eng.fused_fallbacks
        del d_raw, d_o

    total_sectors = world * S * args.steps
    line = {
        "metric": "sectors/sec + achieved HBM GB/s on in/00iq.altb shape, 1/2/4/8 GPUs",
        "value": round(total_sectors / elapsed, 1), "unit": "sectors/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed * 1e3 / args.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": (f"00iq.altb shape (C=2, m=1024, n=512, fp32 complex), {S}-sector elevation "
                                f"sweep per GPU per step, device-resident") if args.shape == "A" else
                               (f"shape B = BASELINE configs[4] (C=2, m=2048 range gates, n=128 pulses, fp32 complex), {S} sectors "
                                f"per GPU per step, device-resident"), "sectors_per_step_per_gpu": S,
                   "parallelism": f"sector-sharded x{world}, no collective (barrier + MAX of the elapsed time over a TCP store)",
                   "launch": "fused" if fused else "two kernels", "untimed_settle_s": args.settle,
                   # a7 (read.cc:290-301) in the timed launch, said in so many words
                   "ma_stage": "row sum as the DC bin of the circular moving average: S = (sum of the taps) x (sum |.|^2); the stage "
                               "08pow itself (direct causal circular 7-tap) is formed by the dump instantiations only -- every ma_count "
                               "stage-tested in tests/test_gpu_batches.py; forming it in the timed launch costs +0.97 % (1024 x 512), "
                               "+0.59 % (wire format), +0.27 % (2048 x 128): profiles/r04/ab_rowsum_dc_bin.log"},
        "achieved_hbm_GBps": round(world * achieved, 1),
        "spot_check_vs_oracle": ok,
        "single_sector_latency_us": single["submit_wait_pinned_us"] if single else None,
        "single_sector": single,
        "roofline": roofline,
    }
    if wire is not None:
        line["wire_format_input"] = wire
    if shape_b is not None:
        line["shape_b"] = shape_b
    if end_to_end is not None:
        line["end_to_end"] = end_to_end
    if caller is not None:
        line["caller_stream"] = caller
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(m, n)
    eng.close()
    if ctl is not None:
        ctl.barrier()
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
