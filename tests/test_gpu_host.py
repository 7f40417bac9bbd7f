"""GPU: the rpv2 binary (C++ RadarProcessor over libwrp.so, wire-format ingest, GPU decode,
framed egress) end to end against the oracle."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RPV2 = os.path.join(ROOT, "weather-radar-processing_amd", "host", "rpv2")


def wire_bytes(iq):
    """[2][m][n] complex (integer valued) -> 12 bytes/sample hhI hhQ vvI vvQ vhI vhQ big-endian int16."""
    m, n = iq.shape[1:]
    s = np.zeros((m * n, 6), dtype=">i2")
    s[:, 0] = iq[0].real.ravel(); s[:, 1] = iq[0].imag.ravel()
    s[:, 2] = iq[1].real.ravel(); s[:, 3] = iq[1].imag.ravel()
    s[:, 4] = 77; s[:, 5] = -77          # VH: on the wire, feeds no product
    return s.tobytes()


@pytest.mark.parametrize("streams,wire8", [(1, False), (3, False), (2, True)])
def test_rpv2_file_replay(tmp_path, oracle, streams, wire8):
    """wire8: the file holds the reference's 12-byte samples either way; --wire8 makes the feeder drop VH on the way into the
    pinned slot (WRP_FLAG_WIRE_8) -- same frames."""
    assert os.path.exists(RPV2), "run `make host`"
    K = 5
    sectors = [oracle.synthetic_sector(s) for s in range(K)]
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        for iq in sectors:
            f.write(wire_bytes(iq))
    r = subprocess.run([RPV2, str(streams), "--in", f"file:{fin}", "--out", f"file:{fout}", "--sectors", str(K)] + (["--wire8"] if wire8 else []),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert f"{K} sectors processed" in r.stderr
    raw = np.fromfile(fout, dtype=np.uint8)
    frame = 4 + 4 * 512
    assert raw.size == K * 2 * frame
    for s in range(K):
        want = oracle.sector(sectors[s][0], sectors[s][1], dtype=np.float64)
        for which in (0, 1):
            fr = raw[(2 * s + which) * frame:(2 * s + which + 1) * frame]
            assert list(fr[:4]) == [0, s, 0, 0]                      # sector BE16, elevation BE16 (rpv2.cu:635-642)
            vals = oracle.abtoaf(fr[4:])
            if which == 0:
                assert np.isneginf(vals[0])
                assert np.max(np.abs(vals[1:] - want[1:, 0]) / np.abs(want[1:, 0])) < 1e-5
            else:
                assert np.max(np.abs(vals - want[:, 1])) < 2e-5


def parse_frames(raw, header):
    """frames of `header` + 512 big-endian floats -> {(which-th frame of (sector, elev)): values}"""
    frame = header + 4 * 512
    assert raw.size % frame == 0
    out = {}
    for k in range(raw.size // frame):
        fr = raw[k * frame:(k + 1) * frame]
        sector = (int(fr[0]) << 8) | int(fr[1])
        elev = ((int(fr[2]) << 8) | int(fr[3])) if header == 4 else 0
        out.setdefault((elev, sector), []).append(fr[header:])
    return out


def test_rpv2_two_gpu_threads_share_one_source(tmp_path, oracle):
    """--devices 0,0: two host threads, two engine handles (here on the one GPU there is), sector s of every
    elevation to thread s mod 2, one source read in acquisition order through the turnstile.  Every sector of
    the scan comes out exactly once, with the values of the single-thread run (configs[3]'s host side)."""
    assert os.path.exists(RPV2), "run `make host`"
    K, SECT, ELEV = 11, 4, 3                      # 11 sectors of a 4 x 3 scan: the last elevation stays incomplete
    sectors = [oracle.synthetic_sector(s) for s in range(3)]
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        for k in range(K):
            f.write(wire_bytes(sectors[k % 3]))
    r = subprocess.run([RPV2, "3", "--devices", "0,0", "--scan", f"{SECT},{ELEV}", "--in", f"file:{fin}", "--out", f"file:{fout}",
                        "--sectors", str(K)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert f"{K} sectors processed" in r.stderr and "2 GPU threads" in r.stderr
    assert "shard 0 of 2): 6 sectors" in r.stderr and "shard 1 of 2): 5 sectors" in r.stderr
    got = parse_frames(np.fromfile(fout, dtype=np.uint8), 4)
    want_keys = {(k // SECT, k % SECT) for k in range(K)}
    assert set(got) == want_keys
    ref = [oracle.sector(s[0], s[1], dtype=np.float64) for s in sectors]
    for k in range(K):
        fr = got[(k // SECT, k % SECT)]
        assert len(fr) == 2                                          # Zdb frame, then Zdr frame
        zdb, zdr = oracle.abtoaf(fr[0]), oracle.abtoaf(fr[1])
        w = ref[k % 3]
        assert np.isneginf(zdb[0]) and np.max(np.abs(zdb[1:] - w[1:, 0]) / np.abs(w[1:, 0])) < 1e-5
        assert np.max(np.abs(zdr - w[:, 1])) < 2e-5


@pytest.mark.parametrize("wire8", [False, True])
def test_rpv2_udp_ingest_and_egress_on_loopback(oracle, wire8):
    """N3, the UDP half: m datagrams of 12 n bytes per sector to port IN (read_single.cc:145-148), products back as
    one datagram per product with the 2-byte sector header (read_single.cc:510-520), here unicast to 127.0.0.1.
    wire8: the same datagrams; every row goes from the socket's buffer into the pinned slot without its VH samples
    (RadarProcessor::set_comms with WRP_FLAG_WIRE_8)."""
    import socket
    import threading
    import time
    assert os.path.exists(RPV2), "run `make host`"
    IN, ZDB, ZDR, K = (19421, 19422, 19423, 3) if wire8 else (19411, 19412, 19413, 3)
    sectors = [oracle.synthetic_sector(s) for s in range(K)]
    rx = []
    for port in (ZDB, ZDR):
        s = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        s.bind(("127.0.0.1", port))
        s.settimeout(60)
        rx.append(s)
    proc = subprocess.Popen([RPV2, "2", "--in", f"udp:{IN}", "--out", f"udp:{ZDB},{ZDR}@127.0.0.1", "--sectors", str(K)] + (["--wire8"] if wire8 else []),
                            stderr=subprocess.PIPE, text=True)
    got = {0: [], 1: []}

    def receive(which):
        for _ in range(K):
            got[which].append(np.frombuffer(rx[which].recvfrom(65536)[0], np.uint8))
    threads = [threading.Thread(target=receive, args=(w,)) for w in (0, 1)]
    for t in threads:
        t.start()
    tx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    line = proc.stderr.readline()                  # the engine (and its socket) come up first
    assert "ready" in line, line
    row = 12 * 512
    for iq in sectors:
        b = wire_bytes(iq)
        for i in range(1024):
            tx.sendto(b[i * row:(i + 1) * row], ("127.0.0.1", IN))
            if i % 8 == 7:
                time.sleep(0.001)                  # loop-back has no flow control: do not overrun the receive buffer
    for t in threads:
        t.join(90)
    err = proc.communicate(timeout=60)[1]
    assert proc.returncode == 0, err
    assert all(len(got[w]) == K for w in (0, 1)), (len(got[0]), len(got[1]), err)
    for k in range(K):
        want = oracle.sector(sectors[k][0], sectors[k][1], dtype=np.float64)
        for which in (0, 1):
            fr = got[which][k]
            assert fr.size == 2 + 4 * 512 and list(fr[:2]) == [0, k]
            vals = oracle.abtoaf(fr[2:])
            if which == 0:
                assert np.isneginf(vals[0]) and np.max(np.abs(vals[1:] - want[1:, 0]) / np.abs(want[1:, 0])) < 1e-5
            else:
                assert np.max(np.abs(vals - want[:, 1])) < 2e-5


READ_ALTB = os.path.join(ROOT, "weather-radar-processing_amd", "host", "read_altb")
M = 1024


@pytest.mark.parametrize("with_ids", [False, True])
def test_read_altb_is_read_cc_with_the_gpu_behind_it(tmp_path, oracle, with_ids):
    """read.cc's own interface: text sectors on stdin (m*n pairs "I Q" of HH, then of VV: read.cc:106-123; the
    multi-sector variant carries a sector id in front of each, gpu_1fp_streamreordered.cu:290-302), one line
    "zdb zdr" per gate on stdout (read.cc:344, ostream default formatting = 6 significant digits).  Three synthetic
    sectors (so that both pinned slots are reused) against the fp64 oracle."""
    assert os.path.exists(READ_ALTB), "run `make host`"
    K = 3
    secs = [oracle.synthetic_sector(10 + k) for k in range(K)]          # [2][m][n] complex64, int16-valued
    parts = []
    for k, s in enumerate(secs):
        flat = s.view(np.float32).reshape(-1).astype(np.int64)           # HH (re, im) row-major, then VV: the text order
        if with_ids:
            parts.append(str(100 + k))
        parts.append(" ".join(map(str, flat.tolist())))
    text = "\n".join(parts) + "\n"
    r = subprocess.run([READ_ALTB], input=text, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = np.array([[float(x) for x in line.split()] for line in r.stdout.strip().split("\n")], dtype=np.float64)
    assert got.shape == (K * M // 2, 2)
    for k, s in enumerate(secs):
        want = oracle.sector(s[0], s[1], dtype=np.float64)
        g = got[k * (M // 2):(k + 1) * (M // 2)]
        assert np.isneginf(g[0, 0])
        # 6 printed digits of values up to ~100 dB: 5e-4 dB of print quantisation on top of the usual tolerances
        assert np.max(np.abs(g[1:, 0] - want[1:, 0])) < 1e-3
        assert np.max(np.abs(g[1:, 1] - want[1:, 1])) < 1e-4
    # a truncated file is refused, not misread
    r = subprocess.run([READ_ALTB], input="1 2 3\n", capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "expected a multiple" in r.stderr


def test_rpv2_host_fill_source_and_process_entry_point():
    """`--in synthetic:copy:T` (every sector memcpy'd from a pageable buffer into its pinned slot by T threads: the host's
    share of an end-to-end run, bench.py's end_to_end.with_host_fill) with the GPU thread bound to its NUMA node, one and
    two GPU threads; and `process` -- the reference's other entry-point name (Makefile:3 there) -- is the same program."""
    import re
    proc = os.path.join(os.path.dirname(RPV2), "process")
    assert os.path.exists(proc) and os.path.samefile(os.path.realpath(proc), RPV2), "run `make process`"
    for exe, devices, threads, want, extra in ((RPV2, "0", 1, 12, []), (proc, "0,0", 3, 24, []), (RPV2, "0", 4, 12, ["--wire8"])):
        r = subprocess.run([exe, "2", "--devices", devices, "--in", f"synthetic:copy:{threads}", "--bind-numa", "--out", "none",
                            "--scan", "6,2", "--sectors", "12"] + extra, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        m = re.search(r"rpv2: (\d+) sectors processed .*host fill by (\d+) thread", r.stderr)
        assert m and int(m.group(1)) == want and int(m.group(2)) == threads, r.stderr       # --sectors counts per GPU thread here
        assert ("VH dropped" in r.stderr) == bool(extra)
