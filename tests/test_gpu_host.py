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


@pytest.mark.parametrize("streams", [1, 3])
def test_rpv2_file_replay(tmp_path, oracle, streams):
    assert os.path.exists(RPV2), "run `make host`"
    K = 5
    sectors = [oracle.synthetic_sector(s) for s in range(K)]
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        for iq in sectors:
            f.write(wire_bytes(iq))
    r = subprocess.run([RPV2, str(streams), "--in", f"file:{fin}", "--out", f"file:{fout}", "--sectors", str(K)],
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
