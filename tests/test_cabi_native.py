"""The C ABI from a plain C program (no Python, no torch in the client): compiled with gcc against include/v2v_hip.h,
linked to libv2v_hip.so + the HIP runtime, checked against the oracle through checksums of the output bytes."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cabi", "cabi_smoke.c")


def _fnv1a(b: bytes) -> int:
    h = 1469598103934665603
    for byte in b:
        h ^= byte
        h = (h * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def _build(tmp_path):
    exe = str(tmp_path / "cabi_smoke")
    cmd = ["gcc", "-O1", "-std=c11", SRC, "-I", os.path.join(ROOT, "include"), "-L", os.path.join(ROOT, "v2v_amd"), "-lv2v_hip",
           "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + os.path.join(ROOT, "v2v_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-3000:]
    return exe


def test_c_client_compiles_and_links(tmp_path):
    import __graft_entry__ as ge
    ge.build()
    _build(tmp_path)                                           # header is valid C11 and every symbol used resolves


@pytest.mark.gpu
def test_c_client_results_match_oracle(tmp_path, oracle_c, luts):
    exe = _build(tmp_path)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert res.returncode == 0, res.stdout + res.stderr
    lines = dict(l.split(" ", 1) for l in res.stdout.strip().splitlines())
    clip = np.fromfile(str(tmp_path / "cabi_clip.bin"), dtype=np.uint8).reshape(2, 11, 24, 32)
    params = np.array([[0.2, 0.2, 0, 0, 0], [0.15, 0.35, 0, 0, 0]])
    s, totals = oracle_c.esim_voxel(clip, params, luts, rng_mode=oracle_c.RNG_NONE, bin_mode=oracle_c.BIN_SUM, num_bins=5)
    want_sum = "%016x" % _fnv1a(s.astype(np.float32).tobytes())
    got = lines["sum"].split()
    assert got[0] == want_sum
    assert [int(v) for v in got[2:6]] == totals.reshape(-1).tolist()
    b, _ = oracle_c.esim_voxel(clip, params, luts, rng_mode=oracle_c.RNG_NONE, bin_mode=oracle_c.BIN_BILINEAR, num_bins=5)
    assert lines["bilinear"].strip() == "%016x" % _fnv1a(b.tobytes())
    assert int(lines["bins_error"]) == -3                       # V2V_ERR_BINS
