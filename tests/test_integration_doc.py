"""The ctypes stubs printed in INTEGRATION.md are executed as they stand against the built library and must return what
the package's own wrappers return (keeps the maintainer-facing binding honest)."""
import ctypes as C  # noqa: F401  (the stubs use it)
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_integration_md_stubs_run_and_match_the_wrappers():
    from v2v_amd import esim, v2e, voxel
    src = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = [b for b in re.findall(r"```python\n(.*?)```", src, re.S) if "C.CDLL" in b or "V2EParams" in b]
    assert len(blocks) == 2
    ns = {}
    exec(blocks[0].replace('C.CDLL("libv2v_hip.so")', 'C.CDLL("%s")' % os.path.join(ROOT, "v2v_amd", "libv2v_hip.so")), ns)
    exec(blocks[1], ns)
    vid = torch.randint(0, 256, (8, 32, 64), dtype=torch.uint8, device="cuda")
    got = ns["video_to_voxel"](vid, 0.2, 0.2, 0.0, 0.0, 0.0, seed=3)
    want = esim.esim_voxel_batch(vid[None], [0.2, 0.2, 0, 0, 0], bin_mode="sum", num_bins=7, frames_per_bin=1, seed=3)[0, 0]
    assert torch.equal(got, want)
    margs = (24, "pn_related", 0.5, 0.1, 0.0, 0.1, 30, 0.1, 0, 5.0, 0.1, 0.1)
    got = ns["v2e_video_to_voxel"](vid, 24, 0, *margs[2:], seed=3)
    want = v2e.v2e_voxel_batch(vid[None], v2e.make_params(*margs), bin_mode="sum", num_bins=7, frames_per_bin=1, seed=3)[0, 0]
    assert torch.equal(got, want)
    g = np.random.default_rng(0)
    n = 1000
    ev = [torch.from_numpy(x).cuda() for x in (np.sort(g.uniform(0, 0.05, n)), g.integers(0, 64, n), g.integers(0, 32, n),
                                               g.integers(0, 2, n).astype(np.float64))]
    # interpolated bins: float64 atomics, summation order differs from launch to launch
    torch.testing.assert_close(ns["make_voxel"](ev[0], ev[1], ev[2], ev[3], 5, 32, 64), voxel.make_voxel(ev, 32, 64, 5, True),
                               rtol=1e-12, atol=1e-12)
    assert torch.equal(ns["make_voxel"](ev[0], ev[1], ev[2], ev[3], 5, 32, 64, False), voxel.make_voxel(ev, 32, 64, 5, False))


@pytest.mark.gpu
def test_readme_usage_snippet_runs():
    src = open(os.path.join(ROOT, "README.md")).read()
    block = re.search(r"## Using it\n\n```python\n(.*?)```", src, re.S).group(1)
    ns = {}
    exec(block, ns)
    assert ns["voxels"].shape == (256, 5, 256, 256) and ns["voxels"].dtype == torch.float32
    assert ns["counts"].shape == (31, 256, 256) and ns["counts"].dtype == np.float64
    assert ns["dvs"].shape == (256, 5, 256, 256)
