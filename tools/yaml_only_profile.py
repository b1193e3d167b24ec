"""Where a sample's 5.9 ms go on the zero-edit path (DataLoader(num_workers=0) over WebvidDatasetV2(output_device: cuda)):
cProfile of 120 __getitem__ calls + default_collate at the training shape, decode excluded (PooledFrameSource)."""
import cProfile
import os
import pstats
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import loader_bench  # noqa: E402

src = loader_bench.PooledFrameSource()
with tempfile.TemporaryDirectory() as tmp:
    ds = loader_bench.make_dataset(tmp, 400, src, defer_sim=False, output_device="cuda")
    from torch.utils.data import default_collate
    for i in range(24):
        ds[i]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range(10):
        default_collate([ds[24 + 12 * b + j] for j in range(12)])
    torch.cuda.synchronize()
    print("samples/s", 120 / (time.perf_counter() - t0))
    pr = cProfile.Profile()
    pr.enable()
    for b in range(10):
        default_collate([ds[150 + 12 * b + j] for j in range(12)])
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
