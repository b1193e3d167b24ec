"""Teardown of spawned DataLoader workers that hand CUDA tensors back: runs `rounds` epochs of train.py's own loader
over v2v_amd.datasets.WebvidDatasetV2 with `worker_start_method: spawn`, `output_device: cuda`, and prints the workers' exit codes as JSON.

    python tools/spawn_teardown_probe.py [--rounds 4] [--workers 9] [--batches 40] [--persistent 1] 2> err.log

A worker that dies in std::terminate shows up as exit code -6 and as "terminate called" in err.log; with
LD_PRELOAD=tools/diag/terminate_trace.so the log also has the native backtrace of the thread that called it.
V2V_WORKER_EXIT_HOOK=0 disables the dataset's atexit hook (the round-5 behaviour)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--workers", type=int, default=9)
    ap.add_argument("--batches", type=int, default=40)
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--persistent", type=int, default=1)
    ap.add_argument("--output-device", default="cuda")
    a = ap.parse_args()
    import torch
    from torch.utils.data import DataLoader, RandomSampler
    import loader_bench
    dev = torch.device("cuda", 0)
    src = loader_bench.PooledFrameSource()
    codes = []
    with tempfile.TemporaryDirectory() as tmp:
        ds = loader_bench.make_dataset(tmp, a.batches * a.batch, src, defer_sim=False, output_device=a.output_device, worker_start_method="spawn")
        for _ in range(a.rounds):
            loader = DataLoader(ds, batch_size=a.batch, sampler=RandomSampler(ds, num_samples=a.batches * a.batch), num_workers=a.workers, persistent_workers=bool(a.persistent),
                                pin_memory=a.output_device == "cpu", drop_last=True, multiprocessing_context=ds.multiprocessing_context)
            it = iter(loader)
            workers = list(it._workers)
            n = 0
            for b in it:                                   # the whole epoch, as train.py's loop runs it: nothing is left prefetched in the workers
                for k, v in b.items():
                    if isinstance(v, torch.Tensor):
                        b[k] = v.to(dev, non_blocking=True)
                n += 1
            assert n == a.batches
            torch.cuda.synchronize(dev)
            del b, v
            it._shutdown_workers()
            codes.append([w.exitcode for w in workers])
            del it, loader
    print(json.dumps({"worker_exit_codes": codes, "all_zero": all(c == 0 for r in codes for c in r)}))


if __name__ == "__main__":
    main()
