"""Two training batches in flight: does the chip have room for batch k+1's simulator under batch k's?

Training shape (config/train_v2v_e2vid_10k.yaml:50-76): B = 12 clips of 201 x 128 x 128 uint8 -> [12,40,5,128,128] SUM bins, dataset-style
parameters, per-clip {seed, clip id} keys.  Same box, interleaved, median of `reps` timings each:
    one_stream      batch A then batch B, one stream (what RingLoader issues today)
    two_streams     batch A on stream 0, batch B on stream 1, launched back to back (the kernels may overlap)
    one_launch_24   A and B as ONE launch of 24 clips
    one_launch_48   four batches as one launch
all three must give bit-identical grids (per-clip keys make a clip's result independent of the launch it rides in).
    python tools/train_two_batches_probe.py [--reps 40]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v2v_amd import esim  # noqa: E402

P = [0.2, 0.3, 0.05, 5e-4, 1.0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=40)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    b, n, h, w = 12, 201, 128, 128
    frames = esim.synth_clips(4 * b, n, h, w, dtype=torch.uint8, seed=3, clip_id0=0, device=dev)
    keys = torch.stack([torch.full((4 * b,), 77, dtype=torch.int64), torch.arange(4 * b, dtype=torch.int64)], 1).to(dev)
    pt = torch.tensor([P] * (4 * b), dtype=torch.float64, device=dev)
    out = torch.empty((4 * b, 40, 5, h, w), dtype=torch.float32, device=dev)
    ref = torch.empty_like(out)

    def launch(lo, hi, dst=out):
        esim.esim_voxel_batch(frames[lo:hi], pt[lo:hi], bin_mode="sum", num_bins=5, clip_keys=keys[lo:hi], out=dst[lo:hi], validate=False, no_noise=False)
    for i in range(4):
        launch(i * b, (i + 1) * b, ref)
    torch.cuda.synchronize()
    s0, s1 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    cur = torch.cuda.current_stream(dev)

    def one_stream():
        launch(0, b)
        launch(b, 2 * b)

    def two_streams():
        s0.wait_stream(cur)
        s1.wait_stream(cur)
        with torch.cuda.stream(s0):
            launch(0, b)
        with torch.cuda.stream(s1):
            launch(b, 2 * b)
        cur.wait_stream(s0)
        cur.wait_stream(s1)

    def one_launch_24():
        launch(0, 2 * b)

    def one_launch_48():
        launch(0, 4 * b)

    modes = {"one_stream": (one_stream, 2), "two_streams": (two_streams, 2), "one_launch_24": (one_launch_24, 2), "one_launch_48": (one_launch_48, 4)}
    res = {k: [] for k in modes}
    for name, (fn, nb) in modes.items():
        out.zero_()
        fn()
        torch.cuda.synchronize()
        assert torch.equal(out[:nb * b], ref[:nb * b]), name                  # bit-identical whatever launch a clip rides in
    for _ in range(3):
        for name, (fn, nb) in modes.items():
            for _ in range(5):
                fn()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
            for s, e in ev:
                s.record()
                fn()
                e.record()
            torch.cuda.synchronize()
            t = sorted(s.elapsed_time(e) for s, e in ev)
            res[name].append(t[len(t) // 2] / nb)
    summary = {k: min(v) for k, v in res.items()}
    base = summary["one_stream"]
    report = {"shape": "12 x 201 x 128 x 128 uint8 -> [12,40,5,128,128] f32 SUM, dataset-style parameters, per-clip keys",
              "ms_per_12_clip_batch": summary, "throughput_vs_one_stream": {k: base / v for k, v in summary.items()},
              "bit_identical": True, "algorithmic_bytes_per_batch": esim.algorithmic_bytes(torch.uint8, b, n, h, w, "sum", 5, 1)}
    report["frac_of_hbm_peak"] = {k: report["algorithmic_bytes_per_batch"] / (v * 1e-3) / 8e12 for k, v in summary.items()}
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(report, open("gpurun_out/r6_train_two_batches.json", "w"), indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
