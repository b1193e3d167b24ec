"""f-2 "normalise in the writer": kernel time of simulator + normalize_batch_voxel at the training shape (12 x 201 x 128 x 128 uint8 ->
[12,40,5,128,128]) and config 4's per-GPU shape, same box, interleaved:
    sim                         the simulator alone (x16-padded layout)
    sim+stats                   the simulator with the writer's statistics (v2v_esim_voxel_stats_hip)
    scales                      k-th values off the statistics (v2v_voxel_scales_hip)
    apply                       the one scaling pass (v2v_voxel_apply_scales_hip, in place)
    sim+count+normalise         round 3: simulator, then histogram pass + pick + scaling pass over the finished tensor
Run on the GPU box: python tools/writer_stats_time.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v2v_amd import _lib, esim, postops  # noqa: E402


def time_ms(fn, reps=40):
    for _ in range(5):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in ev)
    return t[len(t) // 2]


out = {}
for name, (b, n, h, w) in (("train_12x201x128x128", (12, 201, 128, 128)), ("cfg4_24x41x256x256", (24, 41, 256, 256)), ("cfg4_256x41x256x256", (256, 41, 256, 256))):
    frames = esim.synth_clips(b, n, h, w, dtype=torch.uint8, seed=1)
    g = torch.Generator().manual_seed(0)
    p = torch.stack([torch.rand(b, generator=g) * 0.4 + 0.15, torch.rand(b, generator=g) * 0.4 + 0.15, torch.rand(b, generator=g) * 0.1,
                     torch.rand(b, generator=g) * 1e-3, torch.rand(b, generator=g) * 10], 1).double().cuda()
    keys = torch.stack([torch.arange(b) + 5, torch.arange(b)], 1).cuda()
    L = (n - 1) // 5
    vox = torch.empty((b, L, 5, h, w), dtype=torch.float32, device="cuda")
    stats = torch.empty((b, _lib.VOXEL_STATS_WORDS), dtype=torch.int32, device="cuda")
    kw = dict(bin_mode="sum", num_bins=5, clip_keys=keys, out=vox, validate=False, no_noise=False)
    sim = lambda: esim.esim_voxel_batch(frames, p, **kw)                                    # noqa: E731
    sim_st = lambda: esim.esim_voxel_batch(frames, p, stats=stats, **kw)                     # noqa: E731
    row = {}
    for rnd in range(3):
        for key, fn in (("sim", sim), ("sim+stats", sim_st)):
            row[key] = min(row.get(key, 1e9), time_ms(fn))
    sim_st()
    elems = L * 5 * h * w
    row["scales"] = time_ms(lambda: postops.scales_from_stats(stats, elems))
    sc = postops.scales_from_stats(stats, elems)
    work = vox.clone()
    row["apply"] = time_ms(lambda: postops.apply_scales(work, sc, 16, valid_hw=(h, w), inplace=True))

    def old():
        sim()
        postops.normalize_and_pad(vox, True, 16, method="count", valid_hw=(h, w), inplace=True)

    def new():
        sim_st()
        s2 = postops.scales_from_stats(stats, elems)
        postops.apply_scales(vox, s2, 16, valid_hw=(h, w), inplace=True)

    def new_scales_only():
        sim_st()
        postops.scales_from_stats(stats, elems)
    for rnd in range(3):
        for key, fn in (("sim+count+normalise (round 3)", old), ("sim+stats+scales+apply", new), ("sim+stats+scales (consumer scales while reading)", new_scales_only)):
            row[key] = min(row.get(key, 1e9), time_ms(fn))
    row["algorithmic_bytes_sim"] = esim.algorithmic_bytes(torch.uint8, b, n, h, w, "sum", 5)
    out[name] = row
    print(name, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in row.items()}, flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/writer_stats_time.json", "w"), indent=1)
