"""Time the event-list voxelisers (v2v_events.hpp) on synthetic event streams: make_voxel float64 (discrete / interpolated),
the float32 twin, and the segmented form (every image interval of a sequence in one launch).  Run under rocprofv3 by
tools/profile_events.sh for the kernel-level numbers kept in profiles/."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v2v_amd import voxel  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in ev:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in ev)
    return ts[len(ts) // 2]


def main():
    g = np.random.default_rng(0)
    h, w, nb = 260, 346, 5                                       # DAVIS346 geometry, the reference's 5 bins
    for n in (200_000, 2_000_000, 20_000_000):
        ts = np.sort(g.uniform(0, 1, n))
        ts_d = torch.from_numpy(ts).cuda()
        xs, ys = torch.from_numpy(g.integers(0, w, n)).cuda(), torch.from_numpy(g.integers(0, h, n)).cuda()
        ps01 = torch.from_numpy(g.integers(0, 2, n).astype(np.float64)).cuda()           # make_voxel takes polarities in {0,1}
        d = [ts_d, xs, ys, ps01]
        row = {"events": n, "sensor": [h, w], "bins": nb}
        row["make_voxel_discrete_ms"] = timeit(lambda: voxel.make_voxel(d, h, w, nb, interpolate_bins=False))
        row["make_voxel_interpolated_ms"] = timeit(lambda: voxel.make_voxel(d, h, w, nb, interpolate_bins=True))
        tf, pf = ts_d.float(), (ps01 * 2 - 1).float()
        row["events_to_voxel_torch_f32_ms"] = timeit(lambda: voxel.events_to_voxel_torch(xs, ys, tf, pf, nb, sensor_size=(h, w)))
        idx = np.linspace(0, n, 41).astype(np.int64)             # 40 image intervals
        row["segmented_40_intervals_ms"] = timeit(lambda: voxel.make_voxels_segmented(d, idx, h, w, nb, interpolate_bins=True))
        row["Mevents_per_s_interpolated"] = n / row["make_voxel_interpolated_ms"] / 1e3
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
