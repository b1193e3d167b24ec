"""The N>1 path on CPU: two gloo ranks shard a batch with v2v_amd.sharding, each simulates ITS clips (the CPU
oracle stands in for the GPU, which this container lacks), results are gathered and must equal the single-process
run bit for bit -- the property (global clip-id keyed RNG, no exchange) that makes the 8-GPU bench a pure shard."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_partition():
    from v2v_amd.sharding import shard_range, weak_shard
    for total in (0, 1, 7, 256, 257):
        for world in (1, 2, 3, 8):
            shards = [shard_range(total, r, world) for r in range(world)]
            assert shards[0].lo == 0 and shards[-1].hi == total
            assert all(a.hi == b.lo for a, b in zip(shards, shards[1:]))
            assert max(s.count for s in shards) - min(s.count for s in shards) <= 1
    s = weak_shard(256, 3, 8)
    assert (s.lo, s.hi, s.count) == (768, 1024, 256)
    with pytest.raises(ValueError):
        shard_range(4, 4, 4)


WORKER = textwrap.dedent("""
    import os, sys, json, time
    import numpy as np, torch
    sys.path.insert(0, {root!r})
    from v2v_amd import sharding
    from oracle import clib, v2v_oracle as O
    rank, local_rank, world = sharding.env_rank_world()
    dist = sharding.init_process_group("gloo")
    total, n, h, w = 6, 11, 16, 20
    video = np.stack([O.synth_clip_s1(n, h, w, seed=50 + i, dtype=np.uint8) for i in range(total)])
    sh = sharding.shard_range(total, rank, world)
    sharding.barrier(dist)
    t0 = time.perf_counter()
    mine, totals = clib.esim_voxel(video[sh.lo:sh.hi], [0.2, 0.3, 0.04, 0.02, 0.6], O.load_luts(), seed=4242,
                                   clip_id0=sh.lo, bin_mode=clib.BIN_SUM, num_bins=5, frames_per_bin=2)
    sharding.barrier(dist)
    dt = sharding.max_over_ranks(dist, time.perf_counter() - t0)
    parts = [None] * world
    dist.all_gather_object(parts, (sh.lo, mine))
    if rank == 0:
        parts.sort(key=lambda p: p[0])
        np.save({out!r}, np.concatenate([p[1] for p in parts]))
        print(json.dumps({{"elapsed_max": dt, "world": world}}))
    dist.destroy_process_group()
""")


def test_two_gloo_ranks_equal_single_process(tmp_path, oracle_c, luts):
    from oracle import v2v_oracle as O
    out = str(tmp_path / "gathered.npy")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=out))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    got = np.load(out)
    video = np.stack([O.synth_clip_s1(11, 16, 20, seed=50 + i, dtype=np.uint8) for i in range(6)])
    want, _ = oracle_c.esim_voxel(video, [0.2, 0.3, 0.04, 0.02, 0.6], luts, seed=4242, clip_id0=0,
                                  bin_mode=oracle_c.BIN_SUM, num_bins=5, frames_per_bin=2)
    assert got.shape == want.shape and np.array_equal(got, want)
