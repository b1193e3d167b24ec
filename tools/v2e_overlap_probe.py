"""Probe: does the memory-bound frame-sum pre-pass of one chunk overlap with the VALU-bound main kernel of another when the
config-3 batch is split over two HIP streams?  """
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from v2v_amd import esim, v2e  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    wl = bench.WORKLOADS["cfg3_v2e_f32_256x32x256x256_bilinear5"]
    for dtype in (torch.float32, torch.uint8):
        frames = esim.synth_clips(256, 32, 256, 256, dtype=dtype, seed=20240001, clip_id0=0, device=dev)
        params = v2e.make_params(*wl["params"])
        out = torch.empty((256, 5, 256, 256), dtype=torch.float32, device=dev)
        ref = v2e.v2e_voxel_batch(frames, params, bin_mode="bilinear", num_bins=5, seed=1, clip_id0=0).clone()
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]

        def run(chunks, nstreams):
            per = 256 // chunks
            cur = torch.cuda.current_stream()
            if nstreams == 0:
                for c in range(chunks):
                    v2e.v2e_voxel_batch(frames[c * per:(c + 1) * per], params, bin_mode="bilinear", num_bins=5, seed=1, clip_id0=c * per,
                                        out=out[c * per:(c + 1) * per])
                return
            for s in streams[:nstreams]:
                s.wait_stream(cur)
            for c in range(chunks):
                with torch.cuda.stream(streams[c % nstreams]):
                    v2e.v2e_voxel_batch(frames[c * per:(c + 1) * per], params, bin_mode="bilinear", num_bins=5, seed=1, clip_id0=c * per,
                                        out=out[c * per:(c + 1) * per])
            for s in streams[:nstreams]:
                cur.wait_stream(s)

        for chunks, ns in ((1, 0), (2, 0), (4, 0), (2, 2), (4, 2), (8, 2), (16, 2)):
            for _ in range(3):
                run(chunks, ns)
            torch.cuda.synchronize()
            ts = []
            for _ in range(10):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run(chunks, ns)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ts.sort()
            same = bool(torch.equal(out, ref))
            print(f"{str(dtype):14s} chunks={chunks:2d} streams={ns}: median {ts[len(ts)//2]:.3f} ms  min {ts[0]:.3f}  identical={same}", flush=True)


if __name__ == "__main__":
    main()
