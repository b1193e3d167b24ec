"""One rank of tests/test_loader.py::test_create_dataloader_under_two_ddp_ranks: what train.py does per process -- init the process group
(train.py:41-47; gloo here: both ranks share the test box's one GPU), build the dataset through the plugin loader's nesting, call
create_dataloader(dataset, configs, batch_size, local_rank), iterate one epoch -- and write the sample keys it received."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _frames(ds, sample_idx, start, end, crop_before, min_i, min_j, flip, need_h, need_w):
    g = np.random.default_rng(1000 + start + 7 * int(sample_idx))
    base = g.uniform(0, 255, size=(need_h, need_w, 1))
    out = []
    for _ in range(end - start):
        base = np.clip(base + g.normal(0, 6, size=base.shape), 0, 255)
        out.append(base.astype(np.uint8))
    return out


def main():
    out_dir = sys.argv[1]
    rank = int(os.environ["RANK"])
    dist.init_process_group(backend="gloo", init_method="env://")
    from torch.utils.data import ConcatDataset
    from v2v_amd.datasets import WebvidDatasetV2
    from v2v_amd.loader import RingLoader, create_dataloader
    lst = os.path.join(out_dir, "videos.txt")
    if rank == 0:
        with open(lst, "w") as f:
            f.write("".join(f"clip_{i}.mp4 {300 + i} 0.2 0.3\n" for i in range(12)))
    dist.barrier()
    cfg = {"video_list_file": lst, "sequence_length": 4, "crop_size": 32, "data_source_name": "webvid", "frame_source": _frames,
           "video_size": (1280, 720), "video_reader": "opencv", "fixed_seed": 3}
    ds = ConcatDataset([ConcatDataset([WebvidDatasetV2(out_dir, cfg)])])
    loader = create_dataloader(ds, {"num_workers": 2, "persistent_workers": False}, 3, 0)      # LOCAL_RANK 0 on both: one GPU on the box
    assert isinstance(loader, RingLoader) and type(loader.sampler).__name__ == "DistributedSampler"
    seen = []
    for epoch in range(2):
        loader.sampler.set_epoch(epoch)                                                    # train.py does this per epoch under DDP
        keys = []
        for batch in loader:
            assert batch["events"].is_cuda and batch["events"].shape == (3, 4, 5, 32, 32)
            # fixed_seed: a sample is a pure function of its index -> identify it by its parameters
            keys += [round(float(v), 12) for v in batch["v2e_params"]["pos_thres"]]
        seen.append(keys)
    json.dump(seen, open(os.path.join(out_dir, f"rank{rank}.json"), "w"))
    loader.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
