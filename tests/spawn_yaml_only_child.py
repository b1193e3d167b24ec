"""Child process of tests/test_hip_dataset_events.py: the YAML-only route in a FRESH program, as train.py is one -- no multiprocessing start
method fixed yet, the dataset's `worker_start_method: spawn` key fixes it, the DataLoader is built exactly like train.py:52-65 (no
multiprocessing_context).  Prints one JSON line: batches equal the in-process samples, how the workers were started, their exit codes."""
import json
import multiprocessing
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main(tmp, workers, output_device):
    import torch
    from torch.utils.data import ConcatDataset, DataLoader
    from v2v_amd.datasets import WebvidDatasetV2, synthetic_frame_source
    lst = os.path.join(tmp, "videos.txt")
    with open(lst, "w") as f:
        f.write("clip_a.mp4 450 0.2 0.3\nclip_b.mp4 300 0.25 0.25\n")
    configs = {"video_list_file": lst, "sequence_length": 4, "crop_size": 32, "data_source_name": "webvid", "video_size": (1280, 720),
               "video_reader": "opencv", "fixed_seed": 31, "max_samples_per_shot": 4, "step_size": 20, "frame_source": synthetic_frame_source,
               "worker_start_method": "spawn", "output_device": output_device}
    assert multiprocessing.get_start_method(allow_none=True) is None
    ds = WebvidDatasetV2(tmp, configs)
    start_method = multiprocessing.get_start_method(allow_none=True)
    wrapped = ConcatDataset([ConcatDataset([ds])])                                                   # data/data_interface.py:19,21,27
    loader = DataLoader(wrapped, batch_size=2, shuffle=False, num_workers=workers, persistent_workers=True, pin_memory=output_device == "cpu",
                        drop_last=True)                                                              # train.py:52-65
    equal, n, on_cuda = True, 0, True
    procs, popen = [], None
    for _epoch in range(2):
        it = iter(loader)
        procs = list(it._workers)
        popen = type(procs[0]._popen).__module__
        for bi, batch in enumerate(it):
            got = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}  # clone: release the producers' blocks
            del batch
            on_cuda = on_cuda and (got["events"].is_cuda == (output_device == "cuda"))
            for j in range(2):
                ref = wrapped[2 * bi + j]
                equal = equal and torch.equal(got["events"][j].cpu(), ref["events"].cpu()) and torch.equal(got["frame"][j].cpu(), ref["frame"].cpu())
            n += 1
    it._shutdown_workers()
    del it, loader
    print(json.dumps({"equal": bool(equal), "batches": n, "start_method": start_method, "popen": popen, "device_ok": bool(on_cuda),
                      "exit_codes": [p.exitcode for p in procs]}))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3])
