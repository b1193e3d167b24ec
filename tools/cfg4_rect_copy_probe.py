"""Config-4 stream: how fast do the CROP RECTANGLES of page-locked 720p frames reach the GPU by other routes than the zero-copy front-end?
    a) hipMemcpy3DAsync, one call per clip (40 frames x cb rows x cb*3 bytes, pitched on both sides) into a compact device buffer
    b) hipMemcpy2DAsync, one call per frame
    c) the front-end kernel reading host memory (what bench.py --workload cfg4_stream does), for reference: rect bytes / step time
Reports GB/s of rectangle bytes.  python tools/cfg4_rect_copy_probe.py"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class Pos(C.Structure):
    _fields_ = [("x", C.c_size_t), ("y", C.c_size_t), ("z", C.c_size_t)]


class Pitched(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("pitch", C.c_size_t), ("xsize", C.c_size_t), ("ysize", C.c_size_t)]


class Extent(C.Structure):
    _fields_ = [("width", C.c_size_t), ("height", C.c_size_t), ("depth", C.c_size_t)]


class Parms3D(C.Structure):
    _fields_ = [("srcArray", C.c_void_p), ("srcPos", Pos), ("srcPtr", Pitched), ("dstArray", C.c_void_p), ("dstPos", Pos), ("dstPtr", Pitched),
                ("extent", Extent), ("kind", C.c_int)]


def main():
    from v2v_amd import esim, frontend, _lib
    hip = C.CDLL("libamdhip64.so")
    dev = torch.device("cuda", 0)
    b, n, sh, sw, crop = 8, 40, 720, 1280, 256
    raw = torch.empty((b, n, sh, sw, 3), dtype=torch.uint8, device=dev)
    for ch in range(3):
        raw[..., ch] = esim.synth_clips(b, n, sh, sw, dtype=torch.uint8, seed=5 + ch, device=dev)
    host = raw.cpu().pin_memory()
    g = np.random.default_rng(1)
    keep_h = int(sh * 0.54)
    scale = g.uniform(max(crop / keep_h, crop / sw), 1.3, size=b)
    cb = (crop / scale).astype(np.int64)
    table = np.stack([[g.integers(0, keep_h - c + 1), g.integers(0, sw - c + 1), c, 0] for c in cb]).astype(np.int32)
    cbm = int(cb.max())
    rect_bytes = int((cb ** 2).sum()) * 3 * n
    compact = torch.zeros((b, n, cbm, cbm, 3), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    H2D = 1
    out = {"rect_MB_per_step": rect_bytes / 1e6, "whole_frames_MB_per_step": host.numel() / 1e6}

    def copy3d():
        for c in range(b):
            p = Parms3D()
            p.srcPos = Pos(int(table[c, 1]) * 3, int(table[c, 0]), 0)
            p.srcPtr = Pitched(host.data_ptr() + c * n * sh * sw * 3, sw * 3, sw * 3, sh)
            p.dstPos = Pos(0, 0, 0)
            p.dstPtr = Pitched(compact.data_ptr() + c * n * cbm * cbm * 3, cbm * 3, cbm * 3, cbm)
            p.extent = Extent(int(cb[c]) * 3, int(cb[c]), n)
            p.kind = H2D
            rc = hip.hipMemcpy3DAsync(C.byref(p), C.c_void_p(stream))
            assert rc == 0, rc

    def copy2d():
        for c in range(b):
            for t in range(n):
                src = host.data_ptr() + ((c * n + t) * sh * sw + int(table[c, 0]) * sw + int(table[c, 1])) * 3
                dst = compact.data_ptr() + (c * n + t) * cbm * cbm * 3
                rc = hip.hipMemcpy2DAsync(C.c_void_p(dst), C.c_size_t(cbm * 3), C.c_void_p(src), C.c_size_t(sw * 3), C.c_size_t(int(cb[c]) * 3),
                                          C.c_size_t(int(cb[c])), H2D, C.c_void_p(stream))
                assert rc == 0, rc

    idx = np.tile(np.arange(n, dtype=np.int32), (b, 1))
    tab_d, idx_d = torch.as_tensor(table, device=dev), torch.as_tensor(idx, device=dev)

    def zero_copy():
        frontend.prepare_clips_batch(host, tab_d, idx_d, crop, "gray", validate=False, max_crop_before=cbm)

    for name, fn, reps in (("hipMemcpy3DAsync_per_clip", copy3d, 10), ("hipMemcpy2DAsync_per_frame", copy2d, 5), ("front_end_reads_host_memory", zero_copy, 20)):
        try:
            fn()
            torch.cuda.synchronize()
            if name.startswith("hipMemcpy3D"):                                 # check the first clip's first frame landed
                c0 = compact[0, 0, :int(cb[0]), :int(cb[0])].cpu()
                want = host[0, 0, int(table[0, 0]):int(table[0, 0]) + int(cb[0]), int(table[0, 1]):int(table[0, 1]) + int(cb[0])]
                out[name + "_correct"] = bool(torch.equal(c0, want))
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            out[name] = {"ms_per_step": dt * 1e3, "rect_GBps": rect_bytes / dt / 1e9}
        except Exception as exc:  # noqa: BLE001
            out[name] = {"error": f"{type(exc).__name__}: {exc}"}
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/cfg4_rect_copy_probe.json", "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
