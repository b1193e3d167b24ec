"""Where the host time of RingLoader._stage goes: pieces of the H2D enqueue out of page-locked (hipHostRegister) shared memory,
timed on the host (us per call) and on the device (GB/s), against torch-pinned memory."""
import mmap
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def t(fn, reps=200):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    host = (time.perf_counter() - t0) / reps * 1e6
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / reps * 1e6
    return host, total


n = 12 * 201 * 128 * 128 + 4096
m = mmap.mmap(-1, 4 * n)
ring = np.frombuffer(m, dtype=np.uint8).reshape(4, n)
ring[:] = 1
rc = torch.cuda.cudart().cudaHostRegister(ring.ctypes.data, ring.nbytes, 0)
print("register rc", rc)
ring_t = torch.from_numpy(ring)
print("is_pinned(registered):", ring_t.is_pinned())
pin = torch.empty((4, n), dtype=torch.uint8).pin_memory()
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
cs = torch.cuda.Stream()
for name, src in (("registered shared mmap", ring_t), ("torch pinned", pin)):
    def cp(src=src):
        with torch.cuda.stream(cs):
            dev.copy_(src[1], non_blocking=True)
    h, tot = t(cp, 100)
    print(f"{name}: host {h:.1f} us/call, {n / tot / 1e3:.1f} GB/s")
ev = torch.cuda.Event()
print("Event() + record us", t(lambda: torch.cuda.Event().record(cs))[0])
print("wait_event us", t(lambda: cs.wait_event(ev))[0])
print("index us", t(lambda: ring_t[1])[0])
# raw hipMemcpyAsync through the runtime torch loaded
import ctypes as C
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
def raw():
    hip.hipMemcpyAsync(dev.data_ptr(), ring.ctypes.data + n, n, 1, cs.cuda_stream)
h, tot = t(raw, 100)
print(f"raw hipMemcpyAsync registered: host {h:.1f} us/call, {n / tot / 1e3:.1f} GB/s")
def raw2():
    hip.hipMemcpyAsync(dev.data_ptr(), pin[1].data_ptr(), n, 1, cs.cuda_stream)
h, tot = t(raw2, 100)
print(f"raw hipMemcpyAsync torch-pinned: host {h:.1f} us/call, {n / tot / 1e3:.1f} GB/s")
