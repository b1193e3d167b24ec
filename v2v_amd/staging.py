"""Host -> device staging for host-resident clips (BASELINE config 4's stream: decoded frames arrive in host memory).

The C ABI takes device pointers, so a host pipeline pays the PCIe copy first.  `HostStager` keeps that copy off the
simulator's critical path: two page-locked staging buffers and two device buffers per tensor shape, and a dedicated copy
stream -- batch k+1 crosses PCIe while batch k is simulated; the consumer stream only waits on the copy's event.
One process per GPU: every rank owns its stager (and its PCIe link); nothing here is shared between ranks.
"""
from __future__ import annotations

import torch


class HostStager:
    def __init__(self, device="cuda", depth: int = 2):
        self.device = torch.device(device)
        self.depth = depth
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._slots = {}          # (shape, dtype) -> list of [pinned, dev, event]
        self._turn = {}

    def _slot(self, t: torch.Tensor):
        key = (tuple(t.shape), t.dtype)
        if key not in self._slots:
            try:
                mk_pin = lambda: torch.empty(t.shape, dtype=t.dtype).pin_memory()   # noqa: E731
                self._slots[key] = [[mk_pin(), torch.empty(t.shape, dtype=t.dtype, device=self.device), torch.cuda.Event(), False]
                                    for _ in range(self.depth)]
            except RuntimeError:                       # RLIMIT_MEMLOCK too small for page-locked memory: pageable copies
                self._slots[key] = [[None, torch.empty(t.shape, dtype=t.dtype, device=self.device), torch.cuda.Event(), False]
                                    for _ in range(self.depth)]
            self._turn[key] = 0
        i = self._turn[key]
        self._turn[key] = (i + 1) % self.depth
        return self._slots[key][i]

    def stage(self, host: torch.Tensor):
        """Start the asynchronous copy of `host` (CPU tensor) and return a handle for `ready()`.  The slot used `depth`
        calls ago is recycled: its previous consumer must have been enqueued before this call (true for a loop that
        consumes batch k before staging batch k+depth)."""
        if host.is_cuda:
            return (host, None)
        slot = self._slot(host)
        pinned, dev, ev, used = slot
        cur = torch.cuda.current_stream(self.device)
        self.copy_stream.wait_stream(cur)              # the slot's last consumer (enqueued on `cur`) must be done with `dev`
        with torch.cuda.stream(self.copy_stream):
            if host.is_pinned() or pinned is None:
                dev.copy_(host, non_blocking=True)
            else:
                # The DMA out of this page-locked buffer that was queued `depth` calls ago sits behind the compute stream's
                # backlog (wait_stream above is a GPU-side dependency: it does not hold the host back).  A host that runs
                # ahead of the GPU would overwrite the buffer before that DMA has read it -- wait for it ON THE HOST first.
                if used:
                    ev.synchronize()
                pinned.copy_(host)                     # host memcpy into the page-locked buffer (overlaps the GPU's work)
                dev.copy_(pinned, non_blocking=True)
            ev.record(self.copy_stream)
            slot[3] = True
        return (dev, ev)

    def ready(self, handle) -> torch.Tensor:
        """The device tensor of a staged batch; the CURRENT stream waits for its copy (no host synchronisation)."""
        dev, ev = handle
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
        return dev
