"""Read access to event sequences in the Monash HDF5 layout the reference's real-data loaders use
(data/testh5.py:33-50,111-119; data/dataset.py:375-427; written by scripts/*_to_h5.py):

    events/{ts, xs, ys, ps}                         ts float64 seconds, ps in {0, 1}
    images/image%09d            [H, W] uint8        attrs: event_idx (first event AFTER this image), timestamp
    attrs                       sensor_resolution, num_events, num_imgs, source

`open_sequence(path)` returns a store with that access pattern for either container:
  * an .h5 file (needs h5py, which this image does not ship -- imported lazily), or
  * an .npz file holding the same datasets under flattened names (tests/golden/g16_monash_sequence.npz):
        "events/ts" ...            the event arrays
        "images/stack"             [n, H, W] uint8, "images/keys" the image names, "images/event_idx", "images/timestamp"
        "attrs/sensor_resolution", "attrs/num_events", "attrs/num_imgs", "attrs/source"
        "flow/stack" [m, 2, H, W], "flow/keys", "flow/event_idx", "flow/image_idx"      (MVSEC-style sequences with optic flow, optional)
    and voxel caches (data/testh5.py:383-446): top-level datasets "frames" [n,H,W], "events" [n,Tb,H,W], attrs num_bins, interpolate_bins
Host-side IO only; the voxelisation itself is v2v_amd/voxel.py -> HIP.
"""
from __future__ import annotations

import numpy as np


class NpzSequence:
    def __init__(self, path):
        z = np.load(path, allow_pickle=False)
        self._z = {k: z[k] for k in z.files}
        self.image_keys = [str(k) for k in self._z["images/keys"]] if "images/keys" in self._z else []
        self._idx = {k: i for i, k in enumerate(self.image_keys)}

    def image(self, key):
        return self._z["images/stack"][self._idx[key]]

    def image_attr(self, key, name):
        return self._z[f"images/{name}"][self._idx[key]]

    def events(self, name, a=None, b=None):
        return self._z[f"events/{name}"][a:b]

    def attr(self, name, default=None):
        v = self._z.get(f"attrs/{name}")
        if v is None:
            return default
        return str(v) if v.dtype.kind in "US" else (v if v.ndim else v[()])

    def has_flow(self):
        return "flow/stack" in self._z and len(self._z["flow/stack"]) > 0

    @property
    def flow_keys(self):
        return [str(k) for k in self._z["flow/keys"]]

    def flow(self, key):
        return self._z["flow/stack"][self.flow_keys.index(key)]

    def flow_attr(self, key, name):
        return self._z[f"flow/{name}"][self.flow_keys.index(key)]

    def dataset(self, name, a=None, b=None):
        """A top-level dataset (voxel caches: `frames`, `events`)."""
        return self._z[name][a:b]

    def dataset_len(self, name):
        return len(self._z[name])

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class H5Sequence:
    def __init__(self, path):
        import h5py                                           # absent from this image; present where real data lives
        self._f = h5py.File(path, "r")
        self.image_keys = sorted(self._f["images"].keys()) if "images" in self._f.keys() else []

    def image(self, key):
        return self._f["images"][key][()]

    def image_attr(self, key, name):
        return self._f["images"][key].attrs[name]

    def events(self, name, a=None, b=None):
        return self._f[f"events/{name}"][a:b]

    def attr(self, name, default=None):
        return self._f.attrs.get(name, default)

    def has_flow(self):
        return "flow" in self._f.keys() and len(self._f["flow"]) > 0

    @property
    def flow_keys(self):
        return sorted(self._f["flow"].keys())

    def flow(self, key):
        return self._f["flow"][key][()]

    def flow_attr(self, key, name):
        return self._f["flow"][key].attrs[name]

    def dataset(self, name, a=None, b=None):
        return self._f[name][a:b]

    def dataset_len(self, name):
        return self._f[name].shape[0]

    def close(self):
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def open_sequence(path):
    return NpzSequence(path) if str(path).endswith(".npz") else H5Sequence(path)
