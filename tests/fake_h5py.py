"""A stand-in `h5py` for tests ONLY (there is no h5py in this image): just the calls v2v_amd/monash.py:H5Sequence and
v2v_amd/voxel_cache.py:convert make.  A "file" is an .npz behind the scenes: reading maps the flattened names of the Monash fixture
(tests/golden/g16_monash_sequence.npz: "events/ts", "images/stack" + "images/keys" + "images/event_idx" / "images/timestamp", "attrs/*") onto
groups / datasets / attrs; writing collects create_dataset() / attrs and saves them as an .npz at the given path on close.  It tests the
control flow of the .h5 branches (which never ran before round 5), not HDF5."""
import numpy as np


class _Dataset:
    def __init__(self, arr, attrs=None):
        self._a, self.attrs = arr, attrs or {}

    def __getitem__(self, idx):
        return self._a[idx] if idx != () else self._a

    def __len__(self):
        return len(self._a)

    @property
    def shape(self):
        return self._a.shape


class _Group(dict):
    attrs = None


class File:
    def __init__(self, path, mode="r"):
        self.path, self.mode, self.attrs, self._root, self.closed = str(path), mode, {}, _Group(), False
        if mode == "r":
            z = np.load(self.path, allow_pickle=False)
            for k in z.files:
                if k.startswith("attrs/"):
                    v = z[k]
                    self.attrs[k[6:]] = str(v) if v.dtype.kind in "US" else (v if v.ndim else v[()])
            ev = _Group({k[7:]: _Dataset(z[k]) for k in z.files if k.startswith("events/")})
            if ev:
                self._root["events"] = ev
            if "images/keys" in z.files:
                imgs = _Group()
                for i, key in enumerate(z["images/keys"]):
                    imgs[str(key)] = _Dataset(z["images/stack"][i], {"event_idx": z["images/event_idx"][i], "timestamp": z["images/timestamp"][i]})
                self._root["images"] = imgs
            if "flow/keys" in z.files:                           # MVSEC-style sequences: a group of optic-flow maps with event_idx / image_idx attrs
                fl = _Group()
                for i, key in enumerate(z["flow/keys"]):
                    fl[str(key)] = _Dataset(z["flow/stack"][i], {"event_idx": z["flow/event_idx"][i], "image_idx": z["flow/image_idx"][i]})
                self._root["flow"] = fl
            for k in z.files:                                   # flat datasets of a cached-voxel file (frames, flow, events, timestamps, dt)
                if "/" not in k:
                    self._root[k] = _Dataset(z[k])

    def keys(self):
        return self._root.keys()

    def __getitem__(self, name):
        node = self._root
        for part in name.split("/"):
            node = node[part]
        return node

    def create_dataset(self, name, data=None, dtype=None):
        assert self.mode == "w"
        self._root[name] = _Dataset(np.asarray(data, dtype=dtype))

    def close(self):
        if self.mode == "w" and not self.closed:
            np.savez(open(self.path, "wb"), **{k: v._a for k, v in self._root.items()}, **{f"attrs/{k}": np.asarray(v) for k, v in self.attrs.items()})
        self.closed = True

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
