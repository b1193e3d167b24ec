"""v2v_amd -- MI355X-native video -> event-count -> voxel-grid simulator (the one accelerated path of
HYLZ-2019/V2V).  Python host code over a C ABI (include/v2v_hip.h) into hand-written gfx950 HIP kernels.

Surface mirrored from the reference (same names, argument order and error behaviour):
    v2v_amd.esim.EventEmulator            data/v2v_core_esim.py:6-69
    v2v_amd.datasets.imgs_to_voxels       data/v2v_datasets.py:363-410
    v2v_amd.esim.esim_voxel_batch         (batched form: the measured hot path)

There is no CPU fallback: importing the compute modules without the built HIP library raises.
"""
__version__ = "0.1.0"
