"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/v2v_hip.h declares; argument validation works without a GPU; host logic of the wrappers."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    import __graft_entry__ as ge
    ge.build()
    from v2v_amd import _lib
    return _lib


def test_header_symbols_all_exported(L):
    hdr = open(os.path.join(ROOT, "include", "v2v_hip.h")).read()
    declared = set(re.findall(r"\b(v2v_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"v2v_status", "v2v_dtype"}
    assert declared, "no declarations parsed"
    lib = L.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/v2v_hip.h but not exported"
    assert set(L.EXPORTS) == declared


def test_version_and_no_gpu_behaviour(L):
    lib = L.lib()
    assert lib.v2v_version() == L.ABI_VERSION
    assert lib.v2v_device_count() >= 0


def test_luts_shipped_equal_golden(L, luts):
    lib = L.lib()
    a = np.zeros(256, dtype=np.float64)
    b = np.zeros(256, dtype=np.float32)
    c = np.zeros(256, dtype=np.float32)
    assert lib.v2v_lut_get(0, a.ctypes.data_as(C.c_void_p)) == 0
    assert lib.v2v_lut_get(1, b.ctypes.data_as(C.c_void_p)) == 0
    assert lib.v2v_lut_get(2, c.ctypes.data_as(C.c_void_p)) == 0
    assert np.array_equal(a, luts["lut64"]) and np.array_equal(b, luts["lut32"]) and np.array_equal(c, luts["v2e32"])
    assert lib.v2v_lut_get(9, a.ctypes.data_as(C.c_void_p)) == L.ERR_PARAM


def test_argument_validation_needs_no_gpu(L):
    lib = L.lib()
    dummy = C.c_void_p(4096)
    args = lambda **kw: [kw.get("frames", dummy), kw.get("dt", 0), kw.get("B", 1), kw.get("N", 6), 4, 4,
                         kw.get("cs", 4096), kw.get("fs", 16), kw.get("params", dummy), 0, 0, kw.get("rng", 1), 0, 0, None,
                         kw.get("bin", 0), kw.get("nb", 5), 1, dummy, kw.get("odt", 1), None, None]
    assert lib.v2v_esim_voxel_hip(*args(frames=None)) == L.ERR_NULL
    assert lib.v2v_esim_voxel_hip(*args(N=1)) == L.ERR_SHAPE
    assert lib.v2v_esim_voxel_hip(*args(N=7)) == L.ERR_BINS
    assert b"not a multiple" in lib.v2v_last_error()
    assert lib.v2v_esim_voxel_hip(*args(dt=2)) == L.ERR_DTYPE
    assert lib.v2v_esim_voxel_hip(*args(odt=0)) == L.ERR_DTYPE
    assert lib.v2v_esim_voxel_hip(*args(rng=2)) == L.ERR_MODE
    assert lib.v2v_esim_voxel_hip(*args(bin=7)) == L.ERR_MODE
    assert lib.v2v_esim_voxel_hip(*args(fs=8)) == L.ERR_SHAPE
    assert lib.v2v_esim_voxel_hip(*args(B=0)) == L.OK


def test_algorithmic_bytes_match_survey(L):
    lib = L.lib()
    # config 2 (fp32 in, bilinear 5 bins) = 9,699,328 B/clip; u8 variant 3,407,872; training shape 16,400,384
    assert lib.v2v_esim_voxel_bytes(L.F32, 1, 32, 256, 256, L.BIN_BILINEAR, 5, 1, L.F32) == 9_699_328
    assert lib.v2v_esim_voxel_bytes(L.F32, 256, 32, 256, 256, L.BIN_BILINEAR, 5, 1, L.F32) == 2_483_027_968
    assert lib.v2v_esim_voxel_bytes(L.U8, 1, 32, 256, 256, L.BIN_BILINEAR, 5, 1, L.F32) == 3_407_872
    assert lib.v2v_esim_voxel_bytes(L.U8, 1, 201, 128, 128, L.BIN_SUM, 5, 1, L.F32) == 16_400_384
    assert lib.v2v_esim_voxel_bytes(L.U8, 1, 200, 128, 128, L.BIN_SUM, 5, 1, L.F32) == L.ERR_BINS


def test_product_never_imports_oracle():
    """The product path must not route through the oracle (or any CPU fallback)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "v2v_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".inc")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "libv2v_oracle" not in src, f


def test_wrappers_fail_loudly_without_gpu(L):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from v2v_amd import esim
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        esim.esim_voxel_batch(torch.zeros((1, 6, 4, 4), dtype=torch.uint8), [0.2, 0.2, 0, 0, 0])
    with pytest.raises(RuntimeError):
        esim.synth_clips(1, 2, 4, 4)


def test_binding_constants_match_header(L):
    """The ctypes binding's enum values are the header's (a maintainer's stub would be generated from the same header)."""
    hdr = open(os.path.join(ROOT, "include", "v2v_hip.h")).read()
    vals = {k: int(v, 0) for k, v in re.findall(r"\b(V2V_[A-Z0-9_]+)\s*=\s*(-?\d+|0x[0-9a-fA-F]+)", hdr)}
    vals.update({k: int(v.rstrip("u"), 0) for k, v in re.findall(r"#define\s+(V2V_[A-Z0-9_]+)\s+(0x[0-9a-fA-F]+u?|\d+)\b", hdr)})
    pairs = {"V2V_U8": L.U8, "V2V_F32": L.F32, "V2V_F64": L.F64, "V2V_RNG_NONE": L.RNG_NONE, "V2V_RNG_PHILOX": L.RNG_PHILOX,
             "V2V_RNG_REPLAY": L.RNG_REPLAY, "V2V_RNG_PHILOX_FAST": L.RNG_PHILOX_FAST, "V2V_BIN_SUM": L.BIN_SUM,
             "V2V_BIN_BILINEAR": L.BIN_BILINEAR, "V2V_FLAG_NOISE_EXTERNAL": L.FLAG_NOISE_EXTERNAL, "V2V_FLAG_NO_NOISE": L.FLAG_NO_NOISE,
             "V2V_OK": L.OK, "V2V_ERR_NULL": L.ERR_NULL, "V2V_ERR_SHAPE": L.ERR_SHAPE, "V2V_ERR_BINS": L.ERR_BINS,
             "V2V_ERR_DTYPE": L.ERR_DTYPE, "V2V_ERR_MODE": L.ERR_MODE, "V2V_ERR_ALIGN": L.ERR_ALIGN, "V2V_ERR_HIP": L.ERR_HIP,
             "V2V_ERR_PARAM": L.ERR_PARAM, "V2V_ABI_VERSION": L.ABI_VERSION, "V2V_EV_MAKE_VOXEL_DISCRETE": L.EV_MAKE_VOXEL_DISCRETE,
             "V2V_EV_MAKE_VOXEL_INTERP": L.EV_MAKE_VOXEL_INTERP, "V2V_EV_BILINEAR": L.EV_BILINEAR,
             "V2V_V2E_PN_RELATED": L.V2E_MODELS["pn_related"], "V2V_V2E_SPATIAL_INDEPENDENT": L.V2E_MODELS["spatial_independent"],
             "V2V_V2E_SPATIAL_TEMPORAL_INDEPENDENT": L.V2E_MODELS["spatial_temporal_independent"]}
    for name, py in pairs.items():
        assert name in vals, name
        assert vals[name] == py, (name, vals[name], py)
    import ctypes as C
    assert C.sizeof(L.V2EParams) == 8 * 12 + 8 and C.sizeof(L.EsimReplay) == 32 and C.sizeof(L.V2EReplay) == 56
