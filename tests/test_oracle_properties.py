"""Property tests of the CPU oracle (hypothesis): the restated np.floor_divide against NumPy itself, the exact
floor-divide identity the HIP kernel relies on, and invariances of the restated simulator."""
import math

import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import v2v_oracle as O

pos_f = st.floats(min_value=1e-3, max_value=1e3, allow_nan=False, allow_infinity=False)


@settings(max_examples=300, deadline=None)
@given(a=st.floats(min_value=0.0, max_value=1e4, allow_nan=False), b=pos_f)
def test_floor_divide_scalar_equals_numpy(a, b):
    assert O.floor_divide_scalar(a, b) == float(np.floor_divide(np.float64(a), np.float64(b)))


@settings(max_examples=300, deadline=None)
@given(q=st.integers(min_value=0, max_value=5000), b=pos_f, ulps=st.integers(min_value=-3, max_value=3))
def test_near_tie_floor_divide_is_exact_floor_of_real_quotient(q, b, ulps):
    """What the kernel computes (floor of the exact quotient via a sign-exact fma residual) is np.floor_divide."""
    from fractions import Fraction
    a = float(q) * b
    for _ in range(abs(ulps)):
        a = math.nextafter(a, math.inf if ulps > 0 else -math.inf)
    if a < 0:
        return
    exact = Fraction(a) / Fraction(b)
    assert float(np.floor_divide(np.float64(a), np.float64(b))) == float(math.floor(exact))
    # kernel recipe: low-biased reciprocal estimate, fma residual, one-sided correction
    inv = (1.0 / b) * float.fromhex("0x1.ffffffffffffcp-1")
    est = math.floor(a * inv)
    r = float(Fraction(a) - Fraction(est) * Fraction(b))        # the fma residual is exact for est in {floor, floor-1}
    if r >= b:
        est += 1
    assert est == math.floor(exact)


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(min_value=0, max_value=2**31 - 1), cp=st.floats(0.05, 1.0), cn=st.floats(0.05, 1.0))
def test_esim_oracle_invariants(seed, cp, cn):
    video = O.synth_clip_s1(6, 8, 8, seed=seed % 1000, dtype=np.uint8)
    np.random.seed(seed)
    out, on, off = O.esim_video_to_voxel(video, cp, cn, 0.0, 0.0, 0.0, use_lut=True, return_polarity=True)
    assert np.array_equal(out, on - off) and (on >= 0).all() and (off >= 0).all() and not (on * off).any()
    # time reversal symmetry of the log differences: total signed threshold mass tracks the log change within one step
    lut = O.load_luts()["lut64"]
    resid = (lut[video[-1]] - lut[video[0]]) - (on.sum(0) * cp - off.sum(0) * cn)
    assert np.all(np.abs(resid) < cp + cn + 1e-12)
    # a still video emits nothing once the initial potential is inside (-C-, C+)
    np.random.seed(seed)
    still = O.esim_video_to_voxel(np.repeat(video[:1], 5, axis=0), cp, cn, 0.0, 0.0, 0.0, use_lut=True)
    assert not still.any()
