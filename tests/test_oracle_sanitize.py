"""The scalar C oracle under AddressSanitizer + UBSan (CPU build; GPU sanitizers are unavailable on this pool)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_oracle_clean_under_asan_ubsan():
    res = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "sanitize"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "sanitize ok rc=0 q=19" in res.stdout
