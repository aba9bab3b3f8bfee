"""ASan + UBSan over the CPU restatement (sanitizers on the CPU build only).  `make -C oracle
sanitize` compiles oracle/*.c with -fsanitize=address,undefined and runs oracle/selftest.c: every model, shared and
per-voxel start values, a fixed parameter, the T1 factor, the failure sentinels, NNLS with every regulariser."""
from __future__ import annotations

import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_oracle_clean_under_asan_ubsan():
    if not shutil.which("gcc") or not shutil.which("make"):
        pytest.skip("gcc / make not available")
    r = subprocess.run(["make", "-C", os.path.join(HERE, "..", "oracle"), "sanitize"], capture_output=True, text=True,
                       timeout=600)
    if r.returncode and ("cannot find -lasan" in r.stderr or "libasan" in r.stderr and "No such file" in r.stderr):
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "oracle selftest ok" in r.stdout
