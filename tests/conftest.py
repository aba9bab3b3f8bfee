from __future__ import annotations

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` through gpurun)")


def load_golden(name: str):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


# fixture name -> kernel model key
CURVEFIT_FIXTURES = {
    "g1_mono_b16": "mono", "g1_mono_b8": "mono",
    "g2_bi_reduced": "bi_reduced", "g2_bi_s0": "bi_s0", "g2_bi_full": "bi_full",
    "g3_tri_reduced": "tri_reduced", "g3_tri_reduced_maxiter4": "tri_reduced",
    "g3_tri_s0": "tri_s0", "g3_tri_full": "tri_full",
    "g5_bi_pervoxel": "bi_reduced", "g5_tri_pervoxel": "tri_reduced",
}
NNLS_FIXTURES = ["g4_nnls_250_r2", "g4_nnls_250_r1", "g4_nnls_250_r3", "g4_nnls_50_r2", "g4_nnls_50_r0",
                 "g4_nnls_250_r2_maxiter20"]


def golden_p0_bounds(d):
    if "p0_arr" in d.files:
        return d["p0_arr"], d["lo_arr"], d["hi_arr"]
    return d["p0_vals"], d["lo_vals"], d["hi_vals"]


def rel_err(a, b):
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-300)


def pcov_norm_err(pc, pr):
    """max |dpcov_ij| / sqrt(pcov_ii pcov_jj) per voxel (scale-free)."""
    dg = np.sqrt(np.abs(np.einsum("vii->vi", pr)))
    return (np.abs(pc - pr) / (dg[:, :, None] * dg[:, None, :])).max(axis=(1, 2))


@pytest.fixture(scope="session")
def oracle():
    from oracle import pnx_oracle

    pnx_oracle.lib()
    return pnx_oracle


@pytest.fixture(scope="session")
def gpu():
    """The HIP backend; the test is an error (not a skip) if a `gpu`-marked test runs without a device."""
    from pyneapple_amd import _lib, api

    _lib.load()
    assert _lib.device_count() >= 1, "gpu-marked test running without a visible HIP device"
    return api
