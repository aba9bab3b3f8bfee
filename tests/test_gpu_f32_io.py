"""fp32-storage entry points (pnx_curvefit_batch_f32 / pnx_nnls_solve_f32): fp64 arithmetic on float32 data.  The
contract: the result equals the fp64 entry point applied to the float32 inputs widened to float64 (what the reference
computes for a float32 image), rounded to float32."""
from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(n_vox=3000 + 5):
    from pyneapple_amd import synth

    b, y, P = synth.make_numpy("tri_reduced", n_vox, 32, sigma=0.01, seed=21)
    names, p0, lo, hi = synth.shared_arrays("tri_reduced")
    return b, y, p0, lo, hi


def test_curvefit_f32_host_equals_f64_on_widened_inputs(gpu, monkeypatch):
    b, y, p0, lo, hi = _case()
    monkeypatch.setenv("PNX_HOST_CHUNK", "1024")  # several ragged chunks through the ring
    y32 = y.astype(np.float32)
    r32 = gpu.curvefit("tri_reduced", b, y32, p0, lo, hi)
    w = lambda a: np.asarray(a, np.float32).astype(np.float64)
    r64 = gpu.curvefit("tri_reduced", w(b), w(y), w(p0), w(lo), w(hi))
    assert r32["popt"].dtype == np.float32 and r32["pcov"].dtype == np.float32 and r32["cost"].dtype == np.float32
    np.testing.assert_array_equal(r32["status"], r64["status"])
    np.testing.assert_array_equal(r32["nfev"], r64["nfev"])
    np.testing.assert_array_equal(r32["popt"], r64["popt"].astype(np.float32))
    np.testing.assert_array_equal(r32["pcov"], r64["pcov"].astype(np.float32))
    np.testing.assert_array_equal(r32["cost"], r64["cost"].astype(np.float32))
    # and it is the fit of the original data to within the float32 rounding of the signal (1 % noise dominates)
    full = gpu.curvefit("tri_reduced", b, y, p0, lo, hi)
    e = np.abs(r32["popt"] - full["popt"]) / np.abs(full["popt"])
    assert np.median(e) < 1e-5


def test_curvefit_f32_per_voxel_and_fixed(gpu):
    from pyneapple_amd import synth

    n_vox = 2000 + 3
    b, y, P = synth.make_numpy("bi_reduced", n_vox, 24, sigma=0.01, seed=5)
    names, p0s, los, his = synth.shared_arrays("bi_reduced")
    rng = np.random.default_rng(1)
    p0 = (np.tile(p0s[[0, 2], None], (1, n_vox)) * rng.uniform(0.9, 1.1, (2, n_vox))).astype(np.float32)
    lo = np.tile(los[[0, 2], None], (1, n_vox)).astype(np.float32)
    hi = np.tile(his[[0, 2], None], (1, n_vox)).astype(np.float32)
    fixed = P["D1"][None, :].astype(np.float32)
    y32 = y.astype(np.float32)
    kw = dict(fixed_idx=[1], jac="analytic")
    r32 = gpu.curvefit("bi_reduced", b, y32, p0, lo, hi, fixed_vals=fixed, **kw)
    d = lambda a: a.astype(np.float64)
    r64 = gpu.curvefit("bi_reduced", d(b.astype(np.float32)), d(y32), d(p0), d(lo), d(hi), fixed_vals=d(fixed), **kw)
    np.testing.assert_array_equal(r32["status"], r64["status"])
    np.testing.assert_array_equal(r32["popt"], r64["popt"].astype(np.float32))


def test_curvefit_f32_device_pointers(gpu):
    import torch

    b, y, p0, lo, hi = _case(4096)
    dev = torch.device("cuda", 0)
    n, n_vox = 5, y.shape[0]
    yt = torch.tensor(y, dtype=torch.float32, device=dev)
    popt = torch.empty((n, n_vox), dtype=torch.float32, device=dev)
    pcov = torch.empty((n_vox, n, n), dtype=torch.float32, device=dev)
    cost = torch.empty(n_vox, dtype=torch.float32, device=dev)
    status = torch.empty(n_vox, dtype=torch.int8, device=dev)
    nfev = torch.empty(n_vox, dtype=torch.int32, device=dev)
    o = gpu.make_opts("tri_reduced", 32, [], False, False, 250, 1e-8, 1e-8, 1e-8, "fd", 0, 0.0, 0.0)
    gpu.curvefit_device(o, n_vox, b, yt, p0, lo, hi, None, popt, pcov, status, nfev, cost, 0,
                        torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    host = gpu.curvefit("tri_reduced", b, y.astype(np.float32), p0, lo, hi)
    np.testing.assert_array_equal(popt.cpu().numpy(), host["popt"])
    np.testing.assert_array_equal(pcov.cpu().numpy(), host["pcov"])
    np.testing.assert_array_equal(status.cpu().numpy(), host["status"])
    np.testing.assert_array_equal(cost.cpu().numpy(), host["cost"])


def test_nnls_f32_host_and_device(gpu, monkeypatch):
    import torch

    from pyneapple_amd import synth

    _, basis, reg = synth.nnls_matrices(32)
    _, y, _ = synth.make_numpy("tri_reduced", 2000 + 9, 32, sigma=0.01, seed=3, scale=1000.0)
    y32 = y.astype(np.float32)
    plan = gpu.NnlsPlan(basis, reg, 0)
    monkeypatch.setenv("PNX_NNLS_HOST_CHUNK", "1024")
    r32 = plan.solve(y32, 250)
    r64 = plan.solve(y32.astype(np.float64), 250)
    assert r32["coefficients"].dtype == np.float32 and r32["residual"].dtype == np.float32
    np.testing.assert_array_equal(r32["status"], r64["status"])
    np.testing.assert_array_equal(r32["iters"], r64["iters"])
    np.testing.assert_array_equal(r32["coefficients"], r64["coefficients"].astype(np.float32))
    np.testing.assert_array_equal(r32["residual"], r64["residual"].astype(np.float32))
    dev = torch.device("cuda", 0)
    n_vox = y.shape[0]
    yt = torch.tensor(y32, device=dev)
    c = torch.empty((n_vox, 250), dtype=torch.float32, device=dev)
    rn = torch.empty(n_vox, dtype=torch.float32, device=dev)
    st = torch.empty(n_vox, dtype=torch.int8, device=dev)
    it = torch.empty(n_vox, dtype=torch.int32, device=dev)
    plan.solve_device(n_vox, yt, 250, c, rn, st, it, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    plan.close()
    np.testing.assert_array_equal(c.cpu().numpy(), r32["coefficients"])
    np.testing.assert_array_equal(it.cpu().numpy(), r32["iters"])


def test_solver_io_dtype_float32(gpu):
    from pyneapple_amd.models import BiExpModel, NNLSModel
    from pyneapple_amd.solvers import HipCurveFitSolver, HipNNLSSolver

    b = np.linspace(0, 1200, 24)
    rng = np.random.default_rng(0)
    f1, D1, D2 = rng.uniform(0.1, 0.4, 200), rng.uniform(5e-3, 5e-2, 200), rng.uniform(5e-4, 2e-3, 200)
    y = (f1[:, None] * np.exp(-b * D1[:, None]) + (1 - f1[:, None]) * np.exp(-b * D2[:, None])).astype(np.float32)
    kw = dict(model=BiExpModel(), max_iter=250, tol=1e-8, p0={"f1": 0.2, "D1": 0.01, "D2": 0.001},
              bounds={"f1": (0.0, 1.0), "D1": (1e-3, 0.1), "D2": (1e-5, 5e-3)})
    s32 = HipCurveFitSolver(io_dtype="float32", **kw).fit(b, y)
    s64 = HipCurveFitSolver(**kw).fit(b, y)  # default: float64 copy of the float32 image, like the reference
    assert s32.params_["D1"].dtype == np.float32 and s64.params_["D1"].dtype == np.float64
    # start values / bounds are float32 in the first, float64 in the second: same optimum, not the same bits
    np.testing.assert_allclose(s32.params_["D1"], s64.params_["D1"], rtol=2e-4)
    np.testing.assert_allclose(s32.params_["f1"], f1, rtol=2e-3)
    with pytest.raises(ValueError):
        HipCurveFitSolver(io_dtype="float16", **kw)
    n = HipNNLSSolver(model=NNLSModel(d_range=(1e-4, 0.1), n_bins=50), reg_order=2, mu=0.02, io_dtype="float32")
    n.fit(b, y * 1000)
    assert n.params_["coefficients"].dtype == np.float32 and n.params_["coefficients"].shape == (200, 50)
