"""Host-side mirror of the reference's solver interface: construction, validation and state conventions
(modelled on the reference's tests/test_solver_curvefit.py:83-261,775-809 and tests/test_solver_nnls.py)."""
from __future__ import annotations

import numpy as np
import pytest

from pyneapple_amd import _lib
from pyneapple_amd._compat import PixelResultsView
from pyneapple_amd.models import BiExpModel, MonoExpModel, NNLSModel, TriExpModel
from pyneapple_amd.solvers import HipCurveFitSolver, HipNNLSSolver, _split, kernel_model_key

B8 = np.array([0, 50, 100, 200, 400, 600, 800, 1000], float)


def mono_solver(**kw):
    return HipCurveFitSolver(model=MonoExpModel(), max_iter=250, tol=1e-8, p0={"S0": 1000.0, "D": 1e-3},
                             bounds={"S0": (1.0, 5000.0), "D": (1e-5, 0.1)}, **kw)


class TestModels:
    def test_param_orders(self):
        assert MonoExpModel().param_names == ["S0", "D"]
        assert BiExpModel().param_names == ["f1", "D1", "D2"]
        assert BiExpModel(fit_s0=True).param_names == ["f1", "D1", "D2", "S0"]
        assert BiExpModel(fit_reduced=False).param_names == ["f1", "D1", "f2", "D2"]
        assert TriExpModel().param_names == ["f1", "D1", "f2", "D2", "D3"]
        assert TriExpModel(fit_s0=True).param_names == ["f1", "D1", "f2", "D2", "D3", "S0"]
        assert TriExpModel(fit_reduced=False).param_names == ["f1", "D1", "f2", "D2", "f3", "D3"]

    def test_s0_requires_reduced(self):
        with pytest.raises(ValueError):
            TriExpModel(fit_reduced=False, fit_s0=True)

    def test_fixed_params_reduce_free_names(self):
        m = BiExpModel(fixed_params={"D1": 0.02})
        assert m.param_names == ["f1", "D2"] and m._free_indices(m.fixed_params) == [0, 2]
        with pytest.raises(ValueError):
            BiExpModel(fixed_params={"nope": 1.0})

    def test_forward_closed_form(self):
        np.testing.assert_allclose(MonoExpModel().forward(B8, 1000.0, 1e-3), 1000 * np.exp(-B8 * 1e-3))
        f = TriExpModel().forward(np.array([0.0]), 0.2, 0.05, 0.3, 0.005, 0.001)
        assert f[0] == pytest.approx(1.0)

    def test_kernel_model_key(self):
        assert kernel_model_key(TriExpModel()) == "tri_reduced"
        assert kernel_model_key(BiExpModel(fit_s0=True)) == "bi_s0"
        from pyneapple_amd.solvers import kernel_t1

        m = MonoExpModel(fit_t1=True, repetition_time=3000.0)
        assert m.param_names == ["S0", "D", "T1"] and kernel_model_key(m) == "mono"
        assert kernel_t1(m) == dict(t1_mode=1, tr=3000.0, tm=0.0)
        ms = BiExpModel(fit_t1_steam=True, repetition_time=3000.0, mixing_time=20.0)
        assert ms.param_names[-1] == "T1" and kernel_t1(ms)["t1_mode"] == 2
        with pytest.raises(ValueError):
            MonoExpModel(fit_t1=True)

    def test_nnls_model(self):
        m = NNLSModel(d_range=(1e-4, 0.1), n_bins=50)
        assert m.bins.shape == (50,) and m.get_basis(B8).shape == (8, 50)
        assert np.all(m.get_basis(B8)[0] == 1.0)  # b = 0 row
        with pytest.raises(ValueError):
            m.get_basis(B8[None, :])


class TestCurveFitSolverHostSide:
    def test_init_stores_configuration(self):
        s = mono_solver()
        assert s.max_iter == 250 and s.tol == 1e-8 and s.method == "trf"
        assert s.p0 == {"S0": 1000.0, "D": 1e-3} and s.params_ == {} and len(s.pixel_results_) == 0

    def test_missing_p0_name_raises(self):
        with pytest.raises(ValueError):
            HipCurveFitSolver(model=MonoExpModel(), max_iter=10, tol=1e-8, p0={"S0": 1.0},
                              bounds={"S0": (0.0, 2.0), "D": (0.0, 1.0)})

    def test_bounds_must_be_tuples(self):
        with pytest.raises(ValueError):
            HipCurveFitSolver(model=MonoExpModel(), max_iter=10, tol=1e-8, p0={"S0": 1.0, "D": 0.1},
                              bounds={"S0": [0.0, 2.0], "D": [0.0, 1.0]})

    def test_only_trf(self):
        with pytest.raises(ValueError):
            mono_solver(method="dogbox")

    def test_results_before_fit_raise_runtime_error(self):
        s = mono_solver()
        with pytest.raises(RuntimeError):
            s.get_params()
        with pytest.raises(RuntimeError):
            s.get_diagnostics()

    def test_shape_validation(self):
        s = mono_solver()
        with pytest.raises(ValueError):
            s.fit(B8[None, :], np.ones((4, 8)))
        with pytest.raises(ValueError):
            s.fit(B8, np.ones((4, 7)))
        with pytest.raises(ValueError):
            s.fit(B8, np.ones((4, 8)), p0={"S0": np.ones(4), "D": np.ones(4)})
        with pytest.raises(ValueError):
            s.fit(B8, np.ones((4, 8)), p0=np.ones((2, 3)))
        with pytest.raises(ValueError):
            s.fit(B8, np.ones((4, 8)), bounds=([0, 0], [1, 1]))

    def test_p0_bounds_preparation(self):
        s = mono_solver()
        p0, lo, hi, pv = s._prepare_p0_bounds(None, None, 5)
        assert not pv and p0.tolist() == [1000.0, 1e-3] and lo.tolist() == [1.0, 1e-5] and hi.tolist() == [5000.0, 0.1]
        p0, lo, hi, pv = s._prepare_p0_bounds({"S0": 900.0, "D": 2e-3}, None, 5)
        assert p0.tolist() == [900.0, 2e-3]
        arr = np.tile(np.array([[800.0], [1e-3]]), (1, 5))
        p0, lo, hi, pv = s._prepare_p0_bounds(arr, None, 5)
        assert pv and p0.shape == (2, 5) and lo.shape == (2, 5) and (lo[0] == 1.0).all()

    def test_solver_kwargs_are_honoured_or_refused(self):
        """Extra [Fitting.solver] keys reach the constructor as kwargs (io/toml.py:328-338); the reference forwards them to
        curve_fit (curvefit.py:295-306).  sigma (scalar or 1-D) / absolute_sigma -- the two its docstring names (curvefit.py:33) --
        and xtol / gtol are implemented, SciPy defaults are accepted, the rest is refused."""
        kw = dict(model=MonoExpModel(), max_iter=250, tol=1e-8, p0={"S0": 1000.0, "D": 1e-3},
                  bounds={"S0": (1.0, 5000.0), "D": (1e-5, 0.1)})
        s = HipCurveFitSolver(**kw, xtol=1e-10, gtol=1e-6, n_pools=4, multi_threading=True, device=0, n_gpus=1,
                              jacobian="fd", io_dtype="float64", absolute_sigma=False, loss="linear", x_scale=1.0)
        assert s.xtol == 1e-10 and s.gtol == 1e-6 and s.sigma is None and s.absolute_sigma is False
        s = HipCurveFitSolver(**kw, sigma=np.ones(8), absolute_sigma=True)
        assert s.sigma.shape == (8,) and s.absolute_sigma is True
        assert HipCurveFitSolver(**kw, sigma=0.5).sigma.tolist() == [0.5]
        for bad in (dict(sigma=np.eye(8)), dict(loss="soft_l1"), dict(x_scale="jac"), dict(typo_key=1)):
            with pytest.raises(ValueError):
                HipCurveFitSolver(**kw, **bad)
        with pytest.raises(ValueError):
            HipCurveFitSolver(**kw, method="dogbox")

    def test_fit_without_gpu_fails_loudly(self):
        if _lib.device_count() > 0:
            pytest.skip("a HIP device is visible here")
        with pytest.raises(_lib.PnxError):
            mono_solver().fit(B8, np.ones((4, 8)))


class TestNNLSSolverHostSide:
    def test_regularized_basis_and_extension(self):
        m = NNLSModel(d_range=(1e-4, 0.1), n_bins=50)
        s = HipNNLSSolver(model=m, reg_order=2, mu=0.02)
        A = s._build_regularized_basis(B8)
        assert A.shape == (8 + 50, 50)
        np.testing.assert_array_equal(A[:8], m.get_basis(B8))
        assert A[8, 0] == -2 * 0.02 and A[8, 1] == 0.02
        ext = s._extend_signal(np.ones((3, 8)))
        assert ext.shape == (3, 58) and (ext[:, 8:] == 0).all()

    def test_defaults(self):
        s = HipNNLSSolver(model=NNLSModel(d_range=(1e-4, 0.1), n_bins=10))
        assert s.reg_order == 0 and s.mu == 0.02 and s.max_iter == 250


def test_pixel_results_view_behaves_like_a_list():
    params = np.arange(12.0).reshape(4, 3)
    cov = np.zeros((4, 3, 3))
    ok = np.array([True, False, True, True])
    v = PixelResultsView(params, cov, ok, lambda i: None if ok[i] else "boom")
    assert len(v) == 4 and v[0].params.shape[0] == 3 and v[-1].success
    assert [p.success for p in v] == ok.tolist() and v[1].message == "boom" and v[1].n_iterations is None
    assert len(v[1:3]) == 2
    with pytest.raises(IndexError):
        v[4]


def test_split_covers_range_without_overlap():
    for n, p in ((10, 3), (4194304, 8), (5, 8), (1, 1)):
        parts = _split(n, p)
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(parts[:-1], parts[1:]))


def test_solver_n_gpus_sharding_reassembles_rows(monkeypatch):
    """`n_gpus > 1` shards voxels over devices inside one process: every device gets a contiguous row range of the
    signal, of per-voxel p0 / bounds and of the fixed maps, and the results come back in voxel order.  The device call
    is replaced by a recorder here (the kernels are covered by the gpu tests; an 8-GPU node is the driver's)."""
    import numpy as np

    from pyneapple_amd import api, solvers
    from pyneapple_amd.models import BiExpModel

    calls = []

    def fake_curvefit(model, b, y, p0, lo, hi, *, fixed_idx=(), fixed_vals=None, device=0, **kw):
        n_vox = y.shape[0]
        n = p0.shape[0]
        calls.append((device, n_vox, p0.shape, None if fixed_vals is None else np.shape(fixed_vals)))
        # encode (device, row content) so that the re-assembly can be checked
        popt = np.tile(y[:, 0][None, :], (n, 1)) + (p0 if p0.ndim == 2 else p0[:, None]) * 0
        if fixed_vals is not None and np.ndim(fixed_vals) == 2:
            popt[0] = fixed_vals[0]
        res = dict(popt=popt, pcov=np.zeros((n_vox, n, n)) + device, status=np.ones(n_vox, np.int8),
                   nfev=np.full(n_vox, device, np.int32), cost=y[:, 1].copy())
        for key, a in (kw.get("out") or {}).items():  # the shards write their voxel-major results into the caller's row ranges
            assert a.shape == res[key].shape and a.flags.c_contiguous
            a[...] = res[key]
        return res

    monkeypatch.setattr(api, "curvefit", fake_curvefit)
    n_vox = 1003
    b = np.linspace(0, 1000, 8)
    y = np.arange(n_vox * 8, dtype=float).reshape(n_vox, 8)
    s = solvers.HipCurveFitSolver(model=BiExpModel(), max_iter=250, tol=1e-8, p0={"f1": 0.2, "D1": 0.01, "D2": 0.001},
                                  bounds={"f1": (0.0, 1.0), "D1": (1e-3, 0.1), "D2": (1e-5, 5e-3)}, n_gpus=4, device=2)
    d1 = np.linspace(0.005, 0.05, n_vox)
    s.fit(b, y, pixel_fixed_params={"D1": d1})
    assert sorted(c[0] for c in calls) == [2, 3, 4, 5] and sum(c[1] for c in calls) == n_vox  # the shards run in threads: any order
    assert all(c[3] == (1, c[1]) for c in calls)  # the fixed map was sliced with the rows
    np.testing.assert_array_equal(s.params_["f1"], d1)            # row 0 carried the fixed map through
    np.testing.assert_array_equal(s.params_["D2"], y[:, 0])       # voxel order preserved
    np.testing.assert_array_equal(s.diagnostics_["cost"], y[:, 1])
    assert sorted(set(s.diagnostics_["nfev"].tolist())) == [2, 3, 4, 5]
    assert s.diagnostics_["pcov"].shape == (n_vox, 2, 2)
    # per-voxel p0 / bounds arrays are sliced the same way
    calls.clear()
    p0 = np.tile(np.array([0.2, 0.01, 0.001])[:, None], (1, n_vox))
    lo = np.tile(np.array([0.0, 1e-3, 1e-5])[:, None], (1, n_vox))
    hi = np.tile(np.array([1.0, 0.1, 5e-3])[:, None], (1, n_vox))
    s.fit(b, y, p0=p0, bounds=(lo, hi))
    assert all(c[2] == (3, c[1]) for c in calls) and len(calls) == 4


def test_result_arrays_handed_in_by_the_caller_are_checked():
    """`out=` of api.curvefit / NnlsPlan.solve (row ranges of a larger array that several device shards fill): the right shape,
    dtype and contiguity, or a ValueError -- never a silent copy."""
    import numpy as np
    import pytest

    from pyneapple_amd import api

    big = np.zeros((10, 3, 3))
    view = big[2:7]
    assert api._out({"pcov": view}, "pcov", (5, 3, 3), np.float64) is view
    fresh = api._out(None, "pcov", (5, 3, 3), np.float64)
    assert fresh.shape == (5, 3, 3) and fresh.dtype == np.float64
    assert api._out({"status": None}, "status", (4,), np.int8).shape == (4,)
    with pytest.raises(ValueError):
        api._out({"pcov": big[::2]}, "pcov", (5, 3, 3), np.float64)           # not contiguous
    with pytest.raises(ValueError):
        api._out({"pcov": view.astype(np.float32)}, "pcov", (5, 3, 3), np.float64)  # wrong dtype
    with pytest.raises(ValueError):
        api._out({"pcov": big[:4]}, "pcov", (5, 3, 3), np.float64)            # wrong shape
