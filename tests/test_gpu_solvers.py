"""The plugin classes end to end on the GPU, written like the reference's own solver tests
(tests/test_solver_curvefit.py:284-313,395-408,567-576,689-809,849-952; tests/test_solver_nnls.py:414-560)."""
from __future__ import annotations

import numpy as np
import pytest

from pyneapple_amd.models import BiExpModel, MonoExpModel, NNLSModel
from pyneapple_amd.solvers import HipCurveFitSolver, HipNNLSSolver

pytestmark = pytest.mark.gpu
B8 = np.array([0, 50, 100, 200, 400, 600, 800, 1000], float)


def mono_solver(**kw):
    return HipCurveFitSolver(model=MonoExpModel(), max_iter=250, tol=1e-8, p0={"S0": 1000.0, "D": 1e-3},
                             bounds={"S0": (1.0, 5000.0), "D": (1e-5, 0.1)}, **kw)


def bi_solver(model=None, **kw):
    return HipCurveFitSolver(model=model or BiExpModel(), max_iter=250, tol=1e-8,
                             p0={"f1": 0.2, "D1": 0.01, "D2": 0.001},
                             bounds={"f1": (0.0, 1.0), "D1": (1e-3, 0.1), "D2": (1e-5, 5e-3)}, **kw)


class TestCurveFit:
    def test_single_pixel_recovery(self, gpu):
        s = mono_solver().fit(B8, 1200.0 * np.exp(-B8 * 1.5e-3))
        assert isinstance(s.params_["S0"], list) and len(s.params_["S0"]) == 1
        assert s.params_["S0"][0] == pytest.approx(1200.0, rel=1e-3) and s.params_["D"][0] == pytest.approx(1.5e-3, rel=1e-3)
        assert s.diagnostics_["pcov"].shape == (2, 2) and s.diagnostics_["n_pixels"] == 1
        assert len(s.pixel_results_) == 1 and s.pixel_results_[0].success

    def test_multi_voxel_recovery_and_shapes(self, gpu):
        rng = np.random.default_rng(0)
        S0, D = rng.uniform(500, 1500, 40), rng.uniform(5e-4, 3e-3, 40)
        y = S0[:, None] * np.exp(-B8[None, :] * D[:, None])
        s = mono_solver().fit(B8, y)
        np.testing.assert_allclose(s.params_["S0"], S0, rtol=1e-2)
        np.testing.assert_allclose(s.params_["D"], D, rtol=1e-2)
        assert s.diagnostics_["pcov"].shape == (40, 2, 2) and s.get_diagnostics()["n_pixels"] == 40
        assert len(s.pixel_results_) == 40 and s.pixel_results_[3].params.shape == (2,)
        p = s.get_params()
        p["S0"] = None
        assert s.params_["S0"] is not None  # get_params returns a copy

    def test_biexp_recovery(self, gpu):
        rng = np.random.default_rng(1)
        f1, D1, D2 = rng.uniform(0.1, 0.4, 30), rng.uniform(5e-3, 5e-2, 30), rng.uniform(5e-4, 2e-3, 30)
        b = np.linspace(0, 1200, 24)
        y = f1[:, None] * np.exp(-b * D1[:, None]) + (1 - f1)[:, None] * np.exp(-b * D2[:, None])
        s = bi_solver().fit(b, y)
        np.testing.assert_allclose(s.params_["f1"], f1, rtol=1e-2)
        np.testing.assert_allclose(s.params_["D2"], D2, rtol=1e-2)

    def test_nan_pixel_fails_without_raising(self, gpu):
        y = np.tile(1000.0 * np.exp(-B8 * 1e-3), (3, 1))
        y[1, 2] = np.nan
        s = mono_solver().fit(B8, y)
        pr = s.pixel_results_[1]
        assert not pr.success and pr.message and np.isnan(pr.covariance).all()
        np.testing.assert_array_equal(pr.params, [1000.0, 1e-3])  # p0 returned
        assert s.pixel_results_[0].success and s.pixel_results_[2].success

    def test_refit_resets_state_and_per_call_overrides(self, gpu):
        s = mono_solver()
        y = np.tile(1000.0 * np.exp(-B8 * 1e-3), (5, 1))
        s.fit(B8, y)
        s.fit(B8, y[:2], p0={"S0": 900.0, "D": 2e-3}, bounds={"S0": (10.0, 4000.0), "D": (1e-5, 0.05)})
        assert s.diagnostics_["n_pixels"] == 2 and len(s.params_["D"]) == 2
        arr = np.tile(np.array([[900.0], [2e-3]]), (1, 2))
        s.fit(B8, y[:2], p0=arr, bounds=(np.tile([[1.0], [1e-5]], (1, 2)), np.tile([[5000.0], [0.1]], (1, 2))))
        np.testing.assert_allclose(s.params_["D"], 1e-3, rtol=1e-6)

    def test_scalar_and_per_pixel_fixed_params(self, gpu):
        b = np.linspace(0, 1200, 24)
        f1, D1, D2 = 0.3, 0.02, 1e-3
        y = np.tile(f1 * np.exp(-b * D1) + (1 - f1) * np.exp(-b * D2), (6, 1))
        s = HipCurveFitSolver(model=BiExpModel(fixed_params={"D1": D1}), max_iter=250, tol=1e-8,
                              p0={"f1": 0.2, "D2": 0.001}, bounds={"f1": (0.0, 1.0), "D2": (1e-5, 5e-3)})
        s.fit(b, y)
        assert set(s.params_) == {"f1", "D2"}
        np.testing.assert_allclose(s.params_["f1"], f1, rtol=1e-6)
        s2 = bi_solver().fit(b, y, pixel_fixed_params={"D1": np.full(6, D1)})
        assert set(s2.params_) == {"f1", "D2"} and s2.diagnostics_["pcov"].shape == (6, 2, 2)
        np.testing.assert_allclose(s2.params_["D2"], D2, rtol=1e-6)

    def test_matches_oracle_through_the_plugin(self, gpu, oracle):
        from pyneapple_amd import synth

        b, y, _ = synth.make_numpy("bi_reduced", 500, 24, sigma=0.02, seed=5)
        names, p0, lo, hi = synth.shared_arrays("bi_reduced")
        s = bi_solver().fit(b, y)
        o = oracle.curvefit("bi_reduced", b, y, p0, lo, hi)
        got = np.stack([s.params_[n] for n in names])
        assert (np.abs(got - o["popt"]) <= 1e-4 * np.abs(o["popt"])).all()


class TestNNLS:
    def _solver(self, **kw):
        return HipNNLSSolver(model=NNLSModel(d_range=(1e-4, 0.1), n_bins=50), **kw)

    def test_non_negative_and_shapes(self, gpu):
        b = np.array([0, 5, 10, 20, 30, 40, 50, 75, 100, 150, 200, 250, 350, 450, 550, 650], float)
        y = np.tile(0.3 * np.exp(-b * 0.05) + 0.7 * np.exp(-b * 0.001), (4, 1))
        s = self._solver(reg_order=2, mu=0.02).fit(b, y)
        c = s.params_["coefficients"]
        assert c.shape == (4, 50) and (c >= 0).all() and s.diagnostics_["residual"].shape == (4,)
        assert len(s.pixel_results_) == 4 and s.pixel_results_[0].covariance is None
        assert s.pixel_results_[0].residual == pytest.approx(s.diagnostics_["residual"][0])
        s1 = self._solver(reg_order=0).fit(b, y[0])
        assert s1.params_["coefficients"].shape == (1, 50)

    def test_peak_within_a_decade_of_true_D(self, gpu):
        b = np.linspace(0, 1000, 16)
        s = self._solver(reg_order=0).fit(b, np.exp(-b * 2e-3)[None, :])
        peak = s.model.bins[int(np.argmax(s.params_["coefficients"][0]))]
        assert 2e-4 < peak < 2e-2

    def test_failure_path(self, gpu):
        b = np.linspace(0, 1000, 16)
        y = (0.5 * np.exp(-b * 0.03) + 0.5 * np.exp(-b * 0.002))[None, :] * 1000
        s = self._solver(reg_order=2, mu=0.02, max_iter=3).fit(b, y)
        pr = s.pixel_results_[0]
        assert not pr.success and pr.message and (pr.params == 0).all()
        assert pr.residual == pytest.approx(np.linalg.norm(y[0]))

    def test_regularisation_smooths(self, gpu):
        b = np.linspace(0, 1000, 16)
        y = (0.5 * np.exp(-b * 0.03) + 0.5 * np.exp(-b * 0.002))[None, :]
        tv = lambda c: np.abs(np.diff(c)).sum()
        c0 = self._solver(reg_order=0).fit(b, y).params_["coefficients"][0]
        c2 = self._solver(reg_order=2, mu=0.5).fit(b, y).params_["coefficients"][0]
        assert tv(c2) < tv(c0)


def test_t1_model_with_fixed_t1_map_through_the_plugin(gpu):
    """MonoExp + T1 relaxation factor, T1 supplied as a per-pixel fixed map (reference fixture g6_mono_t1_fixed)."""
    from conftest import load_golden

    d = load_golden("g6_mono_t1_fixed")
    model = MonoExpModel(fit_t1=True, repetition_time=3000.0)
    s = HipCurveFitSolver(model=model, max_iter=250, tol=1e-8, p0={"S0": 1000.0, "D": 1e-3, "T1": 1000.0},
                          bounds={"S0": (1.0, 5000.0), "D": (1e-5, 0.1), "T1": (100.0, 5000.0)})
    s.fit(d["bvalues"], d["y"], pixel_fixed_params={"T1": d["fixed_T1"]})
    assert set(s.params_) == {"S0", "D"}
    got = np.stack([s.params_["S0"], s.params_["D"]], axis=1)
    assert (np.abs(got - d["popt"]) <= 1e-8 * np.abs(d["popt"])).all()


def test_xtol_gtol_solver_kwargs_reach_the_kernel(gpu, oracle):
    """`[Fitting.solver] xtol = ...` arrives as a constructor kwarg and the reference forwards it to curve_fit
    (curvefit.py:295-306): a loose xtol / gtol must change the termination exactly as it does in the oracle."""
    from pyneapple_amd import synth

    b, y, _ = synth.make_numpy("bi_reduced", 2000, 24, sigma=0.01, seed=3)
    names, p0, lo, hi = synth.shared_arrays("bi_reduced")
    default = bi_solver().fit(b, y)
    for kw in (dict(xtol=1e-3), dict(gtol=1e-4), dict(xtol=1e-4, gtol=1e-5)):
        s = bi_solver(**kw).fit(b, y)
        ref = oracle.curvefit("bi_reduced", b, y, p0, lo, hi, max_nfev=250, ftol=1e-8, jac="fd", **kw)
        st = np.asarray(s.diagnostics_["status"])
        assert (st == ref["status"]).mean() > 0.995
        assert (np.asarray(s.diagnostics_["nfev"]) == ref["nfev"]).mean() > 0.99
        assert (st != np.asarray(default.diagnostics_["status"])).mean() > 0.05  # the tolerance really took effect
        got = np.stack([s.params_[n] for n in names])
        rel = np.abs(got - ref["popt"]) / np.maximum(np.abs(ref["popt"]), 1e-300)
        assert (rel.max(axis=0) <= 1e-4).mean() >= 0.99


class TestReferenceContractOddsAndEnds:
    """The remaining behaviours the reference's solver tests pin (tests/test_solver_curvefit.py:597-640,811-840;
    tests/test_solver_nnls.py:463-560,663-735)."""

    def test_fit_returns_self_and_key_order(self, gpu):
        s = mono_solver()
        assert s.fit(B8, 1000.0 * np.exp(-B8 * 1e-3)) is s
        assert list(s.get_params().keys()) == s.model.param_names

    def test_identical_voxels_give_identical_results(self, gpu):
        y = np.tile(1000.0 * np.exp(-B8 * 1e-3), (5, 1))
        s = mono_solver().fit(B8, y)
        assert np.ptp(s.params_["S0"]) == 0.0 and np.ptp(s.params_["D"]) == 0.0  # same lanes' arithmetic, bit for bit

    @pytest.mark.parametrize("noise_std", [1.0, 10.0, 50.0])
    def test_noisy_data_within_tolerance(self, gpu, noise_std):
        rng = np.random.default_rng(42)
        y = 1000.0 * np.exp(-B8 * 1e-3) + rng.normal(0, noise_std, B8.size)
        s = mono_solver().fit(B8, y)
        assert s.params_["S0"][0] == pytest.approx(1000.0, rel=noise_std / 1000.0 * 20) and s.params_["D"][0] > 0

    def test_diagnostics_are_copies(self, gpu):
        s = mono_solver().fit(B8, 1000.0 * np.exp(-B8 * 1e-3))
        d = s.get_diagnostics()
        d["n_pixels"] = 99
        assert s.diagnostics_["n_pixels"] == 1

    @pytest.mark.parametrize("fit_reduced,fit_s0,names", [(False, False, ["f1", "D1", "f2", "D2"]),
                                                          (True, False, ["f1", "D1", "D2"]),
                                                          (True, True, ["f1", "D1", "D2", "S0"])])
    def test_biexp_modes_param_count(self, gpu, fit_reduced, fit_s0, names):
        m = BiExpModel(fit_reduced=fit_reduced, fit_s0=fit_s0)
        assert m.param_names == names
        b = np.linspace(0, 1200, 24)
        y = 0.3 * np.exp(-b * 0.02) + 0.7 * np.exp(-b * 1e-3)
        p0 = {"f1": 0.2, "D1": 0.01, "f2": 0.8, "D2": 0.001, "S0": 1.0}
        bd = {"f1": (0.0, 1.0), "D1": (1e-3, 0.1), "f2": (0.0, 1.0), "D2": (1e-5, 5e-3), "S0": (0.1, 10.0)}
        s = HipCurveFitSolver(model=m, max_iter=250, tol=1e-8, p0={k: p0[k] for k in names},
                              bounds={k: bd[k] for k in names}).fit(b, y)
        assert list(s.params_) == names and s.diagnostics_["pcov"].shape == (len(names), len(names))
        assert s.params_["D2"][0] == pytest.approx(1e-3, rel=1e-3)

    @pytest.mark.parametrize("reg_order", [0, 1, 2, 3])
    def test_nnls_completes_for_all_reg_orders(self, gpu, reg_order):
        b = np.array([0, 5, 10, 20, 30, 40, 50, 75, 100, 150, 200, 250, 350, 450, 550, 650], float)
        y = 0.3 * np.exp(-b * 0.05) + 0.7 * np.exp(-b * 0.001)
        s = HipNNLSSolver(model=NNLSModel(d_range=(1e-4, 0.1), n_bins=50), reg_order=reg_order, mu=0.02)
        assert s.fit(b, y) is s
        c = s.params_["coefficients"]
        assert c.shape == (1, 50) and (c >= 0).all() and s.diagnostics_["residual"].shape == (1,)

    @pytest.mark.parametrize("reg_order,n_bins", [(0, 300), (2, 300), (2, 512), (1, 400)])
    def test_nnls_plugin_with_more_than_256_bins_matches_the_golden_reference(self, gpu, reg_order, n_bins):
        """The reference accepts any n_bins (models/nnls.py:37-77).  Through the plugin class, against fixtures the reference
        itself produced (oracle/gen_golden.py g11) where one exists for the shape, else shapes / KKT only."""
        from conftest import GOLDEN, load_golden
        import os

        name = {(2, 300): "g11_nnls_300_r2", (2, 512): "g11_nnls_512_r2", (0, 400): "g11_nnls_400_r0"}.get((reg_order, n_bins))
        if name and os.path.exists(os.path.join(GOLDEN, name + ".npz")):
            d = load_golden(name)
            s = HipNNLSSolver(model=NNLSModel(d_range=tuple(d["d_range"]), n_bins=n_bins), reg_order=reg_order, mu=float(d["mu"]),
                              max_iter=int(d["max_iter"])).fit(d["bvalues"], d["y"])
            c = s.params_["coefficients"]
            ok = d["success"]
            scale = np.abs(d["coefficients"]).max(axis=1, keepdims=True) + 1e-300
            assert (np.abs(c - d["coefficients"]) / scale).max() < 1e-6
            np.testing.assert_allclose(s.diagnostics_["residual"], d["residual"], rtol=1e-10)
            assert [r.success for r in s.pixel_results_] == list(ok)
        else:
            b = np.linspace(0, 1200, 32)
            y = np.tile(300 * np.exp(-b * 0.05) + 700 * np.exp(-b * 0.001), (5, 1))
            s = HipNNLSSolver(model=NNLSModel(d_range=(0.0008, 0.5), n_bins=n_bins), reg_order=reg_order, mu=0.02).fit(b, y)
            c = s.params_["coefficients"]
            assert c.shape == (5, n_bins) and (c >= 0).all() and (c[0] == c[4]).all() and c.sum() > 0

    def test_nnls_results_are_copies_and_refit_resets(self, gpu):
        b = np.linspace(0, 1000, 16)
        y = np.tile(np.exp(-b * 2e-3), (3, 1))
        s = HipNNLSSolver(model=NNLSModel(d_range=(1e-4, 0.1), n_bins=50), reg_order=2, mu=0.02).fit(b, y)
        p = s.get_params()
        p["coefficients"] = np.zeros((1, 1))
        assert s.params_["coefficients"].shape == (3, 50)
        r0 = s.diagnostics_["residual"].copy()
        d = s.get_diagnostics()
        d["residual"] = np.array([99999.0])
        np.testing.assert_array_equal(s.diagnostics_["residual"], r0)
        s.fit(b, y[:1])
        assert s.params_["coefficients"].shape == (1, 50) and s.diagnostics_["residual"].shape == (1,)
