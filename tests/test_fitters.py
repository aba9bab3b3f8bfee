"""HipPixelWiseFitter: mask handling, result assembly and R^2 (modelled on the reference's
tests/test_fitter_pixelwise.py:125-163,207-303)."""
from __future__ import annotations

import numpy as np
import pytest

from pyneapple_amd.fitters import HipPixelWiseFitter, r_squared_from_ss
from pyneapple_amd.models import BiExpModel, MonoExpModel, NNLSModel

B8 = np.array([0, 50, 100, 200, 400, 600, 800, 1000], float)


def test_r_squared_from_ss():
    y = np.array([[1.0, 2.0, 3.0], [2.0, 2.0, 2.0]])
    r2 = r_squared_from_ss(np.array([0.5, 0.0]), y)
    assert r2[0] == pytest.approx(0.75) and np.isnan(r2[1])  # constant signal -> NaN (fitters/base.py:182)


@pytest.mark.gpu
class TestOnGpu:
    def _image(self):
        rng = np.random.default_rng(0)
        S0 = rng.uniform(500, 1500, (6, 5, 2))
        D = rng.uniform(5e-4, 3e-3, (6, 5, 2))
        return S0, D, S0[..., None] * np.exp(-B8 * D[..., None])

    def test_mask_recovery_maps_and_r2(self, gpu):
        from pyneapple_amd.solvers import HipCurveFitSolver

        S0, D, img = self._image()
        seg = np.zeros((6, 5, 2), int)
        seg[1:4, :, 0] = 1
        solver = HipCurveFitSolver(model=MonoExpModel(), max_iter=250, tol=1e-8, p0={"S0": 1000.0, "D": 1e-3},
                                   bounds={"S0": (1.0, 5000.0), "D": (1e-5, 0.1)})
        f = HipPixelWiseFitter(solver).fit(B8, img, segmentation=seg)
        r = f.results_
        assert r.n_pixels == 15 and r.pixel_indices.shape == (15, 3) and r.success.all()
        np.testing.assert_allclose(r.params["D"], D[seg != 0], rtol=1e-3)
        assert r.covariance.shape == (15, 2, 2) and r.mean_r_squared > 0.999999 and r.convergence_rate == 1.0
        maps = f.parameter_maps()
        assert maps["S0"].shape == (6, 5, 2) and maps["S0"].dtype == np.float32
        assert (maps["S0"][seg == 0] == 0).all() and np.allclose(maps["S0"][seg != 0], S0[seg != 0], rtol=1e-3)
        pred = f.predict(B8)  # (X, Y, Z, N) volume, zeros outside the mask (fitters/base.py:90-131)
        assert pred.shape == img.shape and (pred[seg == 0] == 0).all()
        np.testing.assert_allclose(pred[seg != 0], img[seg != 0], rtol=1e-5)
        np.testing.assert_allclose(f.predict_pixels(B8), img[seg != 0], rtol=1e-5)
        assert f.get_fitted_params() is f.fitted_params_
        assert r.solver_name == "HipCurveFitSolver" and r.model_name == "MonoExpModel"

    def test_fixed_param_maps_and_failures(self, gpu):
        from pyneapple_amd.solvers import HipCurveFitSolver

        b = np.linspace(0, 1200, 24)
        f1, D1, D2 = 0.3, 0.02, 1e-3
        img = np.tile(f1 * np.exp(-b * D1) + (1 - f1) * np.exp(-b * D2), (4, 3, 1, 1))
        img[0, 0, 0, 3] = np.nan
        solver = HipCurveFitSolver(model=BiExpModel(), max_iter=250, tol=1e-8, p0={"f1": 0.2, "D1": 0.01, "D2": 0.001},
                                   bounds={"f1": (0.0, 1.0), "D1": (1e-3, 0.1), "D2": (1e-5, 5e-3)})
        f = HipPixelWiseFitter(solver).fit(b, img, fixed_param_maps={"D1": np.full((4, 3, 1), D1)})
        r = f.results_
        assert set(r.params) == {"f1", "D2"} and r.n_converged == 11 and not r.success[0]
        assert r.messages[0] and r.messages[1] is None and np.isnan(r.r_squared[0])
        np.testing.assert_allclose(r.params["f1"][1:], f1, rtol=1e-6)
        with pytest.raises(ValueError):
            HipPixelWiseFitter(solver).fit(b, img, fixed_param_maps={"nope": np.zeros((4, 3, 1))})

    def test_nnls_assembly(self, gpu):
        from pyneapple_amd.solvers import HipNNLSSolver

        b = np.linspace(0, 1000, 16)
        img = np.tile((0.5 * np.exp(-b * 0.03) + 0.5 * np.exp(-b * 0.002)) * 1000, (3, 2, 1, 1))
        solver = HipNNLSSolver(model=NNLSModel(d_range=(1e-4, 0.1), n_bins=50), reg_order=2, mu=0.02)
        f = HipPixelWiseFitter(solver).fit(b, img)
        r = f.results_
        assert r.params["coefficients"].shape == (6, 50) and r.covariance is None and r.residuals.shape == (6,)
        assert r.success.all() and r.mean_r_squared > 0.999
        assert f.parameter_maps()["coefficients"].shape == (3, 2, 1, 50)
