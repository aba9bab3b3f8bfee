"""NNLS spectrum post-processing on the device against the reference's own utility/spectrum.py (fixtures g9_spectrum_*, and
g11_spectrum_* with 300 / 400 / 512 bins:
find_spectrum_peaks + apply_cutoffs run on reference NNLS spectra by oracle/gen_golden.py), plus scipy.signal directly on
adversarial rows (plateaus, peaks at the border, ties)."""
from __future__ import annotations

import glob
import os

import numpy as np
import pytest
from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu
FIXTURES = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "g*_spectrum_*.npz")))


@pytest.mark.parametrize("name", FIXTURES)
def test_peaks_and_cutoffs_match_the_reference(gpu, name):
    d = load_golden(name)
    r = gpu.spectrum_peaks(d["spectrum"], d["bins"], height=float(d["height"]), regularized=bool(d["regularized"]),
                           max_peaks=d["d_values"].shape[1], cutoffs=d["cutoffs"])
    np.testing.assert_array_equal(r["n_peaks"], d["n_peaks"])
    np.testing.assert_array_equal(np.isnan(r["d_values"]), np.isnan(d["d_values"]))
    np.testing.assert_array_equal(np.nan_to_num(r["d_values"]), np.nan_to_num(d["d_values"]))  # bins[peak]: exact
    np.testing.assert_allclose(r["f_values"], d["f_values"], rtol=1e-12, equal_nan=True)
    np.testing.assert_array_equal(np.isnan(r["d_cut"]), np.isnan(d["d_cut"]))
    np.testing.assert_allclose(r["d_cut"], d["d_cut"], rtol=1e-12, equal_nan=True)
    np.testing.assert_allclose(r["f_cut"], d["f_cut"], rtol=1e-12, equal_nan=True)


def test_against_scipy_on_adversarial_rows(gpu):
    from scipy import signal

    rng = np.random.default_rng(0)
    n = 250
    rows = []
    for k in range(300):
        x = np.zeros(n)
        for _ in range(rng.integers(1, 7)):
            c, w, a = rng.integers(0, n), rng.uniform(0.6, 12), rng.uniform(0.05, 50)
            x += a * np.exp(-0.5 * ((np.arange(n) - c) / w) ** 2)
        x[x < 1e-3] = 0.0                      # NNLS spectra are exactly zero between the peaks
        if k % 3 == 0:                         # flat tops: plateaus of equal samples (midpoint rule)
            i = int(np.argmax(x))
            x[max(i - 2, 0):i + 3] = x[i]
        if k % 5 == 0:
            x = np.round(x, 1)                 # many exact ties
        if k % 7 == 0:
            x[0] = x.max() + 1                 # maxima at the border are never peaks
            x[-1] = x.max() + 1
        rows.append(x)
    X = np.array(rows)
    bins = np.logspace(-3, 0, n)
    for height, reg in ((0.1, True), (0.1, False), (2.0, True)):
        r = gpu.spectrum_peaks(X, bins, height=height, regularized=reg, max_peaks=16)
        for i, x in enumerate(X):
            pk, prop = signal.find_peaks(x, height=height)
            assert r["n_peaks"][i] == len(pk)
            k = min(len(pk), 16)
            np.testing.assert_array_equal(r["d_values"][i, :k], bins[pk[:k]])
            f = prop["peak_heights"].copy()
            if reg and len(pk):
                fw = signal.peak_widths(x, pk, rel_height=0.5)[0]
                f = f * fw / (2 * np.sqrt(2 * np.log(2))) * np.sqrt(2 * np.pi)
            if len(pk) and f.sum() > 0 and len(pk) <= 16:
                np.testing.assert_allclose(r["f_values"][i, :k], (f / f.sum())[:k], rtol=1e-12)
            assert np.isnan(r["f_values"][i, k:]).all()


def test_fit_peaks_keeps_the_spectra_on_the_device(gpu):
    """solve + peak analysis in one call == fit, then the reference-style post-processing of the returned spectra."""
    from pyneapple_amd import synth
    from pyneapple_amd.models import NNLSModel
    from pyneapple_amd.solvers import HipNNLSSolver

    b, y, _ = synth.make_numpy("tri_reduced", 3000, 32, sigma=0.01, seed=4, scale=1000.0)
    cuts = [(0.0008, 0.003), (0.003, 0.02), (0.02, 0.5)]
    model = NNLSModel(d_range=(0.0008, 0.5), n_bins=250)
    a = HipNNLSSolver(model=model, reg_order=2, mu=0.02, max_iter=250).fit(b, y)
    ref = gpu.spectrum_peaks(a.params_["coefficients"], model.bins, height=0.1, regularized=True, max_peaks=8, cutoffs=cuts)
    f = HipNNLSSolver(model=model, reg_order=2, mu=0.02, max_iter=250).fit_peaks(b, y, height=0.1, cutoffs=cuts)
    assert "coefficients" not in f.params_
    for k in ("n_peaks", "d_values", "f_values", "d_cut", "f_cut"):
        np.testing.assert_array_equal(f.params_[k], ref[k])
    np.testing.assert_array_equal(f.diagnostics_["residual"], a.diagnostics_["residual"])
    np.testing.assert_array_equal(f.diagnostics_["status"], a.diagnostics_["status"])
    # three compartments recovered as three ranges on (almost) every voxel
    assert (np.isfinite(f.params_["d_cut"]).sum(axis=1) == 3).mean() > 0.5
    has = np.isfinite(f.params_["f_cut"]).any(axis=1)  # a voxel without a peak inside any range stays all-NaN
    assert has.mean() > 0.99
    np.testing.assert_allclose(np.nansum(f.params_["f_cut"], axis=1)[has], 1.0, rtol=1e-12)


def test_parameter_maps_on_device(gpu):
    rng = np.random.default_rng(1)
    shape = (7, 5, 3)
    mask = rng.random(shape) > 0.4
    idx = np.argwhere(mask)
    vals = rng.normal(size=len(idx))
    spec = rng.normal(size=(len(idx), 11))
    m = gpu.scatter_maps(vals, idx, shape)
    assert m.dtype == np.float32 and m.shape == shape and (m[~mask] == 0).all()
    np.testing.assert_array_equal(m[mask], vals.astype(np.float32))
    s = gpu.scatter_maps(spec, idx, shape)
    assert s.shape == shape + (11,) and (s[~mask] == 0).all()
    np.testing.assert_array_equal(s[mask], spec.astype(np.float32))


def test_more_peaks_than_the_table_holds_is_flagged_not_truncated(gpu):
    """The device table holds 64 peaks per spectrum (round 3: 16).  A noisy unregularised spectrum at a low threshold with 20 or
    24 maxima is reported in full, as the reference's find_spectrum_peaks does; one with more than 64 (a 250-bin spectrum has at
    most 124 maxima) gets NaN rows and its count in n_peaks -- never fractions normalised over a part of its peaks."""
    import warnings

    from pyneapple_amd import spectrum

    bins = np.logspace(-3, 0, 250)
    x = np.zeros((4, 250))
    x[0, 10::12] = 1.0  # 20 isolated peaks
    x[1, [30, 90]] = (1.0, 3.0)
    x[2, 5:245:10] = np.linspace(1, 2, 24)  # 24 peaks
    x[3, 2:212:3] = 1.0  # 70 peaks
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        d, f, n = spectrum.find_spectrum_peaks_batch(x, bins, height=0.1, max_peaks=32)
        dc, fc = spectrum.apply_cutoffs_batch(x, bins, [(1e-3, 3e-2), (3e-2, 1.0)], height=0.1)
    assert [int(v) for v in n] == [20, 2, 24, 70] and any("more than 64 peaks" in str(m.message) for m in w)
    np.testing.assert_allclose(d[0, :20], bins[10::12])
    np.testing.assert_allclose(f[0, :20], 1.0 / 20)
    assert np.isnan(d[0, 20:]).all()
    np.testing.assert_allclose(d[2, :24], bins[5:245:10])
    np.testing.assert_allclose(f[2, :24], np.linspace(1, 2, 24) / np.linspace(1, 2, 24).sum())
    np.testing.assert_allclose(f[1, :2], [0.25, 0.75])
    assert np.isnan(d[3]).all() and np.isnan(f[3]).all() and np.isnan(dc[3]).all() and np.isnan(fc[3]).all()
    assert np.isfinite(fc[0]).any() and np.isfinite(fc[1]).any() and np.isfinite(fc[2]).any()
    np.testing.assert_allclose(fc[0].sum(), 1.0)
    with pytest.raises(ValueError, match="70 peaks"):
        spectrum.find_spectrum_peaks(x[3], bins, height=0.1)
    dd, ff = spectrum.find_spectrum_peaks(x[0], bins, height=0.1)  # the reference returns every peak; so does this
    assert dd.shape == (20,)
    np.testing.assert_allclose(ff, 1.0 / 20)
    dd, ff = spectrum.find_spectrum_peaks(x[1], bins, height=0.1)
    np.testing.assert_allclose(ff, [0.25, 0.75])
