"""IDEAL driver: OpenCV-semantics resize pinned by analytic cases (cv2 itself is absent here, so the
restatement is 'parity unpinned' against OpenCV), validation mirrored from the reference's tests/test_fitter_ideal.py,
and an end-to-end 3-level pyramid on the GPU."""
from __future__ import annotations

import numpy as np
import pytest

from pyneapple_amd.ideal import HipIDEALFitter, resize2d, resize_weights
from pyneapple_amd.models import BiExpModel


@pytest.mark.parametrize("method", ["linear", "cubic"])
@pytest.mark.parametrize("n_src,n_dst", [(4, 8), (8, 4), (5, 13), (16, 16), (2, 64), (64, 3)])
def test_resize_weights_partition_of_unity(method, n_src, n_dst):
    W = resize_weights(n_src, n_dst, method)
    assert W.shape == (n_dst, n_src)
    np.testing.assert_allclose(W.sum(axis=1), 1.0, atol=1e-14)  # constants are reproduced exactly
    if n_src == n_dst:
        np.testing.assert_allclose(W, np.eye(n_src), atol=1e-14)  # same size = identity


def test_linear_kernel_reproduces_affine_functions():
    """Bilinear at half-pixel centres is exact for affine functions away from the replicated border."""
    n_src, n_dst = 16, 40
    src = 3.0 + 0.5 * np.arange(n_src)
    out = resize_weights(n_src, n_dst, "linear") @ src
    x = (np.arange(n_dst) + 0.5) * n_src / n_dst - 0.5  # source coordinate of every output sample
    inner = (x > 0) & (x < n_src - 1)
    np.testing.assert_allclose(out[inner], 3.0 + 0.5 * x[inner], atol=1e-12)
    assert out.min() >= src.min() - 1e-12 and out.max() <= src.max() + 1e-12


def test_cubic_kernel_is_opencvs():
    """OpenCV's bicubic uses a = -0.75 (not Catmull-Rom's -0.5): known tap weights, symmetric, interpolating."""
    from pyneapple_amd.ideal import _cubic_coeffs

    np.testing.assert_allclose(_cubic_coeffs(np.array([0.5]))[0], [-0.09375, 0.59375, 0.59375, -0.09375], atol=1e-15)
    np.testing.assert_allclose(_cubic_coeffs(np.array([0.0]))[0], [0, 1, 0, 0], atol=1e-15)
    np.testing.assert_allclose(_cubic_coeffs(np.array([0.25]))[0], _cubic_coeffs(np.array([0.75]))[0][::-1], atol=1e-15)
    # upscaling by 2: output samples sit at +-0.25 of a source pixel; taps come from the replicated border at the edge
    W = resize_weights(4, 8, "cubic")
    c = _cubic_coeffs(np.array([0.75]))[0]
    np.testing.assert_allclose(W[0], [c[0] + c[1] + c[2], c[3], 0, 0], atol=1e-15)  # f = -0.25: taps -2,-1,0 clamp to 0
    np.testing.assert_allclose(W[3], np.r_[_cubic_coeffs(np.array([0.25]))[0]], atol=1e-15)  # f = 1.25: taps 0..3


def test_linear_downscale_by_two_is_block_mean():
    img = np.arange(64.0).reshape(8, 8)
    out = resize2d(img[..., None, None], (4, 4), "linear")[..., 0, 0]
    np.testing.assert_allclose(out, img.reshape(4, 2, 4, 2).mean(axis=(1, 3)))


def test_resize2d_shapes_and_dtypes():
    a = np.random.default_rng(0).uniform(size=(8, 6, 3, 5))
    out = resize2d(a, (16, 12, 3), "cubic")
    assert out.shape == (16, 12, 3, 5) and out.dtype == a.dtype
    seg = np.ones((8, 6, 3, 1), dtype=np.int64)
    s = resize2d(seg, (4, 3, 3), "linear")
    assert s.dtype == np.float32 and np.allclose(s, 1.0)  # integer masks are cast to float32 (ideal.py:309-310)
    with pytest.raises(ValueError):
        resize2d(a, (4, 4), "nearest")


class _FakeSolver:
    """Records the per-voxel arrays each level receives (the reference's test_fitter_ideal.py uses mocks too)."""

    def __init__(self):
        self.model = BiExpModel()
        self.p0 = {"f1": 0.2, "D1": 0.01, "D2": 0.001}
        self.bounds = {"f1": (0.0, 1.0), "D1": (1e-3, 0.1), "D2": (1e-5, 5e-3)}
        self.calls = []
        self.params_ = {}

    def fit(self, xdata, ydata, p0=None, bounds=None, **kw):
        self.calls.append((ydata.shape, p0.copy(), bounds[0].copy(), bounds[1].copy()))
        self.params_ = {n: p0[k] * 1.1 for k, n in enumerate(self.model.param_names)}
        return self


def test_levels_shapes_bounds_and_fallback():
    b = np.linspace(0, 1000, 8)
    image = np.ones((16, 16, 2, 8))
    seg = np.zeros((16, 16, 2), dtype=int)
    seg[0, 0, :] = 1  # vanishes at the coarse levels -> all-voxel fallback there
    s = _FakeSolver()
    f = HipIDEALFitter(s, np.array([[2, 2], [4, 4], [16, 16]]), {"f1": 0.2, "D1": 0.3, "D2": 0.4})
    f.fit(b, image, segmentation=seg)
    assert [c[0] for c in s.calls] == [(2 * 2 * 2, 8), (4 * 4 * 2, 8), (2, 8)]
    assert [p.shape for p in f.step_params] == [(2, 2, 2, 3), (4, 4, 2, 3), (16, 16, 2, 3)]
    p0, lo, hi = s.calls[1][1:]
    assert p0.shape == (3, 32) and np.allclose(p0[0], 0.22)            # level-0 result, resized, clipped
    np.testing.assert_allclose(lo[1], np.clip(p0[1] * 0.7, 1e-3, 0.1))  # p0 * (1 - tol) clipped to global bounds
    np.testing.assert_allclose(hi[2], np.clip(p0[2] * 1.4, 1e-5, 5e-3))
    assert len(f.pixel_indices) == 2 and set(f.fitted_params_) == {"f1", "D1", "D2"}


def test_ideal_fitter_built_the_way_a_toml_config_builds_plugins():
    """io/toml.py:236 builds a plugin fitter as cls(solver=solver): construction must work, fit must name what is missing,
    and assigning the attributes afterwards must be enough."""
    s = _FakeSolver()
    f = HipIDEALFitter(solver=s)
    b = np.linspace(0, 1000, 8)
    img = np.ones((8, 8, 1, 8))
    with pytest.raises(ValueError, match="dim_steps"):
        f.fit(b, img)
    f.dim_steps = [[4, 4], [8, 8]]
    f.step_tol = {"f1": .2, "D1": .2, "D2": .2}
    f.fit(b, img)
    assert f.image_shape == (8, 8, 1, 8) and f.n_measurements == 8 and f.results_.n_pixels == 64
    assert f.predict(b).shape == (8, 8, 1, 8)


def test_validation_errors():
    s = _FakeSolver()
    b = np.linspace(0, 1000, 8)
    img = np.ones((8, 8, 1, 8))
    with pytest.raises(ValueError):
        HipIDEALFitter(s, np.array([[2, 2], [8, 8]]), {"f1": 0.2}).fit(b, img)              # step_tol keys
    with pytest.raises(ValueError):
        HipIDEALFitter(s, np.array([[4, 4], [2, 2], [8, 8]]), {"f1": .2, "D1": .2, "D2": .2}).fit(b, img)  # monotonic
    with pytest.raises(ValueError):
        HipIDEALFitter(s, np.array([[2, 2], [4, 4]]), {"f1": .2, "D1": .2, "D2": .2}).fit(b, img)  # last != image
    with pytest.raises(ValueError):
        HipIDEALFitter(s, np.array([[8, 8]]), {"f1": .2, "D1": .2, "D2": .2}, interpolation_method="area")


@pytest.mark.gpu
def test_three_level_pyramid_on_gpu(gpu):
    from pyneapple_amd.solvers import HipCurveFitSolver

    rng = np.random.default_rng(0)
    b = np.linspace(0, 1200, 24)
    X = Y = 32
    f1 = 0.2 + 0.1 * np.sin(np.linspace(0, 3, X))[:, None] * np.ones((1, Y))
    D1 = np.full((X, Y), 0.02)
    D2 = 0.001 + 0.0005 * np.linspace(0, 1, Y)[None, :] * np.ones((X, 1))
    img = (f1[..., None] * np.exp(-b * D1[..., None]) + (1 - f1[..., None]) * np.exp(-b * D2[..., None]))[:, :, None, :]
    img = img * (1 + 0.005 * rng.standard_normal(img.shape))
    solver = HipCurveFitSolver(model=BiExpModel(), max_iter=250, tol=1e-8, p0={"f1": 0.2, "D1": 0.01, "D2": 0.001},
                               bounds={"f1": (0.0, 1.0), "D1": (1e-3, 0.1), "D2": (1e-5, 5e-3)})
    f = HipIDEALFitter(solver, np.array([[4, 4], [16, 16], [32, 32]]), {"f1": 0.5, "D1": 0.5, "D2": 0.5})
    f.fit(b, img)
    assert [p.shape for p in f.step_params] == [(4, 4, 1, 3), (16, 16, 1, 3), (32, 32, 1, 3)]
    final = f.step_params[-1][:, :, 0, :]
    assert np.median(np.abs(final[..., 0] - f1) / f1) < 0.05
    assert np.median(np.abs(final[..., 2] - D2) / D2) < 0.05
    assert (np.asarray(solver.diagnostics_["status"]) > 0).mean() > 0.95


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["linear", "cubic"])
def test_device_resize_matches_numpy_restatement(gpu, method):
    """pnx_resize2d_f64 against the numpy statement of OpenCV's arithmetic (resize2d), down- and up-scaling, odd sizes."""
    rng = np.random.default_rng(1)
    for src, dst in (((13, 9, 3, 5), (5, 4)), ((6, 7, 2), (19, 16)), ((8, 8, 1, 2), (8, 8)), ((5, 11), (10, 3))):
        a = rng.standard_normal(src)
        want = resize2d(a, dst, method)
        got = gpu.resize2d(a, dst, method)
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=0, atol=4e-15 * np.abs(a).max())


@pytest.mark.gpu
def test_device_resident_pyramid_equals_host_pyramid(gpu):
    """The HBM-resident level loop (device resize, device bounds, device-pointer fits) returns what the numpy resize +
    host-array solver calls return, with and without a segmentation that drops voxels."""
    from pyneapple_amd.solvers import HipCurveFitSolver

    rng = np.random.default_rng(3)
    b = np.linspace(0, 1200, 24)
    X, Y, Z = 24, 20, 2
    f1 = 0.2 + 0.1 * rng.random((X, Y, Z))
    D1 = 0.02 + 0.01 * rng.random((X, Y, Z))
    D2 = 0.001 + 0.0005 * rng.random((X, Y, Z))
    img = f1[..., None] * np.exp(-b * D1[..., None]) + (1 - f1[..., None]) * np.exp(-b * D2[..., None])
    img = img * (1 + 0.01 * rng.standard_normal(img.shape))
    seg = np.zeros((X, Y, Z), int)
    seg[4:20, 3:15, :] = 1
    steps = np.array([[6, 5], [12, 10], [24, 20]])
    tol = {"f1": 0.5, "D1": 0.5, "D2": 0.5}

    def run(resident, segmentation):
        solver = HipCurveFitSolver(model=BiExpModel(), max_iter=250, tol=1e-8, p0={"f1": 0.2, "D1": 0.01, "D2": 0.001},
                                   bounds={"f1": (0.0, 1.0), "D1": (1e-3, 0.1), "D2": (1e-5, 5e-3)})
        f = HipIDEALFitter(solver, steps, tol, device_resident=resident)
        f.fit(b, img, segmentation)
        return f, solver

    for segmentation in (None, seg):
        fh, sh = run(False, segmentation)
        fd, sd = run(True, segmentation)
        assert len(fd.step_params) == 3
        for a, c in zip(fh.step_params, fd.step_params):
            assert a.shape == c.shape
            # same arithmetic up to the summation order inside the resize; the fits amplify 1e-16 to ~1e-9
            np.testing.assert_allclose(c, a, rtol=1e-6, atol=1e-12)
        np.testing.assert_array_equal(fd.pixel_indices, fh.pixel_indices)
        for k in ("f1", "D1", "D2"):
            np.testing.assert_allclose(fd.fitted_params_[k], fh.fitted_params_[k], rtol=1e-6)
        np.testing.assert_array_equal(sd.diagnostics_["status"], sh.diagnostics_["status"])
        assert sd.diagnostics_["pcov"].shape == sh.diagnostics_["pcov"].shape and len(sd.pixel_results_) == len(sh.pixel_results_)


@pytest.mark.gpu
def test_config5_triexp_three_levels_full_size(gpu, oracle):
    """BASELINE configs[4]: IDEAL multi-resolution triexp, 64^2 -> 128^2 -> 256^2, 64 slices, 32 b-values, through the
    HBM-resident pyramid (fitters/ideal.py:167-254 of the reference).  Size-independent properties on all 4.19 M
    voxels, and the oracle on 4 096 random voxels of the last level with the SAME per-voxel p0 / bounds arrays the
    device used (read back from HBM)."""
    import torch

    from pyneapple_amd import synth
    from pyneapple_amd.models import TriExpModel
    from pyneapple_amd.solvers import HipCurveFitSolver

    shape = (256, 256, 64)
    n = int(np.prod(shape))
    b, y = synth.make_torch("tri_reduced", n, 32, torch.device("cuda", 0), sigma=0.01)
    img = y.cpu().numpy().reshape(*shape, 32)
    del y
    names, p0, lo, hi = synth.shared_arrays("tri_reduced")
    solver = HipCurveFitSolver(model=TriExpModel(), max_iter=250, tol=1e-8, p0=dict(zip(names, p0)),
                               bounds={k: (a, c) for k, a, c in zip(names, lo, hi)})
    steps = np.array([[64, 64], [128, 128], [256, 256]])
    tol = {k: 0.5 for k in names}
    f = HipIDEALFitter(solver, steps, tol, interpolation_method="cubic", device_resident=True, keep_level_inputs=True)
    f.fit(b, img)
    # level shapes
    assert [p.shape for p in f.step_params] == [(64, 64, 64, 5), (128, 128, 64, 5), (256, 256, 64, 5)]
    assert [s["n_pixels"] for s in f.level_stats_] == [64 * 64 * 64, 128 * 128 * 64, n]
    assert f.results_.n_pixels == n and f.image_shape == (*shape, 32) and f.pixel_indices.shape == (n, 3)
    # the levels below the first start from the previous level's map: never worse than their start values
    for s in f.level_stats_[1:]:
        assert s["not_worse_than_p0_frac"] == 1.0 and s["cost_mean"] <= s["cost_p0_mean"]
        assert s["converged_frac"] > 0.999
    # every estimate of the last level inside its per-voxel bounds (read back from the device)
    L = f.last_level_inputs_
    popt = np.stack([solver.params_[k] for k in names])          # (5, n)
    lo_d, hi_d, p0_d = (L[k].cpu().numpy() for k in ("lo", "hi", "p0"))
    ok = np.asarray(solver.diagnostics_["status"]) > 0
    assert (popt >= lo_d).all() and (popt <= hi_d).all()
    assert (lo_d >= lo[:, None]).all() and (hi_d <= hi[:, None]).all()   # clipped to the global bounds (ideal.py:176-184)
    np.testing.assert_array_equal(p0_d, np.clip(p0_d, lo[:, None], hi[:, None]))
    # the sweep kernel's cost at p0 against the oracle's residuals at p0 (max_nfev=1: one evaluation, at the start values)
    rng = np.random.default_rng(5)
    sub = np.sort(rng.choice(n, 4096, replace=False))
    ysub = np.ascontiguousarray(img.reshape(-1, 32)[sub])
    args = (np.ascontiguousarray(p0_d[:, sub]), np.ascontiguousarray(lo_d[:, sub]), np.ascontiguousarray(hi_d[:, sub]))
    at_p0 = oracle.curvefit("tri_reduced", b, ysub, *args, max_nfev=1)
    inside = ((args[0] > args[1]) & (args[0] < args[2])).all(axis=0) & (at_p0["status"] >= 0)
    np.testing.assert_allclose(L["cost_p0"].cpu().numpy()[sub][inside], at_p0["cost"][inside], rtol=1e-9)
    # the oracle leg: same rows, same per-voxel start values and bounds -> same estimates
    ref = oracle.curvefit("tri_reduced", b, ysub, *args, max_nfev=250, ftol=1e-8, jac="fd", n_threads=8)
    np.testing.assert_array_equal(ref["status"] > 0, ok[sub])
    good = ref["status"] > 0
    rel = np.abs(popt[:, sub] - ref["popt"]) / np.maximum(np.abs(ref["popt"]), 1e-300)
    within = (rel.max(axis=0) <= 1e-4)[good]
    # per-voxel boxes are tight (p0 * (1 -/+ 0.5)): the optimum usually sits on a face, where TRF's answer is sharp
    assert within.mean() >= 0.995, f"only {within.mean():.4f} of the subset within rtol 1e-4 of the oracle"
    np.testing.assert_allclose(np.asarray(solver.diagnostics_["cost"])[sub][good], ref["cost"][good], rtol=1e-5)
    # R^2 assembled from the kernel's cost and the HBM-side SS_tot reduction
    r2 = f.results_.r_squared
    pred = f.predict_pixels(b)[sub]
    ss_res = ((ysub - pred) ** 2).sum(axis=1)
    ss_tot = ((ysub - ysub.mean(axis=1, keepdims=True)) ** 2).sum(axis=1)
    np.testing.assert_allclose(r2[sub][good], (1 - ss_res / ss_tot)[good], rtol=1e-9, atol=1e-12)


@pytest.mark.gpu
def test_level_plumbing_kernels_match_numpy(gpu):
    """pnx_mask_select_f64 / pnx_gather_rows_f64 / pnx_scatter_rows_t_f64 / pnx_row_ss_tot_f64 against the numpy statements
    of the reference's level loop (fitters/ideal.py:199-254: threshold, `image[mask]`, `param_map[xs, ys, zs, k] = values`)."""
    import torch

    rng = np.random.default_rng(7)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    for shape, thr in (((9, 7, 3), 0.2), ((64, 33, 5), 0.55), ((4, 4, 1), 2.0), ((4, 4, 1), -1.0)):
        mask = rng.random(shape)
        n_all = mask.size
        md = torch.from_numpy(mask).to(dev)
        idx = torch.empty(n_all, dtype=torch.int64, device=dev)
        k = gpu.mask_select_device(md, thr, idx, 0, st)
        want = np.flatnonzero(mask.reshape(-1) > thr)          # C order of np.where
        assert k == len(want)
        np.testing.assert_array_equal(idx[:k].cpu().numpy(), want)
        if k == 0:
            continue
        c = 6
        src = rng.normal(size=(n_all, c))
        sd = torch.from_numpy(src).to(dev)
        dst = torch.empty((k, c), dtype=torch.float64, device=dev)
        gpu.gather_rows_device(sd, c, idx[:k], k, dst, 0, st)
        np.testing.assert_array_equal(dst.cpu().numpy(), src[want])
        popt = rng.normal(size=(3, k))                          # parameter major, as the solver returns it
        pmap = torch.full((n_all, 3), 7.0, dtype=torch.float64, device=dev)
        gpu.scatter_rows_t_device(torch.from_numpy(popt).to(dev), idx[:k], k, 3, n_all, pmap, 0, st)
        ref = np.zeros((n_all, 3))
        ref[want] = popt.T
        np.testing.assert_array_equal(pmap.cpu().numpy(), ref)
        pall = rng.normal(size=(3, n_all))
        gpu.scatter_rows_t_device(torch.from_numpy(pall).to(dev), None, n_all, 3, n_all, pmap, 0, st)
        np.testing.assert_array_equal(pmap.cpu().numpy(), pall.T)
        ss = torch.empty(n_all, dtype=torch.float64, device=dev)
        gpu.row_ss_tot_device(sd, n_all, c, ss, 0, st)
        np.testing.assert_allclose(ss.cpu().numpy(), ((src - src.mean(axis=1, keepdims=True)) ** 2).sum(axis=1), rtol=1e-13)


@pytest.mark.gpu
def test_bulk_copies_round_trip(gpu):
    """pnx_upload / pnx_download: bytes arrive unchanged for sizes below, at and across the 32 MiB piece boundary, with one
    or several threads; a size mismatch is refused on the host side."""
    import torch

    from pyneapple_amd import api

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(11)
    for n, threads in ((0, 0), (1, 1), (1000, 0), ((32 << 20) // 8, 2), ((80 << 20) // 8 + 13, 0), ((80 << 20) // 8 + 13, 1)):
        a = rng.standard_normal(n)
        t = torch.empty(n, dtype=torch.float64, device=dev)
        if n:
            api.upload(a, t, 0, torch.cuda.current_stream().cuda_stream, threads)
        np.testing.assert_array_equal(t.cpu().numpy(), a)
        t2 = torch.from_numpy(a).to(dev) * 2.0          # produced on the stream the download has to wait for
        back = api.download(t2, 0, torch.cuda.current_stream().cuda_stream, threads)
        np.testing.assert_array_equal(back, a * 2.0)
    i8 = torch.arange(-100, 100, dtype=torch.int8, device=dev)
    np.testing.assert_array_equal(api.download(i8, 0), np.arange(-100, 100, dtype=np.int8))
    m = torch.arange(24, dtype=torch.float64, device=dev).reshape(4, 6).t()   # non-contiguous view
    np.testing.assert_array_equal(api.download(m, 0), np.arange(24.0).reshape(4, 6).T)
    with pytest.raises(ValueError):
        api.upload(np.zeros(3), torch.empty(4, dtype=torch.float64, device=dev), 0)


@pytest.mark.gpu
def test_device_pyramid_honours_the_solvers_xtol_gtol(gpu):
    """The HBM-resident level loop builds its own option block: it must carry the solver's xtol / gtol like the host path
    does (a coarse xtol stops every level's fit earlier, on both paths alike)."""
    from pyneapple_amd.solvers import HipCurveFitSolver

    rng = np.random.default_rng(5)
    b = np.linspace(0, 1200, 24)
    X, Y, Z = 16, 16, 1
    f1 = 0.2 + 0.1 * rng.random((X, Y, Z))
    img = f1[..., None] * np.exp(-b * 0.02) + (1 - f1[..., None]) * np.exp(-b * 0.001)
    img = img * (1 + 0.02 * rng.standard_normal(img.shape))
    steps = np.array([[4, 4], [16, 16]])
    tol = {"f1": 0.5, "D1": 0.5, "D2": 0.5}

    def run(resident, **kw):
        solver = HipCurveFitSolver(model=BiExpModel(), max_iter=250, tol=1e-8, p0={"f1": 0.2, "D1": 0.01, "D2": 0.001},
                                   bounds={"f1": (0.0, 1.0), "D1": (1e-3, 0.1), "D2": (1e-5, 5e-3)}, **kw)
        HipIDEALFitter(solver, steps, tol, device_resident=resident).fit(b, img)
        return solver

    coarse_h, coarse_d, fine_d = run(False, xtol=1e-2), run(True, xtol=1e-2), run(True)
    np.testing.assert_array_equal(coarse_d.diagnostics_["nfev"], coarse_h.diagnostics_["nfev"])
    np.testing.assert_allclose(coarse_d.params_["D1"], coarse_h.params_["D1"], rtol=1e-6)
    assert coarse_d.diagnostics_["nfev"].mean() < fine_d.diagnostics_["nfev"].mean()  # the coarse tolerance was in force
