"""Parity of the HIP NNLS path (C ABI) with the reference's golden vectors / the oracle, plus KKT
optimality at BASELINE.json's full size."""
from __future__ import annotations

import numpy as np
import pytest
from conftest import NNLS_FIXTURES, load_golden

pytestmark = pytest.mark.gpu


def _scaled_err(c, cr):
    return (np.abs(c - cr) / (np.abs(cr).max(axis=1, keepdims=True) + 1e-300)).max(axis=1)


@pytest.mark.parametrize("name", NNLS_FIXTURES)
def test_matches_reference_golden(gpu, name):
    d = load_golden(name)
    r = gpu.nnls(d["basis"], d["reg"], d["y"], int(d["max_iter"]))
    assert ((r["status"] == 1) == d["success"]).all()
    # coefficients: rtol 1e-4 of the spectrum's peak (entries that are exactly 0 in one result and 1e-12 in
    # the other make an element-wise rtol meaningless); observed ~1e-8
    assert _scaled_err(r["coefficients"], d["coefficients"]).max() < 1e-6
    np.testing.assert_allclose(r["residual"], d["residual"], rtol=1e-10)
    assert (r["coefficients"] >= 0).all()
    if not d["success"].all():  # failure path: zeros and ||y_ext|| (nnls_solver.py:205-210)
        bad = ~d["success"]
        assert (r["coefficients"][bad] == 0).all()
        np.testing.assert_allclose(r["residual"][bad], np.linalg.norm(d["y"][bad], axis=1), rtol=1e-14)


@pytest.mark.parametrize("order", [1, 2, 3])
def test_matches_oracle_seeded(gpu, oracle, order):
    from pyneapple_amd import synth

    cfg = dict(synth.NNLS_CFG, reg_order=order)
    _, basis, reg = synth.nnls_matrices(32, cfg)
    _, y, _ = synth.make_numpy("tri_reduced", 512, 32, sigma=0.01, seed=99, scale=1000.0)
    r = gpu.nnls(basis, reg, y, 250)
    o = oracle.nnls(basis, reg, y, 250, n_threads=8)
    np.testing.assert_array_equal(r["status"], o["status"])
    ok = o["status"] == 1
    assert _scaled_err(r["coefficients"][ok], o["coefficients"][ok]).max() < 1e-6
    np.testing.assert_allclose(r["residual"][ok], o["residual"][ok], rtol=1e-9)
    # same active-set path: identical iteration counts (a tie decided differently is tolerated on < 1 %)
    assert (r["iters"] == o["iters"]).mean() > 0.99
    assert (((r["coefficients"] > 0) != (o["coefficients"] > 0)).sum(axis=1) == 0).mean() > 0.99


@pytest.mark.parametrize("n_bins", [65, 34, 129, 201, 256])
@pytest.mark.parametrize("order", [1, 2])
def test_last_bin_in_every_register_slot(gpu, oracle, n_bins, order):
    """The block kernel evaluates R^T R x of the tridiagonal regularisers as one five-point stencil with a correction at each end, and
    the last bin sits in another register slot depending on n_bins (bin 64: slot 0, 33: slot 1, 128 / 200: slot 2, 255: slot 3; the
    250 bins of the other tests: slot 3, their 50: slot 1): every slot against the oracle, passive sets on both sides of the
    Gram-form threshold."""
    from pyneapple_amd import synth

    cfg = dict(synth.NNLS_CFG, reg_order=order, n_bins=n_bins)
    _, basis, reg = synth.nnls_matrices(32, cfg)
    assert basis.shape == (32, n_bins)
    _, y, _ = synth.make_numpy("tri_reduced", 384, 32, sigma=0.01, seed=7 + n_bins, scale=1000.0)
    # a third of the voxels: signals made of the first and the last bins, so that x_0 and x_{n-1} -- what the end corrections multiply --
    # are in the solution
    rng = np.random.default_rng(n_bins)
    ends = np.zeros((128, n_bins))
    ends[:, 0] = rng.uniform(100, 1000, 128)
    ends[:, -1] = rng.uniform(100, 1000, 128)
    ends[:, -2] = rng.uniform(0, 300, 128)
    y[:128] = ends @ basis.T * (1 + 0.01 * rng.standard_normal((128, 32)))
    r = gpu.nnls(basis, reg, y, 250)
    o = oracle.nnls(basis, reg, y, 250, n_threads=8)
    np.testing.assert_array_equal(r["status"], o["status"])
    ok = o["status"] == 1
    assert ok.mean() > 0.9
    assert _scaled_err(r["coefficients"][ok], o["coefficients"][ok]).max() < 1e-6
    np.testing.assert_allclose(r["residual"][ok], o["residual"][ok], rtol=1e-9)
    assert (r["iters"] == o["iters"]).mean() > 0.99
    assert (o["coefficients"][ok] > 0).sum(axis=1).max() > 16  # beyond the Gram-form dual's passive sets too
    assert (o["coefficients"][ok][:, -1] > 0).sum() > 10 and (o["coefficients"][ok][:, 0] > 0).sum() > 10


@pytest.mark.parametrize("n_b,n_bins,d_range", [(32, 250, (0.0008, 0.5)), (16, 50, (1e-4, 0.1)), (24, 120, (0.0008, 0.5))])
def test_unregularised_matches_oracle(gpu, oracle, n_b, n_bins, d_range):
    """reg_order = 0, the reference's default (nnls_solver.py:37): A = B alone, rank deficient.  The QR-based kernel
    (pnx_nnls_qr.hip) must follow the oracle's active-set path voxel by voxel -- same supports, same iteration counts."""
    from pyneapple_amd import synth

    bins = np.logspace(np.log10(d_range[0]), np.log10(d_range[1]), n_bins)
    b = np.linspace(0.0, 1200.0, n_b)
    basis = np.exp(-b[:, None] * bins[None, :])
    _, y, _ = synth.make_numpy("tri_reduced", 3000, n_b, sigma=0.01, seed=5, scale=1000.0)
    for reg in (None, np.zeros((n_bins, n_bins))):
        r = gpu.nnls(basis, reg, y, 250)
        o = oracle.nnls(basis, reg, y, 250, n_threads=8)
        np.testing.assert_array_equal(r["status"], o["status"])
        ok = o["status"] == 1
        same = ((r["coefficients"] > 0) == (o["coefficients"] > 0)).all(axis=1)
        assert same[ok].mean() > 0.995, f"supports differ on {(~same[ok]).sum()} voxels"
        assert (r["iters"] == o["iters"])[ok].mean() > 0.995
        assert _scaled_err(r["coefficients"][ok & same], o["coefficients"][ok & same]).max() < 1e-6
        np.testing.assert_allclose(r["residual"][ok], o["residual"][ok], rtol=1e-9)  # the minimum itself: every voxel


def test_edge_shapes_and_failures(gpu, oracle):
    rng = np.random.default_rng(0)
    # no regulariser, more bins than measurements, > 64 bins (several bins per lane), 1 voxel, empty batch
    for (m, n, nv) in [(4, 3, 2), (16, 10, 5), (32, 80, 7), (32, 256, 3), (128, 40, 2)]:
        B = rng.uniform(0, 1, (m, n))
        y = rng.uniform(0, 1, (nv, m))
        r = gpu.nnls(B, None, y, 1000)
        o = oracle.nnls(B, None, y, 1000)
        np.testing.assert_allclose(r["residual"], o["residual"], rtol=1e-7, atol=1e-9)
        assert (r["coefficients"] >= 0).all()
    assert gpu.nnls(np.ones((4, 3)), None, np.empty((0, 4)))["coefficients"].shape == (0, 3)
    d = load_golden("g4_nnls_50_r2")
    y = d["y"][:3].copy()
    y[1, 2] = np.nan
    r = gpu.nnls(d["basis"], d["reg"], y, 5)
    assert r["status"][0] == 0 and r["status"][1] == -2 and (r["coefficients"][:2] == 0).all()
    assert r["residual"][0] == pytest.approx(np.linalg.norm(y[0])) and np.isnan(r["residual"][1])
    with pytest.raises(Exception):
        gpu.nnls(np.ones((4, 600)), None, np.ones((1, 4)))  # > 512 bins: refused, not silently wrong


@pytest.mark.parametrize("n_bins,order,n_b", [(300, 2, 32), (512, 2, 32), (257, 1, 16), (384, 3, 48), (300, 2, 96)])
def test_wide_regularised_plans_match_oracle(gpu, oracle, n_bins, order, n_b):
    """More than 256 bins (the reference takes any n_bins, models/nnls.py:37-77; refused up to round 3): the Gram-form kernel
    with eight bins per lane must walk the oracle's active-set path -- status (512 bins run into max_iter = 250 on some
    voxels), iteration counts, supports."""
    from pyneapple_amd import synth

    cfg = dict(synth.NNLS_CFG, reg_order=order, n_bins=n_bins)
    _, basis, reg = synth.nnls_matrices(n_b, cfg)
    assert basis.shape == (n_b, n_bins)
    _, y, _ = synth.make_numpy("tri_reduced", 384, n_b, sigma=0.01, seed=123, scale=1000.0)
    r = gpu.nnls(basis, reg, y, 250)
    o = oracle.nnls(basis, reg, y, 250, n_threads=8)
    np.testing.assert_array_equal(r["status"], o["status"])
    ok = o["status"] == 1
    assert ok.mean() > 0.5
    assert _scaled_err(r["coefficients"][ok], o["coefficients"][ok]).max() < 1e-6
    np.testing.assert_allclose(r["residual"][ok], o["residual"][ok], rtol=1e-9)
    np.testing.assert_allclose(r["residual"][~ok], np.linalg.norm(y[~ok], axis=1), rtol=1e-14)
    assert (r["coefficients"][~ok] == 0).all()
    assert (r["iters"] == o["iters"]).mean() > 0.99
    assert (((r["coefficients"] > 0) != (o["coefficients"] > 0)).sum(axis=1) == 0).mean() > 0.99


@pytest.mark.parametrize("n_b,n_bins", [(32, 300), (24, 512), (64, 400), (96, 300), (128, 512)])
def test_wide_unregularised_plans_match_oracle(gpu, oracle, n_b, n_bins):
    """reg_order = 0 with more than 256 bins: the QR-form kernels with eight bins per lane (Q / R in LDS up to 32 measurements, in
    the per-wave slab beyond)."""
    from pyneapple_amd import synth

    bins = np.logspace(np.log10(0.0008), np.log10(0.5), n_bins)
    b = np.linspace(0.0, 1200.0, n_b)
    basis = np.exp(-b[:, None] * bins[None, :])
    _, y, _ = synth.make_numpy("tri_reduced", 1000, n_b, sigma=0.01, seed=7, scale=1000.0)
    for reg in (None, np.zeros((n_bins, n_bins))):
        r = gpu.nnls(basis, reg, y, 400)
        o = oracle.nnls(basis, reg, y, 400, n_threads=8)
        np.testing.assert_array_equal(r["status"], o["status"])
        ok = o["status"] == 1
        assert ok.mean() > 0.9
        same = ((r["coefficients"] > 0) == (o["coefficients"] > 0)).all(axis=1)
        assert same[ok].mean() > 0.99, f"supports differ on {(~same[ok]).sum()} voxels"
        assert (r["iters"] == o["iters"])[ok].mean() > 0.99
        assert _scaled_err(r["coefficients"][ok & same], o["coefficients"][ok & same]).max() < 1e-6
        np.testing.assert_allclose(r["residual"][ok], o["residual"][ok], rtol=1e-9)
        assert (r["coefficients"] >= 0).all()


@pytest.mark.parametrize("dense,n_bins", [(False, 300), (True, 300), (False, 420), (True, 500)])
def test_wide_passive_set_beyond_256_positions(gpu, oracle, dense, n_bins):
    """A strongly damped fit of a positive combination of ALL columns: every bin enters.  The first-pass kernels of a wide plan
    (`nnls_kernel<6, 4>` up to 384 bins, `<8, 4>` beyond) keep 256 positions; at the 257th they hand the voxel over to `<8, 8>`,
    where the passive set grows through the fifth .. eighth register slot of the position-indexed vectors and the overflow rows
    of the inverse factor reach row n_bins - 1 of its slab; with a dense (non-Toeplitz) regulariser the generic rows-of-reg
    epilogue runs too."""
    rng = np.random.default_rng(5)
    n_b = 128
    basis = np.abs(rng.standard_normal((n_b, n_bins)))
    reg = 3.0 * np.eye(n_bins)
    if dense:
        reg = reg + 0.05 * rng.standard_normal((n_bins, n_bins))
    x_true = rng.uniform(0.5, 2.0, (48, n_bins))
    x_true[:, ::11] = 0.0
    x_true[1::2] *= rng.random((24, n_bins)) < 0.05  # every other voxel: a sparse combination -- stays in the first-pass kernel
    y = x_true @ basis.T + 1e-3 * rng.standard_normal((48, n_b))
    r = gpu.nnls(basis, reg, y, 2000)
    o = oracle.nnls(basis, reg, y, 2000, n_threads=8)
    np.testing.assert_array_equal(r["status"], o["status"])
    support = (o["coefficients"] > 0).sum(axis=1)
    assert (o["status"] == 1).all() and (support > 256).sum() >= 20 and (support <= 256).sum() >= 10, support
    assert (r["iters"] == o["iters"]).mean() > 0.9
    assert _scaled_err(r["coefficients"], o["coefficients"]).max() < 1e-8
    np.testing.assert_allclose(r["residual"], o["residual"], rtol=1e-9)


@pytest.mark.parametrize("n_b,n_bins", [(96, 250), (65, 250), (128, 256), (100, 60), (48, 250), (33, 120), (64, 250)])
def test_unregularised_beyond_64_measurements_matches_oracle(gpu, oracle, n_b, n_bins):
    """reg_order = 0 (the reference default, nnls_solver.py:37) has no limit on the number of b-values.  Up to round 3 a plan
    with 65..128 measurements was refused (the QR-form kernel keeps one measurement per lane, and the Gram-form kernel selects
    other columns than SciPy on such ill conditioned bases); since round 4 a second QR-form kernel with two measurements per
    lane and Q / R in a global slab follows the oracle's active-set path voxel by voxel -- same supports, same iteration
    counts -- also where the passive set grows past 64 positions (the 100 x 60 basis is well conditioned: every bin enters)."""
    from pyneapple_amd import synth

    bins = np.logspace(np.log10(0.0008), np.log10(0.5), n_bins)
    b = np.linspace(0.0, 1200.0, n_b)
    basis = np.exp(-b[:, None] * bins[None, :])
    _, y, _ = synth.make_numpy("tri_reduced", 1500, n_b, sigma=0.01, seed=5, scale=1000.0)
    for reg in (None, np.zeros((n_bins, n_bins))):
        r = gpu.nnls(basis, reg, y, 400)
        o = oracle.nnls(basis, reg, y, 400, n_threads=8)
        np.testing.assert_array_equal(r["status"], o["status"])
        ok = o["status"] == 1
        assert ok.mean() > 0.9
        same = ((r["coefficients"] > 0) == (o["coefficients"] > 0)).all(axis=1)
        assert same[ok].mean() > 0.99, f"supports differ on {(~same[ok]).sum()} voxels"
        assert (r["iters"] == o["iters"])[ok].mean() > 0.99
        assert _scaled_err(r["coefficients"][ok & same], o["coefficients"][ok & same]).max() < 1e-6
        np.testing.assert_allclose(r["residual"][ok], o["residual"][ok], rtol=1e-9)  # the minimum itself: every voxel
        assert (r["coefficients"] >= 0).all()


def test_unregularised_passive_set_beyond_64_positions(gpu, oracle):
    """A well conditioned tall basis (128 measurements, 100 widely spaced bins) and a signal that is a positive combination of
    ALL columns: the passive set grows to 100 positions, i.e. into the second register slot of every position-indexed vector
    of the two-slot kernel, and removals shift positions across the slot boundary."""
    rng = np.random.default_rng(11)
    n_b, n_bins = 128, 100
    basis = np.abs(rng.standard_normal((n_b, n_bins))) + 0.1 * np.eye(n_b, n_bins)
    x_true = rng.uniform(0.5, 2.0, (64, n_bins))
    x_true[:, ::7] = 0.0
    y = x_true @ basis.T + 1e-3 * rng.standard_normal((64, n_b))
    r = gpu.nnls(basis, None, y, 1000)
    o = oracle.nnls(basis, None, y, 1000, n_threads=8)
    np.testing.assert_array_equal(r["status"], o["status"])
    assert (o["status"] == 1).all() and ((o["coefficients"] > 0).sum(axis=1) > 64).all()
    assert (r["iters"] == o["iters"]).mean() > 0.95
    assert _scaled_err(r["coefficients"], o["coefficients"]).max() < 1e-8
    np.testing.assert_allclose(r["residual"], o["residual"], rtol=1e-9)


def test_builders_on_device(gpu):
    d = load_golden("g4_nnls_250_r2")
    basis = gpu.nnls_basis(d["bvalues"], d["bins"])
    np.testing.assert_allclose(basis, d["basis"], rtol=4e-16, atol=0)  # exp() within 1-2 ulp of numpy's


def test_mfma_gram_step_matches_numpy(gpu):
    """pnx_nnls_aty_f64 (v_mfma_f64_16x16x4 Gram step): aty = y @ basis to fp64 rounding, padding columns zero."""
    import torch

    from pyneapple_amd import synth

    _, basis, reg = synth.nnls_matrices(32)
    _, y, _ = synth.make_numpy("tri_reduced", 1000 + 13, 32, sigma=0.01, seed=1, scale=1000.0)
    plan = gpu.NnlsPlan(basis, reg, 0)
    dev = torch.device("cuda", 0)
    yt = torch.tensor(y, device=dev)
    aty = torch.full((y.shape[0], 256), float("nan"), dtype=torch.float64, device=dev)
    plan.aty_device(y.shape[0], yt, aty, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    plan.close()
    got = aty.cpu().numpy()
    ref = y @ basis
    np.testing.assert_allclose(got[:, :250], ref, rtol=1e-13, atol=1e-13 * np.abs(ref).max())
    assert (got[:, 250:] == 0).all()


def test_full_size_kkt_c4(gpu):
    """BASELINE.json configs[3] at full size: the result of every voxel satisfies the NNLS optimality (KKT)
    conditions  x >= 0,  w = A^T(y_ext - A x) <= tol on x == 0,  |w| <= tol on x > 0 -- size independent and
    checked on the device for all 4 194 304 voxels."""
    import torch

    from pyneapple_amd import api, synth

    dev = torch.device("cuda", 0)
    n_vox, n_b = 256 * 256 * 64, 32
    bins, basis, reg = synth.nnls_matrices(n_b)
    _, y = synth.make_torch("tri_reduced", n_vox, n_b, dev, sigma=0.01, scale=1000.0)
    nb = basis.shape[1]
    coeff = torch.empty((n_vox, nb), dtype=torch.float64, device=dev)
    rnorm = torch.empty(n_vox, dtype=torch.float64, device=dev)
    status = torch.empty(n_vox, dtype=torch.int8, device=dev)
    iters = torch.empty(n_vox, dtype=torch.int32, device=dev)
    plan = api.NnlsPlan(basis, reg, 0)
    plan.solve_device(n_vox, y, 250, coeff, rnorm, status, iters, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ok = status == 1
    assert ok.double().mean().item() > 0.999
    assert bool((coeff >= 0).all())
    Bt = torch.tensor(basis, device=dev)
    Rt = torch.tensor(reg, device=dev)
    G = Bt.T @ Bt + Rt.T @ Rt
    worst_pos, worst_act = 0.0, 0.0
    for s in range(0, n_vox, 1 << 19):
        c = coeff[s:s + (1 << 19)]
        w = y[s:s + (1 << 19)] @ Bt - c @ G
        scale = (y[s:s + (1 << 19)] @ Bt).abs().max(dim=1, keepdim=True).values
        o = ok[s:s + (1 << 19), None]
        worst_pos = max(worst_pos, float(torch.where((c == 0) & o, w / scale, torch.zeros_like(w)).max()))
        worst_act = max(worst_act, float(torch.where((c > 0) & o, (w / scale).abs(), torch.zeros_like(w)).max()))
    assert worst_pos < 1e-9 and worst_act < 1e-9
    # failures keep the reference's sentinel
    bad = ~ok
    if bool(bad.any()):
        assert bool((coeff[bad] == 0).all())
        torch.testing.assert_close(rnorm[bad], torch.linalg.norm(y[bad], dim=1))


@pytest.mark.parametrize("order", [1, 2, 3])
def test_rejected_candidate_columns_are_set_aside_and_come_back(gpu, oracle, monkeypatch, order):
    """Lawson-Hanson answers a rejected candidate with "w_j = 0, take the next largest" (and stops where nothing positive is
    left).  The block kernel has no loop around its append for that: it sets the column's passive flag, remembers the bin in LDS,
    evaluates the dual again and takes the flags back when a column enters; a ninth rejection in one outer iteration hands the
    voxel to the general kernel.  The reference workload never rejects a column (none in 400 fuzz cases either), so a test hook
    shared by kernel and oracle (PNX_NNLS_TEST_REJECT=k,n: with p % k == k - 1 the first n candidates of an outer iteration are
    rejected unseen) forces the path: the kernel must follow the oracle's detours column by column."""
    from pyneapple_amd import synth

    cfg = dict(synth.NNLS_CFG, reg_order=order)
    _, basis, reg = synth.nnls_matrices(32, cfg)
    _, y, _ = synth.make_numpy("tri_reduced", 768, 32, sigma=0.01, seed=5, scale=1000.0)
    plain = gpu.nnls(basis, reg, y, 2000)
    assert (plain["status"] == 1).all()
    changed = 0
    for hook in ("3,1", "2,4", "1,2", "5,8", "4,9"):  # the last one overflows the list of eight: hand-over to the general kernel
        monkeypatch.setenv("PNX_NNLS_TEST_REJECT", hook)
        r = gpu.nnls(basis, reg, y, 2000)
        o = oracle.nnls(basis, reg, y, 2000, n_threads=8)
        np.testing.assert_array_equal(r["status"], o["status"], err_msg=hook)
        ok = o["status"] == 1
        assert _scaled_err(r["coefficients"][ok], o["coefficients"][ok]).max() < 1e-6, hook
        np.testing.assert_allclose(r["residual"][ok], o["residual"][ok], rtol=1e-9, err_msg=hook)
        assert (r["iters"] == o["iters"]).mean() > 0.99, hook
        assert (((r["coefficients"] > 0) != (o["coefficients"] > 0)).sum(axis=1) == 0).mean() > 0.99, hook
        changed += int((r["iters"] != plain["iters"]).any() or (r["coefficients"] != plain["coefficients"]).any())
    assert changed >= 3  # the hook did send the solves down other paths
    monkeypatch.delenv("PNX_NNLS_TEST_REJECT")
    again = gpu.nnls(basis, reg, y, 2000)
    for k in ("coefficients", "residual", "status", "iters"):
        np.testing.assert_array_equal(again[k], plain[k])


def test_developer_switches_are_ignored_without_the_gate(gpu):
    """include/pnx.h, "Environment": the rejection hook (and every A/B kernel switch) is read only in a process started with
    PNX_ENABLE_TEST_HOOKS=1.  Three fresh processes solve the same 256 voxels: hook set but no gate = no hook, bit for bit; with
    the gate the hook sends the solves down other paths (more iterations)."""
    import json
    import os
    import subprocess
    import sys

    from conftest import ROOT

    child = (
        "import sys, json, numpy as np; sys.path.insert(0, %r)\n"
        "from pyneapple_amd import api, synth\n"
        "_, basis, reg = synth.nnls_matrices(32)\n"
        "_, y, _ = synth.make_numpy('tri_reduced', 256, 32, sigma=0.01, seed=5, scale=1000.0)\n"
        "r = api.nnls(basis, reg, y, 2000)\n"
        "print(json.dumps({'iters': int(r['iters'].sum()), 'coef': float(r['coefficients'].sum())}))\n" % ROOT)

    def run(**env):
        e = {k: v for k, v in os.environ.items() if not k.startswith("PNX_")}
        e.update(env)
        out = subprocess.run([sys.executable, "-c", child], env=e, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(out.stdout.strip().splitlines()[-1])

    plain = run()
    ungated = run(PNX_NNLS_TEST_REJECT="2,4", PNX_NNLS_NO_BLK="1", PNX_BLK_ROUTE_PERMILLE="1")
    gated = run(PNX_ENABLE_TEST_HOOKS="1", PNX_NNLS_TEST_REJECT="2,4")
    assert ungated == plain
    assert gated["iters"] > plain["iters"]


def test_host_chunks_defer_the_hand_over_pass(gpu, monkeypatch):
    """Host arrays in several chunks: the block kernel's handed-over voxels (passive set beyond 128 positions) are solved in
    ONE pass at the end of the call and patched into the result arrays.  Same bits as the single-chunk call (hand-over pass
    inside it), as the per-chunk passes (PNX_NNLS_DEFER_CAP=0), and as the overflow path (side buffer too small -> the call
    runs again with per-chunk passes); float32 storage likewise."""
    import torch

    from pyneapple_amd import synth

    _, basis, reg = synth.nnls_matrices(32)
    _, yt = synth.make_torch_rows("tri_reduced", 0, 1 << 14, 32, torch.device("cuda", 0), sigma=0.01, scale=1000.0)
    y = yt.cpu().numpy()
    plan = gpu.NnlsPlan(basis, reg, 0)
    monkeypatch.setenv("PNX_NNLS_HOST_CHUNK", str(1 << 20))
    one = plan.solve(y, 250)
    handed_over = np.flatnonzero((one["coefficients"] > 0).sum(axis=1) > 128)
    assert handed_over.size >= 1  # the sample holds such voxels (tests/test_gpu_parity_large.py compares them with the oracle)
    one32 = plan.solve(y.astype(np.float32), 250)
    monkeypatch.setenv("PNX_NNLS_HOST_CHUNK", "3000")  # six ragged chunks
    for cap in ("16384", "0", str(max(1, handed_over.size - 1)), "1"):  # the last two: more voxels than the side buffer holds -> batches
        monkeypatch.setenv("PNX_NNLS_DEFER_CAP", cap)
        many = plan.solve(y, 250)
        for k in ("coefficients", "residual", "status", "iters"):
            np.testing.assert_array_equal(many[k], one[k], err_msg=f"{k} cap={cap}")
    monkeypatch.setenv("PNX_NNLS_DEFER_CAP", "16384")
    many32 = plan.solve(y.astype(np.float32), 250)
    for k in ("coefficients", "residual", "status", "iters"):
        np.testing.assert_array_equal(many32[k], one32[k], err_msg=f"{k} float32")
    plan.close()


def test_peak_tables_from_host_arrays_in_chunks(gpu, monkeypatch):
    """pnx_nnls_solve_peaks_f64 on host arrays: chunk ring with the peak analysis behind every chunk's solve and the deferred
    hand-over (the handed-over voxels' peak rows are patched at the end).  Equal to the one-chunk call, to the per-chunk
    hand-over, to the old serial loop, and to solve() followed by the peak analysis of the returned spectra."""
    import torch

    from pyneapple_amd import synth

    bins, basis, reg = synth.nnls_matrices(32)
    _, yt = synth.make_torch_rows("tri_reduced", 0, 1 << 14, 32, torch.device("cuda", 0), sigma=0.01, scale=1000.0)
    y = yt.cpu().numpy()
    cuts = [(0.0008, 0.003), (0.003, 0.02), (0.02, 0.5)]
    kw = dict(max_iter=250, height=0.1, regularized=True, max_peaks=8, cutoffs=cuts)
    plan = gpu.NnlsPlan(basis, reg, 0)
    monkeypatch.setenv("PNX_NNLS_HOST_CHUNK", str(1 << 20))
    full = plan.solve(y, 250)
    assert ((full["coefficients"] > 0).sum(axis=1) > 128).any()
    ref = gpu.spectrum_peaks(full["coefficients"], bins, height=0.1, regularized=True, max_peaks=8, cutoffs=cuts)
    monkeypatch.setenv("PNX_NNLS_PEAKS_CHUNK", str(1 << 20))
    one = plan.solve_peaks(y, bins, **kw)
    for k in ("n_peaks", "d_values", "f_values", "d_cut", "f_cut"):
        np.testing.assert_array_equal(one[k], ref[k], err_msg=k)
    np.testing.assert_array_equal(one["residual"], full["residual"])
    monkeypatch.setenv("PNX_NNLS_PEAKS_CHUNK", "3000")
    for cap, ring in (("16384", "1"), ("0", "1"), ("16384", "0"), ("1", "1")):  # the last: one handed-over voxel per batch
        monkeypatch.setenv("PNX_NNLS_DEFER_CAP", cap)
        monkeypatch.setenv("PNX_NNLS_PEAKS_RING", ring)
        many = plan.solve_peaks(y, bins, **kw)
        for k in one:
            np.testing.assert_array_equal(many[k], one[k], err_msg=f"{k} cap={cap} ring={ring}")
    plan.close()


def test_host_pipeline_chunking_is_invisible(gpu, monkeypatch):
    """Chunked, pipelined host staging (solves in order on one stream) returns the single-chunk results."""
    from pyneapple_amd import synth

    _, basis, reg = synth.nnls_matrices(32)
    _, y, _ = synth.make_numpy("tri_reduced", 3000 + 11, 32, sigma=0.01, seed=3, scale=1000.0)
    plan = gpu.NnlsPlan(basis, reg, 0)
    one = plan.solve(y, 250)
    monkeypatch.setenv("PNX_NNLS_HOST_CHUNK", "1024")
    many = plan.solve(y, 250)
    plan.close()
    for k in ("coefficients", "residual", "status", "iters"):
        np.testing.assert_array_equal(many[k], one[k], err_msg=k)
    assert (one["status"] == 1).all()


def test_wide_plan_through_every_host_path(gpu, oracle, monkeypatch):
    """A 300-bin spectrum (refused up to round 3) through the paths a Pyneapple user reaches: host arrays in chunks, float32
    storage, the fused solve + peak tables, device-resident tensors -- all equal to one another, and the plain call to the oracle."""
    import torch

    from pyneapple_amd import synth

    cfg = dict(synth.NNLS_CFG, n_bins=300)
    bins, basis, reg = synth.nnls_matrices(32, cfg)
    _, y, _ = synth.make_numpy("tri_reduced", 5000 + 13, 32, sigma=0.01, seed=31, scale=1000.0)
    o = oracle.nnls(basis, reg, y[:512], 250, n_threads=8)
    plan = gpu.NnlsPlan(basis, reg, 0)
    one = plan.solve(y, 250)
    np.testing.assert_array_equal(one["status"][:512], o["status"])
    assert _scaled_err(one["coefficients"][:512], o["coefficients"]).max() < 1e-6
    assert (one["iters"][:512] == o["iters"]).mean() > 0.99
    monkeypatch.setenv("PNX_NNLS_HOST_CHUNK", "1024")
    many = plan.solve(y, 250)
    for k in ("coefficients", "residual", "status", "iters"):
        np.testing.assert_array_equal(many[k], one[k], err_msg=k)
    y32 = y.astype(np.float32)
    f32 = plan.solve(y32, 250)
    assert f32["coefficients"].dtype == np.float32
    ref32 = plan.solve(y32.astype(np.float64), 250)
    np.testing.assert_array_equal(f32["coefficients"], ref32["coefficients"].astype(np.float32))
    np.testing.assert_array_equal(f32["iters"], ref32["iters"])
    # device-resident
    dev = torch.device("cuda", 0)
    yd = torch.from_numpy(y).to(dev)
    n = y.shape[0]
    coeff = torch.empty((n, 300), dtype=torch.float64, device=dev)
    rn = torch.empty(n, dtype=torch.float64, device=dev)
    st = torch.empty(n, dtype=torch.int8, device=dev)
    it = torch.empty(n, dtype=torch.int32, device=dev)
    plan.solve_device(n, yd, 250, coeff, rn, st, it, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(coeff.cpu().numpy(), one["coefficients"])
    np.testing.assert_array_equal(it.cpu().numpy(), one["iters"])
    # fused solve + peak tables
    cuts = [(0.0008, 0.003), (0.003, 0.02), (0.02, 0.5)]
    ref = gpu.spectrum_peaks(one["coefficients"], bins, height=0.1, regularized=True, max_peaks=8, cutoffs=cuts)
    monkeypatch.setenv("PNX_NNLS_PEAKS_CHUNK", "2000")
    pk = plan.solve_peaks(y, bins, max_iter=250, height=0.1, regularized=True, max_peaks=8, cutoffs=cuts)
    for k in ("n_peaks", "d_values", "f_values", "d_cut", "f_cut"):
        np.testing.assert_array_equal(pk[k], ref[k], err_msg=k)
    np.testing.assert_array_equal(pk["residual"], one["residual"])
    plan.close()
    with pytest.raises(Exception):  # the MFMA Gram step on its own is a 256-column layout: refused for a wide plan, loudly
        p2 = gpu.NnlsPlan(basis, reg, 0)
        try:
            p2.aty_device(16, yd[:16], None, torch.cuda.current_stream().cuda_stream)
        finally:
            p2.close()


def _strong_regulariser():
    from pyneapple_amd import synth

    cfg = dict(synth.NNLS_CFG, reg_order=1, mu=1.0)  # supports beyond 128 bins on more than half of the voxels
    _, basis, reg = synth.nnls_matrices(32, cfg)
    return basis, reg


def test_pilot_routes_a_strong_regulariser_to_the_four_slot_kernel(gpu, oracle, monkeypatch, capfd):
    """The two-slot block kernel pays twice for a voxel it hands over (passive set beyond 128 positions).  A call of >= 4 x 12 288
    voxels therefore solves a pilot of 12 288 first, and when more than 15 % of those are handed over the rest of the call goes to
    the four-slot instantiation (256 positions) directly -- decided on the device, from the pilot's voxels only.  Same results as
    the two-slot kernel first for everything (PNX_BLK_ROUTE_PERMILLE=0) and as the oracle; the reference's own regulariser stays
    on the two-slot kernel."""
    from pyneapple_amd import synth

    basis, reg = _strong_regulariser()
    _, y, _ = synth.make_numpy("tri_reduced", 4 * 12288 + 777, 32, sigma=0.01, seed=11, scale=1000.0)
    monkeypatch.setenv("PNX_BLK_ROUTE_DEBUG", "1")
    capfd.readouterr()
    routed = gpu.nnls(basis, reg, y, 250)
    err = capfd.readouterr().err
    assert "pnx nnls pilot" in err and "-> four-slot block kernel" in err, err
    monkeypatch.setenv("PNX_BLK_ROUTE_PERMILLE", "0")
    blk = gpu.nnls(basis, reg, y, 250)
    assert "pnx nnls pilot" not in capfd.readouterr().err
    monkeypatch.delenv("PNX_BLK_ROUTE_PERMILLE")
    np.testing.assert_array_equal(routed["status"], blk["status"])
    ok = blk["status"] == 1
    assert _scaled_err(routed["coefficients"][ok], blk["coefficients"][ok]).max() < 1e-9
    np.testing.assert_allclose(routed["residual"], blk["residual"], rtol=1e-12)
    assert (routed["iters"] == blk["iters"]).mean() > 0.999
    # pilot voxels and routed voxels against the oracle
    pick = np.r_[0:768, 12288:12288 + 768, y.shape[0] - 512:y.shape[0]]
    o = oracle.nnls(basis, reg, y[pick], 250, n_threads=8)
    np.testing.assert_array_equal(routed["status"][pick], o["status"])
    oko = o["status"] == 1
    assert _scaled_err(routed["coefficients"][pick][oko], o["coefficients"][oko]).max() < 1e-6
    assert (routed["iters"][pick] == o["iters"]).mean() > 0.99
    # the same call twice: the route depends on the pilot's voxels only
    again = gpu.nnls(basis, reg, y, 250)
    for k in ("coefficients", "residual", "status", "iters"):
        np.testing.assert_array_equal(again[k], routed[k], err_msg=k)
    # the reference's regulariser: the pilot keeps the block kernel
    _, basis2, reg2 = synth.nnls_matrices(32)
    capfd.readouterr()
    gpu.nnls(basis2, reg2, y, 250)
    err = capfd.readouterr().err
    assert "-> block kernel" in err, err


def test_pilot_route_holds_for_the_later_chunks_of_a_host_call(gpu, monkeypatch, capfd):
    """Host arrays in several chunks: the first chunk runs the pilot, the later ones follow its route (same stream, behind it);
    the pilot's handed-over voxels go through the deferred pass.  Same bits as the single-chunk call."""
    from pyneapple_amd import synth

    basis, reg = _strong_regulariser()
    _, y, _ = synth.make_numpy("tri_reduced", 130000, 32, sigma=0.01, seed=12, scale=1000.0)
    plan = gpu.NnlsPlan(basis, reg, 0)
    monkeypatch.setenv("PNX_NNLS_HOST_CHUNK", str(1 << 20))
    one = plan.solve(y, 250)
    monkeypatch.setenv("PNX_NNLS_HOST_CHUNK", "50000")  # three chunks: pilot in the first
    monkeypatch.setenv("PNX_BLK_ROUTE_DEBUG", "1")
    capfd.readouterr()
    many = plan.solve(y, 250)
    err = capfd.readouterr().err
    assert err.count("pnx nnls pilot") == 1 and "-> four-slot block kernel" in err, err
    for k in ("coefficients", "residual", "status", "iters"):
        np.testing.assert_array_equal(many[k], one[k], err_msg=k)
    many32 = plan.solve(y.astype(np.float32), 250)
    one32 = None
    monkeypatch.setenv("PNX_NNLS_HOST_CHUNK", str(1 << 20))
    one32 = plan.solve(y.astype(np.float32), 250)
    for k in ("coefficients", "residual", "status", "iters"):
        np.testing.assert_array_equal(many32[k], one32[k], err_msg=f"{k} float32")
    plan.close()


def test_many_handed_over_voxels_go_through_the_side_buffer_in_batches(gpu, monkeypatch, capfd):
    """A regulariser five to ten times the reference's: several per cent of the voxels are handed over, more than the deferred
    pass's side buffer holds.  The first batch is solved behind the last chunk, the others follow from the caller's array --
    the call does not run again, and the bits are those of the single-chunk call."""
    from pyneapple_amd import synth

    cfg = dict(synth.NNLS_CFG, reg_order=2, mu=0.2)
    _, basis, reg = synth.nnls_matrices(32, cfg)
    _, y, _ = synth.make_numpy("tri_reduced", 110000, 32, sigma=0.01, seed=13, scale=1000.0)
    plan = gpu.NnlsPlan(basis, reg, 0)
    monkeypatch.setenv("PNX_NNLS_HOST_CHUNK", str(1 << 20))
    monkeypatch.setenv("PNX_BLK_ROUTE_DEBUG", "1")
    capfd.readouterr()
    one = plan.solve(y, 250)
    assert "-> block kernel" in capfd.readouterr().err  # 5 - 10 % handed over: below the pilot's threshold of 15 %
    n_over = int(((one["coefficients"] > 0).sum(axis=1) > 128).sum())
    assert n_over > 3000
    monkeypatch.setenv("PNX_NNLS_HOST_CHUNK", "50000")
    monkeypatch.setenv("PNX_NNLS_DEFER_CAP", "2048")
    many = plan.solve(y, 250)
    assert capfd.readouterr().err.count("pnx nnls pilot") == 1  # one pass over the chunks
    for k in ("coefficients", "residual", "status", "iters"):
        np.testing.assert_array_equal(many[k], one[k], err_msg=k)
    many32 = plan.solve(y.astype(np.float32), 250)
    monkeypatch.setenv("PNX_NNLS_HOST_CHUNK", str(1 << 20))
    one32 = plan.solve(y.astype(np.float32), 250)
    for k in ("coefficients", "residual", "status", "iters"):
        np.testing.assert_array_equal(many32[k], one32[k], err_msg=f"{k} float32")
    plan.close()
