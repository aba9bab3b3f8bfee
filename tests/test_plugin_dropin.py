"""The real drop-in: the reference's own TOML loader resolves `hip_curvefit` / `hip_nnls` / `hip_pixelwise` / `hip_ideal`
through entry points and builds the fitter (io/toml.py:106-148, 202-236, 328-338), with the extra `[Fitting.solver]`
keys forwarded as constructor kwargs.  Modelled on the reference's tests/test_io_plugin_discovery.py:84-148.

Runs only where the reference is present (/root/reference/src: the build container), in a child process -- the plugin
classes pick their base classes at import time (pyneapple_amd/_compat.py), and this test process has already imported
them without the reference.  The two `sys.modules` stand-ins for loguru / cv2 are SURVEY.md Appendix B's; they live
here, in test code, not in the product."""
from __future__ import annotations

import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/src"

CHILD = r'''
import os, sys, types
sys.dont_write_bytecode = True
class _Noop:
    def __getattr__(self, name):
        return lambda *a, **k: None
loguru = types.ModuleType("loguru"); loguru.logger = _Noop(); sys.modules["loguru"] = loguru
cv2 = types.ModuleType("cv2"); cv2.INTER_LINEAR = 1; cv2.INTER_CUBIC = 2; sys.modules["cv2"] = cv2
for name in ("nibabel", "h5py"):
    try:
        __import__(name)
    except Exception:
        sys.modules[name] = types.ModuleType(name)
sys.path.insert(0, REF_SRC); sys.path.insert(0, ROOT)
import numpy as np
try:
    import tomllib
except ModuleNotFoundError:
    import tomli as tomllib
from importlib.metadata import EntryPoint
import pyneapple.io.toml as T
from pyneapple.fitters.base import BaseFitter
from pyneapple.solvers import CurveFitSolver, NNLSSolver

# what `pip install pyneapple-amd` registers: the entry points of OUR pyproject.toml, discovered the reference's way
with open(os.path.join(ROOT, "pyproject.toml"), "rb") as fh:
    eps = tomllib.load(fh)["project"]["entry-points"]
def fake_entry_points(group):
    return [EntryPoint(name=k, value=v, group=group) for k, v in eps.get(group, {}).items()]
T.entry_points = fake_entry_points
T._discover_plugins("pyneapple.solvers", T._SOLVER_REGISTRY)
T._discover_plugins("pyneapple.fitters", T._FITTER_REGISTRY)
assert {"hip_curvefit", "hip_nnls"} <= set(T._SOLVER_REGISTRY) and {"hip_pixelwise", "hip_ideal"} <= set(T._FITTER_REGISTRY)
assert isinstance(T._SOLVER_REGISTRY["hip_curvefit"], EntryPoint)   # lazy: nothing of the plugin imported yet
assert "pyneapple_amd.solvers" not in sys.modules

cfg = os.path.join(TMP, "tri.toml")
open(cfg, "w").write("""
[Fitting]
fitter = "hip_pixelwise"
[Fitting.model]
type = "triexp"
[Fitting.solver]
type = "hip_curvefit"
max_iter = 250
tol = 1e-8
device = 0
n_gpus = 1
jacobian = "fd"
io_dtype = "float64"
xtol = 1e-9
multi_threading = true
n_pools = 4
[Fitting.solver.p0]
f1 = 0.2
D1 = 0.05
f2 = 0.3
D2 = 0.005
D3 = 0.001
[Fitting.solver.bounds]
f1 = [0.0, 1.0]
D1 = [0.01, 0.5]
f2 = [0.0, 1.0]
D2 = [0.002, 0.01]
D3 = [0.00001, 0.002]
""")
fitter = T.load_config(cfg).build_fitter()
from pyneapple_amd import _compat, _lib
from pyneapple_amd.fitters import HipPixelWiseFitter
from pyneapple_amd.ideal import HipIDEALFitter
from pyneapple_amd.solvers import HipCurveFitSolver, HipNNLSSolver
assert _compat.HAVE_PYNEAPPLE
assert type(fitter) is HipPixelWiseFitter and isinstance(fitter, BaseFitter)
s = fitter.solver
assert type(s) is HipCurveFitSolver and isinstance(s, CurveFitSolver)
assert issubclass(HipNNLSSolver, NNLSSolver)      # fitters/base.py:162-169 picks the R^2 path with isinstance
assert (s.device, s.n_gpus, s.jacobian_mode, s.xtol, s.n_pools, s.multi_threading) == (0, 1, "fd", 1e-9, 4, True)
assert s.max_iter == 250 and s.tol == 1e-8 and s.p0["D3"] == 0.001 and s.bounds["D2"] == (0.002, 0.01)
assert T._SOLVER_REGISTRY["hip_curvefit"] is HipCurveFitSolver   # cached after the first resolve

cfg = os.path.join(TMP, "nnls.toml")
open(cfg, "w").write("""
[Fitting]
fitter = "pixelwise"
[Fitting.model]
type = "nnls"
d_range = [0.0008, 0.5]
n_bins = 250
[Fitting.solver]
type = "hip_nnls"
reg_order = 2
mu = 0.02
max_iter = 250
tol = 1e-8
multi_threading = false
device = 0
""")
f2 = T.load_config(cfg).build_fitter()             # the reference's own PixelWiseFitter around the plugin solver
assert type(f2.solver) is HipNNLSSolver and isinstance(f2.solver, NNLSSolver)
assert (f2.solver.reg_order, f2.solver.mu, f2.solver.device) == (2, 0.02, 0)
np.testing.assert_array_equal(f2.solver.get_regularization_matrix().shape, (250, 250)) if _lib.device_count() else None

# a plugin fitter other than the built-in "ideal" is built as cls(solver=solver) (io/toml.py:236)
cfg = os.path.join(TMP, "ideal.toml")
open(cfg, "w").write(open(os.path.join(TMP, "tri.toml")).read().replace('"hip_pixelwise"', '"hip_ideal"'))
f3 = T.load_config(cfg).build_fitter()
assert type(f3) is HipIDEALFitter and isinstance(f3, BaseFitter) and f3.dim_steps is None

# an unsupported curve_fit argument in [Fitting.solver] is refused at construction, not dropped
bad = os.path.join(TMP, "bad.toml")
open(bad, "w").write(open(os.path.join(TMP, "tri.toml")).read().replace("xtol = 1e-9", 'loss = "soft_l1"'))
try:
    T.load_config(bad).build_fitter()
    raise SystemExit("loss='soft_l1' was accepted")
except ValueError as e:
    assert "loss" in str(e)

# no GPU here: the fit fails loudly through the reference's fitter (no CPU fallback, no oracle)
if _lib.device_count() == 0:
    b = np.linspace(0, 1200, 32)
    img = np.ones((2, 2, 1, 32))
    for f in (fitter, f2):
        try:
            f.fit(b, img)
            raise SystemExit("fit succeeded without a GPU")
        except _lib.PnxError as e:
            assert e.code == -3
assert "oracle" not in " ".join(sys.modules)
print("DROPIN-OK")
'''


@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="the reference is not present on this host")
def test_reference_toml_loader_builds_the_plugins(tmp_path):
    code = f"REF_SRC = {REF_SRC!r}\nROOT = {ROOT!r}\nTMP = {str(tmp_path)!r}\n" + textwrap.dedent(CHILD)
    env = dict(os.environ, PYNEAPPLE_QUIET="1", PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "DROPIN-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
