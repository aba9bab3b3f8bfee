"""Reference base classes when Pyneapple is importable, interface-identical stand-ins otherwise.

The GPU box (and CI without the reference installed) has no `pyneapple`; there the plugin classes derive
from the stand-ins below, which mirror the reference's solver contract:
  BaseSolver / _PixelFitResult  -- src/pyneapple/solvers/base.py:14-90
When `pyneapple` IS importable the plugin classes derive from the reference's own `CurveFitSolver` /
`NNLSSolver`, because the reference's fitter picks the R^2 path with `isinstance(solver, NNLSSolver)`
(src/pyneapple/fitters/base.py:162-169).
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Any

import numpy as np

try:  # pragma: no cover - exercised only where the reference is installed
    from pyneapple.solvers.base import BaseSolver as RefBaseSolver  # type: ignore
    from pyneapple.solvers.base import _PixelFitResult  # type: ignore
    from pyneapple.solvers.curvefit import CurveFitSolver as RefCurveFitSolver  # type: ignore
    from pyneapple.solvers.nnls_solver import NNLSSolver as RefNNLSSolver  # type: ignore

    HAVE_PYNEAPPLE = True
except Exception:  # ImportError or a missing transitive dependency (loguru, ...)
    HAVE_PYNEAPPLE = False
    RefCurveFitSolver = None
    RefNNLSSolver = None

    @dataclass
    class _PixelFitResult:  # solvers/base.py:14-38
        params: np.ndarray
        covariance: np.ndarray | None = None
        success: bool = True
        message: str | None = None
        n_iterations: int | None = None
        residual: float | None = None

    class RefBaseSolver(ABC):  # solvers/base.py:41-90
        def __init__(self, model: Any, max_iter: int = 250, tol: float = 1e-8, verbose: bool = False,
                     **solver_kwargs):
            self.model = model
            self.max_iter = max_iter
            self.tol = tol
            self.verbose = verbose
            self.diagnostics_: dict[str, Any] = {}
            self.params_: dict[str, Any] = {}
            self.pixel_results_ = []

        @abstractmethod
        def fit(self, *args, **kwargs):
            return self

        def get_diagnostics(self) -> dict[str, Any]:
            if len(self.diagnostics_) == 0:
                raise RuntimeError(
                    "No diagnostics available. Ensure fit() has been called and diagnostics are stored.")
            return self.diagnostics_.copy()

        def get_params(self) -> dict[str, Any]:
            if len(self.params_) == 0:
                raise RuntimeError("No parameters available. Ensure fit() has been called and parameters are stored.")
            return self.params_.copy()

        def _reset_state(self):
            self.diagnostics_ = {}
            self.params_ = {}
            self.pixel_results_ = []


CurveFitBase = RefCurveFitSolver if HAVE_PYNEAPPLE else RefBaseSolver
NNLSBase = RefNNLSSolver if HAVE_PYNEAPPLE else RefBaseSolver


class PixelResultsView:
    """Array-backed, lazily materialised stand-in for `list[_PixelFitResult]`.

    The reference's fitter only needs len(), iteration and indexing of `solver.pixel_results_`
    (fitters/base.py:211-253); building 4.2 M dataclass objects eagerly would dwarf the GPU time.
    """

    def __init__(self, params, covariance, success, messages, residual=None, n_iterations=None):
        self._params = params          # (n_px, n_free)
        self._cov = covariance         # (n_px, n, n) or None
        self._success = success        # (n_px,) bool
        self._messages = messages      # callable i -> str | None
        self._residual = residual      # (n_px,) or None
        self._n_iter = n_iterations    # (n_px,) or None

    def __len__(self):
        return int(self._params.shape[0])

    def _make(self, i: int) -> _PixelFitResult:
        return _PixelFitResult(
            params=self._params[i],
            covariance=None if self._cov is None else self._cov[i],
            success=bool(self._success[i]),
            message=self._messages(i),
            n_iterations=None if self._n_iter is None else int(self._n_iter[i]),
            residual=None if self._residual is None else float(self._residual[i]),
        )

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._make(k) for k in range(*i.indices(len(self)))]
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError(i)
        return self._make(i)

    def __iter__(self):
        for i in range(len(self)):
            yield self._make(i)

    def __bool__(self):
        return len(self) > 0
