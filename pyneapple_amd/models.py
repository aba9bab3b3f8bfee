"""Model descriptors with the reference's model interface (parameter names/order, modes, forward()).

Used where Pyneapple itself is not installed (tests on the GPU box, bench.py).  The plugin solvers accept
either these or the reference's own model objects: they only read `_all_param_names`, `param_names`,
`fixed_params`, the mode flags and (NNLS) `bins`, `n_bins`, `get_basis`.
Interface mirrored from: src/pyneapple/models/base.py:79-230,325-373, monoexp.py, biexp.py, triexp.py, nnls.py.
"""
from __future__ import annotations

import numpy as np


class _Parametric:
    _base_names: list[str] = []

    def __init__(self, fixed_params: dict[str, float] | None = None, fit_t1: bool = False, fit_t1_steam: bool = False,
                 repetition_time: float | None = None, mixing_time: float | None = None, **model_kwargs):
        self.model_kwargs = model_kwargs
        self.fixed_params = dict(fixed_params) if fixed_params else {}
        if fit_t1_steam:
            fit_t1 = True  # STEAM always implies the standard T1 factor (models/monoexp.py:38-39)
        if fit_t1 and repetition_time is None:
            raise ValueError("repetition_time is required when fit_t1=True.")
        if fit_t1_steam and mixing_time is None:
            raise ValueError("mixing_time is required when fit_t1_steam=True.")
        self.fit_t1 = fit_t1
        self.fit_t1_steam = fit_t1_steam
        self.repetition_time = repetition_time
        self.mixing_time = mixing_time

    def _t1(self, signal, T1):
        # model_functions/multiexp.py:210-241
        signal = signal * (1 - np.exp(-self.repetition_time / T1))
        if self.fit_t1_steam:
            signal = signal * np.exp(-self.mixing_time / T1)
        return signal

    def _validate_fixed_params(self):
        if self.fixed_params:
            unknown = set(self.fixed_params) - set(self._all_param_names)
            if unknown:
                raise ValueError(f"Unknown fixed parameters: {unknown}. Valid: {self._all_param_names}")
            if len(self.fixed_params) >= len(self._all_param_names):
                raise ValueError("At least one parameter must remain free.")

    def _base(self) -> list[str]:
        return list(self._base_names)

    @property
    def _all_param_names(self) -> list[str]:
        names = self._base()
        if self.fit_t1 or self.fit_t1_steam:
            names.append("T1")
        return names

    @property
    def param_names(self) -> list[str]:
        if not self.fixed_params:
            return self._all_param_names
        return [p for p in self._all_param_names if p not in self.fixed_params]

    @property
    def n_params(self) -> int:
        return len(self.param_names)

    def _free_indices(self, fixed: dict[str, float]) -> list[int]:
        return [i for i, name in enumerate(self._all_param_names) if name not in fixed]

    def _inject_fixed(self, free_params, fixed):
        if not fixed:
            return tuple(free_params)
        it = iter(free_params)
        return tuple(float(fixed[n]) if n in fixed else next(it) for n in self._all_param_names)

    def forward_with_fixed(self, xdata, fixed_dict, *free_params):
        return self.forward(xdata, *self._inject_fixed(free_params, fixed_dict))


class MonoExpModel(_Parametric):
    """S(b) = S0 exp(-b D); params [S0, D]."""

    _base_names = ["S0", "D"]

    def __init__(self, fixed_params=None, **kw):
        super().__init__(fixed_params, **kw)
        self._validate_fixed_params()

    def forward(self, xdata, *p):
        s = p[0] * np.exp(-xdata * p[1])
        return self._t1(s, p[2]) if self.fit_t1 else s


class BiExpModel(_Parametric):
    """Reduced [f1,D1,D2] (default) / S0 [f1,D1,D2,S0] / full [f1,D1,f2,D2]."""

    def __init__(self, fit_reduced: bool = True, fit_s0: bool = False, fixed_params=None, **kw):
        if fit_s0 and not fit_reduced:
            raise ValueError("fit_s0=True requires fit_reduced=True. Full model with independent fractions and S0 "
                             "is over-parameterized.")
        self.fit_reduced, self.fit_s0 = fit_reduced, fit_s0
        super().__init__(fixed_params, **kw)
        self._validate_fixed_params()

    def _base(self):
        if self.fit_reduced:
            return ["f1", "D1", "D2", "S0"] if self.fit_s0 else ["f1", "D1", "D2"]
        return ["f1", "D1", "f2", "D2"]

    def forward(self, xdata, *p):
        if self.fit_s0:
            s = p[3] * (p[0] * np.exp(-xdata * p[1]) + (1 - p[0]) * np.exp(-xdata * p[2]))
        elif self.fit_reduced:
            s = p[0] * np.exp(-xdata * p[1]) + (1 - p[0]) * np.exp(-xdata * p[2])
        else:
            s = p[0] * np.exp(-xdata * p[1]) + p[2] * np.exp(-xdata * p[3])
        return self._t1(s, p[len(self._base())]) if self.fit_t1 else s


class TriExpModel(_Parametric):
    """Reduced [f1,D1,f2,D2,D3] (default) / S0 [...,S0] / full [f1,D1,f2,D2,f3,D3]."""

    def __init__(self, fit_reduced: bool = True, fit_s0: bool = False, fixed_params=None, **kw):
        if fit_s0 and not fit_reduced:
            raise ValueError("fit_s0=True requires fit_reduced=True. Full model with independent fractions and S0 "
                             "is over-parameterized.")
        self.fit_reduced, self.fit_s0 = fit_reduced, fit_s0
        super().__init__(fixed_params, **kw)
        self._validate_fixed_params()

    def _base(self):
        if self.fit_reduced:
            return ["f1", "D1", "f2", "D2", "D3", "S0"] if self.fit_s0 else ["f1", "D1", "f2", "D2", "D3"]
        return ["f1", "D1", "f2", "D2", "f3", "D3"]

    def forward(self, xdata, *p):
        e = lambda D: np.exp(-xdata * D)
        if self.fit_s0:
            s = p[5] * (p[0] * e(p[1]) + p[2] * e(p[3]) + (1 - p[0] - p[2]) * e(p[4]))
        elif self.fit_reduced:
            s = p[0] * e(p[1]) + p[2] * e(p[3]) + (1 - p[0] - p[2]) * e(p[4])
        else:
            s = p[0] * e(p[1]) + p[2] * e(p[3]) + p[4] * e(p[5])
        return self._t1(s, p[len(self._base())]) if self.fit_t1 else s


class NNLSModel:
    """Distribution model: log-spaced bins, basis[i,j] = exp(-b_i D_j) (models/nnls.py:37-77)."""

    def __init__(self, d_range: tuple[float, float], n_bins: int, **model_kwargs):
        self.model_kwargs = model_kwargs
        self.d_range = d_range
        self.n_bins = n_bins

    @property
    def bins(self) -> np.ndarray:
        return np.logspace(np.log10(self.d_range[0]), np.log10(self.d_range[1]), self.n_bins)

    def get_basis(self, xdata: np.ndarray) -> np.ndarray:
        if xdata.ndim != 1:
            raise ValueError(f"xdata must be a 1D array of shape (n_measurements,), but got shape {xdata.shape}")
        return np.exp(-xdata.reshape(-1, 1) * self.bins.reshape(1, -1))

    def forward(self, xdata, *spectrum):
        return self.get_basis(xdata) @ np.asarray(spectrum)
