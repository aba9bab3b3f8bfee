"""pyneapple_amd -- MI355X (gfx950) backend for Pyneapple's per-voxel fitting hot path.

Importing the package is cheap (no HIP initialisation); the shared library is loaded on first use and
there is deliberately no CPU fallback.
"""
from __future__ import annotations

__version__ = "0.1.0"

__all__ = ["HipCurveFitSolver", "HipNNLSSolver", "MonoExpModel", "BiExpModel", "TriExpModel", "NNLSModel"]


def __getattr__(name):
    if name in ("HipCurveFitSolver", "HipNNLSSolver"):
        from . import solvers

        return getattr(solvers, name)
    if name in ("MonoExpModel", "BiExpModel", "TriExpModel", "NNLSModel"):
        from . import models

        return getattr(models, name)
    raise AttributeError(name)
