"""NNLS spectrum post-processing for whole volumes on the device.

Array-level counterparts of `pyneapple.utility.spectrum` (reference src/pyneapple/utility/spectrum.py:13-215): the
reference analyses one spectrum per call (scipy.signal.find_peaks / peak_widths in a Python loop over voxels); here one
call handles every voxel (`pnx_nnls_spectrum_peaks_f64`), and `HipNNLSSolver.fit_peaks` goes from signals to peak tables
without the (n_voxels, n_bins) spectra ever leaving the GPU.  Same names, same conventions (including
`geometric_mean_peak` returning log10 of the merged position); results per voxel are NaN padded to a fixed width.
"""
from __future__ import annotations

import numpy as np

from . import api


def find_spectrum_peaks_batch(spectra, bins, height: float = 0.1, regularized: bool = False, max_peaks: int = 8,
                              rel_height: float = 0.5, device: int = 0):
    """(d_values, f_values, n_peaks): rows are `find_spectrum_peaks(spectra[i], bins, height, regularized)` NaN padded to
    `max_peaks` columns; n_peaks[i] is the number of peaks found (spectrum.py:51-104)."""
    r = api.spectrum_peaks(spectra, bins, height=height, regularized=regularized, rel_height=rel_height, max_peaks=max_peaks,
                           device=device)
    return r["d_values"], r["f_values"], r["n_peaks"]


def find_spectrum_peaks(spectrum, bins, height: float = 0.1, regularized: bool = False):
    """One spectrum, the reference's signature and return shape (arrays of the detected peaks only)."""
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        d, f, n = find_spectrum_peaks_batch(np.asarray(spectrum, float)[None, :], bins, height, regularized,
                                            max_peaks=api.SPECTRUM_MAX_PEAKS)
    if int(n[0]) > api.SPECTRUM_MAX_PEAKS or (int(n[0]) > 0 and np.isnan(d[0, 0])):
        raise ValueError(f"the spectrum has {int(n[0])} peaks at height {height}; the device kernel keeps at most "
                         f"{api.SPECTRUM_MAX_PEAKS} per voxel (16 for a spectrum with a flat-topped rise): raise `height`, or "
                         f"regularise the fit")
    k = int(n[0])
    return d[0, :k].copy(), f[0, :k].copy()


def apply_cutoffs_batch(spectra, bins, cutoffs, height: float = 0.1, regularized: bool = False, device: int = 0):
    """(d_cut, f_cut), each (n_voxels, len(cutoffs)): `apply_cutoffs(*find_spectrum_peaks(...), cutoffs)` per voxel
    (spectrum.py:142-215): NaN where a range holds no peak, merged peaks as geometric_mean_peak."""
    r = api.spectrum_peaks(spectra, bins, height=height, regularized=regularized, max_peaks=0, cutoffs=cutoffs, device=device)
    return r["d_cut"], r["f_cut"]
