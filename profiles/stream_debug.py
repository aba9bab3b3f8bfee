"""Streamed host path vs chunk ring: where do they differ?  (debug aid for pnx_api.hip curvefit_streamed)"""
import os as _os; _os.environ.setdefault("PNX_ENABLE_TEST_HOOKS", "1")  # this script drives developer switches of the library (include/pnx.h, "Environment")
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyneapple_amd import api, synth

n_vox = int(os.environ.get("N", 40037))
model, n_b = "tri_reduced", 32
b, y, _ = synth.make_numpy(model, n_vox, n_b, sigma=0.01, seed=11)
names, p0, lo, hi = synth.shared_arrays(model)
os.environ["PNX_HOST_STREAM"] = "0"
ring = api.curvefit(model, b, y, p0, lo, hi)
os.environ["PNX_HOST_STREAM"] = "1"
for shift, piece in ((10, 1024), (10, 1024), (12, 3000), (14, 16384), (10, 1 << 20)):
    os.environ["PNX_STREAM_GRANULE_SHIFT"] = str(shift)
    os.environ["PNX_STREAM_IN_CHUNK"] = str(piece)
    st = api.curvefit(model, b, y, p0, lo, hi)
    bad = np.flatnonzero((st["popt"] != ring["popt"]).any(axis=0) | (st["status"] != ring["status"]) | (st["nfev"] != ring["nfev"]))
    print(f"shift {shift} piece {piece}: {bad.size} voxels differ", flush=True)
    for v in bad[:12]:
        print(f"   vox {v} (granule {v >> shift}, offset {v & ((1 << shift) - 1)}): stream popt {st['popt'][:, v]} status {st['status'][v]} nfev {st['nfev'][v]} cost {st['cost'][v]:.3e}"
              f" | ring popt {ring['popt'][:, v]} status {ring['status'][v]} nfev {ring['nfev'][v]}")
    pc = np.flatnonzero(~np.isclose(st["pcov"], ring["pcov"], rtol=0, atol=0, equal_nan=True).all(axis=(1, 2)))
    print(f"   pcov differs on {pc.size} voxels; first {pc[:8]}")
