"""C3 from numpy arrays (PCIe inclusive): streamed host path (one persistent kernel, pnx_api.hip curvefit_streamed) against
the chunk ring, over granule sizes and upload piece sizes.  python profiles/stream_sweep.py [f32]"""
import os as _os; _os.environ.setdefault("PNX_ENABLE_TEST_HOOKS", "1")  # this script drives developer switches of the library (include/pnx.h, "Environment")
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyneapple_amd import api, synth

dt = np.float32 if "f32" in sys.argv[1:] else np.float64
n = 256 * 256 * 64
b, y, _ = synth.make_numpy("tri_reduced", n, 32, sigma=0.01)
y = y.astype(dt)
names, p0, lo, hi = synth.shared_arrays("tri_reduced")


def run(reps=4):
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); r = api.curvefit("tri_reduced", b, y, p0, lo, hi); ts.append(time.perf_counter() - t)
        keep = r["popt"][:, ::4099].copy(); del r
    return ts, keep


os.environ["PNX_HOST_STREAM"] = "0"
run(1)
ts, ref = run()
print(f"ring: {[round(t * 1e3, 1) for t in ts]} ms", flush=True)
os.environ["PNX_HOST_STREAM"] = "1"
for shift, piece, outs in ((17, 1 << 17, 1), (17, 1 << 17, 2), (17, 1 << 17, 3), (18, 1 << 17, 1), (18, 1 << 17, 2), (18, 1 << 17, 3),
                           (19, 1 << 17, 2), (19, 1 << 19, 2), (16, 1 << 17, 2), (16, 1 << 17, 4), (18, 1 << 18, 2)):
    os.environ["PNX_STREAM_GRANULE_SHIFT"] = str(shift)
    os.environ["PNX_STREAM_IN_CHUNK"] = str(piece)
    os.environ["PNX_STREAM_OUT_THREADS"] = str(outs)
    ts, keep = run()
    print(f"streamed, granule 2^{shift}, upload piece {piece >> 10} Ki, {outs} download threads: {[round(t * 1e3, 1) for t in ts]} ms  best {n / min(ts) / 1e6:.1f} M voxels/s"
          f"  same as ring: {bool(np.array_equal(keep, ref))}", flush=True)
os.environ["PNX_STREAM_GRANULE_SHIFT"] = "17"; os.environ["PNX_STREAM_IN_CHUNK"] = str(1 << 17); os.environ["PNX_STREAM_OUT_THREADS"] = "2"
os.environ["PNX_HOST_TRACE"] = "1"
run(1)
