import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyneapple_amd import api, synth, _lib
_lib.load()
dev = torch.device("cuda", 0)
names, p0, lo, hi = synth.shared_arrays("tri_reduced")
for n in (1 << 22, 3 << 20, 1 << 21, 3 << 19, 1 << 20, 1 << 19, 1 << 18, 1 << 17):
    b, y = synth.make_torch_rows("tri_reduced", 0, n, 32, dev, sigma=0.01)
    nn = len(names)
    popt = torch.empty((nn, n), dtype=torch.float64, device=dev); pcov = torch.empty((n, nn, nn), dtype=torch.float64, device=dev)
    st = torch.empty(n, dtype=torch.int8, device=dev); nf = torch.empty(n, dtype=torch.int32, device=dev); cost = torch.empty(n, dtype=torch.float64, device=dev)
    opts = api.make_opts("tri_reduced", 32, [], False, False, 250, 1e-8, 1e-8, 1e-8, "fd", 0, 0.0, 0.0)
    s = torch.cuda.current_stream().cuda_stream
    for pc in (pcov, None):
        for _ in range(2):
            api.curvefit_device(opts, n, b, y, p0, lo, hi, None, popt, pc, st, nf, cost, 0, s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            api.curvefit_device(opts, n, b, y, p0, lo, hi, None, popt, pc, st, nf, cost, 0, s)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(n, "pcov" if pc is not None else "nopcov", f"{ms:.3f} ms  {n / ms / 1e3:.1f} M voxels/s  mean nfev {nf.double().mean().item():.2f}", flush=True)
    del y, popt, pcov, st, nf, cost
