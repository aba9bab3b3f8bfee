import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
from pyneapple_amd import _lib, api, synth
_lib.load()
dev = torch.device("cuda", 0)
model, n_b, shape = synth.WORKLOADS["triexp"]
n_vox = int(np.prod(shape))
b, y = synth.make_torch_rows(model, 0, n_vox, n_b, dev, sigma=0.01, dtype=torch.float32)
names, p0, _, _ = synth.shared_arrays(model)
n = len(names); ntri = n * (n + 1) // 2
params = torch.tensor(p0, dtype=torch.float32, device=dev)[:, None].repeat(1, n_vox).contiguous()
params *= 1.0 + 0.05 * torch.rand_like(params)
cost = torch.empty(n_vox, dtype=torch.float32, device=dev); g = torch.empty((n, n_vox), dtype=torch.float32, device=dev); h = torch.empty((ntri, n_vox), dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
def grp(reps):
    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ea.record()
    for _ in range(reps):
        api.sweep_device(model, n_vox, b, y, params, cost, g, h, 0, st)
    eb.record(); torch.cuda.synchronize()
    return ea.elapsed_time(eb) / reps
grp(2)
print("groups of 20:", [round(grp(20), 4) for _ in range(5)])
print("groups of 500:", [round(grp(500), 4) for _ in range(8)])
