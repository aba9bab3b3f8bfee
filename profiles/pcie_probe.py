import time, torch, numpy as np, os
from concurrent.futures import ThreadPoolExecutor
n = 1 << 27  # 1 GiB of float64
dev = torch.device("cuda", 0)
a = torch.from_numpy(np.random.default_rng(0).random(n))
d = torch.empty(n, dtype=torch.float64, device=dev)
def t(f, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
gb = n * 8 / 1e9
print("cpus", len(os.sched_getaffinity(0)))
print("H2D pageable  %.1f GB/s" % (gb / t(lambda: d.copy_(a))))
out = torch.empty(n, dtype=torch.float64)
print("D2H pageable  %.1f GB/s" % (gb / t(lambda: out.copy_(d))))
t0 = time.perf_counter(); p = torch.empty(n, dtype=torch.float64).pin_memory(); print("pin_memory alloc 1GiB %.1f ms" % ((time.perf_counter() - t0) * 1e3))
print("H2D pinned    %.1f GB/s" % (gb / t(lambda: d.copy_(p, non_blocking=True))))
print("D2H pinned    %.1f GB/s" % (gb / t(lambda: p.copy_(d, non_blocking=True))))
rt = torch.cuda.cudart()
b = torch.from_numpy(np.random.default_rng(1).random(n))
t0 = time.perf_counter(); r = rt.cudaHostRegister(b.data_ptr(), n * 8, 0); dt = time.perf_counter() - t0
print("hostRegister 1GiB rc=%s %.1f ms" % (r, dt * 1e3))
print("H2D registered %.1f GB/s" % (gb / t(lambda: d.copy_(b, non_blocking=True))))
t0 = time.perf_counter(); rt.cudaHostUnregister(b.data_ptr()); print("unregister %.1f ms" % ((time.perf_counter() - t0) * 1e3))
# CPU memcpy speed, 1..8 threads
src = a.numpy(); dst = p.numpy()
for nt in (1, 2, 4, 8):
    parts = np.array_split(np.arange(n), nt)
    bounds = [(int(x[0]), int(x[-1]) + 1) for x in parts]
    def cp(bd): dst[bd[0]:bd[1]] = src[bd[0]:bd[1]]
    with ThreadPoolExecutor(nt) as ex:
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); list(ex.map(cp, bounds)); best = min(best, time.perf_counter() - t0)
    print("memcpy pageable->pinned %d threads %.1f GB/s" % (nt, gb / best))
# bidirectional concurrently on two streams
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
p2 = torch.empty(n, dtype=torch.float64).pin_memory(); d2 = torch.empty(n, dtype=torch.float64, device=dev)
def both():
    with torch.cuda.stream(s1): d.copy_(p, non_blocking=True)
    with torch.cuda.stream(s2): p2.copy_(d2, non_blocking=True)
print("bidir pinned  %.1f GB/s each way" % (gb / t(both)))
