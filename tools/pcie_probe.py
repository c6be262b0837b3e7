"""Raw PCIe rates of the box (pinned host memory): H2D, D2H, both at once."""
import time
import torch
n = 64 * 1024 * 1024   # doubles: 512 MiB
h1 = torch.empty(n, dtype=torch.float64).pin_memory()
h2 = torch.empty(n, dtype=torch.float64).pin_memory()
d1 = torch.empty(n, dtype=torch.float64, device="cuda")
d2 = torch.empty(n, dtype=torch.float64, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def t(f, rep=3):
    best = 1e9
    for _ in range(rep):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best
gb = n * 8 / 1e9
a = t(lambda: d1.copy_(h1, non_blocking=True))
b = t(lambda: h2.copy_(d2, non_blocking=True))
def both():
    with torch.cuda.stream(s1):
        d1.copy_(h1, non_blocking=True)
    with torch.cuda.stream(s2):
        h2.copy_(d2, non_blocking=True)
c = t(both)
print("H2D %.1f GB/s, D2H %.1f GB/s, both at once %.1f GB/s aggregate (%.1f ms for 2 x %.2f GB)" % (gb / a, gb / b, 2 * gb / c, c * 1e3, gb))
# many small copies: 64 x 8 MiB
parts = [(h1[i * (n // 64):(i + 1) * (n // 64)], d1[i * (n // 64):(i + 1) * (n // 64)]) for i in range(64)]
e = t(lambda: [d.copy_(h, non_blocking=True) for h, d in parts])
print("H2D in 64 pieces %.1f GB/s" % (gb / e))
hp = torch.empty(n, dtype=torch.float64)
f = t(lambda: d1.copy_(hp))
g = t(lambda: hp.copy_(d1))
print("pageable H2D %.1f GB/s, D2H %.1f GB/s" % (gb / f, gb / g))
# registered (hipHostRegister) pageable memory, as SPARTACUS_PIN_HOST=1 produces
import numpy as np
rt = torch.cuda.cudart()
a1 = np.empty(n); a2 = np.empty(n)
a1[:] = 1.0; a2[:] = 2.0
assert int(rt.cudaHostRegister(a1.ctypes.data, a1.nbytes, 0)) == 0
assert int(rt.cudaHostRegister(a2.ctypes.data, a2.nbytes, 0)) == 0
r1, r2 = torch.from_numpy(a1), torch.from_numpy(a2)
print("registered: is_pinned", r1.is_pinned())
a = t(lambda: d1.copy_(r1, non_blocking=True))
b = t(lambda: r2.copy_(d2, non_blocking=True))
def both_r():
    with torch.cuda.stream(s1):
        d1.copy_(r1, non_blocking=True)
    with torch.cuda.stream(s2):
        r2.copy_(d2, non_blocking=True)
c = t(both_r)
print("registered: H2D %.1f GB/s, D2H %.1f GB/s, both at once %.1f GB/s aggregate" % (gb / a, gb / b, 2 * gb / c))
# is the H2D call itself asynchronous? time of the call vs time to completion
torch.cuda.synchronize(); t0 = time.perf_counter(); d1.copy_(r1, non_blocking=True); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("registered H2D: call returns after %.2f ms, done after %.2f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t0)))
torch.cuda.synchronize(); t0 = time.perf_counter(); d1.copy_(h1, non_blocking=True); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("hipHostMalloc H2D: call returns after %.2f ms, done after %.2f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t0)))
