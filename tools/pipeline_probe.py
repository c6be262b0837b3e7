"""Can H2D, kernels and D2H of alternating streams overlap on this box?  (torch, pinned memory)
Emulates the host entry: per chunk 35 uploads (102 MB), ~1.3 ms of kernels, 60 downloads (146 MB)."""
import time
import torch
torch.cuda.init()
dev = "cuda"
def mk(nbytes, pieces, pin=True):
    n = nbytes // 8 // pieces
    hs = [torch.empty(n, dtype=torch.float64).pin_memory() if pin else torch.empty(n, dtype=torch.float64) for _ in range(pieces)]
    ds = [torch.empty(n, dtype=torch.float64, device=dev) for _ in range(pieces)]
    return hs, ds
import sys
nchunk = 4
PU, PD = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (35, 60)
up = [mk(102_000_000, PU) for _ in range(2)]
dn = [mk(146_000_000, PD) for _ in range(2)]
a = torch.randn(2048, 2048, device=dev)
def kernels():
    x = a
    for _ in range(8):
        x = x @ a
    return x
print("pieces per chunk: %d up, %d down" % (PU, PD))
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def chunk(s, do_up=True, do_k=True, do_dn=True):
    with torch.cuda.stream(streams[s]):
        if do_up:
            for h, d in zip(*up[s]):
                d.copy_(h, non_blocking=True)
        if do_k:
            kernels()
        if do_dn:
            for h, d in zip(*dn[s]):
                h.copy_(d, non_blocking=True)
def run(**kw):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(nchunk):
            s = k & 1
            streams[s].synchronize()
            chunk(s, **kw)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3
print("kernels only      %.2f ms" % run(do_up=False, do_dn=False))
print("uploads only      %.2f ms" % run(do_k=False, do_dn=False))
print("downloads only    %.2f ms" % run(do_up=False, do_k=False))
print("up + kernels      %.2f ms" % run(do_dn=False))
print("up + down         %.2f ms" % run(do_k=False))
print("all three         %.2f ms" % run())
