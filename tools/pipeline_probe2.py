"""Three-stream pipeline (upload stream, compute stream, download stream, events between them)
against the two alternating all-in-one streams of pipeline_probe.py."""
import time
import torch
dev = "cuda"
def mk(nbytes, pieces):
    n = nbytes // 8 // pieces
    return ([torch.empty(n, dtype=torch.float64).pin_memory() for _ in range(pieces)],
            [torch.empty(n, dtype=torch.float64, device=dev) for _ in range(pieces)])
nchunk, nslot = 4, 2
up = [mk(102_000_000, 8) for _ in range(nslot)]
dn = [mk(146_000_000, 8) for _ in range(nslot)]
a = torch.randn(2048, 2048, device=dev)
def kernels():
    x = a
    for _ in range(8):
        x = x @ a
U, C, D = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
def run3():
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        done = [None] * nslot
        for k in range(nchunk):
            s = k % nslot
            with torch.cuda.stream(U):
                if done[s] is not None:
                    U.wait_event(done[s])          # slot free: its previous results are home
                for h, d in zip(*up[s]):
                    d.copy_(h, non_blocking=True)
                eu = torch.cuda.Event(); eu.record(U)
            with torch.cuda.stream(C):
                C.wait_event(eu)
                kernels()
                ec = torch.cuda.Event(); ec.record(C)
            with torch.cuda.stream(D):
                D.wait_event(ec)
                for h, d in zip(*dn[s]):
                    h.copy_(d, non_blocking=True)
                done[s] = torch.cuda.Event(); done[s].record(D)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3
print("three streams (U, C, D) with events: %.2f ms for %d chunks" % (run3(), nchunk))
# pure direction streams without dependencies: 4 uploads on U, 4 downloads on D at once
def pure():
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(nchunk):
            s = k % nslot
            with torch.cuda.stream(U):
                for h, d in zip(*up[s]):
                    d.copy_(h, non_blocking=True)
            with torch.cuda.stream(D):
                for h, d in zip(*dn[s]):
                    h.copy_(d, non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3
print("independent upload and download streams, no kernels: %.2f ms (sum of the two alone: ~17.4)" % pure())
