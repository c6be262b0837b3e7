"""The driver's default pattern (driver/spartacus_surface_driver.F90:206-234): radsurf called with blocks of
`nblock` columns, one after the other.  Prints the time per call with SPARTACUS_HOST_TIMING on one block.
usage: python tools/host_entry_blocks.py [nblock] [ncol]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from spartacus_surface_amd import radsurf, types
nblock = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
case = bench.make_case("c3", ncol, 999)
cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
bc = types.BoundaryCondsOut(); bc.allocate(cp.ncol, cfg.nsw, cfg.nlw)
fl = []
for nspec, direct in ((cfg.nsw, True), (cfg.nsw, False), (cfg.nlw, False), (cfg.nlw, False)):
    f = types.CanopyFlux(); f.allocate(cfg, cp.ncol, cp.ntotlay, nspec, use_direct=direct, do_save_flux_profile=True); fl.append(f)
radsurf.radsurf(cfg, cp, sw, lw, bc, 1, nblock, *fl)      # plans, kernels loaded
for rep in range(2):
    t0 = time.perf_counter()
    for i0 in range(1, ncol + 1, nblock):
        radsurf.radsurf(cfg, cp, sw, lw, bc, i0, min(ncol, i0 + nblock - 1), *fl)
    dt = time.perf_counter() - t0
    print("%d-column blocks: %.3f ms per call, %.3e columns/s" % (nblock, 1e3 * dt / ((ncol + nblock - 1) // nblock), ncol / dt), flush=True)
os.environ["SPARTACUS_HOST_TIMING"] = "1"
radsurf.radsurf(cfg, cp, sw, lw, bc, 17, 32, *fl)

# concurrent callers (the driver's OpenMP loop): T threads take blocks from a shared counter
import threading
os.environ.pop("SPARTACUS_HOST_TIMING", None)
for T in (2, 4, 8, 16):
    nblocks = (ncol + nblock - 1) // nblock
    counter = iter(range(nblocks))
    lock = threading.Lock()
    def work():
        while True:
            with lock:
                b = next(counter, None)
            if b is None:
                return
            i0 = 1 + b * nblock
            radsurf.radsurf(cfg, cp, sw, lw, bc, i0, min(ncol, i0 + nblock - 1), *fl)
    ts = [threading.Thread(target=work) for _ in range(T)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.perf_counter() - t0
    print("%2d threads, %d-column blocks: %.3e columns/s" % (T, nblock, ncol / dt), flush=True)
