"""All call times of the host entry in its three locking modes, twice over (GPU box).
usage: python tools/host_entry_modes.py [ncol]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from spartacus_surface_amd import radsurf, types
ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
case = bench.make_case("c3", ncol, 999)
cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
bc = types.BoundaryCondsOut(); bc.allocate(cp.ncol, cfg.nsw, cfg.nlw)
fl = []
for nspec, direct in ((cfg.nsw, True), (cfg.nsw, False), (cfg.nlw, False), (cfg.nlw, False)):
    f = types.CanopyFlux(); f.allocate(cfg, cp.ncol, cp.ntotlay, nspec, use_direct=direct, do_save_flux_profile=False); fl.append(f)
def calls(n):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); radsurf.radsurf(cfg, cp, sw, lw, bc, 1, cp.ncol, *fl); ts.append(1e3 * (time.perf_counter() - t0))
    return " ".join("%.1f" % t for t in ts)
for rep in range(2):
    os.environ.pop("SPARTACUS_PIN_HOST", None)
    print("default (call-owned locks for large calls):", calls(5))
    os.environ["SPARTACUS_PIN_HOST"] = "0"
    print("pageable (SPARTACUS_PIN_HOST=0):", calls(4))
    os.environ.pop("SPARTACUS_PIN_HOST", None)
    arrays = radsurf.host_arrays(cfg, cp, sw, lw, bc, *fl)
    for a in arrays: radsurf.host_pin(a)
    print("caller-owned locks (ss_host_pin):", calls(5))
    radsurf.host_unpin_all()
