import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from spartacus_surface_amd import radsurf, types
ncol = 200000
case = bench.make_case("c3", ncol, 999)
cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
bc = types.BoundaryCondsOut(); bc.allocate(cp.ncol, cfg.nsw, cfg.nlw)
fl = []
for nspec, direct in ((cfg.nsw, True), (cfg.nsw, False), (cfg.nlw, False), (cfg.nlw, False)):
    f = types.CanopyFlux(); f.allocate(cfg, cp.ncol, cp.ntotlay, nspec, use_direct=direct, do_save_flux_profile=False); fl.append(f)
for i in range(4):
    t0 = time.perf_counter()
    radsurf.radsurf(cfg, cp, sw, lw, bc, 1, cp.ncol, *fl)
    print(os.environ.get("SPARTACUS_HOST_ONE_STREAM"), os.environ.get("SPARTACUS_HOST_CHUNK"), "call", i, "%.2f ms" % (1e3 * (time.perf_counter() - t0)), flush=True)
