import os, time, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from spartacus_surface_amd import radsurf, workloads, types
from oracle import checker
radsurf.load_library()
case = workloads.c3_vegetated_urban(ncol=200000, seed=1)
cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
out = checker.alloc_outputs(cfg, cp, False)
for rep in range(3):
    t0 = time.perf_counter()
    plan = radsurf.Plan(cfg, cp, sw, lw, want_profiles=False)
    plan.synchronize()
    t1 = time.perf_counter()
    plan.solve(); plan.synchronize()
    t2 = time.perf_counter()
    plan.download(out["bc_out"], out["sw_norm_dir"], out["sw_norm_diff"], out["lw_internal"], out["lw_norm"])
    t3 = time.perf_counter()
    plan.close()
    t4 = time.perf_counter()
    print("create+upload %.1f ms  solve %.1f  download %.1f  destroy %.1f" % (1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(t4-t3)))
t0 = time.perf_counter()
radsurf.radsurf(cfg, cp, sw, lw, out["bc_out"], 1, cp.ncol, out["sw_norm_dir"], out["sw_norm_diff"], out["lw_internal"], out["lw_norm"])
print("host entry %.1f ms" % (1e3*(time.perf_counter()-t0)))
