#!/usr/bin/env python3
"""Where a seeded workload of tools/fuzz_parity.py differs most from exact arithmetic on the GPU (GPU box, TEST tool).
usage: fuzz_worst.py WORKLOAD SEED NCOL VARIABLE [VARIABLE ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import checker                      # noqa: E402
from spartacus_surface_amd import radsurf, workloads   # noqa: E402
from tests import fixture_io                     # noqa: E402

name, seed, ncol = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
build = {"c3": lambda: workloads.c3_vegetated_urban(ncol=ncol, seed=seed),
         "c4": lambda: workloads.c4_mixed(ncol=ncol, seed=seed, nsw=3, nlw=3),
         "ragged": lambda: workloads.ragged_all_tiles(ncol=ncol, seed=seed)}[name]
case = build()
cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
plan = radsurf.Plan(cfg, cp, sw, lw, want_profiles=True)
plan.solve()
out = checker.alloc_outputs(cfg, cp, True)
plan.download(out["bc_out"], out["sw_norm_dir"], out["sw_norm_diff"], out["lw_internal"], out["lw_norm"])
plan.close()
got = fixture_io.flatten_outputs(out)
ref = fixture_io.flatten_outputs(checker.run_oracle(cfg, cp, sw, lw, want_profiles=True, nthreads=16)[0])
truth = fixture_io.flatten_outputs(checker.run_oracle(cfg, cp, sw, lw, want_profiles=True, nthreads=16, quad=True)[0])
for k in sys.argv[4:]:
    g, r, t = got[k], ref[k], truth[k]
    sc = np.max(np.abs(r))
    eg, er = np.abs(g - t) / sc, np.abs(r - t) / sc
    order = np.argsort(-eg.ravel())[:3]
    print(k, "scale %.3e  GPU-vs-exact max %.3e  reference-vs-exact max %.3e" % (sc, eg.max(), er.max()))
    for o in order:
        i = np.unravel_index(o, eg.shape)
        lay = int(i[0])
        col = int(np.searchsorted(np.asarray(cp.istartlay) - 1, lay, side="right") - 1) if g.shape[0] == cp.ntotlay else lay
        l0, n = int(cp.istartlay[col]) - 1, int(cp.nlay[col])
        print("  element", tuple(int(x) for x in i), "GPU %.10e ref %.10e exact %.10e | GPU err %.2e ref err %.2e | column %d repr %d layer %d of %d cos_sza %.4f" % (
            g[i], r[i], t[i], eg[i], er[i], col, int(cp.i_representation[col]), lay - l0, n, float(cp.cos_sza[col])))
        for nm in ("dz", "veg_fraction", "veg_ext", "building_fraction"):
            a = getattr(cp, nm)
            if a is not None:
                print("     ", nm, np.array2string(np.asarray(a)[l0:l0 + n], precision=5, max_line_width=200))
