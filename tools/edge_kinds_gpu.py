#!/usr/bin/env python3
"""Per kind of edge input: outputs that are not finite on the GPU where the reference's are (GPU box, TEST tool).
usage: edge_kinds_gpu.py WORKLOAD SEED [NCOL [kind+kind ...]]   (WORKLOAD: c3 | c4 | ragged | c3ns4)"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tools", "fuzz_parity.py"))
fz = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fz)
from oracle import checker                      # noqa: E402
from spartacus_surface_amd import radsurf, workloads   # noqa: E402
from tests import fixture_io                     # noqa: E402

name, seed = sys.argv[1], int(sys.argv[2])
ncol = int(sys.argv[3]) if len(sys.argv) > 3 else 512
build = {"c3": lambda: workloads.c3_vegetated_urban(ncol=ncol, seed=seed),
         "c3ns4": lambda: workloads.c3_vegetated_urban(ncol=ncol, seed=seed, nstream=4),
         "c4": lambda: workloads.c4_mixed(ncol=ncol, seed=seed, nsw=3, nlw=3),
         "ragged": lambda: workloads.ragged_all_tiles(ncol=ncol, seed=seed)}[name]
combos = [tuple(a.split("+")) for a in sys.argv[4:]] if len(sys.argv) > 4 else [(k,) for k in fz.EDGE_KINDS] + [fz.EDGE_KINDS]
for kinds in combos:
    case = fz.edge_inputs(build(), seed, kinds=kinds)
    cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
    plan = radsurf.Plan(cfg, cp, sw, lw, want_profiles=True)
    plan.solve()
    out = checker.alloc_outputs(cfg, cp, True)
    plan.download(out["bc_out"], out["sw_norm_dir"], out["sw_norm_diff"], out["lw_internal"], out["lw_norm"])
    plan.close()
    got = fixture_io.flatten_outputs(out)
    ref = fixture_io.flatten_outputs(checker.run_oracle(cfg, cp, sw, lw, want_profiles=True, nthreads=16)[0])
    bad = {k: int(np.count_nonzero(~np.isfinite(got[k]) & np.isfinite(ref[k]))) for k in ref}
    bad = {k: v for k, v in bad.items() if v}
    nref = sum(int(np.count_nonzero(~np.isfinite(v))) for v in ref.values())
    print("%-22s GPU non-finite where the reference is finite %6d | reference non-finite %6d | %s" % (
        "+".join(kinds) if len(kinds) < len(fz.EDGE_KINDS) else "all", sum(bad.values()), nref, sorted(bad.items(), key=lambda kv: -kv[1])[:3]), flush=True)
