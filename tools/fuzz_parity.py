#!/usr/bin/env python3
"""Seeded sweep of the synthetic workloads through the C-ABI against the oracle (GPU box; TEST tool, like tests/).

usage: fuzz_parity.py [NSEEDS [NCOL]]  ->  one line per (workload, seed): columns, eigen failures, NaN/inf count, largest
error where the bar is 1e-5, number of variables with a widened bar, violations of the parity bar (tests/fixture_io.py);
exit code 1 if any seed violates the bar, has a non-finite output or an eigen failure.
The pytest suite checks a handful of fixed seeds; this runs many, at sizes the fp64 + binary128 oracle finishes in
seconds, to look for inputs the fixed seeds do not reach (rare region layouts, layer counts, optical depths).
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import checker                      # noqa: E402  (the checker)
from spartacus_surface_amd import radsurf, workloads   # noqa: E402
from tests import fixture_io                     # noqa: E402


def run(case, label):
    cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
    prof = True
    plan = radsurf.Plan(cfg, cp, sw, lw, want_profiles=prof)
    plan.solve()
    out = checker.alloc_outputs(cfg, cp, prof)
    plan.download(out["bc_out"], out["sw_norm_dir"], out["sw_norm_diff"], out["lw_internal"], out["lw_norm"])
    nfail = plan.eigen_failures()
    plan.close()
    got = fixture_io.flatten_outputs(out)
    a = (cfg, cp, sw, lw)
    ref = fixture_io.flatten_outputs(checker.run_oracle(*a, want_profiles=prof, nthreads=16)[0])
    truth = fixture_io.flatten_outputs(checker.run_oracle(*a, want_profiles=prof, nthreads=16, quad=True)[0])
    noise, noise_arr = fixture_io.noise_from_truth(ref, truth)
    report = {}
    bad = fixture_io.parity_violations(got, ref, noise, truth=truth, noise_arr=noise_arr, collect=report, label=label)
    nonfinite = int(sum(np.count_nonzero(~np.isfinite(v)) for v in got.values()))
    rep = report.get(label, {})
    tight = [v["err"] for v in rep.values() if v.get("bar", 1.0) <= 1.0000001e-5]
    widened = sum(1 for v in rep.values() if v.get("bar", 0.0) > 1.0000001e-5)
    farther = sum(1 for v in rep.values() if v.get("err", 0.0) > 1e-5 and v.get("gpu_vs_truth", 0.0) > v.get("ref_vs_truth", 1e300))
    return {"label": label, "columns": int(cp.ncol), "eigen_failures": int(nfail), "nonfinite": nonfinite,
            "max_err_where_bar_is_1e-5": max(tight) if tight else 0.0, "variables_with_widened_bar": widened,
            "differ_by_more_than_1e-5_and_farther_from_exact_than_reference": farther,
            "violations": [(b[0], float(b[1])) for b in bad[:5]], "nviolations": len(bad)}


def main():
    nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
    radsurf.load_library()
    builders = [
        ("c3", lambda s: workloads.c3_vegetated_urban(ncol=ncol, seed=s)),
        ("c3ns4", lambda s: workloads.c3_vegetated_urban(ncol=max(256, ncol // 4), seed=s, nstream=4)),
        ("c3ns3v1", lambda s: workloads.c3_vegetated_urban(ncol=max(256, ncol // 2), seed=s, nstream=3, n_vegetation_region=1)),
        ("c4", lambda s: workloads.c4_mixed(ncol=max(256, ncol // 3), seed=s, nsw=3, nlw=3)),
        ("c4ns4", lambda s: workloads.c4_mixed(ncol=max(128, ncol // 8), seed=s, nsw=2, nlw=2, nstream=4)),
        ("ragged", lambda s: workloads.ragged_all_tiles(ncol=max(256, ncol // 2), seed=s)),
        ("c3_14lay", lambda s: workloads.with_more_layers(workloads.c3_vegetated_urban(ncol=max(256, ncol // 3), seed=s), 4)),
    ]
    results, worst = [], 0
    t0 = time.time()
    for seed in range(1000, 1000 + nseeds):
        for name, build in builders:
            r = run(build(seed), "%s/seed%d" % (name, seed))
            results.append(r)
            flag = r["nviolations"] or r["nonfinite"] or r["eigen_failures"]
            worst |= 1 if flag else 0
            print("%-18s cols %5d eigfail %d nonfinite %d max_err(tight) %.2e widened %3d farther %d violations %d %s  [%.0f s]" % (
                r["label"], r["columns"], r["eigen_failures"], r["nonfinite"], r["max_err_where_bar_is_1e-5"],
                r["variables_with_widened_bar"], r["differ_by_more_than_1e-5_and_farther_from_exact_than_reference"],
                r["nviolations"], r["violations"][:2] if r["nviolations"] else "", time.time() - t0), flush=True)
    out = os.path.join(ROOT, "gpurun_out", "fuzz_parity.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(results, open(out, "w"), indent=1)
    print("seeds %d, runs %d, with violations / non-finite / eigen failures: %d" % (
        nseeds, len(results), sum(1 for r in results if r["nviolations"] or r["nonfinite"] or r["eigen_failures"])))
    return worst


if __name__ == "__main__":
    sys.exit(main())
