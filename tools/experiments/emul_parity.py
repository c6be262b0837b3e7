"""CPU check of a kernel variant (TEST infrastructure): the host build of the kernel headers
(tests/emul) on a bench workload against the oracle in fp64 and binary128.
usage: python tools/experiments/emul_parity.py [workload] [ncol] [-DFLAG ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import checker                 # noqa: E402
from tests import emul, fixture_io         # noqa: E402
import bench                               # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
flags = [a for a in sys.argv[3:] if a.startswith("-D")]
emul.build(force=True, extra_flags=flags)
case = bench.make_case(wl, ncol, 12345)
a = (case["config"], case["canopy"], case["sw"], case["lw"])
out, nfail = emul.run_emul(*a, want_profiles=False)
got = fixture_io.flatten_outputs(out)
ref = fixture_io.flatten_outputs(checker.run_oracle(*a, want_profiles=False, nthreads=8)[0])
truth = fixture_io.flatten_outputs(checker.run_oracle(*a, want_profiles=False, nthreads=8, quad=True)[0])
noise, noise_arr = fixture_io.noise_from_truth(ref, truth)
rep = {}
bad = fixture_io.parity_violations(got, ref, noise, truth=truth, noise_arr=noise_arr, collect=rep, label="x")
rows = sorted(rep["x"].items(), key=lambda kv: -kv[1]["err"])
print("eigen failures", nfail, " violations", len(bad))
print("%-40s %10s %10s %10s %10s" % ("variable", "err", "bar", "gpu-exact", "ref-exact"))
for k, r in rows[:12]:
    print("%-40s %10.2e %10.2e %10s %10s" % (k, r["err"], r["bar"],
          "%.2e" % r["gpu_vs_truth"] if r.get("gpu_vs_truth") is not None else "-",
          "%.2e" % r["ref_vs_truth"] if r.get("ref_vs_truth") is not None else "-"))
# overall distance from exact arithmetic, all variables
def dist(x):
    worst = 0.0
    for k, t in truth.items():
        sc = max(np.max(np.abs(t)) * 1e-3, 1e-300)
        d = np.max(np.abs(x[k] - t) / np.maximum(np.abs(t), sc)) if t.size else 0.0
        worst = max(worst, float(d))
    return worst
print("max distance from binary128: emulated kernels %.3e, reference algorithm fp64 %.3e" % (dist(got), dist(ref)))
