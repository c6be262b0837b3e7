"""Inputs of one column of a reference test/rami5 run (as stored by tests/golden/make_golden.py) -> the
package's data directory, for workloads.rami5_replicated (bench.py --workload rami5).
usage: python tools/extract_rami5_column.py [scene]"""
import os
import sys

import numpy as np

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
scene = sys.argv[1] if len(sys.argv) > 1 else "HET09_JBS_SUM-41-direct"
z = np.load(os.path.join(root, "tests", "golden", "rami5__%s.npz" % scene))
keep = {k: np.array(z[k]) for k in z.files if k == "meta" or k.startswith(("canopy.", "sw.", "lw.", "driver."))}
out = os.path.join(root, "spartacus_surface_amd", "data", "rami5__%s_inputs.npz" % scene)
np.savez_compressed(out, **keep)
print(out, os.path.getsize(out), "bytes,", len(keep), "arrays")
