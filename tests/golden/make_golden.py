"""Generate the golden fixtures in tests/golden/ from the REAL reference.

Run in the build container (needs /root/reference for the test inputs and the
compiled reference in oracle/_ref):

    python tests/golden/make_golden.py            # the reference's own test runs
    python tests/golden/make_golden.py --extra    # + stream counts its tests never select (tests/cases.py)
    python tests/golden/make_golden.py [--extra] --only=SUBSTRING   # just the cases whose name matches

For every run of the reference's own `make test` that exercises the hot path
(tests/cases.py) this freezes the solver inputs and the reference's fp64
outputs (all members of bc_out and of the four canopy_flux objects, before
scale/sum) into one compressed .npz.  The rami5 set is thinned to keep the
repository small: every scene with diffuse illumination (real soil) and its
first direct-beam angle with real and black soil.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import checker            # noqa: E402
from tests import cases, fixture_io   # noqa: E402


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    checker.build()
    total = 0
    todo = cases.extra_cases() if "--extra" in sys.argv else cases.all_cases(rami5="subset")
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]
    for name, build in todo.items():
        if only and not any(o in name for o in only):
            continue
        case = build()
        outputs, _ = checker.run_ref(case["config"], case["canopy"], case["sw"], case["lw"],
                                     want_profiles=True,
                                     do_simple_spectrum=case["do_simple_spectrum"])
        # the reference's own rounding error per variable, against the binary128
        # build of the restatement (exact-arithmetic yardstick)
        truth, _ = checker.run_oracle(case["config"], case["canopy"], case["sw"], case["lw"],
                                      want_profiles=True, quad=True, nthreads=8,
                                      do_simple_spectrum=case["do_simple_spectrum"])
        ref_flat = fixture_io.flatten_outputs(outputs)
        noise = fixture_io.rel_errors(ref_flat, fixture_io.flatten_outputs(truth))
        # ... and its sensitivity to a last-bit change of the inputs (layer
        # thickness scaled by 1 +- few ulp), which also exposes configurations
        # where the algorithm itself is singular (test_noscat) and the
        # binary128 value is meaningless
        import copy
        for eps in (1.0e-15, -1.0e-15, 3.0e-15, -4.0e-15, 1.0e-13, -1.0e-13):
            cp2 = copy.deepcopy(case["canopy"])
            cp2.dz = cp2.dz * (1.0 + eps)
            pert, _ = checker.run_oracle(case["config"], cp2, case["sw"], case["lw"],
                                         want_profiles=True, nthreads=8,
                                         do_simple_spectrum=case["do_simple_spectrum"])
            for k, v in fixture_io.rel_errors(fixture_io.flatten_outputs(pert), ref_flat).items():
                noise[k] = max(noise.get(k, 0.0), v)
        path = os.path.join(out_dir, name.replace("/", "__") + ".npz")
        fixture_io.save_case(path, case, outputs, want_profiles=True, noise=noise)
        total += os.path.getsize(path)
        print("%-50s %7.1f KB" % (name, os.path.getsize(path) / 1024.0))
    print("total %.2f MB" % (total / 1048576.0))


if __name__ == "__main__":
    main()
