"""Generate the golden fixtures in tests/golden/ from the REAL reference.

Run in the build container (needs /root/reference for the test inputs and the
compiled reference in oracle/_ref):

    python tests/golden/make_golden.py            # the reference's own test runs
    python tests/golden/make_golden.py --extra    # + stream counts its tests never select (tests/cases.py)
    python tests/golden/make_golden.py [--extra] --only=SUBSTRING   # just the cases whose name matches

For every run of the reference's own `make test` that exercises the hot path
(tests/cases.py) this freezes the solver inputs and the reference's fp64
outputs (all members of bc_out and of the four canopy_flux objects, before
scale/sum) into one compressed .npz, together with the reference's own rounding
noise per variable and - for the variables whose parity bar that noise widens -
the binary128 result and the per-element noise (tests/fixture_io.parity_report).
All illuminations of test/rami5/Makefile:37-45 are kept (36 runs, ~6 MB).
"""
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import checker            # noqa: E402
from tests import cases, fixture_io   # noqa: E402


# Cases at which the reference's algorithm is singular in EXACT arithmetic, not merely ill-conditioned in fp64:
# test/simple/test_noscat sets every single-scattering albedo to exactly zero, Gamma2 vanishes, the eigenvector
# blocks G2 = X - X of radtool_calc_matrices_lw_eig.F90:152-176 are exactly zero, and what the formulas then
# divide by is decided by rounding - in binary128 too: its evaluation AT zero jumps by 30 % from the limit
# ssa -> 0+ (lw_internal%veg_abs -246.5 against -183.4, continuous from 1e-4 down to 1e-14).  The yardstick
# stored for such a case is that limit: binary128 at albedos of 1e-15 in place of the exact zeros.
SINGULAR_LIMIT = {"simple/test_noscat": 1.0e-15}
# A second singular point of the reference's algorithm: a vegetated layer whose vegetation has an extinction of
# EXACTLY zero (simple/test_surfaces, column 2, layer 1).  Its two vegetated regions are then identical clear air,
# the layer's coefficient matrix has repeated eigenvalues, and formulas that invert the eigenvector matrix return
# whatever rounding decides - in any precision: the binary128 evaluation AT zero jumps by 6.8e-4 from its own
# limit (lw_norm%veg_air_abs 9.83937e-6 at zero against 9.84610e-6 for every extinction from 1e-20 to 1e-9), the
# fp64 reference sits 5.5e-5 from that limit (and 3 - 5 % from it with a hundred times the air extinction).  The
# yardstick ("truth") stored for such a case is the limit: binary128 with an extinction of 1e-15 in place of the
# exact zeros.  The EXPECTED values stay the reference's own, from the unmodified inputs.
DEGENERATE_LIMIT = {"simple/test_surfaces": 1.0e-15, "simple/test_overhang": 1.0e-15}


def _regularised(props, eps):
    p = copy.deepcopy(props)
    for name in ("air_ssa", "veg_ssa"):
        a = getattr(p, name, None)
        if a is not None and np.size(a):
            a = np.asarray(a, dtype=np.float64)
            setattr(p, name, np.where(a == 0.0, eps, a))
    return p


def _nondegenerate(cp, eps):
    c = copy.deepcopy(cp)
    ve = np.asarray(c.veg_ext, dtype=np.float64)
    vf = np.asarray(c.veg_fraction, dtype=np.float64)
    c.veg_ext = np.where((ve == 0.0) & (vf > 0.0), eps, ve)
    return c


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    checker.build()
    total = 0
    todo = cases.extra_cases() if "--extra" in sys.argv else cases.all_cases(rami5="all")
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
        truth_canopy = case["canopy"]
        if name in DEGENERATE_LIMIT:
            truth_canopy = _nondegenerate(case["canopy"], DEGENERATE_LIMIT[name])
        truth, _ = checker.run_oracle(case["config"], truth_canopy, case["sw"], case["lw"],
                                      want_profiles=True, quad=True, nthreads=8,
                                      do_simple_spectrum=case["do_simple_spectrum"])
        ref_flat = fixture_io.flatten_outputs(outputs)
        truth_flat = fixture_io.flatten_outputs(truth)
        if name in DEGENERATE_LIMIT:
            # a variable that the reference's own guards make DIScontinuous at the degenerate point (the sunlit
            # fraction of the vegetation: absorbed direct flux over max(epsilon, its clear-sky value), 0 / epsilon at
            # zero extinction) keeps the evaluation at the point itself
            at_point, _ = checker.run_oracle(case["config"], case["canopy"], case["sw"], case["lw"],
                                             want_profiles=True, quad=True, nthreads=8,
                                             do_simple_spectrum=case["do_simple_spectrum"])
            at_point = fixture_io.flatten_outputs(at_point)
            jump = fixture_io.rel_errors(truth_flat, at_point)
            for k, e in jump.items():
                if not e <= 10.0:
                    print("   %s: discontinuous at the degenerate point (%.3g), evaluated there" % (k, e))
                    truth_flat[k] = at_point[k]
        # per element, in the metric of the parity bar (denominators from the reference's values)
        noise_arr = {k: v for k, v in fixture_io.rel_error_arrays(truth_flat, ref_flat).items() if v is not None}
        # ... and further realisations of its rounding error: the same run with the layer
        # thicknesses changed in their last bits (a few ulp - no more than the rounding of
        # the inputs themselves; round 1 also used 1e-13, i.e. ~900 ulp, which inflated the
        # bars of ill-conditioned variables by the genuine change of the exact result).
        # Each is compared with the binary128 result of the UNPERTURBED inputs, so the
        # noise of a variable is the largest distance from exact arithmetic that the
        # reference shows over five evaluations that are equivalent at input precision.
        for eps in (1.0e-15, -1.0e-15, 3.0e-15, -4.0e-15):
            cp2 = copy.deepcopy(case["canopy"])
            cp2.dz = cp2.dz * (1.0 + eps)
            pert, _ = checker.run_oracle(case["config"], cp2, case["sw"], case["lw"],
                                         want_profiles=True, nthreads=8,
                                         do_simple_spectrum=case["do_simple_spectrum"])
            pf = fixture_io.flatten_outputs(pert)
            # distance from the binary128 result, in the metric of the reference's values
            for k, v in fixture_io.rel_error_arrays({k: pf[k] - truth_flat[k] + ref_flat[k] for k in ref_flat},
                                                    ref_flat).items():
                if v is not None and k in noise_arr:
                    noise_arr[k] = np.maximum(noise_arr[k], v)
        noise = {k: float(np.max(v)) for k, v in noise_arr.items()}
        for k, v in ref_flat.items():
            if v.size and k not in noise:
                noise[k] = float("inf")
        limit = None
        if name in SINGULAR_LIMIT:
            eps = SINGULAR_LIMIT[name]
            lim, _ = checker.run_oracle(case["config"], case["canopy"], _regularised(case["sw"], eps),
                                        _regularised(case["lw"], eps), want_profiles=True, quad=True, nthreads=8,
                                        do_simple_spectrum=case["do_simple_spectrum"])
            limit = fixture_io.flatten_outputs(lim)
        path = os.path.join(out_dir, name.replace("/", "__") + ".npz")
        fixture_io.save_case(path, case, outputs, want_profiles=True, noise=noise, truth=truth_flat,
                             noise_arr=noise_arr, limit=limit)
        total += os.path.getsize(path)
        print("%-50s %7.1f KB" % (name, os.path.getsize(path) / 1024.0))
    print("total %.2f MB" % (total / 1048576.0))


if __name__ == "__main__":
    main()
