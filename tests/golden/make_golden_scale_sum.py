"""Golden vectors of canopy_flux%scale / %sum from
the REAL reference: for a few of the golden cases the compiled reference runs
radsurf and then scales / sums the flux objects exactly as its driver does
(driver/spartacus_surface_driver.F90:250-261) with the top-of-canopy fluxes of
the case's input file.  Stored per case: the three factors, the normalised fluxes of the same run, the
four scaled objects and the two sums.

    python tests/golden/make_golden_scale_sum.py      (needs oracle/_ref)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import checker            # noqa: E402
from tests import fixture_io          # noqa: E402

CASES = ["simple__test_surfaces", "urban__2stream", "rami4pilps__vis-med-0.3", "single_layer__sp"]


def top_fluxes(case):
    cfg, cp = case["config"], case["canopy"]
    def full(a, n, default):
        if a is None:
            return np.full((cp.ncol, n), default)
        a = np.asarray(a, dtype=np.float64)
        return np.ascontiguousarray(np.broadcast_to(a.reshape(cp.ncol, -1), (cp.ncol, n)))
    return (full(case.get("top_flux_dn_sw"), cfg.nsw, 500.0),
            full(case.get("top_flux_dn_direct_sw"), cfg.nsw, 300.0),
            full(case.get("top_flux_dn_lw"), cfg.nlw, 350.0))


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    out_dir = os.path.join(here, "scale_sum")
    os.makedirs(out_dir, exist_ok=True)
    checker.build()
    for name in CASES:
        case, _ = fixture_io.load_case(os.path.join(here, name + ".npz"))
        if case["do_simple_spectrum"]:
            from spartacus_surface_amd.inputs import calc_simple_spectrum_lw
            calc_simple_spectrum_lw(case["config"], case["canopy"], case["lw"])
        t_sw, t_dir, t_lw = top_fluxes(case)
        out, sw_flux, lw_flux = checker.run_ref_scale_sum(case["config"], case["canopy"], case["sw"],
                                                          case["lw"], t_sw, t_dir, t_lw)
        norm, _ = checker.run_ref(case["config"], case["canopy"], case["sw"], case["lw"],
                                  nblocksize=max(1, case["canopy"].ncol))
        d = {"top_flux_dn_sw": t_sw, "top_flux_dn_direct_sw": t_dir, "top_flux_dn_lw": t_lw}
        for k, v in fixture_io.flatten_outputs(norm).items():
            d["norm." + k] = v
        for k, v in fixture_io.flatten_outputs(out).items():
            d["scaled." + k] = v
        if case["config"].do_sw:
            for k, v in sw_flux.fields().items():
                d["sw_flux." + k] = v
        if case["config"].do_lw:
            for k, v in lw_flux.fields().items():
                d["lw_flux." + k] = v
        path = os.path.join(out_dir, name + ".npz")
        np.savez_compressed(path, **d)
        print("%-40s %6.1f KB" % (name, os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
