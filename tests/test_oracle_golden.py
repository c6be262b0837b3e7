"""The CPU restatement (oracle/spartacus_oracle.c) against the committed golden
vectors, i.e. against outputs of the real reference on every input its own
test suite ships (tests/golden/make_golden.py).  Runs without a GPU and without
/root/reference."""
import os

import pytest

from tests import fixture_io
from tests.conftest import golden_files

import numpy as np

# The restatement keeps the reference's operation order and is compiled without floating-point contraction:
# it reproduces the outputs of the compiled reference BIT FOR BIT on every fixture (gcc build against flang -O2
# build, this image).  SPARTACUS_ORACLE_TOL=<rtol> relaxes that for another compiler / libm.
TOL = float(os.environ.get("SPARTACUS_ORACLE_TOL", "0"))


def assert_same(got, expected, what=""):
    """bit-identical (NaN == NaN), or within SPARTACUS_ORACLE_TOL where that is set"""
    for k, v in expected.items():
        a, b = np.asarray(got[k]), np.asarray(v)
        if TOL > 0.0:
            assert np.allclose(a, b, rtol=TOL, atol=0.0, equal_nan=True), (what, k)
        else:
            same = (a == b) | (np.isnan(a) & np.isnan(b))
            assert same.all(), "%s %s: %d of %d elements differ" % (what, k, int((~same).sum()), same.size)


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_matches_golden(path, checkers):
    case, expected = fixture_io.load_case(path)
    out, _ = checkers.run_oracle(case["config"], case["canopy"], case["sw"], case["lw"],
                                 want_profiles=case["want_profiles"],
                                 do_simple_spectrum=case["do_simple_spectrum"])
    assert_same(fixture_io.flatten_outputs(out), expected, os.path.basename(path)[:-4])


def test_golden_set_is_complete():
    names = {os.path.basename(p)[:-4] for p in golden_files()}
    # every family of the reference's `make test`
    for fam in ("simple__", "rami4pilps__", "urban__", "rami5__", "single_layer__"):
        assert any(n.startswith(fam) for n in names), fam
    assert len(names) >= 45


@pytest.mark.parametrize("name", ["simple__test_surfaces", "simple__test_overhang"])
def test_limit_yardstick_of_the_degenerate_layer(name, checkers):
    """The second singular point of the reference's algorithm (tests/golden/make_golden.py: DEGENERATE_LIMIT):
    a vegetated layer whose vegetation has an extinction of exactly zero.  Pins what the fixtures store and why:
    (i) the stored truth IS the binary128 evaluation with 1e-15 in place of the zeros, (ii) it does not depend
    on that number (1e-12 gives the same to 1e-9: a limit), (iii) the binary128 evaluation AT zero is farther from
    it than the fp64 reference is - the reason it cannot serve as the yardstick there."""
    import copy
    path = [p for p in golden_files() if os.path.basename(p)[:-4] == name][0]
    case, expected = fixture_io.load_case(path)
    cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]

    def quad(eps):
        c = copy.deepcopy(cp)
        ve, vf = np.asarray(cp.veg_ext, dtype=np.float64), np.asarray(cp.veg_fraction, dtype=np.float64)
        assert ((ve == 0.0) & (vf > 0.0)).any()
        if eps is not None:
            c.veg_ext = np.where((ve == 0.0) & (vf > 0.0), eps, ve)
        out, _ = checkers.run_oracle(cfg, c, sw, lw, want_profiles=case["want_profiles"], quad=True,
                                     do_simple_spectrum=case["do_simple_spectrum"])
        return fixture_io.flatten_outputs(out)

    truth = case["ref_truth"]
    k = "lw_norm.veg_air_abs"
    assert k in truth
    lim15, lim12, at0 = quad(1.0e-15), quad(1.0e-12), quad(None)
    scale = np.max(np.abs(expected[k]))
    assert np.max(np.abs(lim15[k] - truth[k])) <= 1.0e-13 * scale
    assert np.max(np.abs(lim12[k] - lim15[k])) <= 1.0e-9 * scale
    ref_off = np.max(np.abs(expected[k] - truth[k])) / scale
    jump = np.max(np.abs(at0[k] - truth[k])) / scale
    assert jump > 3.0 * ref_off > 3.0e-5, (jump, ref_off)
