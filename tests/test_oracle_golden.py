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
