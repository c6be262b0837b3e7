"""The CPU restatement (oracle/spartacus_oracle.c) against the committed golden
vectors, i.e. against outputs of the real reference on every input its own
test suite ships (tests/golden/make_golden.py).  Runs without a GPU and without
/root/reference."""
import os

import pytest

from tests import fixture_io
from tests.conftest import golden_files

# The restatement keeps the reference's operation order, so it reproduces the
# reference to rounding; 1e-11 leaves room for a different libm/compiler.
TOL = 1.0e-11


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_matches_golden(path, checkers):
    case, expected = fixture_io.load_case(path)
    out, _ = checkers.run_oracle(case["config"], case["canopy"], case["sw"], case["lw"],
                                 want_profiles=case["want_profiles"],
                                 do_simple_spectrum=case["do_simple_spectrum"])
    err, where = fixture_io.max_rel_error(fixture_io.flatten_outputs(out), expected)
    assert err <= TOL, "max rel error %.3e at %s" % (err, where)


def test_golden_set_is_complete():
    names = {os.path.basename(p)[:-4] for p in golden_files()}
    # every family of the reference's `make test`
    for fam in ("simple__", "rami4pilps__", "urban__", "rami5__", "single_layer__"):
        assert any(n.startswith(fam) for n in names), fam
    assert len(names) >= 45
