"""CPU coverage of the kernel bodies and of the host-side plan logic.

tests/emul/ss_emul.cpp compiles the SAME kernel headers (spartacus_surface_amd/
csrc/*.hpp) and the same plan (ss_plan_impl.hpp: column classification, chunking,
scratch layout, launch geometry) with g++ behind a backend that runs every
"lane" as a plain function call.  This is TEST infrastructure (it is never
loaded by the product path, which fails loudly without the HIP library); it
lets the -m "not gpu" suite check the arithmetic and the indexing of the
device code against the reference's golden vectors before a GPU is involved.
"""
import os

import pytest

from tests import fixture_io
from tests.conftest import golden_files
from tests.test_hip_parity import REFERENCE_SINGULAR, check_against_limit

# one case per code path, kept small so the CPU suite stays within minutes
CASES = ["simple__test_surfaces", "simple__test_surfaces_1stream", "simple__test_empty_layers",
         "simple__test_nearly_empty_layers", "simple__test_overhang", "simple__test_noscat",
         "simple__test_closed", "single_layer__sp", "single_layer__exp", "single_layer__inf",
         "extra__single_layer-mixed", "extra__ragged-tiles", "urban__1stream", "urban__2stream",
         "urban__4stream", "urban__single", "rami4pilps__vis-med-0.3", "rami4pilps__vis-snw-0.3-2-4",
         "rami4pilps__vis-snw-0.3-2-8",   # order 24: symmetrised route with rolled loops
         "rami5__HET07_JPS_SUM-56-direct",
         # stream counts beyond the reference's own tests (tests/cases.py: extra_cases)
         "extra__urban-3stream", "extra__urban-5stream", "extra__urban-6stream",
         "extra__rami4pilps-vis-snw-0.3-2-3", "extra__rami4pilps-vis-snw-0.3-1-7"]


def _available():
    have = {os.path.basename(p)[:-4]: p for p in golden_files()}
    return [have[c] for c in CASES if c in have] or list(have.values())[:4]


@pytest.fixture(scope="module")
def emul():
    from tests import emul as e
    e.build()
    return e


@pytest.mark.parametrize("path", _available(), ids=lambda p: os.path.basename(p)[:-4])
def test_emulated_kernels_vs_golden(emul, path):
    name = os.path.basename(path)[:-4]
    case, expected = fixture_io.load_case(path)
    cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
    out, nfail = emul.run_emul(cfg, cp, sw, lw, want_profiles=case["want_profiles"],
                               do_simple_spectrum=case["do_simple_spectrum"])
    assert nfail == 0
    got = fixture_io.flatten_outputs(out)
    noise = dict(case["ref_noise"])
    if name in REFERENCE_SINGULAR:
        lim, bar = REFERENCE_SINGULAR[name]
        expected = {k: v for k, v in expected.items() if noise.get(k, 0.0) <= lim}
        noise = {k: bar / fixture_io.NOISE_FACTOR for k in noise}
    singular = name in REFERENCE_SINGULAR
    bad = fixture_io.parity_violations(got, expected, noise,
                                       truth=None if singular else case["ref_truth"],
                                       noise_arr=None if singular else case["ref_noise_arr"], slack=2.0)
    assert not bad, "outside the parity bar: %s" % bad[:5]
    if singular:
        # the reference's algorithm is singular here; the yardstick is the limit stored in the fixture
        # (gcc arithmetic without fused multiply-adds: twice the device's allowance)
        far = check_against_limit(got, case, tol=2.0e-5)
        assert not far, "farther than 2e-5 from the limit of the singular case: %s" % far[:5]


def test_emulated_ragged_all_tile_types_vs_oracle(emul):
    """All six tile types incl. the single-layer urban ones next to multi-layer
    columns (where the compiled reference cannot serve as the checker, DESIGN.md
    section 7): emulated device code against the CPU restatement."""
    from oracle import checker
    from spartacus_surface_amd import workloads
    case = workloads.ragged_all_tiles(ncol=96, seed=2024)
    a = (case["config"], case["canopy"], case["sw"], case["lw"])
    out, nfail = emul.run_emul(*a, want_profiles=True)
    assert nfail == 0
    ref = fixture_io.flatten_outputs(checker.run_oracle(*a, want_profiles=True)[0])
    truth = fixture_io.flatten_outputs(checker.run_oracle(*a, want_profiles=True, quad=True)[0])
    noise, noise_arr = fixture_io.noise_from_truth(ref, truth)
    bad = fixture_io.parity_violations(fixture_io.flatten_outputs(out), ref, noise, truth=truth,
                                       noise_arr=noise_arr, slack=2.0)
    assert not bad, bad[:5]


def test_emulated_deep_columns_vs_oracle(emul):
    """14-layer columns: more layers than the per-lane slots of the downward sweeps can stage for every field
    (csrc/ss_column_sw.hpp: PUTL), so staged and directly stored fields mix."""
    from oracle import checker
    from spartacus_surface_amd import workloads
    case = workloads.with_more_layers(workloads.c3_vegetated_urban(ncol=24, seed=777), 4)
    a = (case["config"], case["canopy"], case["sw"], case["lw"])
    out, nfail = emul.run_emul(*a, want_profiles=False)
    assert nfail == 0
    ref = fixture_io.flatten_outputs(checker.run_oracle(*a, want_profiles=False)[0])
    truth = fixture_io.flatten_outputs(checker.run_oracle(*a, want_profiles=False, quad=True)[0])
    noise, noise_arr = fixture_io.noise_from_truth(ref, truth)
    bad = fixture_io.parity_violations(fixture_io.flatten_outputs(out), ref, noise, truth=truth,
                                       noise_arr=noise_arr, slack=2.0)
    assert not bad, bad[:5]


@pytest.mark.parametrize("flag,what", [
    # the shortwave downward sweep with both passes on one lane (csrc/ss_down_sw_both.hpp): off by default on the
    # device (slower, measured), kept correct: same bits as the paired-lane form under gcc
    ("-DSS_DOWN_BOTH", "bits"),
    # the three-kernel form of round 3 (layer records, upward sweep as a kernel of its own) against the fused
    # kernels of round 4 (ss_layer_up_sw2.hpp, sym::layer_up_lw2): the same algebra in another order
    ("-DSS_NO_SYM_FUSE_SW2", 1.0e-8)])
@pytest.mark.skipif(os.environ.get("SPARTACUS_SLOW_TESTS") != "1",
                    reason="two more builds of the emulator (3 minutes each): SPARTACUS_SLOW_TESTS=1")
def test_alternative_kernel_forms_agree(flag, what, tmp_path):
    """Build the emulator a second time with the flag and run three goldens through both (both pass: round 4)."""
    import ctypes as C
    import subprocess
    import numpy as np
    from tests import emul as e
    e.build()
    alt = str(tmp_path / "libss_emul_alt.so")
    subprocess.run(["g++"] + e.CXXFLAGS + [flag, "-o", alt, os.path.join(os.path.dirname(e.__file__), "ss_emul.cpp")],
                   check=True)
    have = {os.path.basename(p)[:-4]: p for p in golden_files()}
    names = [n for n in ("urban__2stream", "extra__ragged-tiles", "simple__test_surfaces") if n in have]
    res = {}
    keep = e._lib
    try:
        for tag, lib in (("a", e.LIB), ("b", alt)):
            e._lib = C.CDLL(lib)
            for n in names:
                case, _ = fixture_io.load_case(have[n])
                out, nfail = e.run_emul(case["config"], case["canopy"], case["sw"], case["lw"],
                                        want_profiles=case["want_profiles"],
                                        do_simple_spectrum=case["do_simple_spectrum"])
                assert nfail == 0
                res[(tag, n)] = {k: np.array(v, copy=True) for k, v in fixture_io.flatten_outputs(out).items()}
    finally:
        e._lib = keep
    for n in names:
        a, b = res[("a", n)], res[("b", n)]
        for k in a:
            if what == "bits":
                assert np.array_equal(a[k], b[k], equal_nan=True), (n, k)
            else:
                den = np.maximum(np.abs(a[k]), 1.0e-3 * max(1.0e-300, float(np.max(np.abs(a[k])))) if a[k].size else 1.0)
                assert np.all(np.abs(a[k] - b[k]) <= what * den), (n, k)


@pytest.mark.parametrize("name", ["simple__test_surfaces", "urban__2stream", "extra__ragged-tiles"])
def test_emulated_general_route_of_declined_units(emul, name, monkeypatch):
    """SPARTACUS_DECLINE_SYM=1: every unit of the orders 3..6 is declined by the symmetrised route and recomputed by
    the out-of-line general route of the layer units (layer_unit_sw_cold / layer_unit_lw_cold), then replayed through
    the fused kernels' writers - the path of units whose matrices are not positive definite, which no input of the
    reference's test-suite takes (on the device: tests/test_hip_parity.py::test_golden_with_symmetric_route_declined)."""
    have = {os.path.basename(p)[:-4]: p for p in golden_files()}
    if name not in have:
        pytest.skip("fixture not present")
    monkeypatch.setenv("SPARTACUS_DECLINE_SYM", "1")
    case, expected = fixture_io.load_case(have[name])
    out, nfail = emul.run_emul(case["config"], case["canopy"], case["sw"], case["lw"], want_profiles=case["want_profiles"],
                               do_simple_spectrum=case["do_simple_spectrum"])
    assert nfail == 0
    bad = fixture_io.parity_violations(fixture_io.flatten_outputs(out), expected, dict(case["ref_noise"]),
                                       truth=case["ref_truth"], noise_arr=case["ref_noise_arr"], slack=2.0)
    assert not bad, "outside the parity bar: %s" % bad[:5]
