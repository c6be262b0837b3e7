"""Parity tests proper: the HIP path on a real MI355X, through the C-ABI
(spartacus_radsurf_hip and the ss_plan_* entry points), against

* the committed golden vectors = fp64 outputs of the real reference on every
  input its own test suite ships (bar: fixture_io.PARITY_TOL = 1e-5 relative,
  widened per variable only where the reference's own rounding error, measured
  with the binary128 build of the oracle, is larger than that);
* the CPU oracle on seeded synthetic many-column inputs (the bench workloads);
* size-independent properties at full bench size (column independence /
  permutation invariance, replicated columns give replicated results).
"""
import os

import numpy as np
import pytest

from tests import fixture_io
from tests.conftest import golden_files

pytestmark = pytest.mark.gpu

# simple__test_noscat sets every single-scattering albedo to exactly zero, which
# makes the reference's eigenvector route singular (G2 = X - X) - in exact
# arithmetic too: its binary128 evaluation AT zero jumps by 30 % from the limit
# albedo -> 0+ (tests/golden/make_golden.py: SINGULAR_LIMIT), and its fp64 output
# is rounding noise for several variables (lw_norm%veg_abs 0.44 where the limit
# is 0.105).  Against the reference this case is a sanity check only (variables
# the reference itself gets right to 1e-3 must agree to 5e-2); the tight check
# is against that limit - binary128 at albedos of 1e-15, stored in the fixture
# for every variable -, within LIMIT_TOL.
REFERENCE_SINGULAR = {"simple__test_noscat": (1.0e-3, 5.0e-2)}
LIMIT_TOL = 1.0e-5


def check_against_limit(got, case, tol=LIMIT_TOL, collect=None, label=None):
    """[(variable, error)] beyond tol against the stored limit of a reference-singular case."""
    limit = case.get("ref_limit") or {}
    assert limit, "fixture of a reference-singular case without its limit values"
    # (variables that vanish in the limit - upwelling flux above a black ground without scattering - are rounding
    # residue, 1e-14 here, 1e-12 in the reference: held to 1e-5 x 1e-6 of the scale of their flux object)
    errs = fixture_io.rel_errors({k: got[k] for k in limit}, limit, zero_floor=1.0e-6)
    if collect is not None and label is not None:
        collect[label] = {k: {"err": e, "bar": tol, "ref_noise": 0.0, "ok": bool(e <= tol)} for k, e in errs.items()}
    return sorted(((k, e) for k, e in errs.items() if not e <= tol), key=lambda kv: -kv[1])

# Every comparison of this module leaves its full report here: per case and variable
# {measured error, bar, the reference's own noise, both distances from binary128}.
# Written to gpurun_out/parity_report.json at the end of the session; the copy of the
# round's final GPU run is committed as profiles/parity_r3.json.
_REPORT = {}


@pytest.fixture(scope="module", autouse=True)
def _dump_parity_report():
    yield
    if not _REPORT:
        return
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    worst = {}
    for label, rep in _REPORT.items():
        over = {k: r for k, r in rep.items() if r["bar"] > fixture_io.PARITY_TOL}
        worst[label] = {"variables": len(rep), "max_err": max(r["err"] for r in rep.values()),
                        "widened_bars": len(over),
                        "max_err_where_bar_is_1e-5": max([r["err"] for r in rep.values()
                                                          if r["bar"] <= fixture_io.PARITY_TOL] or [0.0])}
    # the honest headline: how many variables differ from the reference by more than the tolerance of north_star
    # (the GPU path is more accurate than, not identical to, the reference: DESIGN.md section 2), and of those how
    # many are farther from the binary128 evaluation than the reference itself is
    allv = [r for label, rep in _REPORT.items() if not label.startswith("limit/") for r in rep.values()]
    over_tol = [r for r in allv if r["err"] > fixture_io.PARITY_TOL]
    farther = [r for r in over_tol if "gpu_vs_truth" in r and r["gpu_vs_truth"] > r["ref_vs_truth"]]
    totals = {"cases": len(_REPORT), "variables": len(allv), "differ_from_reference_by_more_than_tol": len(over_tol),
              "of_those_with_binary128_data": sum(1 for r in over_tol if "gpu_vs_truth" in r),
              "of_those_farther_from_binary128_than_the_reference": len(farther),
              "largest_difference_from_reference": max([r["err"] for r in allv] or [0.0])}
    print("\nparity: %(differ_from_reference_by_more_than_tol)d of %(variables)d variables (%(cases)d cases) differ from "
          "the reference by more than 1e-5, all within its measured fp64 noise; %(of_those_farther_from_binary128_than_the_reference)d "
          "of them are farther from the binary128 evaluation than the reference is" % totals)
    with open(os.path.join(out_dir, "parity_report.json"), "w") as f:
        json.dump({"tol": fixture_io.PARITY_TOL, "noise_factor": fixture_io.NOISE_FACTOR, "totals": totals,
                   "summary": worst, "cases": _REPORT}, f, indent=1, sort_keys=True)


def _hip():
    from spartacus_surface_amd import radsurf
    radsurf.load_library()          # raises if the extension is missing: no fallback
    assert radsurf.device_count() >= 1
    return radsurf


def _alloc(case):
    from oracle import checker      # test-only helper for output allocation
    return checker.alloc_outputs(case["config"], case["canopy"], case["want_profiles"])


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_golden_host_entry(path):
    """spartacus_radsurf_hip (host buffers in/out) vs the reference's outputs."""
    radsurf = _hip()
    name = os.path.basename(path)[:-4]
    case, expected = fixture_io.load_case(path)
    cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
    if case["do_simple_spectrum"]:
        from spartacus_surface_amd.inputs import calc_simple_spectrum_lw
        calc_simple_spectrum_lw(cfg, cp, lw)
    out = _alloc(case)
    radsurf.radsurf(cfg, cp, sw, lw, out["bc_out"], 1, cp.ncol, out["sw_norm_dir"],
                    out["sw_norm_diff"], out["lw_internal"], out["lw_norm"])
    got = fixture_io.flatten_outputs(out)
    noise = dict(case["ref_noise"])
    if name in REFERENCE_SINGULAR:
        lim, bar = REFERENCE_SINGULAR[name]
        expected = {k: v for k, v in expected.items() if noise.get(k, 0.0) <= lim}
        noise = {k: bar / fixture_io.NOISE_FACTOR for k in noise}
    singular = name in REFERENCE_SINGULAR
    bad = fixture_io.parity_violations(got, expected, noise,
                                       truth=None if singular else case["ref_truth"],
                                       noise_arr=None if singular else case["ref_noise_arr"],
                                       collect=_REPORT, label="golden/" + name)
    assert not bad, "outside the parity bar: %s" % bad[:5]
    if singular:
        far = check_against_limit(got, case, collect=_REPORT, label="limit/" + name)
        assert not far, "farther than %g from the limit of the singular case: %s" % (LIMIT_TOL, far[:5])


@pytest.mark.parametrize("name", ["simple__test_surfaces", "simple__test_overhang", "urban__2stream", "urban__4stream",
                                  "rami4pilps__vis-med-0.3", "extra__ragged-tiles", "extra__single_layer-mixed",
                                  "single_layer__sp"])
def test_golden_host_entry_chunk_pipeline(name, monkeypatch):
    """Calls of up to 2048 columns take the entry's small-call path (a slot of their own, one slab download);
    the same goldens through the chunk pipeline of the large calls (SPARTACUS_SMALL_CALL_COLUMNS=0), in
    chunks of 8 columns where the case has that many."""
    radsurf = _hip()
    have = [p for p in golden_files() if os.path.basename(p)[:-4] == name]
    if not have:
        pytest.skip("fixture not present")
    monkeypatch.setenv("SPARTACUS_SMALL_CALL_COLUMNS", "0")
    monkeypatch.setenv("SPARTACUS_HOST_CHUNK", "8")
    case, expected = fixture_io.load_case(have[0])
    cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
    if case["do_simple_spectrum"]:
        from spartacus_surface_amd.inputs import calc_simple_spectrum_lw
        calc_simple_spectrum_lw(cfg, cp, lw)
    out = _alloc(case)
    radsurf.radsurf(cfg, cp, sw, lw, out["bc_out"], 1, cp.ncol, out["sw_norm_dir"],
                    out["sw_norm_diff"], out["lw_internal"], out["lw_norm"])
    bad = fixture_io.parity_violations(fixture_io.flatten_outputs(out), expected, dict(case["ref_noise"]),
                                       truth=case["ref_truth"], noise_arr=case["ref_noise_arr"])
    assert not bad, "outside the parity bar: %s" % bad[:5]


@pytest.mark.parametrize("name", ["simple__test_surfaces", "simple__test_closed", "urban__2stream", "urban__1stream",
                                  "extra__ragged-tiles", "extra__urban-3stream", "rami4pilps__vis-med-0.3"])
def test_golden_with_symmetric_route_declined(name, monkeypatch):
    """Orders 3..6: a unit whose scaled matrices are not positive definite (a single-scattering albedo of exactly 1 in
    the longwave next to vegetation of zero extinction, say) is declined by the symmetrised route and computed again by
    the general route in an OUT-OF-LINE unit of the fused kernels (layer_unit_sw_cold / layer_unit_lw_cold).  No input
    of the reference's test-suite reaches that unit, and from round 3 until the edge-input sweep of round 4
    (tools/fuzz_parity.py --edge) it read its launch descriptor from address 0 on the device
    (csrc/ss_common.hpp: SS_LAUNCH_PTR_FOR_CALLEE).  SPARTACUS_DECLINE_SYM=1 sends EVERY unit there: same goldens,
    same bar.  (Not simple__test_overhang: at its degenerate layer the general route - eigenvector matrices inverted
    as the reference does - has the reference's kind of error, 3.8e-5 from the limit against the reference's 2.6e-5,
    1.1 x the element-wise bar; the symmetrised route is 6e-13 from it.)"""
    radsurf = _hip()
    have = [p for p in golden_files() if os.path.basename(p)[:-4] == name]
    if not have:
        pytest.skip("fixture not present")
    monkeypatch.setenv("SPARTACUS_DECLINE_SYM", "1")
    case, expected = fixture_io.load_case(have[0])
    cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
    out = _alloc(case)
    radsurf.radsurf(cfg, cp, sw, lw, out["bc_out"], 1, cp.ncol, out["sw_norm_dir"],
                    out["sw_norm_diff"], out["lw_internal"], out["lw_norm"])
    got = fixture_io.flatten_outputs(out)
    bad = fixture_io.parity_violations(got, expected, dict(case["ref_noise"]), truth=case["ref_truth"],
                                       noise_arr=case["ref_noise_arr"])
    assert not bad, "outside the parity bar: %s" % bad[:5]


@pytest.mark.parametrize("how", ["1", "2"])
@pytest.mark.parametrize("name", ["urban__4stream", "rami5__HET07_JPS_SUM-56-direct", "rami4pilps__vis-snw-0.3-2-4",
                                  "extra__urban-5stream"])
def test_golden_with_cooperative_route_declined(name, how, monkeypatch):
    """Orders 7..16: the second pass of the layer kernel (one lane per unit, general eigen-solver, on the
    plan's fallback stream) is where units without the symmetric structure go; well-formed inputs hardly
    ever reach it, so SPARTACUS_DECLINE_COOP=1 sends every full problem there.  Same goldens, same bar
    (the upward and downward sweeps stay on DPP rows).  =2 lets the cooperative route run and then treats every
    Jacobi iteration as not converged: the hand-over of such a problem to the second pass (status 0 + counter,
    csrc/ss_eigen_coop.hpp) is the same code a real non-convergence takes."""
    radsurf = _hip()
    have = [p for p in golden_files() if os.path.basename(p)[:-4] == name]
    if not have:
        pytest.skip("fixture not present")
    monkeypatch.setenv("SPARTACUS_DECLINE_COOP", how)
    case, expected = fixture_io.load_case(have[0])
    cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
    out = _alloc(case)
    radsurf.radsurf(cfg, cp, sw, lw, out["bc_out"], 1, cp.ncol, out["sw_norm_dir"],
                    out["sw_norm_diff"], out["lw_internal"], out["lw_norm"])
    got = fixture_io.flatten_outputs(out)
    bad = fixture_io.parity_violations(got, expected, dict(case["ref_noise"]), truth=case["ref_truth"],
                                       noise_arr=case["ref_noise_arr"])
    assert not bad, "outside the parity bar: %s" % bad[:5]


@pytest.mark.parametrize("name", ["simple__test_surfaces", "urban__2stream", "rami4pilps__vis-med-0.3"])
def test_golden_plan_api_with_device_spectrum(name):
    """ss_plan_*: upload once, calc_simple_spectrum_lw + solve on device,
    download; solving twice gives identical results; no eigen failures."""
    radsurf = _hip()
    path = [p for p in golden_files() if os.path.basename(p)[:-4] == name][0]
    case, expected = fixture_io.load_case(path)
    cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
    plan = radsurf.Plan(cfg, cp, sw, lw, want_profiles=True)
    outs = []
    for _ in range(2):
        if case["do_simple_spectrum"]:
            plan.calc_simple_spectrum_lw()
        plan.solve()
        out = _alloc(case)
        plan.download(out["bc_out"], out["sw_norm_dir"], out["sw_norm_diff"],
                      out["lw_internal"], out["lw_norm"])
        outs.append(fixture_io.flatten_outputs(out))
    assert plan.eigen_failures() == 0
    assert plan.kernel_ms("total") > 0.0
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k
    bad = fixture_io.parity_violations(outs[0], expected, case["ref_noise"], truth=case["ref_truth"],
                                       noise_arr=case["ref_noise_arr"])
    assert not bad, bad[:5]
    plan.close()


@pytest.mark.parametrize("name,nblock", [("simple__test_surfaces", None), ("urban__2stream", 16),
                                         ("rami4pilps__vis-med-0.3", 7),
                                         ("extra__single_layer-mixed", 5)])
def test_fortran_shim_boundary(name, nblock):
    """Reference derived types -> module radsurf_interface (ISO_C_BINDING shim,
    spartacus_surface_amd/fortran) -> spartacus_radsurf_hip, driven like the
    reference driver (OpenMP blocks of nblock columns, calc_simple_spectrum_lw
    per block on the host)."""
    from oracle import checker
    _hip()
    if not checker.have_shim():
        pytest.skip("oracle/_ref/libspartacus_shim.so not built")
    path = [p for p in golden_files() if os.path.basename(p)[:-4] == name][0]
    case, expected = fixture_io.load_case(path)
    out, _ = checker.run_shim(case["config"], case["canopy"], case["sw"], case["lw"],
                              want_profiles=True, nblocksize=nblock,
                              do_simple_spectrum=case["do_simple_spectrum"])
    bad = fixture_io.parity_violations(fixture_io.flatten_outputs(out), expected, case["ref_noise"],
                                       truth=case["ref_truth"], noise_arr=case["ref_noise_arr"])
    assert not bad, bad[:5]


def test_column_range_slices_only():
    """Only (:, istartcol:iendcol) and the matching layer slices are written
    (radsurf_interface.F90:105; driver blocks, spartacus_surface_driver.F90:206-233)."""
    radsurf = _hip()
    path = [p for p in golden_files() if p.endswith("urban__2stream.npz")][0]
    case, expected = fixture_io.load_case(path)
    cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
    out = _alloc(case)
    sentinel = -12345.0
    for obj in ("sw_norm_dir", "sw_norm_diff"):
        for a in out[obj].fields().values():
            a[...] = sentinel
    out["bc_out"].sw_albedo[...] = sentinel
    i0, i1 = 11, 30
    radsurf.radsurf(cfg, cp, sw, lw, out["bc_out"], i0, i1, out["sw_norm_dir"],
                    out["sw_norm_diff"], out["lw_internal"], out["lw_norm"])
    got = fixture_io.flatten_outputs(out)
    l0 = cp.istartlay[i0 - 1] - 1
    l1 = cp.istartlay[i1 - 1] - 1 + cp.nlay[i1 - 1]
    for k, ref in expected.items():
        if not k.startswith(("sw_norm", "bc_out.sw_albedo")) or k == "bc_out.sw_albedo_dir":
            continue
        per_col = ref.shape[0] == cp.ncol
        lo, hi = (i0 - 1, i1) if per_col else (l0, l1)
        inside, g = ref[lo:hi], got[k][lo:hi]
        assert np.allclose(g, inside, rtol=1e-5, atol=1e-9), k
        outside = np.concatenate([got[k][:lo].ravel(), got[k][hi:].ravel()])
        assert np.all(outside == sentinel), k


def test_error_behaviour():
    """A simple-urban tile with more than one layer is the reference's own
    fatal error (radsurf_interface.F90:281-284)."""
    radsurf = _hip()
    path = [p for p in golden_files() if p.endswith("urban__single.npz")][0]
    case, _ = fixture_io.load_case(path)
    cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
    assert cp.nlay.max() > 1
    cp.i_representation = np.full(cp.ncol, 4, dtype=np.int32)
    out = _alloc(case)
    with pytest.raises(radsurf.SpartacusHipError, match="SIMPLE_URBAN_LAYERS"):
        radsurf.radsurf(cfg, cp, sw, lw, out["bc_out"], 1, cp.ncol, out["sw_norm_dir"],
                        out["sw_norm_diff"], out["lw_internal"], out["lw_norm"])


def _check_against_oracle(case, ncheck, label=None):
    """Solve the whole case on the GPU (plan API) and compare its first ncheck
    columns with the CPU oracle under the parity bar; the reference's own
    rounding error per variable is measured on the spot with the binary128
    build of the oracle.  Returns (gpu outputs, [violations])."""
    from oracle import checker
    from spartacus_surface_amd import workloads
    radsurf = _hip()
    cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
    prof = case.get("want_profiles", False)
    plan = radsurf.Plan(cfg, cp, sw, lw, want_profiles=prof)
    plan.solve()
    out = checker.alloc_outputs(cfg, cp, prof)
    plan.download(out["bc_out"], out["sw_norm_dir"], out["sw_norm_diff"], out["lw_internal"],
                  out["lw_norm"])
    assert plan.eigen_failures() == 0
    plan.close()
    got = fixture_io.flatten_outputs(out)
    head = workloads.head_columns(case, ncheck)
    a = (head["config"], head["canopy"], head["sw"], head["lw"])
    ref = fixture_io.flatten_outputs(checker.run_oracle(*a, want_profiles=prof, nthreads=8)[0])
    truth = fixture_io.flatten_outputs(checker.run_oracle(*a, want_profiles=prof, nthreads=8,
                                                          quad=True)[0])
    noise, noise_arr = fixture_io.noise_from_truth(ref, truth)
    nl = int(head["canopy"].ntotlay)
    sl = lambda a: a[:ncheck] if a.shape[0] == cp.ncol else a[:nl]   # noqa: E731
    bad = fixture_io.parity_violations({k: sl(v) for k, v in got.items()}, ref, noise, truth=truth,
                                       noise_arr=noise_arr, collect=_REPORT if label else None, label=label)
    return got, bad


def test_synthetic_c3_vs_oracle():
    """Bench workload C3 (vegetated urban, 3 regions x 10 layers, SW+LW) at a
    size the oracle finishes in seconds."""
    from spartacus_surface_amd import workloads
    case = workloads.c3_vegetated_urban(ncol=4096, seed=12345)
    _, bad = _check_against_oracle(case, 4096, "synthetic/c3_4096")
    assert not bad, bad[:5]


def test_synthetic_c2_vs_oracle():
    from spartacus_surface_amd import workloads
    case = workloads.c2_rami4pilps(ncol=4600)
    _, bad = _check_against_oracle(case, 4600, "synthetic/c2_4600")
    assert not bad, bad[:5]


def test_synthetic_c4_mixed_multiband_vs_oracle():
    from spartacus_surface_amd import workloads
    case = workloads.c4_mixed(ncol=512, seed=12345)
    _, bad = _check_against_oracle(case, 512, "synthetic/c4_512")
    assert not bad, bad[:5]


def test_synthetic_c3_four_streams_vs_oracle():
    """C3 with 4 streams (n = 12: the kernels of the orders 7..16 on DPP rows), jittered many-column sample."""
    from spartacus_surface_amd import workloads
    case = workloads.c3_vegetated_urban(ncol=2048, seed=12345, nstream=4)
    _, bad = _check_against_oracle(case, 2048, "synthetic/c3ns4_2048")
    assert not bad, bad[:5]


@pytest.mark.parametrize("order,nveg,ns,urban_only", [(8, 1, 4, False), (16, 1, 8, False), (7, 2, 7, True)])
def test_synthetic_other_row_orders_vs_oracle(order, nveg, ns, urban_only):
    """The ends of the range of the row-of-16 kernels on the jittered C3 column set: order 8 (2 regions x 4
    streams), 16 (2 x 8: all 16 lanes of a row live) and 7 (urban tile without vegetation, 1 region x 7 streams);
    the goldens cover 9, 10, 12, 14 and 15."""
    from spartacus_surface_amd import types, workloads
    case = workloads.c3_vegetated_urban(ncol=256, seed=4321, nstream=ns, n_vegetation_region=nveg,
                                        tile=types.ITileUrban if urban_only else None)
    _, bad = _check_against_oracle(case, 256, "synthetic/c3_order%d_256" % order)
    assert not bad, bad[:5]


def test_synthetic_c3_fourteen_layers_vs_oracle():
    """Columns deeper than the per-lane LDS slots of the downward sweeps can stage (80 slots: 8 fields x 10
    layers): with 14 layers only five fields are staged and the others are stored as they are computed."""
    from spartacus_surface_amd import workloads
    case = workloads.with_more_layers(workloads.c3_vegetated_urban(ncol=512, seed=777), 4)
    _, bad = _check_against_oracle(case, 512, "synthetic/c3_14layers_512")
    assert not bad, bad[:5]


def test_synthetic_c4_four_streams_multiband_vs_oracle():
    """C4 with 4 streams: mixed forest / vegetated-urban columns, 14 SW + 16 LW bands, n = 12 in both tile kinds."""
    from spartacus_surface_amd import workloads
    case = workloads.c4_mixed(ncol=256, seed=12345, nstream=4)
    _, bad = _check_against_oracle(case, 256, "synthetic/c4ns4_256")
    assert not bad, bad[:5]


def test_ragged_all_tile_types_vs_oracle():
    """Every tile type (flat, forest, urban, vegetated urban, simple urban, infinite
    street) with 1..10 layers per column, several bands, day and night, in one call."""
    from spartacus_surface_amd import workloads
    case = workloads.ragged_all_tiles(ncol=640, seed=99)
    case["want_profiles"] = True
    _, bad = _check_against_oracle(case, 640, "synthetic/ragged_all_tiles_640")
    assert not bad, bad[:5]


def test_empty_column_range_is_a_no_op():
    """iendcol < istartcol: nothing is read or written (the reference's loop runs zero times)."""
    radsurf = _hip()
    path = [p for p in golden_files() if p.endswith("simple__test_surfaces.npz")][0]
    case, _ = fixture_io.load_case(path)
    out = _alloc(case)
    flat0 = {k: np.array(v, copy=True) for k, v in fixture_io.flatten_outputs(out).items()}
    radsurf.radsurf(case["config"], case["canopy"], case["sw"], case["lw"], out["bc_out"], 3, 2,
                    out["sw_norm_dir"], out["sw_norm_diff"], out["lw_internal"], out["lw_norm"])
    for k, v in fixture_io.flatten_outputs(out).items():
        assert np.array_equal(v, flat0[k]), k


def test_full_size_properties_c3():
    """Size-independent properties at the full bench size (1e6 columns x 10
    layers): (i) columns are independent, so the first 2048 columns of the big
    run equal a 2048-column run bit for bit and the oracle to 1e-5; (ii) every
    flux is finite; (iii) energy closes as well as in the reference
    (ground + air + walls + roofs + vegetation - top net ~ 0)."""
    from spartacus_surface_amd import workloads
    big = workloads.c3_vegetated_urban(ncol=1_000_000, seed=12345)
    got_big, bad = _check_against_oracle(big, 2048)
    assert not bad, bad[:5]
    small = workloads.head_columns(big, 2048)
    got_small, _ = _check_against_oracle(small, 2048)
    cp = small["canopy"]
    nl = int(cp.ntotlay)
    for k, v in got_small.items():
        head = got_big[k][:2048] if v.shape[0] == 2048 else got_big[k][:nl]
        assert np.array_equal(head, v), k
    for k, v in got_big.items():
        assert np.all(np.isfinite(v)), k
    # energy budget of the normalized direct SW fluxes, all 1e6 columns
    cpb = big["canopy"]
    L = int(cpb.nlay[0])
    lay = lambda n: got_big["sw_norm_dir." + n].reshape(cpb.ncol, L, -1).sum(axis=(1, 2))   # noqa: E731
    resid = (got_big["sw_norm_dir.ground_net"].sum(axis=1) + lay("clear_air_abs") + lay("veg_abs")
             + lay("veg_air_abs") + lay("roof_net") + lay("wall_net")
             - got_big["sw_norm_dir.top_net"].sum(axis=1))
    assert np.max(np.abs(resid)) < 5.0e-3


def test_pipes_and_chunking_do_not_change_results(monkeypatch):
    """The schedule (1 / 2 / 4 concurrent streams, one big chunk or many small
    ones) is invisible in the results: bit-identical outputs."""
    from oracle import checker
    from spartacus_surface_amd import workloads
    radsurf = _hip()
    case = workloads.c4_mixed(ncol=3000, seed=4242, nsw=3, nlw=2)
    cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]

    def run(pipes, scratch_mb):
        if scratch_mb is None:
            monkeypatch.delenv("SPARTACUS_SCRATCH_MB", raising=False)
        else:
            monkeypatch.setenv("SPARTACUS_SCRATCH_MB", str(scratch_mb))
        plan = radsurf.Plan(cfg, cp, sw, lw, want_profiles=True)
        plan.set_pipes(pipes)
        plan.solve()
        out = checker.alloc_outputs(cfg, cp, True)
        plan.download(out["bc_out"], out["sw_norm_dir"], out["sw_norm_diff"], out["lw_internal"],
                      out["lw_norm"])
        plan.close()
        return fixture_io.flatten_outputs(out)

    base = run(1, None)
    for pipes, mb in ((2, None), (4, 16), (1, 16), (2, 4)):
        other = run(pipes, mb)
        for k, v in base.items():
            assert np.array_equal(v, other[k]), (pipes, mb, k)


def test_coop_list_and_chunk_floor_do_not_change_results(monkeypatch):
    """Orders 7..16: the factor, Jacobi and layer kernels take their units from the dense list the first pass
    writes (default) or find them by position (SPARTACUS_NO_COOP_LIST=1); chunks are cut with a floor of 65 536
    column-units (default) or a smaller one (SPARTACUS_MIN_CHUNK_UNITS).  Neither is visible in the results:
    bit-identical outputs, on a column set with mixed tile kinds (cooperative and one-lane units in every wave)."""
    from oracle import checker
    from spartacus_surface_amd import workloads
    radsurf = _hip()
    case = workloads.c4_mixed(ncol=1500, seed=77, nsw=2, nlw=2, nstream=4)
    cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]

    def run(env):
        for k in ("SPARTACUS_NO_COOP_LIST", "SPARTACUS_MIN_CHUNK_UNITS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        plan = radsurf.Plan(cfg, cp, sw, lw, want_profiles=True)
        plan.solve()
        out = checker.alloc_outputs(cfg, cp, True)
        plan.download(out["bc_out"], out["sw_norm_dir"], out["sw_norm_diff"], out["lw_internal"],
                      out["lw_norm"])
        plan.close()
        return fixture_io.flatten_outputs(out)

    base = run({})
    for env in ({"SPARTACUS_NO_COOP_LIST": "1"}, {"SPARTACUS_MIN_CHUNK_UNITS": "512"},
                {"SPARTACUS_MIN_CHUNK_UNITS": "512", "SPARTACUS_NO_COOP_LIST": "1"}):
        other = run(env)
        for k, v in base.items():
            assert np.array_equal(v, other[k]), (env, k)


def _host_entry(radsurf, case, out, i0=1, i1=None):
    cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
    radsurf.radsurf(cfg, cp, sw, lw, out["bc_out"], i0, cp.ncol if i1 is None else i1, out["sw_norm_dir"],
                    out["sw_norm_diff"], out["lw_internal"], out["lw_norm"])
    return fixture_io.flatten_outputs(out)


def _sentinel_outputs(case, value=12345.678):
    out = _alloc(case)
    for a in fixture_io.flatten_outputs(out).values():
        a[...] = value
    return out


def test_host_entry_chunks_blocks_and_seeding(monkeypatch):
    """spartacus_radsurf_hip's pipeline is invisible in the results:
    (i) the column chunks that alternate between its two device plans (forced small
    here) give the same bits as one chunk; (ii) a driver that calls with a different
    16-column block every time (driver/spartacus_surface_driver.F90:206-233) gets the
    same bits and re-uses the two plans - none is re-created; (iii) skipping the upload of
    the caller's output arrays where every element is written anyway (canopy columns in
    daylight) equals always seeding them, on sentinel-filled outputs: no element of the
    range is left unwritten."""
    from spartacus_surface_amd import workloads
    radsurf = _hip()
    mixed = workloads.ragged_all_tiles(ncol=700, seed=5)
    mixed["want_profiles"] = True
    day = workloads.c4_mixed(ncol=700, seed=6, nsw=2, nlw=2)     # canopy columns only, sun up
    day["want_profiles"] = True
    assert float(day["canopy"].cos_sza.min()) > 0.0
    for case in (mixed, day):
        monkeypatch.delenv("SPARTACUS_HOST_CHUNK", raising=False)
        monkeypatch.setenv("SPARTACUS_ALWAYS_SEED", "1")
        base = _host_entry(radsurf, case, _sentinel_outputs(case))
        monkeypatch.delenv("SPARTACUS_ALWAYS_SEED")
        lean = _host_entry(radsurf, case, _sentinel_outputs(case))
        # the chunk pipeline of the large calls (the small-call path switched off), forced into small chunks
        monkeypatch.setenv("SPARTACUS_SMALL_CALL_COLUMNS", "0")
        monkeypatch.setenv("SPARTACUS_HOST_CHUNK", "96")
        chunked = _host_entry(radsurf, case, _sentinel_outputs(case))
        monkeypatch.delenv("SPARTACUS_HOST_CHUNK")
        n0 = radsurf.host_entry_plan_creations()
        out = _sentinel_outputs(case)
        ncol = case["canopy"].ncol
        for i0 in range(1, ncol + 1, 16):
            blocks_big = _host_entry(radsurf, case, out, i0, min(ncol, i0 + 15))
        assert radsurf.host_entry_plan_creations() == n0, "block calls re-created the device plans"
        # ... and the same blocks through the small-call path (a slot with a slab of its own)
        monkeypatch.delenv("SPARTACUS_SMALL_CALL_COLUMNS")
        n0 = radsurf.host_entry_plan_creations()
        out = _sentinel_outputs(case)
        for i0 in range(1, ncol + 1, 16):
            blocks = _host_entry(radsurf, case, out, i0, min(ncol, i0 + 15))
        assert radsurf.host_entry_plan_creations() == n0, "block calls re-created the device plans"
        for k, v in base.items():
            assert np.array_equal(v, lean[k]), ("seed", k)
            assert np.array_equal(v, chunked[k]), ("chunk", k)
            assert np.array_equal(v, blocks_big[k]), ("blocks, chunk pipeline", k)
            assert np.array_equal(v, blocks[k]), ("blocks, small-call path", k)


def test_plan_follows_its_device_and_streams_are_per_plan():
    """Two plans alive at once own their streams (no process-global stream state):
    interleaved solves give the same bits as solving them one after the other, and
    ss_set_device does not disturb a live plan."""
    from oracle import checker
    from spartacus_surface_amd import workloads
    radsurf = _hip()
    cases = [workloads.c3_vegetated_urban(ncol=3000, seed=s) for s in (1, 2)]

    def dl(plan, case):
        out = checker.alloc_outputs(case["config"], case["canopy"], False)
        plan.download(out["bc_out"], out["sw_norm_dir"], out["sw_norm_diff"], out["lw_internal"], out["lw_norm"])
        return fixture_io.flatten_outputs(out)

    solo = []
    for c in cases:
        p = radsurf.Plan(c["config"], c["canopy"], c["sw"], c["lw"], want_profiles=False)
        p.solve()
        solo.append(dl(p, c))
        p.close()
    plans = [radsurf.Plan(c["config"], c["canopy"], c["sw"], c["lw"], want_profiles=False) for c in cases]
    assert plans[0].device() == 0
    for _ in range(3):
        for p in plans:
            p.solve()               # queued on each plan's own streams, no synchronisation in between
    radsurf.set_device(0)
    both = [dl(p, c) for p, c in zip(plans, cases)]
    for p in plans:
        p.close()
    for a, b in zip(solo, both):
        for k, v in a.items():
            assert np.array_equal(v, b[k]), k
