"""canopy_flux%scale / %sum (reference
radsurf_canopy_flux.F90:212-282, 399-460 as applied by
spartacus_surface_driver.F90:250-261).

Pins: tests/golden/scale_sum/*.npz hold what the REAL reference produces
(make_golden_scale_sum.py).  The products and sums are single roundings, so
every check of the scale/sum step itself is bit-exact: the step is applied to
the normalised fluxes the implementation under test produced.
"""
import glob
import os

import numpy as np
import pytest

from tests import fixture_io

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "scale_sum", "*.npz")))


def _load(path):
    name = os.path.basename(path)[:-4]
    case, expected = fixture_io.load_case(os.path.join(HERE, "golden", name + ".npz"))
    if case["do_simple_spectrum"]:
        from spartacus_surface_amd.inputs import calc_simple_spectrum_lw
        calc_simple_spectrum_lw(case["config"], case["canopy"], case["lw"])
    z = np.load(path)
    return name, case, expected, {k: np.array(z[k]) for k in z.files}


def _mirror(case, flat, t_sw, t_dir, t_lw):
    """The host mirror (types.CanopyFlux.scale / sum) applied to normalised fluxes
    given as a flat dict; returns {'scaled.<obj>.<x>', 'sw_flux.<x>', 'lw_flux.<x>'}."""
    from spartacus_surface_amd import types
    cfg, cp = case["config"], case["canopy"]
    objs = {}
    for obj, nspec, direct in (("sw_norm_dir", cfg.nsw, True), ("sw_norm_diff", cfg.nsw, True),
                               ("lw_internal", cfg.nlw, False), ("lw_norm", cfg.nlw, False)):
        f = types.CanopyFlux()
        f.allocate(cfg, cp.ncol, cp.ntotlay, nspec, use_direct=direct)
        for n in list(f.fields()):
            key = "%s.%s" % (obj, n)
            if key in flat:
                getattr(f, n)[...] = flat[key]
            else:
                setattr(f, n, None)
        objs[obj] = f
    res = {}
    if cfg.do_sw:
        objs["sw_norm_dir"].scale(cp.nlay, t_dir)
        objs["sw_norm_diff"].scale(cp.nlay, t_sw - t_dir)
        s = types.CanopyFlux()
        s.allocate(cfg, cp.ncol, cp.ntotlay, cfg.nsw, use_direct=True)
        s.sum(objs["sw_norm_dir"], objs["sw_norm_diff"])
        res.update({"sw_flux." + k: v for k, v in s.fields().items()})
    if cfg.do_lw:
        objs["lw_norm"].scale(cp.nlay, t_lw)
        s = types.CanopyFlux()
        s.allocate(cfg, cp.ncol, cp.ntotlay, cfg.nlw, use_direct=False)
        s.sum(objs["lw_internal"], objs["lw_norm"])
        res.update({"lw_flux." + k: v for k, v in s.fields().items()})
    for obj, f in objs.items():
        if (obj.startswith("sw") and cfg.do_sw) or (obj.startswith("lw") and cfg.do_lw):
            res.update({"scaled.%s.%s" % (obj, k): v for k, v in f.fields().items()})
    return res


def _assert_bitwise(got, want, what):
    for k, w in want.items():
        if k not in got:
            continue
        assert np.array_equal(np.asarray(got[k]).reshape(w.shape), w), "%s: %s differs" % (what, k)


@pytest.mark.parametrize("path", FILES, ids=lambda p: os.path.basename(p)[:-4])
def test_host_mirror_reproduces_reference(path):
    """types.CanopyFlux.scale/sum applied to the reference's normalised fluxes ==
    the reference's own scaled objects and sums, bit for bit."""
    name, case, expected, gold = _load(path)
    norm = {k[5:]: v for k, v in gold.items() if k.startswith("norm.")}
    res = _mirror(case, norm, gold["top_flux_dn_sw"], gold["top_flux_dn_direct_sw"],
                  gold["top_flux_dn_lw"])
    want = {k: v for k, v in gold.items() if k.startswith(("sw_flux.", "lw_flux.", "scaled."))
            and not k.startswith("scaled.bc_out")}
    assert want
    _assert_bitwise(res, want, name)


@pytest.mark.parametrize("path", FILES, ids=lambda p: os.path.basename(p)[:-4])
def test_emulated_device_scale_sum(path):
    """The device code of scale_and_sum (host build of the same source) == the
    host mirror applied to the same normalised fluxes, bit for bit."""
    from tests import emul
    name, case, _, gold = _load(path)
    t = (gold["top_flux_dn_sw"], gold["top_flux_dn_direct_sw"], gold["top_flux_dn_lw"])
    plain, _ = emul.run_emul(case["config"], case["canopy"], case["sw"], case["lw"], want_profiles=True)
    out, sw_flux, lw_flux = emul.run_scale_sum_emul(case["config"], case["canopy"], case["sw"], case["lw"], *t)
    want = _mirror(case, fixture_io.flatten_outputs(plain), *t)
    got = {"scaled." + k: v for k, v in fixture_io.flatten_outputs(out).items()}
    if case["config"].do_sw:
        got.update({"sw_flux." + k: v for k, v in sw_flux.fields().items()})
    if case["config"].do_lw:
        got.update({"lw_flux." + k: v for k, v in lw_flux.fields().items()})
    _assert_bitwise(got, want, name)


def test_emulated_scale_sum_on_a_column_range():
    """Only the plan's column range [istartcol, iendcol] is scaled and summed
    (odd offsets exercise the unaligned ends of the 16-byte pair kernel)."""
    from tests import emul
    path = [p for p in FILES if "test_surfaces" in p][0]
    name, case, _, gold = _load(path)
    t = (gold["top_flux_dn_sw"], gold["top_flux_dn_direct_sw"], gold["top_flux_dn_lw"])
    cp = case["canopy"]
    i0, i1 = 2, 3
    full_out, full_sw, full_lw = emul.run_scale_sum_emul(case["config"], cp, case["sw"], case["lw"], *t)
    out, sw_flux, lw_flux = emul.run_scale_sum_emul(case["config"], cp, case["sw"], case["lw"], *t,
                                                    istartcol=i0, iendcol=i1)
    l0 = int(cp.istartlay[i0 - 1] - 1)
    l1 = int(cp.istartlay[i1 - 1] - 1 + cp.nlay[i1 - 1])
    for got, ref in ((sw_flux, full_sw), (lw_flux, full_lw)):
        for k, v in got.fields().items():
            r = ref.fields()[k]
            if v.shape[0] == cp.ncol:
                assert np.array_equal(v[i0 - 1:i1], r[i0 - 1:i1]), k
                assert not np.any(v[:i0 - 1]) and not np.any(v[i1:]), k      # untouched outside
            else:
                assert np.array_equal(v[l0:l1], r[l0:l1]), k
                assert not np.any(v[:l0]) and not np.any(v[l1:]), k


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=lambda p: os.path.basename(p)[:-4])
def test_device_scale_sum(path):
    """ss_plan_scale_and_sum on the GPU == the host mirror applied to the plan's own
    normalised results (bit for bit), and within the parity bar of the reference's sums."""
    from oracle import checker
    from spartacus_surface_amd import radsurf, types
    radsurf.load_library()
    name, case, _, gold = _load(path)
    cfg, cp = case["config"], case["canopy"]
    t = (gold["top_flux_dn_sw"], gold["top_flux_dn_direct_sw"], gold["top_flux_dn_lw"])
    plan = radsurf.Plan(cfg, cp, case["sw"], case["lw"], want_profiles=True)
    plan.solve()
    plain = checker.alloc_outputs(cfg, cp, True)
    plan.download(plain["bc_out"], plain["sw_norm_dir"], plain["sw_norm_diff"], plain["lw_internal"], plain["lw_norm"])
    plan.scale_and_sum(*t)
    out = checker.alloc_outputs(cfg, cp, True)
    plan.download(out["bc_out"], out["sw_norm_dir"], out["sw_norm_diff"], out["lw_internal"], out["lw_norm"])
    sums = []
    for nspec, direct in ((cfg.nsw, True), (cfg.nlw, False)):
        f = types.CanopyFlux()
        f.allocate(cfg, cp.ncol, cp.ntotlay, nspec, use_direct=direct)
        sums.append(f)
    plan.download_sums(sums[0] if cfg.do_sw else None, sums[1] if cfg.do_lw else None)
    plan.close()
    want = _mirror(case, fixture_io.flatten_outputs(plain), *t)
    got = {"scaled." + k: v for k, v in fixture_io.flatten_outputs(out).items()}
    if cfg.do_sw:
        got.update({"sw_flux." + k: v for k, v in sums[0].fields().items()})
    if cfg.do_lw:
        got.update({"lw_flux." + k: v for k, v in sums[1].fields().items()})
    _assert_bitwise(got, want, name)
    # ... and against the reference's own sums.  A sum can cancel (LW absorption
    # minus emission), so the bar is the solver's parity bar of the two summands
    # (max(1e-5, 4 x reference noise) of each, relative to its own scale), added.
    noise = case["ref_noise"]
    pairs = {"sw_flux": ("sw_norm_dir", "sw_norm_diff"), "lw_flux": ("lw_internal", "lw_norm")}
    for k, ref in gold.items():
        if not k.startswith(("sw_flux.", "lw_flux.")):
            continue
        obj, member = k.split(".", 1)
        tol = 0.0
        for part in pairs[obj]:
            sk = "scaled.%s.%s" % (part, member)
            if sk in gold:
                bar = max(fixture_io.PARITY_TOL, fixture_io.NOISE_FACTOR * noise.get("%s.%s" % (part, member), 0.0))
                tol += bar * float(np.max(np.abs(gold[sk]))) if gold[sk].size else 0.0
        diff = float(np.max(np.abs(np.asarray(got[k]).reshape(ref.shape) - ref))) if ref.size else 0.0
        assert diff <= tol + 1.0e-300, (k, diff, tol)
