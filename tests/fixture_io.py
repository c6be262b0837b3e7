"""Golden-fixture (de)serialisation: one .npz per case holding the solver
INPUTS (config scalars + the packed arrays of canopy_props / sw / lw spectral
properties) and the EXPECTED OUTPUTS of the real reference (bc_out + the four
canopy_flux objects, fp64, before scale/sum).  Pure data - no reference code.
"""
import json
from dataclasses import fields

import numpy as np

from spartacus_surface_amd import abi
from spartacus_surface_amd.types import (BoundaryCondsOut, CanopyFlux, CanopyProperties, Config,
                                         LwSpectralProperties, SwSpectralProperties)

_CFG_SKIP = {"lg_sw_urban", "lg_sw_forest", "lg_lw_urban", "lg_lw_forest"}
OUT_OBJECTS = ("sw_norm_dir", "sw_norm_diff", "lw_internal", "lw_norm")


def pack_case(case, outputs=None, want_profiles=True):
    cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
    d = {}
    meta = {f.name: getattr(cfg, f.name) for f in fields(Config) if f.name not in _CFG_SKIP}
    meta["do_simple_spectrum"] = bool(case.get("do_simple_spectrum", False))
    meta["want_profiles"] = bool(want_profiles)
    d["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    for n in abi.CANOPY_INT_FIELDS + abi.CANOPY_REAL_FIELDS:
        a = getattr(cp, n)
        if a is not None:
            d["canopy." + n] = a
    for n in abi.SW_FIELDS:
        a = getattr(sw, n)
        if a is not None:
            d["sw." + n] = a
    for n in abi.LW_FIELDS:
        a = getattr(lw, n)
        if a is not None:
            d["lw." + n] = a
    for k in ("top_flux_dn_sw", "top_flux_dn_direct_sw", "top_flux_dn_lw"):
        if case.get(k) is not None:
            d["driver." + k] = case[k]
    if outputs is not None:
        for n in abi.BC_FIELDS:
            d["out.bc_out." + n] = getattr(outputs["bc_out"], n)
        for obj in OUT_OBJECTS:
            for n, a in outputs[obj].fields().items():
                d["out.%s.%s" % (obj, n)] = a
    return d


def save_case(path, case, outputs, want_profiles=True):
    np.savez_compressed(path, **pack_case(case, outputs, want_profiles))


def load_case(path):
    """Returns (case dict like tests.cases builders, expected-outputs dict
    {'bc_out.sw_albedo': array, 'sw_norm_dir.ground_dn': array, ...})."""
    z = np.load(path)
    meta = json.loads(bytes(z["meta"]).decode())
    do_simple = meta.pop("do_simple_spectrum")
    want_profiles = meta.pop("want_profiles")
    cfg = Config(**meta)
    cfg.consolidate()
    cp = CanopyProperties()
    sw = SwSpectralProperties(nspec=cfg.nsw)
    lw = LwSpectralProperties(nspec=cfg.nlw)
    expected, drv = {}, {}
    for k in z.files:
        if k.startswith("canopy."):
            setattr(cp, k[7:], np.array(z[k]))
        elif k.startswith("sw."):
            setattr(sw, k[3:], np.array(z[k]))
        elif k.startswith("lw."):
            setattr(lw, k[3:], np.array(z[k]))
        elif k.startswith("driver."):
            drv[k[7:]] = np.array(z[k])
        elif k.startswith("out."):
            expected[k[4:]] = np.array(z[k])
    cp.ncol = int(cp.nlay.size)
    cp.ntotlay = int(cp.nlay.sum())
    sw.ncol = lw.ncol = cp.ncol
    sw.ntotlay = lw.ntotlay = cp.ntotlay
    cp.normalise(); sw.normalise(); lw.normalise()
    case = dict(config=cfg, canopy=cp, sw=sw, lw=lw, do_simple_spectrum=do_simple,
                want_profiles=want_profiles, **drv)
    return case, expected


def flatten_outputs(outputs):
    """outputs dict of objects -> {'bc_out.x': arr, 'sw_norm_dir.y': arr}."""
    flat = {}
    for n in abi.BC_FIELDS:
        a = getattr(outputs["bc_out"], n)
        if a is not None:
            flat["bc_out." + n] = a
    for obj in OUT_OBJECTS:
        for n, a in outputs[obj].fields().items():
            flat["%s.%s" % (obj, n)] = a
    return flat


def max_rel_error(got, expected, floor=1.0e-3):
    """Metric of SURVEY.md section 8(d): max over every output variable of
    |got-ref| / max(|ref|, floor * max|ref of that variable|).  Returns
    (error, name of the worst variable).  NaN anywhere -> inf."""
    worst, where = 0.0, None
    for k, ref in expected.items():
        g = got[k]
        if ref.size == 0:
            continue
        scale = max(float(np.max(np.abs(ref))), 1.0e-300)
        err = np.abs(g - ref) / np.maximum(np.abs(ref), floor * scale)
        e = float(np.max(err)) if np.all(np.isfinite(g)) else float("inf")
        if not np.isfinite(e):
            e = float("inf")
        if e > worst:
            worst, where = e, k
    return worst, where
