"""Synthetic many-column workloads C2-C4 of SURVEY.md section 8(d) /
BASELINE.json "configs".  Deterministic (seeded numpy Generator); the base
profiles are DATA taken from the reference's test inputs (rami4pilps base
profile, test/urban/russell_square.nc, the rami5 HET09 band table) so the
generators also work where /root/reference is absent (the GPU box).

Each generator returns the same dict the test-case builders return:
config, canopy, sw, lw (+ top-of-canopy fluxes for scale/sum).
"""
import numpy as np

from .inputs import COS_SZA_46
from .types import (CanopyProperties, Config, LwSpectralProperties, StefanBoltzmann,
                    SwSpectralProperties, ITileForest, ITileVegetatedUrban)

_f32 = lambda x: float(np.float32(x))   # noqa: E731  (values stored as float32 in the inputs)

# ---- test/urban/russell_square.nc (8 layers) ----
RSQ_HEIGHT = [0.0, 4.0, 12.0, 20.0, 28.0, 36.0, 44.0, 56.0, 80.0]
RSQ_BUILDING_FRACTION = [0.46444010734558105, 0.4606631100177765, 0.4114302098751068,
                         0.2802385091781616, 0.13588160276412964, 0.04295969009399414,
                         0.020455999299883842, 0.003069968894124031]
RSQ_VEG_FRACTION = [0.07981766015291214, 0.07981766015291214, 0.0647466629743576,
                    0.024303359910845757, 0.009215861558914185, 0.00034636061172932386, 0.0, 0.0]
RSQ_BUILDING_SCALE = [31.932449340820312, 32.131038665771484, 33.43254089355469,
                      37.870059967041016, 39.07585906982422, 37.0328483581543,
                      36.56882095336914, 36.15266036987305]
RSQ_VEG_SCALE = [11.290510177612305, 13.262040138244629, 15.478409767150879,
                 18.937580108642578, 20.658449172973633, 21.0, 21.0, 21.0]

# ---- rami5 HET09 band table (14 SW bands; last = "black" band) ----
HET09_VEG_SSA = [0.05660540983080864, 0.06309626996517181, 0.1607513725757599,
                 0.06567180156707764, 0.06794755160808563, 0.3296867907047272,
                 0.838228166103363, 0.86658775806427, 0.8735336065292358, 0.7953718304634094,
                 0.480884850025177, 0.22311468422412872, 0.25327175855636597, 0.0]
HET09_GROUND_ALBEDO = [0.014841999858617783, 0.018151000142097473, 0.04529799893498421,
                       0.027396999299526215, 0.03082199953496456, 0.09007299691438675,
                       0.19145700335502625, 0.20870399475097656, 0.21650800108909607,
                       0.1846349984407425, 0.09819599986076355, 0.041593998670578,
                       0.04797299951314926, 0.0]


def russell_square_10_layers():
    """The 8-layer Russell Square profile extended to 10 layers by splitting its
    two thickest layers (44-56 m and 56-80 m) into halves with the properties of
    the parent layer.  Returns dict of per-layer arrays (10,)."""
    h = np.array(RSQ_HEIGHT)
    dz = np.diff(h)
    order = list(range(8))
    split = set(np.argsort(-dz, kind="stable")[:2].tolist())
    idx, new_dz = [], []
    for l in order:
        if l in split:
            idx += [l, l]
            new_dz += [0.5 * dz[l], 0.5 * dz[l]]
        else:
            idx.append(l)
            new_dz.append(dz[l])
    idx = np.array(idx)
    return dict(dz=np.array(new_dz),
                building_fraction=np.array(RSQ_BUILDING_FRACTION)[idx],
                veg_fraction=np.array(RSQ_VEG_FRACTION)[idx],
                building_scale=np.array(RSQ_BUILDING_SCALE)[idx],
                veg_scale=np.array(RSQ_VEG_SCALE)[idx])


def _finish(cfg, cp, sw, lw, **extra):
    cp.normalise(); sw.normalise(); lw.normalise()
    sw.ncol = lw.ncol = cp.ncol
    sw.ntotlay = lw.ntotlay = cp.ntotlay
    d = dict(config=cfg, canopy=cp, sw=sw, lw=lw, do_simple_spectrum=False, want_profiles=False)
    d.update(extra)
    return d


def c2_rami4pilps(ncol=100_000, nstream=1, n_vegetation_region=1):
    """C2: RAMI4PILPS open forest canopy, scenario vis-med-0.3
    (test/rami4pilps/Makefile:36, expand_args.sh): forest, SW only, 1 band,
    1 stream, 2 regions, 2 layers; the base column replicated with cos_sza
    cycling through the 46 values of duplicate_profiles.sh:19.  nstream=8, n_vegetation_region=2 is the
    method run "-2-8" of test/rami4pilps/Makefile:31-32 (3 regions x 8 streams: layer problems of order 24)."""
    cfg = Config(do_sw=True, do_lw=False, do_vegetation=True, do_urban=False,
                 n_vegetation_region_forest=n_vegetation_region, n_vegetation_region_urban=1, nsw=1, nlw=1,
                 n_stream_sw_forest=nstream, n_stream_sw_urban=2,
                 use_symmetric_vegetation_scale_forest=False).consolidate()
    L = 2
    cp = CanopyProperties().set_layers(np.full(ncol, L, dtype=np.int32))
    cp.i_representation = np.full(ncol, ITileForest, dtype=np.int32)
    cp.cos_sza = np.resize(COS_SZA_46, ncol).astype(np.float64)
    cp.dz = np.tile([4.0, 10.0], ncol)
    cp.veg_fraction = np.full(ncol * L, 0.3)
    cp.veg_scale = np.full(ncol * L, 10.0)
    cp.veg_ext = np.tile([0.0, 0.25], ncol)
    cp.veg_fsd = np.full(ncol * L, 0.29289)
    sw = SwSpectralProperties(nspec=1)
    sw.air_ext = np.full((ncol * L, 1), 1.0e-5)
    sw.air_ssa = np.full((ncol * L, 1), 0.999)
    sw.veg_ssa = np.full((ncol * L, 1), 0.1301)
    sw.ground_albedo = np.full((ncol, 1), 0.1217)
    lw = LwSpectralProperties(nspec=1)
    return _finish(cfg, cp, sw, lw, top_flux_dn_sw=np.ones((ncol, 1)),
                   top_flux_dn_direct_sw=np.ones((ncol, 1)))


def rami5_replicated(ncol=20_000, scene="HET09_JBS_SUM-41-direct"):
    """BASELINE config 5 as a throughput workload: one column of the reference's test/rami5 run
    `scene` (its inputs as stored in tests/golden/rami5__<scene>.npz by tests/golden/make_golden.py:
    vegetated tile with 3 regions x 4 streams - 12 x 12 matrices -, 40-62 layers, 14 shortwave bands,
    no longwave) replicated `ncol` times."""
    import json
    import os
    # inputs of that one column (data of the reference's test/rami5 scene files, extracted by
    # tools/extract_rami5_column.py): the package does not read the test fixtures
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "rami5__%s_inputs.npz" % scene)
    z = np.load(path)
    meta = json.loads(bytes(z["meta"]).decode())
    meta.pop("do_simple_spectrum")
    meta.pop("want_profiles")
    cfg = Config(**meta)
    cfg.consolidate()
    nlay0 = int(np.array(z["canopy.nlay"])[0])
    assert np.array(z["canopy.nlay"]).size == 1
    cp = CanopyProperties().set_layers(np.full(ncol, nlay0, dtype=np.int32))
    sw = SwSpectralProperties(nspec=cfg.nsw)
    lw = LwSpectralProperties(nspec=cfg.nlw)
    extra = {}
    for k in z.files:
        for prefix, obj in (("canopy.", cp), ("sw.", sw), ("lw.", lw), ("driver.", None)):
            if not k.startswith(prefix):
                continue
            name, a = k[len(prefix):], np.array(z[k])
            if name in ("nlay", "istartlay"):
                continue
            if a.ndim >= 1 and a.shape[0] in (1, nlay0):
                a = np.tile(a, (ncol,) + (1,) * (a.ndim - 1))
            if obj is None:
                extra[name] = a
            else:
                setattr(obj, name, a)
    return _finish(cfg, cp, sw, lw, **extra)


def _urban_columns(rng, ncol, nsw, nlw):
    """ncol jittered Russell-Square columns (10 layers): canopy + SW + LW arrays."""
    base = russell_square_10_layers()
    L = 10
    jit = lambda: rng.uniform(0.9, 1.1, size=(ncol, 1))   # noqa: E731  one factor per column
    bf = base["building_fraction"][None, :] * jit()
    vf = base["veg_fraction"][None, :] * jit()
    tot = bf + vf
    over = tot > 0.98
    scale = np.where(over, 0.98 / np.maximum(tot, 1e-30), 1.0)
    bf, vf = bf * scale, vf * scale
    cp = dict(
        dz=np.tile(base["dz"], (ncol, 1)),
        building_fraction=bf, veg_fraction=vf,
        building_scale=base["building_scale"][None, :] * jit(),
        veg_scale=base["veg_scale"][None, :] * jit(),
        veg_ext=np.full((ncol, L), 0.25), veg_fsd=np.full((ncol, L), 0.5))
    cp["veg_contact_fraction"] = np.minimum(1.0, cp["veg_fraction"] / np.maximum(
        _f32(1.0e-6), 1.0 - cp["building_fraction"]))
    alb = lambda v, n: np.clip(v * jit()[:, :, None] * np.ones((ncol, L, n)), 0.0, 0.95)   # noqa: E731
    return base, cp, alb


def c3_vegetated_urban(ncol=1_000_000, seed=12345, nstream=2, n_vegetation_region=2, tile=None):
    """C3: test/urban vegetated-urban column on 10 layers, 3 regions, 2 streams,
    1 SW + 1 LW band, SW+LW; per-column multiplicative jitter U(0.9,1.1) on
    building/vegetation fractions, scales and albedos (building+veg clipped to
    0.98), cos_sza ~ U(0.05,1); temperatures of test_surfaces_in.nc.
    n_vegetation_region (1: two regions) and tile (ITileUrban: one region, the vegetation arrays are ignored)
    give the other matrix orders on the same column set (parity tests of the orders 7, 8, 16)."""
    rng = np.random.default_rng(seed)
    cfg = Config(do_sw=True, do_lw=True, do_vegetation=True, do_urban=True,
                 n_vegetation_region_forest=n_vegetation_region, n_vegetation_region_urban=n_vegetation_region,
                 nsw=1, nlw=1,
                 n_stream_sw_forest=nstream, n_stream_sw_urban=nstream, n_stream_lw_forest=nstream,
                 n_stream_lw_urban=nstream).consolidate()
    L = 10
    base, cpa, alb = _urban_columns(rng, ncol, 1, 1)
    cp = CanopyProperties().set_layers(np.full(ncol, L, dtype=np.int32))
    cp.i_representation = np.full(ncol, ITileVegetatedUrban if tile is None else tile, dtype=np.int32)
    cp.cos_sza = rng.uniform(0.05, 1.0, size=ncol)
    for k, v in cpa.items():
        setattr(cp, k, v.reshape(-1))
    nl = ncol * L
    sw = SwSpectralProperties(nspec=1)
    sw.air_ext = np.full((nl, 1), 1.0e-5)
    sw.air_ssa = np.full((nl, 1), 0.999)
    sw.veg_ssa = np.full((nl, 1), _f32(0.13))
    sw.ground_albedo = np.clip(_f32(0.2) * rng.uniform(0.9, 1.1, size=(ncol, 1)), 0.0, 0.95)
    sw.roof_albedo = alb(_f32(0.2), 1).reshape(nl, 1)
    sw.roof_albedo_dir = sw.roof_albedo.copy()
    sw.wall_albedo = alb(_f32(0.2), 1).reshape(nl, 1)
    sw.wall_specular_frac = np.zeros((nl, 1))
    lw = LwSpectralProperties(nspec=1)
    lw.air_ext = np.full((nl, 1), 1.0e-5)
    lw.air_ssa = np.zeros((nl, 1))
    lw.veg_ssa = np.full((nl, 1), _f32(0.01))
    t_air, t_surf, t_sky = _f32(278.15), _f32(283.15), _f32(268.15)
    em = _f32(0.9)
    lw.clear_air_planck = np.full((nl, 1), StefanBoltzmann * t_air ** 4)
    lw.veg_planck = lw.clear_air_planck.copy()
    lw.veg_air_planck = lw.clear_air_planck.copy()
    lw.ground_emissivity = np.full((ncol, 1), em)
    lw.roof_emissivity = np.full((nl, 1), em)
    lw.wall_emissivity = np.full((nl, 1), em)
    lw.ground_emission = np.full((ncol, 1), StefanBoltzmann * em * t_surf ** 4)
    lw.roof_emission = np.full((nl, 1), StefanBoltzmann * em * t_surf ** 4)
    lw.wall_emission = lw.roof_emission.copy()
    return _finish(cfg, cp, sw, lw, top_flux_dn_sw=np.full((ncol, 1), 500.0),
                   top_flux_dn_direct_sw=np.full((ncol, 1), 400.0),
                   top_flux_dn_lw=np.full((ncol, 1), StefanBoltzmann * t_sky ** 4))


def with_more_layers(case, extra):
    """The same columns with their lowest `extra` layers repeated on top (a test input: columns deeper than
    the 10 layers of the bench workloads, e.g. for the paths of the kernels that depend on the layer count)."""
    from . import abi
    cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
    ncol = cp.ncol
    L = int(cp.nlay[0])
    assert np.all(np.asarray(cp.nlay) == L) and 0 < extra <= L
    for obj, names in ((cp, abi.CANOPY_REAL_FIELDS), (sw, abi.SW_FIELDS), (lw, abi.LW_FIELDS)):
        for n in names:
            a = getattr(obj, n)
            if a is not None and a.shape[0] == ncol * L:
                b = a.reshape((ncol, L) + a.shape[1:])
                b = np.concatenate([b, b[:, :extra]], axis=1)
                setattr(obj, n, np.ascontiguousarray(b.reshape((ncol * (L + extra),) + a.shape[1:])))
    cp.set_layers(np.full(ncol, L + extra, dtype=np.int32))
    return case


def c4_mixed(ncol=1_000_000, seed=12345, nsw=14, nlw=16, nstream=2):
    """C4: 50/50 forest (10 layers, 3 regions, `nstream` streams) and C3-like vegetated
    urban columns, 14 SW + 16 LW bands.  SW band values from the rami5 HET09
    table; LW Planck / emission split uniformly over the bands (the reference
    has no LW band model: calc_simple_spectrum_lw aborts for nspec > 1, so the
    arrays are supplied directly, as the API allows)."""
    rng = np.random.default_rng(seed)
    cfg = Config(do_sw=True, do_lw=True, do_vegetation=True, do_urban=True,
                 n_vegetation_region_forest=2, n_vegetation_region_urban=2, nsw=nsw, nlw=nlw,
                 n_stream_sw_forest=nstream, n_stream_sw_urban=nstream, n_stream_lw_forest=nstream,
                 n_stream_lw_urban=nstream).consolidate()
    L = 10
    base, cpa, alb = _urban_columns(rng, ncol, nsw, nlw)
    is_forest = (np.arange(ncol) % 2 == 0)
    cp = CanopyProperties().set_layers(np.full(ncol, L, dtype=np.int32))
    cp.i_representation = np.where(is_forest, ITileForest, ITileVegetatedUrban).astype(np.int32)
    cp.cos_sza = rng.uniform(0.05, 1.0, size=ncol)
    # forest columns: C2-like canopy on 10 layers, lowest 3 layers sparse trunk
    # space.  (Exactly zero extinction in a vegetated LW layer makes Gamma2 = 0 and
    # the reference's own result rounding noise - cf. test_noscat - so it is avoided.)
    f_vf = np.tile(np.array([0.3] * L), (ncol, 1)) * rng.uniform(0.9, 1.1, size=(ncol, 1))
    f_ext = np.tile(np.array([0.05] * 3 + [0.25] * 7), (ncol, 1))
    cpa["veg_fraction"] = np.where(is_forest[:, None], f_vf, cpa["veg_fraction"])
    cpa["veg_ext"] = np.where(is_forest[:, None], f_ext, cpa["veg_ext"])
    cpa["veg_scale"] = np.where(is_forest[:, None], 10.0, cpa["veg_scale"])
    cpa["dz"] = np.where(is_forest[:, None], 1.4, cpa["dz"])
    for k, v in cpa.items():
        setattr(cp, k, v.reshape(-1))
    nl = ncol * L
    ssa = np.resize(np.array(HET09_VEG_SSA), nsw)
    galb = np.resize(np.array(HET09_GROUND_ALBEDO), nsw)
    sw = SwSpectralProperties(nspec=nsw)
    sw.air_ext = np.full((nl, nsw), 1.0e-5)
    sw.air_ssa = np.full((nl, nsw), 0.999)
    sw.veg_ssa = np.tile(ssa, (nl, 1))
    sw.ground_albedo = np.clip(galb[None, :] * rng.uniform(0.9, 1.1, size=(ncol, 1)), 0.0, 0.95)
    sw.roof_albedo = alb(_f32(0.2), nsw).reshape(nl, nsw)
    sw.roof_albedo_dir = sw.roof_albedo.copy()
    sw.wall_albedo = alb(_f32(0.2), nsw).reshape(nl, nsw)
    sw.wall_specular_frac = np.zeros((nl, nsw))
    lw = LwSpectralProperties(nspec=nlw)
    lw.air_ext = np.full((nl, nlw), 1.0e-5)
    lw.air_ssa = np.zeros((nl, nlw))
    lw.veg_ssa = np.full((nl, nlw), _f32(0.01))
    t_air, t_surf, t_sky = _f32(278.15), _f32(283.15), _f32(268.15)
    em = _f32(0.9)
    lw.clear_air_planck = np.full((nl, nlw), StefanBoltzmann * t_air ** 4 / nlw)
    lw.veg_planck = lw.clear_air_planck.copy()
    lw.veg_air_planck = lw.clear_air_planck.copy()
    lw.ground_emissivity = np.full((ncol, nlw), em)
    lw.roof_emissivity = np.full((nl, nlw), em)
    lw.wall_emissivity = np.full((nl, nlw), em)
    lw.ground_emission = np.full((ncol, nlw), StefanBoltzmann * em * t_surf ** 4 / nlw)
    lw.roof_emission = np.full((nl, nlw), StefanBoltzmann * em * t_surf ** 4 / nlw)
    lw.wall_emission = lw.roof_emission.copy()
    return _finish(cfg, cp, sw, lw, top_flux_dn_sw=np.full((ncol, nsw), 500.0 / nsw),
                   top_flux_dn_direct_sw=np.full((ncol, nsw), 400.0 / nsw),
                   top_flux_dn_lw=np.full((ncol, nlw), StefanBoltzmann * t_sky ** 4 / nlw))


def ragged_all_tiles(ncol=96, seed=2024, nsw=2, nlw=3, simple_urban=True):
    """Edge-case workload (not a BASELINE config): every tile type of radsurf in
    one call - flat (no layers), forest, urban, vegetated urban, simple urban and
    infinite street (one layer each) - with a different number of layers per
    column (1..10), several bands, and a few night-time columns.  Built from the
    C4 columns by keeping only the lowest nlay layers of each.
    simple_urban=False leaves types 4/5 out: next to multi-layer columns the
    reference's simple_urban_* index other columns' results,
    so only that variant can be pinned to the compiled reference."""
    from .types import (ITileFlat, ITileUrban, ITileSimpleUrban, ITileInfiniteStreet)
    from . import abi
    case = c4_mixed(ncol, seed, nsw, nlw)
    cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
    rng = np.random.default_rng(seed + 1)
    L = 10
    types = np.array([ITileFlat, ITileForest, ITileUrban, ITileVegetatedUrban, ITileSimpleUrban,
                      ITileInfiniteStreet, ITileVegetatedUrban, ITileForest], dtype=np.int32)
    if not simple_urban:
        types = types[(types != ITileSimpleUrban) & (types != ITileInfiniteStreet)]
    irep = types[np.arange(ncol) % types.size]
    nlay = rng.integers(1, L + 1, size=ncol).astype(np.int32)
    nlay[irep == ITileFlat] = 0
    nlay[(irep == ITileSimpleUrban) | (irep == ITileInfiniteStreet)] = 1
    keep = (np.arange(L)[None, :] < nlay[:, None]).reshape(-1)       # packed (column, layer) order
    ntot_rect = ncol * L
    for obj, names in ((cp, abi.CANOPY_REAL_FIELDS), (sw, abi.SW_FIELDS), (lw, abi.LW_FIELDS)):
        for n in names:
            a = getattr(obj, n)
            if a is not None and a.shape[0] == ntot_rect:
                setattr(obj, n, np.ascontiguousarray(a[keep]))
    cp.set_layers(nlay)
    cp.i_representation = irep
    cz = rng.uniform(0.05, 1.0, size=ncol)
    cz[5::11] = -0.3            # night
    cz[7::13] = 0.0
    cp.cos_sza = cz
    return _finish(cfg, cp, sw, lw)


# Algorithmic HBM bytes per column: fp64 inputs the solver
# reads + outputs it writes, no flux profiles.
def algorithmic_bytes_per_column(case, family=None):
    """(bytes in, bytes out) per column of SURVEY.md section 8(d); family "sw" / "lw": what one spectral
    region alone reads and writes (the band-independent geometry counted with each)."""
    cfg, cp = case["config"], case["canopy"]
    S = cfg.nsw if (cfg.do_sw and family != "lw") else 0
    W = cfg.nlw if (cfg.do_lw and family != "sw") else 0
    L = float(cp.ntotlay) / max(1, cp.ncol)
    urban = float(np.mean(cp.i_representation != ITileForest))
    # per layer: geometry doubles g, SW spectral s, LW spectral w
    g = 40 * (1 - urban) + 64 * urban
    s = 3 * (1 - urban) + 7 * urban
    w = 6 * (1 - urban) + 10 * urban
    a = 2 if cfg.use_sw_direct_albedo else 1
    bytes_in = 20 + 8 * S * a + 16 * W + L * (g + 8 * (s * S + w * W))
    pS = 4 * (1 - urban) + 10 * urban
    q = 1 * (1 - urban) + 3 * urban
    pW = 3 * (1 - urban) + 7 * urban
    bytes_out = 0.0
    if S:
        bytes_out += 16 * S + 2 * (56 * S + 8 + L * (8 * pS * S + 8 * q))
    if W:
        bytes_out += 16 * W + 2 * (40 * W + L * 8 * pW * W)
    return bytes_in, bytes_out


def head_columns(case, ncol):
    """The first ncol columns of a workload as a case of its own (columns are
    independent, so solving them alone must reproduce the big run)."""
    import copy
    from . import abi
    cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
    nl = int(cp.istartlay[ncol - 1] - 1 + cp.nlay[ncol - 1])
    cp2, sw2, lw2 = copy.copy(cp), copy.copy(sw), copy.copy(lw)
    cp2.set_layers(cp.nlay[:ncol].copy())
    for n in abi.CANOPY_REAL_FIELDS + ("i_representation",):
        a = getattr(cp, n)
        if a is not None:
            setattr(cp2, n, a[:ncol].copy() if a.shape[0] == cp.ncol else a[:nl].copy())
    for obj, obj2, names in ((sw, sw2, abi.SW_FIELDS), (lw, lw2, abi.LW_FIELDS)):
        for n in names:
            a = getattr(obj, n)
            if a is not None:
                setattr(obj2, n, a[:ncol].copy() if a.shape[0] == cp.ncol else a[:nl].copy())
    out = dict(case)
    out.update(config=cfg, canopy=cp2, sw=sw2, lw=lw2)
    return _finish(cfg, cp2, sw2, lw2, **{k: v for k, v in out.items()
                                           if k not in ("config", "canopy", "sw", "lw")})
