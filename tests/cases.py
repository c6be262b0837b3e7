"""The reference's own test cases, rebuilt as in-memory inputs.

Counterpart of the reference's Makefile-driven harness (test/*/Makefile,
change_namelist.sh, duplicate_profiles.sh, expand_args.sh - SURVEY.md section 4).
Needs /root/reference (test inputs and namelists are DATA read from there); it
is used by tests that run in the build container and by
tests/golden/make_golden.py, which freezes inputs + reference outputs into
fixtures that travel to the GPU box.
"""
import os

from spartacus_surface_amd import inputs

REF_ROOT = os.environ.get("SPARTACUS_REFERENCE", "/root/reference")
TEST = os.path.join(REF_ROOT, "test")


def have_reference_inputs():
    return os.path.isdir(TEST)


def _nam(path):
    with open(path) as f:
        return f.read()


def _build(nam_text, nc_path, replicate=None):
    cfg, drv = inputs.configs_from_namelist(nam_text)
    cp, sw, lw, top_sw, top_dir, top_lw = inputs.read_input(nc_path, cfg, drv,
                                                            replicate_cos_sza=replicate)
    if cfg.do_lw:
        lw.calc_monochromatic_emission(cp)   # driver :190
    return dict(config=cfg, driver=drv, canopy=cp, sw=sw, lw=lw, top_flux_dn_sw=top_sw,
                top_flux_dn_direct_sw=top_dir, top_flux_dn_lw=top_lw,
                do_simple_spectrum=bool(cfg.do_lw))


# ---- test/simple (Makefile:6-17) ----
SIMPLE = ["test_surfaces", "test_consistency", "test_empty_layers",
          "test_nearly_empty_layers", "test_noscat", "test_overhang", "test_closed"]


def simple_case(name, one_stream=False):
    d = os.path.join(TEST, "simple")
    nam = _nam(os.path.join(d, "config_1stream.nam" if one_stream else "config.nam"))
    return _build(nam, os.path.join(d, name + "_in.nc"))


# ---- test/rami4pilps (Makefile:27-53, expand_args.sh:19-53) ----
RAMI4_PARAMS = {"vis": (0.1301, {"med": 0.1217, "snw": 0.9640}),
                "nir": (0.8058, {"med": 0.2142, "snw": 0.5568})}
RAMI4_SCENES = ["%s-%s-%s" % (b, s, c) for b in ("vis", "nir") for s in ("med", "snw")
                for c in ("0.1", "0.3", "0.5")]
RAMI4_METHODS = ["vis-snw-0.3-%d-%d" % (v, s) for v in (1, 2) for s in (1, 2, 4, 8)]


def rami4pilps_case(target):
    """target = BAND-SURFACE-COVER[-NVEGREG[-NSTREAM]] as in the Makefile."""
    d = os.path.join(TEST, "rami4pilps")
    parts = target.split("-")
    band, surf, frac = parts[0], parts[1], parts[2]
    ssa, albs = RAMI4_PARAMS[band]
    kv = dict(vegetation_fraction=frac, ground_sw_albedo=albs[surf], vegetation_sw_ssa=ssa)
    if len(parts) > 3:
        kv["n_vegetation_region_forest"] = parts[3]
    if len(parts) > 4:
        kv["n_stream_sw_forest"] = parts[4]
    nam = inputs.change_namelist(_nam(os.path.join(d, "config.nam")), **kv)
    return _build(nam, os.path.join(d, "rami4pilps_base_profile.nc"),
                  replicate=inputs.COS_SZA_46)


# ---- test/urban (Makefile:12-24) ----
def urban_case(kind):
    """kind = 'single' or '<N>stream' with N in 1,2,4."""
    d = os.path.join(TEST, "urban")
    nam = _nam(os.path.join(d, "config.nam"))
    if kind == "single":
        nam = inputs.change_namelist(nam, cos_solar_zenith_angle=0.5)
        return _build(nam, os.path.join(d, "russell_square.nc"))
    n = int(kind.replace("stream", ""))
    nam = inputs.change_namelist(nam, n_stream_sw_urban=n, n_stream_lw_urban=n)
    return _build(nam, os.path.join(d, "russell_square.nc"), replicate=inputs.COS_SZA_46)


# ---- test/rami5 (Makefile:24-82) ----
RAMI5_SCENES = {
    "HET09_JBS_SUM": ("diffuse", "56", "41"),
    "HET15_JBS_WIN": ("diffuse", "76", "56"),
    "HET07_JPS_SUM": ("diffuse", "56", "41"),
    "HET08_OPS_WIN": ("diffuse", "76", "56"),
    "HET14_WCO_UND": ("diffuse", "42", "60", "67"),
}


def rami5_case(scene, illumination, black_soil=False):
    d = os.path.join(TEST, "rami5")
    nam = _nam(os.path.join(d, "config.nam"))
    kv = {}
    if illumination != "diffuse":
        kv.update(top_flux_dn_direct_sw=1.0, solar_zenith_angle=illumination)
    if black_soil:
        kv.update(ground_sw_albedo=0.0)
    if kv:
        nam = inputs.change_namelist(nam, **kv)
    return _build(nam, os.path.join(d, "scene_nc", "rami5_%s_scene.nc" % scene))


# ---- test/single_layer (Makefile:8-20): the SPARTACUS variant (type 2) and the
# two single-layer Harman variants (types 4/5) ----
def single_layer_case(isurfacetype=2):
    d = os.path.join(TEST, "single_layer")
    nam = inputs.change_namelist(_nam(os.path.join(d, "config.nam")), isurfacetype=isurfacetype)
    return _build(nam, os.path.join(d, "test_single_layer.nc"))


def all_cases(rami5="all"):
    """name -> zero-argument builder for every run of the reference's
    'make test' that exercises the hot path."""
    c = {}
    for n in SIMPLE:
        c["simple/" + n] = (lambda n=n: simple_case(n))
    c["simple/test_surfaces_1stream"] = lambda: simple_case("test_surfaces", one_stream=True)
    for t in RAMI4_SCENES + RAMI4_METHODS:
        c["rami4pilps/" + t] = (lambda t=t: rami4pilps_case(t))
    for k in ("single", "1stream", "2stream", "4stream"):
        c["urban/" + k] = (lambda k=k: urban_case(k))
    for scene, ills in RAMI5_SCENES.items():
        for ill in ills:
            for bs in (False, True):
                if rami5 == "subset" and not (ill == ills[1] or (ill == "diffuse" and not bs)):
                    continue
                tag = "%s-%s%s" % (scene, ill if ill == "diffuse" else ill + "-direct",
                                   "_blacksoil" if bs else "")
                c["rami5/" + tag] = (lambda s=scene, i=ill, b=bs: rami5_case(s, i, b))
    c["single_layer/sp"] = lambda: single_layer_case(2)
    c["single_layer/exp"] = lambda: single_layer_case(4)
    c["single_layer/inf"] = lambda: single_layer_case(5)
    return c


# ---- beyond the reference's own "make test": the same inputs with the stream
# counts its Makefiles never select (3, 5, 6, 7 per hemisphere), so that every
# compiled (regions, streams) instance has a golden vector from the reference ----
def extra_cases():
    c = {}

    def urban_streams(n):
        d = os.path.join(TEST, "urban")
        nam = inputs.change_namelist(_nam(os.path.join(d, "config.nam")),
                                     n_stream_sw_urban=n, n_stream_lw_urban=n)
        return _build(nam, os.path.join(d, "russell_square.nc"), replicate=inputs.COS_SZA_46[::9])

    # (test/simple's forest column is no use here: with 5+ streams the reference's
    # own longwave result for it is NaN / an emissivity of 2.7)
    for n in (3, 5, 6, 7):
        c["extra/urban-%dstream" % n] = (lambda n=n: urban_streams(n))

    # test/single_layer with the three urban representations interleaved in one
    # call and a spread of sun angles (incl. night); every column has one layer,
    # so the reference's layer-for-column indexing in simple_urban_* is harmless
    def single_layer_mixed():
        import numpy as np
        case = single_layer_case(4)
        cp = case["canopy"]
        cp.i_representation = np.array([(4, 5, 2)[j % 3] for j in range(cp.ncol)], dtype=np.int32)
        cz = np.array(inputs.COS_SZA_46, dtype=np.float64)[:cp.ncol].copy()
        cz[7] = -0.2
        cz[8] = 0.0
        cz[11] = 1.0
        cp.cos_sza = cz
        return case

    c["extra/single_layer-mixed"] = single_layer_mixed

    # synthetic: flat, forest, urban and vegetated-urban tiles with 1..10 layers per
    # column in one call, 2 SW + 3 LW bands, day and night (workloads.ragged_all_tiles)
    def ragged():
        from spartacus_surface_amd import workloads
        case = workloads.ragged_all_tiles(simple_urban=False)
        case.setdefault("driver", None)
        case["do_simple_spectrum"] = False
        return case

    c["extra/ragged-tiles"] = ragged
    for t in ("vis-snw-0.3-1-3", "vis-snw-0.3-2-3", "vis-snw-0.3-1-5", "vis-snw-0.3-2-6", "vis-snw-0.3-1-7"):
        c["extra/rami4pilps-" + t] = (lambda t=t: rami4pilps_case(t))
    return c
