"""The reference's untouched offline driver running on the product libraries:

    namelist + netCDF input  ->  driver / read_input (reference, as is)
      -> easy_netcdf (reference, as is) -> `module netcdf` stand-in + classic-format engine (this repo)
      -> radsurf: `module radsurf_interface` ISO_C_BINDING shim (this repo) -> C-ABI -> HIP kernels
      -> canopy_flux%scale/%sum, radsurf_save (reference, as is) -> netCDF output through the stand-in

The executable (oracle/_ref/spartacus_surface_hip, recipe in oracle/Makefile) is
the test SUBJECT.  What it writes is compared with the golden fixtures - outputs
of the reference compiled without any stand-in - after the host mirror of
scale/sum (itself pinned bit-exactly to the reference, tests/test_scale_sum.py)
and the unpacking of radsurf_save.F90:424-600.  The input files are the data
files of the reference's test/simple and test/single_layer (fixtures under
tests/golden/driver_inputs); the namelists below restate those tests' settings.
"""
import os
import subprocess

import numpy as np
import pytest
from scipy.io import netcdf_file

from tests import fixture_io
from tests.conftest import golden_files
from tests.test_scale_sum import _mirror

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "spartacus_surface_hip")
INPUTS = os.path.join(ROOT, "tests", "golden", "driver_inputs")

NAM_SIMPLE = """&radsurf_driver
do_parallel = true,
cos_solar_zenith_angle=0.5,
vegetation_fsd = 0.5,
do_conservation_check = true,
iverbose = 1,
/
&radsurf
iverbose = 1,
do_sw = true,
do_lw = true,
use_sw_direct_albedo = false,
do_vegetation = true,
do_urban = true,
n_vegetation_region_forest = 2,
n_vegetation_region_urban = 2,
nsw = 1,
nlw = 1,
n_stream_sw_forest = 2,
n_stream_sw_urban = 2,
n_stream_lw_forest = 2,
n_stream_lw_urban = 2,
do_save_flux_profile = true,
/
"""

NAM_SINGLE_LAYER = """&radsurf_driver
do_parallel = true,
cos_solar_zenith_angle = 0.5,
nrepeat = 1,
iverbose = 1,
nblocksize = %(nblocksize)d,
isurfacetype = %(isurfacetype)d,
/
&radsurf
iverbose = 1,
do_sw = true,
do_lw = true,
do_vegetation = false,
do_urban = true,
n_vegetation_region_urban = 0,
nsw = 1,
nlw = 1,
n_stream_sw_urban = 8,
n_stream_lw_urban = 8,
do_save_flux_profile = false,
/
"""

# canopy_flux member -> output variable (radsurf_save.F90:168-300, 424-600); %s = "sw" / "lw"
# (the longwave broadband variables are only present when the reference's
# save_canopy_fluxes happens to find its never-assigned do_broadband_lw true,
# radsurf_save.F90:64-70; whatever the file holds is compared)
PER_COLUMN = {"ground_dn": "ground_flux_dn_%s", "ground_net": "ground_flux_net_%s",
              "top_dn": "top_flux_dn_%s", "top_net": "top_flux_net_%s",
              "ground_dn_dir": "ground_flux_dn_direct_%s", "top_dn_dir": "top_flux_dn_direct_%s"}
PER_LAYER = {"roof_in": "roof_flux_in_%s", "roof_net": "roof_flux_net_%s",
             "wall_in": "wall_flux_in_%s", "wall_net": "wall_flux_net_%s",
             "roof_in_dir": "roof_flux_in_direct_%s", "wall_in_dir": "wall_flux_in_direct_%s",
             "clear_air_abs": "clear_air_absorption_%s", "veg_abs": "veg_absorption_%s",
             "veg_air_abs": "veg_air_absorption_%s", "veg_abs_dir": "veg_absorption_direct_%s",
             "flux_dn_layer_top": "flux_dn_layer_top_%s", "flux_up_layer_top": "flux_up_layer_top_%s",
             "flux_dn_layer_base": "flux_dn_layer_base_%s", "flux_up_layer_base": "flux_up_layer_base_%s",
             "flux_dn_dir_layer_top": "flux_dn_direct_layer_top_%s",
             "flux_dn_dir_layer_base": "flux_dn_direct_layer_base_%s"}
SUNLIT = {"roof_sunlit_frac": "roof_sunlit_fraction", "wall_sunlit_frac": "wall_sunlit_fraction",
          "veg_sunlit_frac": "veg_sunlit_fraction"}
FILL = -9999.0


def _unpack(packed, cp, nmaxlay):
    """(ntotlay) -> (ncol, nmaxlay) with the fill value, radsurf_save.F90:629-649"""
    out = np.full((cp.ncol, nmaxlay), FILL)
    for j in range(cp.ncol):
        n, i0 = int(cp.nlay[j]), int(cp.istartlay[j]) - 1
        out[j, :n] = packed[i0:i0 + n]
    return out


def _run_driver(tmp_path, nam_text, input_name, env=None):
    nam = tmp_path / "config.nam"
    nam.write_text(nam_text)
    out = tmp_path / "out.nc"
    r = subprocess.run([EXE, str(nam), os.path.join(INPUTS, input_name), str(out)],
                       capture_output=True, text=True, timeout=600,
                       env=None if env is None else dict(os.environ, **env))
    assert r.returncode == 0 and out.exists(), r.stdout[-2000:] + r.stderr[-2000:]
    return netcdf_file(str(out), "r", mmap=False)


def _compare(nc, golden_name):
    path = [p for p in golden_files() if os.path.basename(p)[:-4] == golden_name][0]
    case, expected = fixture_io.load_case(path)
    cfg, cp = case["config"], case["canopy"]
    if case["do_simple_spectrum"]:
        from spartacus_surface_amd.inputs import calc_simple_spectrum_lw
        calc_simple_spectrum_lw(cfg, cp, case["lw"])
    want = _mirror(case, expected, case["top_flux_dn_sw"], case["top_flux_dn_direct_sw"],
                   case["top_flux_dn_lw"])
    nmaxlay = nc.dimensions["layer"]
    assert nc.dimensions["column"] == cp.ncol and nmaxlay == int(cp.nlay.max())
    assert np.array_equal(nc.variables["nlayer"][:], cp.nlay)
    assert np.array_equal(nc.variables["surface_type"][:], cp.i_representation)
    nchecked = 0

    noise = case["ref_noise"]
    pairs = {"sw": ("sw_norm_dir", "sw_norm_diff"), "lw": ("lw_internal", "lw_norm")}

    def summand_bar(band, member):
        """A sum of two scaled objects can cancel (LW absorption minus emission), so
        the bar is the solver's parity bar of the two summands (max(1e-5, 4 x the
        reference's own noise), relative to each summand's scale), added - the bar
        tests/test_scale_sum.py uses for the device sums."""
        tol = 0.0
        for part in pairs[band]:
            v = want.get("scaled.%s.%s" % (part, member))
            if v is not None and np.size(v):
                bar = max(fixture_io.PARITY_TOL, fixture_io.NOISE_FACTOR * noise.get("%s.%s" % (part, member), 0.0))
                tol += bar * float(np.max(np.abs(v)))
        return tol

    def close(name, got, ref, tol):
        nonlocal nchecked
        assert got.shape == ref.shape, name
        # + float32 rounding of the file (easy_netcdf.F90:1773)
        assert np.all(np.abs(got - ref) <= tol + 1.2e-7 * np.abs(ref)), (name, float(np.abs(got - ref).max()), tol)
        nchecked += 1

    for band in ("sw", "lw"):
        if not getattr(cfg, "do_" + band):
            continue
        pre = band + "_flux."
        for member, pattern in PER_COLUMN.items():
            if pre + member in want and want[pre + member] is not None and (pattern % band) in nc.variables:
                close(pattern % band, nc.variables[pattern % band][:].astype(np.float64),
                      np.asarray(want[pre + member]).reshape(cp.ncol, -1).sum(axis=1), summand_bar(band, member))
        vert = "ground_flux_vertical_diffuse_sw" if band == "sw" else "ground_flux_vertical_lw"
        if vert in nc.variables:
            close(vert, nc.variables[vert][:].astype(np.float64),
                  np.asarray(want[pre + "ground_vertical_diff"]).reshape(cp.ncol, -1).sum(axis=1),
                  summand_bar(band, "ground_vertical_diff"))
        for member, pattern in PER_LAYER.items():
            name = pattern % band
            if want.get(pre + member) is None or name not in nc.variables:
                continue
            packed = np.asarray(want[pre + member]).reshape(cp.ntotlay, -1).sum(axis=1)
            close(name, nc.variables[name][:].astype(np.float64), _unpack(packed, cp, nmaxlay),
                  summand_bar(band, member))
    if cfg.do_sw:
        close("ground_sunlit_fraction", nc.variables["ground_sunlit_fraction"][:].astype(np.float64),
              np.asarray(want["sw_flux.ground_sunlit_frac"]), 1e-5)
        for member, name in SUNLIT.items():
            if want.get("sw_flux." + member) is not None and name in nc.variables:
                close(name, nc.variables[name][:].astype(np.float64),
                      _unpack(np.asarray(want["sw_flux." + member]), cp, nmaxlay), 1e-5)
    return nchecked


pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/spartacus_surface_hip not built")]


def test_reference_driver_on_hip_test_surfaces(tmp_path):
    """test/simple 'make test_surfaces_out.nc': flat, forest, urban and vegetated-urban tiles"""
    nc = _run_driver(tmp_path, NAM_SIMPLE, "test_surfaces_in.nc")
    assert nc.title.startswith(b"Radiative fluxes from the SPARTACUS-Surface")
    assert nc.variables["ground_flux_dn_sw"].units == b"W m-2"
    n = _compare(nc, "simple__test_surfaces")
    assert n >= 25


@pytest.mark.parametrize("golden,isurfacetype,nblocksize", [("single_layer__sp", 2, 100),
                                                            ("single_layer__exp", 4, 100),
                                                            ("single_layer__inf", 5, 8)])
def test_reference_driver_on_hip_single_layer(tmp_path, golden, isurfacetype, nblocksize):
    """test/single_layer 'make test_sp test_exp test_inf' (46 columns); the last one in
    OpenMP blocks of 8 columns, i.e. concurrent radsurf calls on column ranges"""
    nc = _run_driver(tmp_path, NAM_SINGLE_LAYER % dict(isurfacetype=isurfacetype, nblocksize=nblocksize),
                     "test_single_layer.nc")
    n = _compare(nc, golden)
    assert n >= 12


def test_reference_driver_on_hip_sharded_over_devices(tmp_path):
    """The untouched driver reaching several devices from its ONE process: SPARTACUS_DEVICES lists the devices
    of the host entry, which cuts the call into one contiguous column shard per entry (one host thread each).
    The test box has one GPU: two shards on device 0 (23 columns each, the chunk pipeline instead of the
    small-call path) - same output file as the unsharded run."""
    nam = NAM_SINGLE_LAYER % dict(isurfacetype=2, nblocksize=100)
    nc = _run_driver(tmp_path, nam, "test_single_layer.nc",
                     env={"SPARTACUS_DEVICES": "0,0", "SPARTACUS_SMALL_CALL_COLUMNS": "0",
                          "SPARTACUS_SHARD_MIN_COLUMNS": "8"})
    n = _compare(nc, "single_layer__sp")
    assert n >= 12
