"""Host-side input handling: namelists, the reference's netCDF test inputs and
synthetic many-column workloads.

This is this repository's counterpart of the reference's offline driver input
stage (driver/spartacus_surface_read_input.F90:20-384 and
driver/spartacus_surface_config.F90:76-165) and of its test harness scripts
(test/*/change_namelist.sh, duplicate_profiles.sh, expand_args.sh).  It only
*reads* classic CDF-1 files (scipy.io.netcdf_file); no netCDF library is
needed.  Nothing here touches the GPU.
"""
import math
import re
from dataclasses import dataclass, fields
from typing import Optional

import numpy as np

from .legendre_gauss import Pi
from .types import (CanopyProperties, Config, LwSpectralProperties,
                    StefanBoltzmann, SwSpectralProperties)

# The 46 cosines of test/rami4pilps/duplicate_profiles.sh:19 (also used by
# test/urban/duplicate_profiles.sh); stored as float32 in the netCDF file.
COS_SZA_46 = np.array([
    1.0, 0.999391, 0.997564, 0.994522, 0.990268, 0.984808, 0.978148, 0.970296,
    0.961262, 0.951057, 0.939693, 0.927184, 0.913545, 0.898794, 0.882948,
    0.866025, 0.848048, 0.829038, 0.809017, 0.788011, 0.766044, 0.743145,
    0.71934, 0.694658, 0.669131, 0.642788, 0.615661, 0.587785, 0.559193,
    0.529919, 0.5, 0.469472, 0.438371, 0.406737, 0.374607, 0.34202, 0.309017,
    0.275637, 0.241922, 0.207912, 0.173648, 0.139173, 0.104528, 0.0697565,
    0.0348995, 0.01], dtype=np.float32).astype(np.float64)


@dataclass
class DriverConfig:
    """driver_config_type (driver/spartacus_surface_config.F90:20-60)."""
    do_parallel: bool = True
    nblocksize: int = 16
    nrepeat: int = 1
    istartcol: int = 1
    iendcol: int = 0
    iverbose: int = 3
    do_conservation_check: bool = False
    cos_solar_zenith_angle: float = -1.0   # cos_sza_override
    solar_zenith_angle: float = -100.0
    ground_sw_albedo: float = -1.0
    roof_sw_albedo: float = -1.0
    wall_sw_albedo: float = -1.0
    ground_lw_emissivity: float = -1.0
    roof_lw_emissivity: float = -1.0
    wall_lw_emissivity: float = -1.0
    vegetation_fraction: float = -1.0
    vegetation_extinction: float = -1.0
    vegetation_extinction_scaling: float = -1.0
    vegetation_fsd: float = -1.0
    vegetation_sw_ssa: float = -1.0
    vegetation_lw_ssa: float = -1.0
    top_flux_dn_sw: float = -1.0
    top_flux_dn_direct_sw: float = -1.0
    top_flux_dn_lw: float = -1.0
    isurfacetype: int = -1

    @property
    def cos_sza_override(self):
        # spartacus_surface_config.F90:155-161
        if self.cos_solar_zenith_angle == -1.0 and 0.0 <= self.solar_zenith_angle <= 180.0:
            return math.cos(self.solar_zenith_angle * Pi / 180.0)
        return self.cos_solar_zenith_angle


def parse_namelist(text):
    """Very small Fortran-namelist reader: returns {group: {key: value}}."""
    groups, cur = {}, None
    for raw in text.splitlines():
        line = raw.split("!")[0].strip()
        if not line:
            continue
        if line.startswith("&"):
            cur = groups.setdefault(line[1:].strip().lower(), {})
            continue
        if line.startswith("/"):
            cur = None
            continue
        if cur is None:
            continue
        for item in line.rstrip(",").split(","):
            if "=" not in item:
                continue
            k, v = item.split("=", 1)
            cur[k.strip().lower()] = _nml_value(v.strip())
    return groups


def _nml_value(v):
    lv = v.lower().strip(".")
    if lv in ("true", "t"):
        return True
    if lv in ("false", "f"):
        return False
    try:
        return int(v)
    except ValueError:
        pass
    try:
        return float(v.lower().replace("d", "e"))
    except ValueError:
        return v.strip("'\"")


def change_namelist(text, **kv):
    """test/*/change_namelist.sh: replace the (possibly commented-out) line that
    starts with ``key`` by ``key=value,``."""
    for key, value in kv.items():
        if isinstance(value, bool):
            value = "true" if value else "false"
        pat = re.compile(r"^[\s!]*" + re.escape(key) + r".*$", re.M)
        if not pat.search(text):
            raise KeyError("%s not found in namelist" % key)
        text = pat.sub("%s=%s," % (key, value), text)
    return text


def configs_from_namelist(text):
    """(Config, DriverConfig) from the text of a namelist file holding the
    &radsurf and &radsurf_driver groups (radsurf_config.F90:153-161,
    spartacus_surface_config.F90:100-106)."""
    g = parse_namelist(text)
    cfg, drv = Config(), DriverConfig()
    for k, v in g.get("radsurf", {}).items():
        if hasattr(cfg, k):
            setattr(cfg, k, type(getattr(cfg, k))(v))
    names = {f.name for f in fields(DriverConfig)}
    for k, v in g.get("radsurf_driver", {}).items():
        if k in names:
            setattr(drv, k, type(getattr(drv, k))(v))
    cfg.consolidate()
    return cfg, drv


class _NcFile:
    """The subset of easy_netcdf the reader needs, on scipy.io.netcdf_file.
    File dims (column, layer[, wavelength]) in C order are the Fortran
    (wavelength, layer, column)."""

    def __init__(self, source):
        from scipy.io import netcdf_file
        self.nc = netcdf_file(source, "r", mmap=False)

    def exists(self, name):
        return name in self.nc.variables

    def get(self, name):
        return np.array(self.nc.variables[name].data, dtype=np.float64)

    def rank(self, name):
        return len(self.nc.variables[name].dimensions)


def _pack(unpacked, nlay):
    """read_packed_1d/2d (spartacus_surface_read_input.F90:417-496):
    (column, layer[, band]) -> (ntotlay[, band]) keeping nlay[j] layers of
    column j."""
    parts = [unpacked[j, :n] for j, n in enumerate(nlay)]
    out = np.concatenate(parts, axis=0) if parts else unpacked[:0, 0]
    return np.ascontiguousarray(out, dtype=np.float64)


def _packed_2d(f, name, nlay):
    a = f.get(name)
    p = _pack(a, nlay)
    return p.reshape(-1, 1) if a.ndim == 2 else p


def _read_2d(f, name):
    a = f.get(name)
    return a.reshape(-1, 1) if a.ndim == 1 else np.ascontiguousarray(a)


def read_input(source, config, driver_config, replicate_cos_sza=None):
    """read_input (driver/spartacus_surface_read_input.F90:20-384).

    ``replicate_cos_sza``: optional 1-D array; emulates
    duplicate_profiles.sh (record-concatenate the file's columns once per
    value and overwrite cos_solar_zenith_angle).

    Returns (canopy_props, sw_spectral_props, lw_spectral_props,
    top_flux_dn_sw, top_flux_dn_direct_sw, top_flux_dn_lw).
    """
    f = _NcFile(source)
    cfg, drv = config, driver_config
    rep = 1 if replicate_cos_sza is None else len(replicate_cos_sza)

    def col(a):      # replicate along the column dimension
        return np.concatenate([a] * rep, axis=0) if rep > 1 else a

    class R:         # file view with replicated columns
        @staticmethod
        def get(name):
            return col(f.get(name))
        exists = staticmethod(f.exists)

    cp = CanopyProperties()
    nlay = np.array(R.get("nlayer"), dtype=np.int32)
    cp.set_layers(nlay)
    ncol, ntotlay = cp.ncol, cp.ntotlay

    if cfg.do_sw:
        if drv.cos_sza_override >= 0.0:
            cp.cos_sza = np.full(ncol, drv.cos_sza_override)
        elif replicate_cos_sza is not None:
            base = f.get("cos_solar_zenith_angle").size
            cp.cos_sza = np.repeat(np.asarray(replicate_cos_sza, dtype=np.float64), base)
        else:
            cp.cos_sza = R.get("cos_solar_zenith_angle")

    height = R.get("height")
    cp.dz = _pack(height[:, 1:] - height[:, :-1], nlay)

    if drv.isurfacetype >= 0:
        cp.i_representation = np.full(ncol, drv.isurfacetype, dtype=np.int32)
    else:
        cp.i_representation = np.array(R.get("surface_type"), dtype=np.int32)

    if cfg.do_urban:
        cp.building_fraction = _pack(R.get("building_fraction"), nlay)
        cp.building_scale = _pack(R.get("building_scale"), nlay)
    if cfg.do_vegetation:
        if drv.vegetation_fraction >= 0.0:
            cp.veg_fraction = np.full(ntotlay, drv.vegetation_fraction)
        else:
            cp.veg_fraction = _pack(R.get("veg_fraction"), nlay)
        cp.veg_ext = _pack(R.get("veg_extinction"), nlay)
        if drv.vegetation_extinction >= 0.0:
            cp.veg_ext = np.full(ntotlay, drv.vegetation_extinction)
        elif drv.vegetation_extinction_scaling >= 0.0:
            cp.veg_ext = cp.veg_ext * drv.vegetation_extinction_scaling
        cp.veg_scale = _pack(R.get("veg_scale"), nlay)
        if drv.vegetation_fsd >= 0.0:
            cp.veg_fsd = np.full(ntotlay, drv.vegetation_fsd)
        else:
            cp.veg_fsd = _pack(R.get("veg_fsd"), nlay)
        if cfg.do_urban:
            if R.exists("veg_contact_fraction"):
                cp.veg_contact_fraction = _pack(R.get("veg_contact_fraction"), nlay)
            else:
                cp.veg_contact_fraction = np.minimum(
                    1.0, cp.veg_fraction / np.maximum(cfg.min_vegetation_fraction,
                                                      1.0 - cp.building_fraction))

    sw = SwSpectralProperties(nspec=cfg.nsw, ncol=ncol, ntotlay=ntotlay)
    lw = LwSpectralProperties(nspec=cfg.nlw, ncol=ncol, ntotlay=ntotlay)
    top_sw = top_dir = top_lw = None

    class RP:        # packed readers on the replicated view
        @staticmethod
        def p2(name):
            a = R.get(name)
            p = _pack(a, nlay)
            return p.reshape(-1, 1) if a.ndim == 2 else p

        @staticmethod
        def r2(name):
            a = R.get(name)
            return a.reshape(-1, 1) if a.ndim == 1 else np.ascontiguousarray(a)

    if cfg.do_lw:
        cp.ground_temperature = R.get("ground_temperature")
        if cfg.do_urban:
            cp.roof_temperature = _pack(R.get("roof_temperature"), nlay)
            cp.wall_temperature = _pack(R.get("wall_temperature"), nlay)
        lw.ground_emissivity = RP.r2("ground_lw_emissivity")
        if drv.ground_lw_emissivity >= 0.0:
            lw.ground_emissivity[...] = drv.ground_lw_emissivity
        if cfg.do_urban:
            lw.roof_emissivity = RP.p2("roof_lw_emissivity")
            if drv.roof_lw_emissivity >= 0.0:
                lw.roof_emissivity[...] = drv.roof_lw_emissivity
            lw.wall_emissivity = RP.p2("wall_lw_emissivity")
            if drv.wall_lw_emissivity >= 0.0:
                lw.wall_emissivity[...] = drv.wall_lw_emissivity
        if cfg.do_vegetation:
            lw.veg_ssa = RP.p2("veg_lw_ssa")
            if drv.vegetation_lw_ssa >= 0.0:
                lw.veg_ssa[...] = drv.vegetation_lw_ssa
        if cfg.do_vegetation or cfg.do_urban:
            if R.exists("clear_air_temperature"):
                cp.clear_air_temperature = _pack(R.get("clear_air_temperature"), nlay)
                if cfg.do_vegetation:
                    cp.veg_air_temperature = _pack(R.get("veg_air_temperature"), nlay)
            else:
                cp.clear_air_temperature = _pack(R.get("air_temperature"), nlay)
                if cfg.do_vegetation:
                    cp.veg_air_temperature = cp.clear_air_temperature.copy()
            if cfg.do_vegetation:
                if R.exists("veg_temperature"):
                    cp.veg_temperature = _pack(R.get("veg_temperature"), nlay)
                else:
                    cp.veg_temperature = cp.clear_air_temperature.copy()
            lw.air_ext = np.full((ntotlay, cfg.nlw), 1.0e-5)
            lw.air_ssa = np.zeros((ntotlay, cfg.nlw))
            lw.clear_air_planck = np.zeros((ntotlay, cfg.nlw))
            if cfg.do_vegetation:
                lw.veg_planck = np.zeros((ntotlay, cfg.nlw))
                lw.veg_air_planck = np.zeros((ntotlay, cfg.nlw))
        # 'top_flux_lw_sw' typo of the reference (read_input.F90:273): the
        # sky_temperature branch is always taken for the shipped inputs
        if R.exists("top_flux_lw_sw"):
            top_lw = RP.r2("top_flux_dn_lw")
        else:
            top_lw = StefanBoltzmann * RP.r2("sky_temperature") ** 4
        # emission arrays the LW solver needs allocated (filled by
        # calc_monochromatic_emission / calc_simple_spectrum_lw)
        lw.ground_emission = np.zeros((ncol, cfg.nlw))
        if cfg.do_urban:
            lw.roof_emission = np.zeros((ntotlay, cfg.nlw))
            lw.wall_emission = np.zeros((ntotlay, cfg.nlw))

    if cfg.do_sw:
        sw.ground_albedo = RP.r2("ground_sw_albedo")
        if drv.ground_sw_albedo >= 0.0:
            sw.ground_albedo[...] = drv.ground_sw_albedo
        if R.exists("ground_sw_albedo_direct"):
            sw.ground_albedo_dir = RP.r2("ground_sw_albedo_direct")
        if cfg.do_urban:
            sw.roof_albedo = RP.p2("roof_sw_albedo")
            if drv.roof_sw_albedo >= 0.0:
                sw.roof_albedo[...] = drv.roof_sw_albedo
            if R.exists("roof_sw_albedo_direct"):
                sw.roof_albedo_dir = RP.p2("roof_sw_albedo_direct")
            else:
                sw.roof_albedo_dir = sw.roof_albedo.copy()
            sw.wall_albedo = RP.p2("wall_sw_albedo")
            if drv.wall_sw_albedo >= 0.0:
                sw.wall_albedo[...] = drv.wall_sw_albedo
            if R.exists("wall_sw_specular_fraction"):
                sw.wall_specular_frac = RP.p2("wall_sw_specular_fraction")
            else:
                sw.wall_specular_frac = np.zeros_like(sw.roof_albedo)
        if cfg.do_vegetation:
            sw.veg_ssa = RP.p2("veg_sw_ssa")
            if drv.vegetation_sw_ssa >= 0.0:
                sw.veg_ssa[...] = drv.vegetation_sw_ssa
        if cfg.do_vegetation or cfg.do_urban:
            sw.air_ext = np.full((ntotlay, cfg.nsw), 1.0e-5)
            sw.air_ssa = np.full((ntotlay, cfg.nsw), 0.999)
        if drv.top_flux_dn_sw >= 0.0:
            top_sw = np.full((ncol, cfg.nsw), drv.top_flux_dn_sw)
        else:
            top_sw = RP.r2("top_flux_dn_sw")
        if drv.top_flux_dn_direct_sw >= 0.0:
            top_dir = np.full((ncol, cfg.nsw), drv.top_flux_dn_direct_sw)
        else:
            top_dir = RP.r2("top_flux_dn_direct_sw")

    cp.normalise()
    sw.normalise()
    lw.normalise()
    return cp, sw, lw, top_sw, top_dir, top_lw


def calc_simple_spectrum_lw(config, canopy_props, lw, istartcol=1, iendcol=None):
    """Host mirror of calc_simple_spectrum_lw
    (radsurf/radsurf_simple_spectrum.F90:41-66); the device version is
    ``Plan.calc_simple_spectrum_lw``."""
    cp = canopy_props
    if iendcol is None:
        iendcol = cp.ncol
    if lw.nspec > 1:
        raise RuntimeError("Simple longwave spectrum only possible with one "
                           "input spectral interval")
    c = slice(istartcol - 1, iendcol)
    l1 = cp.istartlay[istartcol - 1] - 1
    l2 = cp.istartlay[iendcol - 1] - 1 + cp.nlay[iendcol - 1]
    lay = slice(l1, l2)
    lw.ground_emission[c, 0] = (StefanBoltzmann * lw.ground_emissivity[c, 0]
                                * cp.ground_temperature[c] ** 4)
    if l2 > l1:
        # (the reference dereferences roof/wall arrays unconditionally here and
        # therefore needs do_urban=true whenever do_lw=true)
        lw.roof_emission[lay, 0] = (StefanBoltzmann * lw.roof_emissivity[lay, 0]
                                    * cp.roof_temperature[lay] ** 4)
        lw.wall_emission[lay, 0] = (StefanBoltzmann * lw.wall_emissivity[lay, 0]
                                    * cp.wall_temperature[lay] ** 4)
        lw.clear_air_planck[lay, 0] = StefanBoltzmann * cp.clear_air_temperature[lay] ** 4
        if config.do_vegetation:
            lw.veg_planck[lay, 0] = StefanBoltzmann * cp.veg_temperature[lay] ** 4
            lw.veg_air_planck[lay, 0] = StefanBoltzmann * cp.veg_air_temperature[lay] ** 4
