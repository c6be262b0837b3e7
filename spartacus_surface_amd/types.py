"""Host-side mirrors of the reference's derived types.

Same names, same members and the same "unallocated == feature off" convention
(``None`` here) as the Fortran types, so host code written against the
reference's API reads the same:

* :class:`Config`                 <- config_type                 (radsurf/radsurf_config.F90:32-113)
* :class:`CanopyProperties`       <- canopy_properties_type      (radsurf/radsurf_canopy_properties.F90:53-110)
* :class:`SwSpectralProperties`   <- sw_spectral_properties_type (radsurf/radsurf_sw_spectral_properties.F90:24-49)
* :class:`LwSpectralProperties`   <- lw_spectral_properties_type (radsurf/radsurf_lw_spectral_properties.F90:24-57)
* :class:`BoundaryCondsOut`       <- boundary_conds_out_type     (radsurf/radsurf_boundary_conds_out.F90:24-39)
* :class:`CanopyFlux`             <- canopy_flux_type            (radsurf/radsurf_canopy_flux.F90:27-91)

Array layout: a Fortran array ``x(nspec, n)`` is held as a C-contiguous numpy
array of shape ``(n, nspec)`` - identical bytes, so the arrays cross the C-ABI
without copies.
"""
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import abi
from .legendre_gauss import LegendreGauss

# Tile representation codes (radsurf_canopy_properties.F90:26-33)
ITileFlat, ITileForest, ITileUrban, ITileVegetatedUrban, ITileSimpleUrban, \
    ITileInfiniteStreet = range(6)
TileRepresentationName = ("Flat", "Forest", "Urban", "VegetatedUrban",
                          "SimpleUrban", "InfiniteStreet")

StefanBoltzmann = 5.67037321e-8  # radtool/radiation_constants.F90:26


@dataclass
class Config:
    """config_type with the reference's defaults (radsurf_config.F90:36-95)."""
    do_sw: bool = True
    do_lw: bool = True
    use_sw_direct_albedo: bool = False
    do_vegetation: bool = True
    do_urban: bool = True
    n_vegetation_region_forest: int = 1
    n_vegetation_region_urban: int = 1
    nsw: int = 1
    nlw: int = 1
    n_stream_sw_forest: int = 4
    n_stream_sw_urban: int = 4
    n_stream_lw_forest: int = 4
    n_stream_lw_urban: int = 4
    use_symmetric_vegetation_scale_forest: bool = True
    use_symmetric_vegetation_scale_urban: bool = True
    vegetation_isolation_factor_forest: float = 0.0
    vegetation_isolation_factor_urban: float = 0.0
    # the reference writes these defaults as default-real literals "1.0e-6"
    # (radsurf_config.F90:84,87), i.e. the float32-rounded value
    min_vegetation_fraction: float = float(np.float32(1.0e-6))
    min_building_fraction: float = float(np.float32(1.0e-6))
    do_save_broadband_flux: bool = True
    do_save_spectral_flux: bool = False
    do_save_flux_profile: bool = False
    iverbose: int = 3
    # computed by consolidate()
    nswinternal: int = 0
    nlwinternal: int = 0
    lg_sw_urban: Optional[LegendreGauss] = None
    lg_sw_forest: Optional[LegendreGauss] = None
    lg_lw_urban: Optional[LegendreGauss] = None
    lg_lw_forest: Optional[LegendreGauss] = None

    def consolidate(self):
        """consolidate_config (radsurf_config.F90:250-270)."""
        self.nswinternal = self.nsw
        self.nlwinternal = self.nlw
        self.lg_sw_forest = LegendreGauss(self.n_stream_sw_forest)
        self.lg_sw_urban = LegendreGauss(self.n_stream_sw_urban)
        self.lg_lw_forest = LegendreGauss(self.n_stream_lw_forest)
        self.lg_lw_urban = LegendreGauss(self.n_stream_lw_urban)
        return self

    def to_c(self):
        if self.lg_sw_forest is None:
            self.consolidate()
        c = abi.ConfigC()
        for name in ("do_sw", "do_lw", "use_sw_direct_albedo", "do_vegetation",
                     "do_urban", "n_vegetation_region_forest",
                     "n_vegetation_region_urban",
                     "use_symmetric_vegetation_scale_forest",
                     "use_symmetric_vegetation_scale_urban", "iverbose"):
            setattr(c, name, int(getattr(self, name)))
        c.nsw = int(self.nswinternal)
        c.nlw = int(self.nlwinternal)
        for name in ("vegetation_isolation_factor_forest",
                     "vegetation_isolation_factor_urban",
                     "min_vegetation_fraction", "min_building_fraction"):
            setattr(c, name, float(getattr(self, name)))
        for name in ("lg_sw_forest", "lg_sw_urban", "lg_lw_forest", "lg_lw_urban"):
            setattr(c, name, getattr(self, name).to_c())
        return c


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


@dataclass
class CanopyProperties:
    ncol: int = 0
    ntotlay: int = 0
    nlay: Optional[np.ndarray] = None
    istartlay: Optional[np.ndarray] = None        # 1-based, as in Fortran
    i_representation: Optional[np.ndarray] = None
    cos_sza: Optional[np.ndarray] = None
    dz: Optional[np.ndarray] = None
    building_fraction: Optional[np.ndarray] = None
    veg_fraction: Optional[np.ndarray] = None
    building_scale: Optional[np.ndarray] = None
    veg_scale: Optional[np.ndarray] = None
    veg_ext: Optional[np.ndarray] = None
    veg_fsd: Optional[np.ndarray] = None
    veg_contact_fraction: Optional[np.ndarray] = None
    ground_temperature: Optional[np.ndarray] = None
    roof_temperature: Optional[np.ndarray] = None
    wall_temperature: Optional[np.ndarray] = None
    clear_air_temperature: Optional[np.ndarray] = None
    veg_temperature: Optional[np.ndarray] = None
    veg_air_temperature: Optional[np.ndarray] = None

    def set_layers(self, nlay):
        """Fill nlay / istartlay / ncol / ntotlay from per-column layer counts
        (the packing of driver/spartacus_surface_read_input.F90:96-106)."""
        self.nlay = _i32(nlay)
        self.ncol = int(self.nlay.size)
        self.ntotlay = int(self.nlay.sum())
        self.istartlay = _i32(1 + np.concatenate(([0], np.cumsum(self.nlay)[:-1])))
        return self

    def normalise(self):
        for n in abi.CANOPY_INT_FIELDS:
            setattr(self, n, _i32(getattr(self, n)))
        for n in abi.CANOPY_REAL_FIELDS:
            setattr(self, n, _f64(getattr(self, n)))
        return self

    def to_c(self):
        self.normalise()
        c = abi.CanopyC()
        c.ncol, c.ntotlay = int(self.ncol), int(self.ntotlay)
        for n in abi.CANOPY_INT_FIELDS:
            setattr(c, n, abi.iptr(getattr(self, n)))
        for n in abi.CANOPY_REAL_FIELDS:
            setattr(c, n, abi.dptr(getattr(self, n)))
        return c


class _Spectral:
    _fields = ()

    def normalise(self):
        for n in self._fields:
            a = _f64(getattr(self, n))
            if a is not None and a.ndim == 1:
                a = a.reshape(-1, 1)
            setattr(self, n, a)
        return self

    def _to_c(self, ctype):
        self.normalise()
        c = ctype()
        c.nspec = int(self.nspec)
        for n in self._fields:
            a = getattr(self, n)
            if a is not None:
                assert a.shape[1] == self.nspec, (n, a.shape, self.nspec)
            setattr(c, n, abi.dptr(a))
        return c


@dataclass
class SwSpectralProperties(_Spectral):
    nspec: int = 1
    ncol: int = 0
    ntotlay: int = 0
    air_ext: Optional[np.ndarray] = None
    air_ssa: Optional[np.ndarray] = None
    veg_ssa: Optional[np.ndarray] = None
    ground_albedo: Optional[np.ndarray] = None
    roof_albedo: Optional[np.ndarray] = None
    wall_albedo: Optional[np.ndarray] = None
    ground_albedo_dir: Optional[np.ndarray] = None
    roof_albedo_dir: Optional[np.ndarray] = None
    wall_specular_frac: Optional[np.ndarray] = None
    _fields = abi.SW_FIELDS

    def to_c(self):
        return self._to_c(abi.SwSpectralC)


@dataclass
class LwSpectralProperties(_Spectral):
    nspec: int = 1
    ncol: int = 0
    ntotlay: int = 0
    air_ext: Optional[np.ndarray] = None
    air_ssa: Optional[np.ndarray] = None
    veg_ssa: Optional[np.ndarray] = None
    clear_air_planck: Optional[np.ndarray] = None
    veg_planck: Optional[np.ndarray] = None
    veg_air_planck: Optional[np.ndarray] = None
    ground_emissivity: Optional[np.ndarray] = None
    roof_emissivity: Optional[np.ndarray] = None
    wall_emissivity: Optional[np.ndarray] = None
    ground_emission: Optional[np.ndarray] = None
    roof_emission: Optional[np.ndarray] = None
    wall_emission: Optional[np.ndarray] = None
    _fields = abi.LW_FIELDS

    def to_c(self):
        return self._to_c(abi.LwSpectralC)

    def calc_monochromatic_emission(self, canopy_props):
        """radsurf_lw_spectral_properties.F90:161-199 (single band)."""
        self.normalise()
        cp = canopy_props
        if cp.ground_temperature is not None and self.ground_emissivity is not None:
            if self.ground_emission is None:
                self.ground_emission = np.zeros((cp.ncol, 1))
            self.ground_emission[:, 0] = (StefanBoltzmann * self.ground_emissivity[:, 0]
                                          * np.asarray(cp.ground_temperature) ** 4)
        if cp.roof_temperature is not None and self.roof_emissivity is not None:
            if self.roof_emission is None:
                self.roof_emission = np.zeros((cp.ntotlay, 1))
            self.roof_emission[:, 0] = (StefanBoltzmann * self.roof_emissivity[:, 0]
                                        * np.asarray(cp.roof_temperature) ** 4)
            if self.wall_emission is None:
                self.wall_emission = np.zeros((cp.ntotlay, 1))
            self.wall_emission[:, 0] = (StefanBoltzmann * self.wall_emissivity[:, 0]
                                        * np.asarray(cp.wall_temperature) ** 4)


@dataclass
class BoundaryCondsOut:
    sw_albedo: Optional[np.ndarray] = None
    sw_albedo_dir: Optional[np.ndarray] = None
    lw_emissivity: Optional[np.ndarray] = None
    lw_emission: Optional[np.ndarray] = None

    def allocate(self, ncol, nsw, nlw):
        """allocate_boundary_conds_out (radsurf_boundary_conds_out.F90:43-58)."""
        self.sw_albedo = np.zeros((ncol, nsw))
        self.sw_albedo_dir = np.zeros((ncol, nsw))
        self.lw_emissivity = np.zeros((ncol, nlw))
        self.lw_emission = np.zeros((ncol, nlw))
        return self

    def to_c(self):
        c = abi.BcOutC()
        for n in abi.BC_FIELDS:
            setattr(c, n, abi.dptr(getattr(self, n)))
        return c


class CanopyFlux:
    """canopy_flux_type.  Members are created by :meth:`allocate` following
    allocate_canopy_flux (radsurf_canopy_flux.F90:96-163): which arrays exist
    depends on config.do_urban / do_vegetation, ``use_direct`` and
    ``do_save_flux_profile`` (default True, as in the reference)."""

    def __init__(self):
        self.nspec = self.ncol = self.ntotlay = 0
        for n, _ in abi.FLUX_FIELDS:
            setattr(self, n, None)

    def allocate(self, config, ncol, ntotlay, nspec, use_direct=True,
                 do_save_flux_profile=True):
        self.nspec, self.ncol, self.ntotlay = nspec, ncol, ntotlay
        for n, kind in abi.FLUX_FIELDS:
            present = True
            if n in abi.FLUX_DIRECT_ONLY and not use_direct:
                present = False
            if n in abi.FLUX_URBAN_ONLY and not config.do_urban:
                present = False
            if n in abi.FLUX_VEG_ONLY and not config.do_vegetation:
                present = False
            if n in abi.FLUX_PROFILE and not do_save_flux_profile:
                present = False
            if not present:
                setattr(self, n, None)
                continue
            shape = {"sc": (ncol, nspec), "sl": (ntotlay, nspec),
                     "c": (ncol,), "l": (ntotlay,)}[kind]
            setattr(self, n, np.zeros(shape))
        return self

    def zero_all(self):
        for n, _ in abi.FLUX_FIELDS:
            a = getattr(self, n)
            if a is not None:
                a[...] = 0.0

    def fields(self):
        return {n: getattr(self, n) for n, _ in abi.FLUX_FIELDS
                if getattr(self, n) is not None}

    def to_c(self):
        c = abi.FluxC()
        c.nspec, c.ncol, c.ntotlay = int(self.nspec), int(self.ncol), int(self.ntotlay)
        for n, _ in abi.FLUX_FIELDS:
            setattr(c, n, abi.dptr(getattr(self, n)))
        return c

    # --- scale / sum stay on the host in round 1 ---
    def scale(self, nlay, factor):
        """scale_canopy_flux (radsurf_canopy_flux.F90:212-282); sunlit
        fractions are not scaled."""
        factor = np.asarray(factor, dtype=np.float64)
        if factor.ndim == 1:
            factor = factor.reshape(-1, 1)
        if factor.shape[1] != self.nspec:
            raise ValueError("spectral resolution mismatch when scaling canopy fluxes")
        indcol = np.repeat(np.arange(self.ncol), np.asarray(nlay))
        for n, kind in abi.FLUX_FIELDS:
            a = getattr(self, n)
            if a is None:
                continue
            if kind == "sc":
                a *= factor
            elif kind == "sl":
                a *= factor[indcol]

    def sum(self, flux1, flux2):
        """sum_canopy_flux (radsurf_canopy_flux.F90:399-460): this = flux1 + flux2
        for every member allocated in this object (a missing member of an
        operand counts as zero)."""
        for n, _ in abi.FLUX_FIELDS:
            a = getattr(self, n)
            if a is None:
                continue
            b1, b2 = getattr(flux1, n), getattr(flux2, n)
            a[...] = (0.0 if b1 is None else b1) + (0.0 if b2 is None else b2)
