"""ctypes view of include/spartacus_surface_hip.h.

The structures here are field-for-field the C structs of the drop-in boundary
(which in turn mirror the reference's Fortran derived types, see the header for
file:line citations).  They are shared by the product binding
(:mod:`spartacus_surface_amd.radsurf`) and by the test-only bindings of the two
checkers under ``oracle/`` so that all three run on byte-identical inputs.
"""
import ctypes as C

import numpy as np

SS_MAX_STREAMS = 16

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)


class LegendreGaussC(C.Structure):
    _fields_ = [
        ("nstream", C.c_int32),
        ("pad_", C.c_int32),
        ("mu", C.c_double * SS_MAX_STREAMS),
        ("sin_ang", C.c_double * SS_MAX_STREAMS),
        ("tan_ang", C.c_double * SS_MAX_STREAMS),
        ("weight", C.c_double * SS_MAX_STREAMS),
        ("hweight", C.c_double * SS_MAX_STREAMS),
        ("vweight", C.c_double * SS_MAX_STREAMS),
        ("vadjustment", C.c_double),
        ("vadjustment2", C.c_double),
    ]


class ConfigC(C.Structure):
    _fields_ = [
        ("do_sw", C.c_int32),
        ("do_lw", C.c_int32),
        ("use_sw_direct_albedo", C.c_int32),
        ("do_vegetation", C.c_int32),
        ("do_urban", C.c_int32),
        ("n_vegetation_region_forest", C.c_int32),
        ("n_vegetation_region_urban", C.c_int32),
        ("nsw", C.c_int32),
        ("nlw", C.c_int32),
        ("use_symmetric_vegetation_scale_forest", C.c_int32),
        ("use_symmetric_vegetation_scale_urban", C.c_int32),
        ("iverbose", C.c_int32),
        ("vegetation_isolation_factor_forest", C.c_double),
        ("vegetation_isolation_factor_urban", C.c_double),
        ("min_vegetation_fraction", C.c_double),
        ("min_building_fraction", C.c_double),
        ("lg_sw_forest", LegendreGaussC),
        ("lg_sw_urban", LegendreGaussC),
        ("lg_lw_forest", LegendreGaussC),
        ("lg_lw_urban", LegendreGaussC),
    ]


CANOPY_INT_FIELDS = ("nlay", "istartlay", "i_representation")
CANOPY_REAL_FIELDS = (
    "cos_sza", "dz", "building_fraction", "veg_fraction", "building_scale",
    "veg_scale", "veg_ext", "veg_fsd", "veg_contact_fraction",
    "ground_temperature", "roof_temperature", "wall_temperature",
    "clear_air_temperature", "veg_temperature", "veg_air_temperature",
)


class CanopyC(C.Structure):
    _fields_ = ([("ncol", C.c_int32), ("ntotlay", C.c_int32)]
                + [(n, c_ip) for n in CANOPY_INT_FIELDS]
                + [(n, c_dp) for n in CANOPY_REAL_FIELDS])


SW_FIELDS = ("air_ext", "air_ssa", "veg_ssa", "ground_albedo", "roof_albedo",
             "wall_albedo", "ground_albedo_dir", "roof_albedo_dir",
             "wall_specular_frac")
# which SW / LW members are per column (nspec,ncol); the others are (nspec,ntotlay)
SW_PER_COLUMN = ("ground_albedo", "ground_albedo_dir")


class SwSpectralC(C.Structure):
    _fields_ = ([("nspec", C.c_int32), ("pad_", C.c_int32)]
                + [(n, c_dp) for n in SW_FIELDS])


LW_FIELDS = ("air_ext", "air_ssa", "veg_ssa", "clear_air_planck", "veg_planck",
             "veg_air_planck", "ground_emissivity", "roof_emissivity",
             "wall_emissivity", "ground_emission", "roof_emission",
             "wall_emission")
LW_PER_COLUMN = ("ground_emissivity", "ground_emission")


class LwSpectralC(C.Structure):
    _fields_ = ([("nspec", C.c_int32), ("pad_", C.c_int32)]
                + [(n, c_dp) for n in LW_FIELDS])


BC_FIELDS = ("sw_albedo", "sw_albedo_dir", "lw_emissivity", "lw_emission")


class BcOutC(C.Structure):
    _fields_ = [(n, c_dp) for n in BC_FIELDS]


# canopy_flux_type members in declaration order, with their shape class:
# "sc" = (nspec,ncol), "sl" = (nspec,ntotlay), "c" = (ncol), "l" = (ntotlay)
FLUX_FIELDS = (
    ("ground_dn", "sc"), ("ground_dn_dir", "sc"), ("ground_net", "sc"),
    ("top_dn", "sc"), ("top_dn_dir", "sc"), ("top_net", "sc"),
    ("roof_in", "sl"), ("roof_in_dir", "sl"), ("roof_net", "sl"),
    ("wall_in", "sl"), ("wall_in_dir", "sl"), ("wall_net", "sl"),
    ("roof_sunlit_frac", "l"), ("wall_sunlit_frac", "l"),
    ("ground_sunlit_frac", "c"), ("ground_vertical_diff", "sc"),
    ("clear_air_abs", "sl"), ("veg_abs", "sl"), ("veg_air_abs", "sl"),
    ("veg_abs_dir", "sl"), ("veg_sunlit_frac", "l"),
    ("flux_dn_layer_top", "sl"), ("flux_dn_dir_layer_top", "sl"),
    ("flux_up_layer_top", "sl"), ("flux_dn_layer_base", "sl"),
    ("flux_dn_dir_layer_base", "sl"), ("flux_up_layer_base", "sl"),
)
FLUX_DIRECT_ONLY = ("ground_dn_dir", "top_dn_dir", "ground_sunlit_frac",
                    "roof_in_dir", "wall_in_dir", "roof_sunlit_frac",
                    "wall_sunlit_frac", "veg_abs_dir", "veg_sunlit_frac",
                    "flux_dn_dir_layer_top", "flux_dn_dir_layer_base")
FLUX_URBAN_ONLY = ("roof_in", "roof_in_dir", "roof_net", "wall_in",
                   "wall_in_dir", "wall_net", "roof_sunlit_frac",
                   "wall_sunlit_frac")
FLUX_VEG_ONLY = ("veg_abs", "veg_air_abs", "veg_abs_dir", "veg_sunlit_frac")
FLUX_PROFILE = ("flux_dn_layer_top", "flux_dn_dir_layer_top",
                "flux_up_layer_top", "flux_dn_layer_base",
                "flux_dn_dir_layer_base", "flux_up_layer_base")


class FluxC(C.Structure):
    _fields_ = ([("nspec", C.c_int32), ("ncol", C.c_int32),
                 ("ntotlay", C.c_int32), ("pad_", C.c_int32)]
                + [(n, c_dp) for n, _ in FLUX_FIELDS])


def dptr(a):
    """double* of a C-contiguous float64 array, or NULL for None."""
    if a is None:
        return c_dp()
    assert isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous, \
        "arrays crossing the C-ABI must be contiguous float64"
    return a.ctypes.data_as(c_dp)


def iptr(a):
    if a is None:
        return c_ip()
    assert isinstance(a, np.ndarray) and a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(c_ip)
