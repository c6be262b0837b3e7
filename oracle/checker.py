"""Python bindings of the two CHECKERS - test infrastructure only.

* ``run_ref``    -> oracle/_ref/libspartacus_ref.so : the real reference Fortran
                    compiled from /root/reference (kind "reference")
* ``run_oracle`` -> oracle/liboracle.so : this repo's plain-C restatement
                    (kind "port")

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  The product path
(spartacus_surface_amd.radsurf -> libspartacus_surface_hip.so) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from spartacus_surface_amd import abi
from spartacus_surface_amd.types import BoundaryCondsOut, CanopyFlux

HERE = os.path.dirname(os.path.abspath(__file__))
REF_LIB = os.path.join(HERE, "_ref", "libspartacus_ref.so")
SHIM_LIB = os.path.join(HERE, "_ref", "libspartacus_shim.so")
ORACLE_LIB = os.path.join(HERE, "liboracle.so")
ORACLE_QUAD_LIB = os.path.join(HERE, "liboracle_quad.so")

_libs = {}


def build(ref=True, oracle=True):
    """Compile the checkers (make).  The reference is only rebuilt where its
    sources exist; elsewhere the prebuilt oracle/_ref/ is used."""
    targets = []
    if oracle:
        targets.append("oracle")
    if ref:
        targets.append("ref")
    subprocess.run(["make", "-C", HERE] + targets, check=True,
                   stdout=subprocess.DEVNULL)


def have_ref():
    return os.path.exists(REF_LIB)


def have_oracle():
    return os.path.exists(ORACLE_LIB) and os.path.exists(ORACLE_QUAD_LIB)


def _raise_stack_limit():
    """The reference keeps all per-column work arrays as Fortran automatics
    (~10 MB per thread for the rami5 scenes: 14 bands x 12x12 x 62 layers), so
    it needs what its users get from `ulimit -s unlimited`: raise the main
    thread's soft stack limit and give OpenMP workers a large stack (the latter
    must happen before libomp initialises, i.e. before the first CDLL)."""
    import resource
    os.environ.setdefault("OMP_STACKSIZE", "1G")
    try:
        soft, hard = resource.getrlimit(resource.RLIMIT_STACK)
        if soft != resource.RLIM_INFINITY and (hard == resource.RLIM_INFINITY or hard > soft):
            resource.setrlimit(resource.RLIMIT_STACK, (hard, hard))
    except (ValueError, OSError):
        pass


def _load(path):
    if path not in _libs:
        if path == REF_LIB:
            _raise_stack_limit()
        _libs[path] = C.CDLL(path)
    return _libs[path]


def alloc_outputs(config, cp, want_profiles=True):
    """Allocate bc_out and the four flux objects the way the reference's
    driver does (driver/spartacus_surface_driver.F90:166-184)."""
    bc = BoundaryCondsOut().allocate(cp.ncol, config.nsw, config.nlw)
    out = {"bc_out": bc}
    for name, nspec, direct, on in (
            ("sw_norm_dir", config.nsw, True, config.do_sw),
            ("sw_norm_diff", config.nsw, True, config.do_sw),
            ("lw_internal", config.nlw, False, config.do_lw),
            ("lw_norm", config.nlw, False, config.do_lw)):
        f = CanopyFlux()
        if on:
            f.allocate(config, cp.ncol, cp.ntotlay, nspec, use_direct=direct,
                       do_save_flux_profile=want_profiles)
        out[name] = f
    return out


def _common_args(config, cp, sw, lw, out, istartcol, iendcol):
    keep = [config.to_c(), cp.to_c(), sw.to_c(), lw.to_c(), out["bc_out"].to_c(),
            out["sw_norm_dir"].to_c(), out["sw_norm_diff"].to_c(),
            out["lw_internal"].to_c(), out["lw_norm"].to_c()]
    args = [C.byref(keep[0]), C.byref(keep[1]), C.byref(keep[2]), C.byref(keep[3]),
            C.byref(keep[4]), C.c_int32(istartcol), C.c_int32(iendcol),
            C.byref(keep[5]), C.byref(keep[6]), C.byref(keep[7]), C.byref(keep[8])]
    return keep, args


_NOPOST = [C.c_void_p(None)] * 5     # no scale / sum post-processing


def run_ref(config, cp, sw, lw, want_profiles=True, do_simple_spectrum=False,
            nblocksize=16, nrepeat=1, istartcol=1, iendcol=None, out=None):
    """Run the REAL reference radsurf() (OpenMP block loop of the reference
    driver).  Returns (outputs dict, seconds spent in the radsurf loop)."""
    lib = _load(REF_LIB)
    if iendcol is None:
        iendcol = cp.ncol
    if out is None:
        out = alloc_outputs(config, cp, want_profiles)
    keep, args = _common_args(config, cp, sw, lw, out, istartcol, iendcol)
    elapsed = C.c_double(0.0)
    lib.spartacus_radsurf_ref.restype = C.c_int
    rc = lib.spartacus_radsurf_ref(*args, C.c_int32(nblocksize), C.c_int32(nrepeat),
                                   C.c_int32(int(do_simple_spectrum)),
                                   C.c_int32(int(want_profiles)), C.byref(elapsed), *_NOPOST)
    if rc != 0:
        raise RuntimeError("reference harness returned %d" % rc)
    return out, elapsed.value


def run_ref_scale_sum(config, cp, sw, lw, top_sw, top_sw_dir, top_lw, want_profiles=True):
    """radsurf() followed by the REAL reference's canopy_flux%scale / %sum as its
    driver applies them (spartacus_surface_driver.F90:250-261).  Returns
    (scaled outputs dict, sw_flux, lw_flux)."""
    from spartacus_surface_amd import types
    lib = _load(REF_LIB)
    out = alloc_outputs(config, cp, want_profiles)
    keep, args = _common_args(config, cp, sw, lw, out, 1, cp.ncol)
    sums = []
    for nspec, direct in ((config.nsw, True), (config.nlw, False)):
        f = types.CanopyFlux()
        f.allocate(config, cp.ncol, cp.ntotlay, nspec, use_direct=direct,
                   do_save_flux_profile=want_profiles)
        sums.append(f)
    cs = [f.to_c() for f in sums]
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (top_sw, top_sw_dir, top_lw)]
    elapsed = C.c_double(0.0)
    lib.spartacus_radsurf_ref.restype = C.c_int
    rc = lib.spartacus_radsurf_ref(*args, C.c_int32(max(1, cp.ncol)), C.c_int32(1), C.c_int32(0),
                                   C.c_int32(int(want_profiles)), C.byref(elapsed),
                                   *[C.c_void_p(a.ctypes.data) for a in arrs],
                                   C.cast(C.byref(cs[0]), C.c_void_p), C.cast(C.byref(cs[1]), C.c_void_p))
    if rc != 0:
        raise RuntimeError("reference harness returned %d" % rc)
    return out, sums[0], sums[1]


def have_shim():
    return os.path.exists(SHIM_LIB)


def run_shim(config, cp, sw, lw, want_profiles=True, do_simple_spectrum=False,
             nblocksize=None, istartcol=1, iendcol=None, out=None):
    """The Fortran drop-in boundary end to end: the harness fills the
    REFERENCE's derived types and calls THIS repository's module
    radsurf_interface (ISO_C_BINDING shim), which calls the HIP library.
    Needs a GPU.  nblocksize defaults to all columns in one call."""
    _raise_stack_limit()
    lib = _load(SHIM_LIB)
    if iendcol is None:
        iendcol = cp.ncol
    if nblocksize is None:
        nblocksize = max(1, cp.ncol)
    if out is None:
        out = alloc_outputs(config, cp, want_profiles)
    keep, args = _common_args(config, cp, sw, lw, out, istartcol, iendcol)
    elapsed = C.c_double(0.0)
    lib.spartacus_radsurf_shim.restype = C.c_int
    rc = lib.spartacus_radsurf_shim(*args, C.c_int32(nblocksize), C.c_int32(1),
                                    C.c_int32(int(do_simple_spectrum)),
                                    C.c_int32(int(want_profiles)), C.byref(elapsed), *_NOPOST)
    if rc != 0:
        raise RuntimeError("shim harness returned %d" % rc)
    return out, elapsed.value


def run_oracle(config, cp, sw, lw, want_profiles=True, do_simple_spectrum=False,
               istartcol=1, iendcol=None, out=None, nthreads=1, quad=False):
    """Run the plain-C restatement.  Returns (outputs dict, seconds).
    quad=True runs the same source compiled in IEEE binary128: the exact-
    arithmetic yardstick against which rounding noise is measured."""
    lib = _load(ORACLE_QUAD_LIB if quad else ORACLE_LIB)
    if iendcol is None:
        iendcol = cp.ncol
    if out is None:
        out = alloc_outputs(config, cp, want_profiles)
    keep, args = _common_args(config, cp, sw, lw, out, istartcol, iendcol)
    elapsed = C.c_double(0.0)
    lib.spartacus_radsurf_oracle.restype = C.c_int
    rc = lib.spartacus_radsurf_oracle(*args, C.c_int32(int(do_simple_spectrum)),
                                      C.c_int32(nthreads), C.byref(elapsed))
    if rc != 0:
        raise RuntimeError("oracle returned %d" % rc)
    return out, elapsed.value


def ref_legendre_gauss(nstream):
    lib = _load(REF_LIB)
    lg = abi.LegendreGaussC()
    lib.ref_legendre_gauss(C.c_int32(nstream), C.byref(lg))
    return lg


# ---- kernel-level entries (stacked matrices, Fortran layout (nmat,i,j)) ----

def _f(a):
    return np.asfortranarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(abi.c_dp)


def eigen_decomposition_real(amat, which="ref"):
    """amat: (nmat, n, n).  Returns (eigenvalue (nmat,n), eigenvector (nmat,n,n))."""
    amat = _f(amat)
    nmat, n, _ = amat.shape
    val = np.zeros((nmat, n), order="F")
    vec = np.zeros((nmat, n, n), order="F")
    lib = _load(REF_LIB if which == "ref" else ORACLE_LIB)
    fn = getattr(lib, "ref_eigen_decomposition_real" if which == "ref"
                 else "oracle_eigen_decomposition_real")
    fn(C.c_int32(n), C.c_int32(nmat), _p(amat), _p(val), _p(vec))
    return val, vec


def calc_matrices_sw_eig(dz, mu0, gamma0, gamma1, gamma2, gamma3, which="ref"):
    g0, g1, g2, g3 = _f(gamma0), _f(gamma1), _f(gamma2), _f(gamma3)
    nmat, nd, _ = g1.shape
    nr = g0.shape[1]
    z = lambda *s: np.zeros(s, order="F")  # noqa: E731
    o = dict(reflectance=z(nmat, nd, nd), transmittance=z(nmat, nd, nd),
             s_up=z(nmat, nd, nr), s_dn=z(nmat, nd, nr), trans_dir=z(nmat, nr, nr),
             int_dir=z(nmat, nr, nr), int_diff=z(nmat, nd, nd),
             int_dir_diff=z(nmat, nd, nr))
    lib = _load(REF_LIB if which == "ref" else ORACLE_QUAD_LIB if which == "quad" else ORACLE_LIB)
    fn = getattr(lib, ("ref_" if which == "ref" else "oracle_") + "calc_matrices_sw_eig")
    fn(C.c_int32(nmat), C.c_int32(nd), C.c_int32(nr), C.c_double(dz), C.c_double(mu0),
       _p(g0), _p(g1), _p(g2), _p(g3), *[_p(o[k]) for k in
                                           ("reflectance", "transmittance", "s_up", "s_dn",
                                            "trans_dir", "int_dir", "int_diff", "int_dir_diff")])
    return o


def calc_matrices_lw_eig(dz, gamma1, gamma2, emiss_rate, which="ref"):
    g1, g2, b = _f(gamma1), _f(gamma2), _f(emiss_rate)
    nmat, n, _ = g1.shape
    z = lambda *s: np.zeros(s, order="F")  # noqa: E731
    o = dict(reflectance=z(nmat, n, n), transmittance=z(nmat, n, n), source=z(nmat, n),
             int_flux=z(nmat, n, n), int_flux_source=z(nmat, n))
    lib = _load(REF_LIB if which == "ref" else ORACLE_QUAD_LIB if which == "quad" else ORACLE_LIB)
    fn = getattr(lib, ("ref_" if which == "ref" else "oracle_") + "calc_matrices_lw_eig")
    fn(C.c_int32(nmat), C.c_int32(n), C.c_double(dz), _p(g1), _p(g2), _p(b),
       *[_p(o[k]) for k in ("reflectance", "transmittance", "source", "int_flux",
                            "int_flux_source")])
    return o
