"""Host-side mirror of the reference's solver entry point, bound to the HIP library.

``radsurf(config, canopy_props, sw_spectral_props, lw_spectral_props, bc_out,
istartcol, iendcol, sw_norm_dir, sw_norm_diff, lw_internal, lw_norm)`` has the
argument list and meaning of ``radsurf_interface::radsurf``
(radsurf/radsurf_interface.F90:20-25): the caller allocates everything, slices
``(:, istartcol:iendcol)`` and the matching packed layer ranges of the outputs
are written.  It goes through the C-ABI ``spartacus_radsurf_hip`` of
``include/spartacus_surface_hip.h`` - host buffers in, host buffers out.

:class:`Plan` is the device-resident form (inputs and outputs stay in HBM
between calls) used by ``bench.py`` and by hosts that keep their fields on the
GPU.

There is NO CPU fallback: if ``libspartacus_surface_hip.so`` has not been built
or no MI355X is visible, these raise.
"""
import ctypes as C
import os

import numpy as np

from . import abi

# SPARTACUS_HIP_LIB selects another build of the SAME HIP library (kernel tuning
# experiments under lib/variants/); it is never anything but a HIP build.
_LIB_PATH = os.environ.get("SPARTACUS_HIP_LIB") or os.path.join(
    os.path.dirname(os.path.abspath(__file__)), "lib", "libspartacus_surface_hip.so")
_lib = None

STATUS = {
    1: "SS_ERR_NO_DEVICE", 2: "SS_ERR_UNSUPPORTED", 3: "SS_ERR_SIMPLE_URBAN_LAYERS",
    4: "SS_ERR_VEG_DISABLED", 5: "SS_ERR_BAD_ARGUMENT", 6: "SS_ERR_HIP",
    7: "SS_ERR_TILE_NOT_ON_DEVICE",
}

EXPORTS = (
    "spartacus_radsurf_hip", "ss_plan_create", "ss_plan_upload",
    "ss_plan_calc_simple_spectrum_lw", "ss_plan_solve", "ss_plan_synchronize",
    "ss_plan_download", "ss_plan_last_kernel_ms", "ss_plan_eigen_failures", "ss_plan_set_pipes", "ss_plan_scale_and_sum", "ss_plan_download_sums",
    "ss_plan_device_ptr", "ss_plan_device", "ss_plan_destroy", "ss_host_entry_plan_creations",
    "ss_host_pin", "ss_host_unpin", "ss_host_unpin_all", "ss_host_pinned_bytes",
    "ss_device_count", "ss_set_device", "ss_set_devices", "ss_get_devices",
    "ss_plan_create_multi", "ss_multi_plan_count", "ss_multi_plan_part", "ss_multi_plan_solve",
    "ss_multi_plan_synchronize", "ss_multi_plan_download", "ss_multi_plan_destroy",
    "ss_last_error", "ss_version", "ss_build_id",
    "ss_calc_matrices_sw_eig", "ss_calc_matrices_lw_eig", "ss_eigen_decomposition_real",
    "ss_units_last_error",
)


class SpartacusHipError(RuntimeError):
    pass


def library_path():
    return _LIB_PATH


def load_library():
    """dlopen the in-tree HIP library (built by ``__graft_entry__.build()`` /
    ``make -C spartacus_surface_amd/csrc``).  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise SpartacusHipError(
                "HIP extension not built: %s is missing (run `python -c 'import "
                "__graft_entry__ as g; g.build()'`); there is no CPU fallback" % _LIB_PATH)
        _lib = C.CDLL(_LIB_PATH)
        _lib.ss_last_error.restype = C.c_char_p
        _lib.ss_version.restype = C.c_char_p
        _lib.ss_build_id.restype = C.c_char_p
        _lib.ss_units_last_error.restype = C.c_char_p
        for name in EXPORTS:
            if name not in ("ss_last_error", "ss_version", "ss_build_id", "ss_units_last_error"):
                getattr(_lib, name).restype = C.c_int
    return _lib


def _check(rc):
    if rc != 0:
        msg = load_library().ss_last_error()
        raise SpartacusHipError("%s (%d): %s" % (STATUS.get(rc, "error"), rc,
                                                 msg.decode() if msg else ""))


def build_id():
    """hash of the sources the loaded library was built from (ss_build_id)"""
    return load_library().ss_build_id().decode()


def device_count():
    n = C.c_int32(0)
    load_library().ss_device_count(C.byref(n))
    return n.value


def set_device(index):
    _check(load_library().ss_set_device(C.c_int32(index)))


def set_devices(devices):
    """Devices of the host entry (ss_set_devices): large calls of radsurf() are cut into one contiguous column
    shard per listed device, one host thread each.  [] = the calling thread's current device; a device may be
    listed more than once."""
    arr = (C.c_int32 * max(1, len(devices)))(*devices)
    _check(load_library().ss_set_devices(arr, C.c_int32(len(devices))))


def get_devices():
    n = C.c_int32(0)
    arr = (C.c_int32 * 64)()
    _check(load_library().ss_get_devices(arr, C.c_int32(64), C.byref(n)))
    return [arr[k] for k in range(min(64, n.value))]


def host_pin(array):
    """Page-lock one numpy array for the host entry (ss_host_pin).  CONTRACT: keep the array alive
    until host_unpin(array) / host_unpin_all() - freeing it first makes a later copy fault the GPU."""
    _check(load_library().ss_host_pin(C.c_void_p(array.ctypes.data), C.c_int64(array.nbytes)))


def host_unpin(array):
    _check(load_library().ss_host_unpin(C.c_void_p(array.ctypes.data)))


def host_arrays(config, canopy_props, sw_spectral_props, lw_spectral_props, bc_out, *fluxes):
    """Every numpy array of a radsurf argument list that the host entry copies (for host_pin)."""
    import numpy as np
    out = []
    for obj in (canopy_props, sw_spectral_props if config.do_sw else None,
                lw_spectral_props if config.do_lw else None, bc_out) + tuple(fluxes):
        if obj is None:
            continue
        for v in vars(obj).values():
            if isinstance(v, np.ndarray) and v.dtype == np.float64 and v.size:
                out.append(v)
    return out


def host_unpin_all():
    """Unlock every host array page-locked through ss_host_pin."""
    _check(load_library().ss_host_unpin_all())


def host_pinned_bytes():
    n = C.c_int64(0)
    _check(load_library().ss_host_pinned_bytes(C.byref(n)))
    return n.value


def host_entry_plan_creations():
    """Device plans the host entry has built so far in this process."""
    n = C.c_int64(0)
    _check(load_library().ss_host_entry_plan_creations(C.byref(n)))
    return n.value


def _flux_ptr(f):
    return C.byref(f.to_c()) if f is not None else None


def radsurf(config, canopy_props, sw_spectral_props, lw_spectral_props, bc_out,
            istartcol=None, iendcol=None, sw_norm_dir=None, sw_norm_diff=None,
            lw_internal=None, lw_norm=None):
    """radsurf_interface::radsurf on the MI355X (host buffers in / out)."""
    lib = load_library()
    if istartcol is None:
        istartcol = 1                       # radsurf_interface.F90:86-90
    if iendcol is None:
        iendcol = canopy_props.ncol         # :92-96
    keep = [config.to_c(), canopy_props.to_c(), sw_spectral_props.to_c(),
            lw_spectral_props.to_c(), bc_out.to_c()]
    fl = [f.to_c() if f is not None and f.nspec else None
          for f in (sw_norm_dir, sw_norm_diff, lw_internal, lw_norm)]
    args = [C.byref(k) for k in keep] + [C.c_int32(istartcol), C.c_int32(iendcol)]
    args += [C.byref(f) if f is not None else None for f in fl]
    _check(lib.spartacus_radsurf_hip(*args))


# ---- kernel-level entry points (radtool's routines on the device) ----
def _fstack(a):
    return np.asfortranarray(a, dtype=np.float64)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _check_units(rc):
    if rc != 0:
        msg = load_library().ss_units_last_error()
        raise SpartacusHipError("%s (%d): %s" % (STATUS.get(rc, "error"), rc, msg.decode() if msg else ""))


def calc_matrices_sw_eig(dz, mu0, gamma0, gamma1, gamma2, gamma3, dscale0=None, dscale=None):
    """radtool's calc_matrices_sw_eig on the device: stacks (nmat, n, n) etc. in, dict of
    stacks out (plus 'sym_route': 1 where the symmetric eigen-route was taken)."""
    lib = load_library()
    g0, g1, g2, g3 = _fstack(gamma0), _fstack(gamma1), _fstack(gamma2), _fstack(gamma3)
    nmat, nd, _ = g1.shape
    nr = g0.shape[1]
    d0 = _fstack(dscale0) if dscale0 is not None else None
    d1 = _fstack(dscale) if dscale is not None else None
    z = lambda *s: np.zeros(s, order="F")  # noqa: E731
    o = dict(reflectance=z(nmat, nd, nd), transmittance=z(nmat, nd, nd), s_up=z(nmat, nd, nr),
             s_dn=z(nmat, nd, nr), trans_dir=z(nmat, nr, nr), int_dir=z(nmat, nr, nr),
             int_diff=z(nmat, nd, nd), int_dir_diff=z(nmat, nd, nr))
    route = np.zeros(nmat, dtype=np.int32)
    _check_units(lib.ss_calc_matrices_sw_eig(
        C.c_int32(nmat), C.c_int32(nd), C.c_int32(nr), C.c_double(dz), C.c_double(mu0),
        _fp(g0), _fp(g1), _fp(g2), _fp(g3), _fp(d0), _fp(d1),
        *[_fp(o[k]) for k in ("reflectance", "transmittance", "s_up", "s_dn", "trans_dir", "int_dir",
                              "int_diff", "int_dir_diff")],
        route.ctypes.data_as(C.POINTER(C.c_int32))))
    o["sym_route"] = route
    return o


def calc_matrices_lw_eig(dz, gamma1, gamma2, emiss_rate, dscale=None):
    lib = load_library()
    g1, g2, b = _fstack(gamma1), _fstack(gamma2), _fstack(emiss_rate)
    nmat, n, _ = g1.shape
    d1 = _fstack(dscale) if dscale is not None else None
    z = lambda *s: np.zeros(s, order="F")  # noqa: E731
    o = dict(reflectance=z(nmat, n, n), transmittance=z(nmat, n, n), source=z(nmat, n),
             int_flux=z(nmat, n, n), int_flux_source=z(nmat, n))
    _check_units(lib.ss_calc_matrices_lw_eig(
        C.c_int32(nmat), C.c_int32(n), C.c_double(dz), _fp(g1), _fp(g2), _fp(b), _fp(d1),
        *[_fp(o[k]) for k in ("reflectance", "transmittance", "source", "int_flux", "int_flux_source")]))
    return o


def eigen_decomposition_real(amat, force_general_loops=False):
    """(eigenvalues (nmat, n), eigenvectors (nmat, n, n), ierr (nmat)) of a stack of matrices."""
    lib = load_library()
    a = _fstack(amat)
    nmat, n, _ = a.shape
    val = np.zeros((nmat, n), order="F")
    vec = np.zeros((nmat, n, n), order="F")
    ierr = np.zeros(nmat, dtype=np.int32)
    _check_units(lib.ss_eigen_decomposition_real(
        C.c_int32(n), C.c_int32(nmat), _fp(a), _fp(val), _fp(vec), C.c_int32(int(force_general_loops)),
        ierr.ctypes.data_as(C.POINTER(C.c_int32))))
    return val, vec, ierr


class MultiPlan:
    """One column set on several devices from one process (ss_plan_create_multi): a plan per contiguous
    column shard, each on its device; solve queues every part, download lets every part write its slices
    of the caller's full-size objects."""

    def __init__(self, config, canopy_props, sw_spectral_props, lw_spectral_props, devices,
                 istartcol=1, iendcol=None, want_profiles=True):
        self._lib = load_library()
        self._h = C.c_void_p()
        if iendcol is None:
            iendcol = canopy_props.ncol
        self._keep = [config.to_c(), canopy_props.to_c(), sw_spectral_props.to_c(), lw_spectral_props.to_c()]
        arr = (C.c_int32 * len(devices))(*devices)
        _check(self._lib.ss_plan_create_multi(C.byref(self._h), *[C.byref(k) for k in self._keep],
                                              C.c_int32(istartcol), C.c_int32(iendcol),
                                              C.c_int32(int(want_profiles)), arr, C.c_int32(len(devices))))

    def parts(self):
        """[(device, istartcol, iendcol)] of the shards"""
        n = C.c_int32(0)
        _check(self._lib.ss_multi_plan_count(self._h, C.byref(n)))
        out = []
        for k in range(n.value):
            d, a, b = C.c_int32(0), C.c_int32(0), C.c_int32(0)
            _check(self._lib.ss_multi_plan_part(self._h, C.c_int32(k), None, C.byref(d), C.byref(a), C.byref(b)))
            out.append((d.value, a.value, b.value))
        return out

    def solve(self):
        _check(self._lib.ss_multi_plan_solve(self._h))

    def synchronize(self):
        _check(self._lib.ss_multi_plan_synchronize(self._h))

    def download(self, bc_out, sw_norm_dir=None, sw_norm_diff=None, lw_internal=None, lw_norm=None):
        keep = [bc_out.to_c()] + [f.to_c() if f is not None and f.nspec else None
                                  for f in (sw_norm_dir, sw_norm_diff, lw_internal, lw_norm)]
        _check(self._lib.ss_multi_plan_download(
            self._h, *[C.byref(k) if k is not None else None for k in keep]))

    def close(self):
        if self._h:
            self._lib.ss_multi_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Plan:
    """Device-resident column set: create -> [upload] -> solve ... -> download."""

    def __init__(self, config, canopy_props, sw_spectral_props, lw_spectral_props,
                 istartcol=1, iendcol=None, want_profiles=True):
        self._lib = load_library()
        self.config = config
        self.canopy = canopy_props
        self.sw = sw_spectral_props
        self.lw = lw_spectral_props
        self._h = C.c_void_p()
        if iendcol is None:
            iendcol = canopy_props.ncol
        self._keep = [config.to_c(), canopy_props.to_c(), sw_spectral_props.to_c(),
                      lw_spectral_props.to_c()]
        _check(self._lib.ss_plan_create(C.byref(self._h), *[C.byref(k) for k in self._keep],
                                        C.c_int32(istartcol), C.c_int32(iendcol),
                                        C.c_int32(int(want_profiles))))

    def upload(self):
        self._keep[1:] = [self.canopy.to_c(), self.sw.to_c(), self.lw.to_c()]
        _check(self._lib.ss_plan_upload(self._h, *[C.byref(k) for k in self._keep[1:]]))

    def calc_simple_spectrum_lw(self):
        _check(self._lib.ss_plan_calc_simple_spectrum_lw(self._h))

    def solve(self):
        _check(self._lib.ss_plan_solve(self._h))

    def synchronize(self):
        _check(self._lib.ss_plan_synchronize(self._h))

    def download(self, bc_out, sw_norm_dir=None, sw_norm_diff=None, lw_internal=None,
                 lw_norm=None):
        keep = [bc_out.to_c()] + [f.to_c() if f is not None and f.nspec else None
                                  for f in (sw_norm_dir, sw_norm_diff, lw_internal, lw_norm)]
        _check(self._lib.ss_plan_download(
            self._h, *[C.byref(k) if k is not None else None for k in keep]))

    def kernel_ms(self, name):
        ms = C.c_double(0.0)
        _check(self._lib.ss_plan_last_kernel_ms(self._h, name.encode(), C.byref(ms)))
        return ms.value

    def scale_and_sum(self, top_flux_dn_sw=None, top_flux_dn_direct_sw=None, top_flux_dn_lw=None):
        """canopy_flux%scale / %sum on the device, as the reference's driver applies
        them after radsurf (driver/spartacus_surface_driver.F90:250-261)."""
        def ptr(a):
            if a is None:
                return None, None
            a = np.ascontiguousarray(a, dtype=np.float64)
            return a, a.ctypes.data_as(C.POINTER(C.c_double))
        k1, p1 = ptr(top_flux_dn_sw)
        k2, p2 = ptr(top_flux_dn_direct_sw)
        k3, p3 = ptr(top_flux_dn_lw)
        _check(self._lib.ss_plan_scale_and_sum(self._h, p1, p2, p3))

    def download_sums(self, sw_flux=None, lw_flux=None):
        cs = sw_flux.to_c() if sw_flux is not None else None
        cl = lw_flux.to_c() if lw_flux is not None else None
        _check(self._lib.ss_plan_download_sums(self._h, C.byref(cs) if cs is not None else None,
                                               C.byref(cl) if cl is not None else None))

    def set_pipes(self, npipes):
        """Concurrent streams used by solve(): 1 = serial kernels, 2 = overlapped chunks."""
        _check(self._lib.ss_plan_set_pipes(self._h, C.c_int32(int(npipes))))

    def eigen_failures(self):
        n = C.c_int64(0)
        _check(self._lib.ss_plan_eigen_failures(self._h, C.byref(n)))
        return n.value

    def device_ptr(self, name):
        p = C.c_void_p()
        n = C.c_int64(0)
        _check(self._lib.ss_plan_device_ptr(self._h, name.encode(), C.byref(p), C.byref(n)))
        return p.value, n.value

    def device(self):
        d = C.c_int32(-1)
        _check(self._lib.ss_plan_device(self._h, C.byref(d)))
        return d.value

    def close(self):
        if self._h:
            self._lib.ss_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
