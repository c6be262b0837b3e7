"""TEST-ONLY host build of the kernel bodies (see ss_emul.cpp)."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libss_emul.so")
_lib = None

CXXFLAGS = ["-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas",
            # the "lane" type of the row-of-16 kernels for host builds (a pack of 16 doubles): TEST-ONLY
            "-I", HERE, '-DSS_ROW16_PACK_HEADER="row16_pack.hpp"']


def build(force=False, extra_flags=()):
    src = os.path.join(HERE, "ss_emul.cpp")
    csrc = os.path.join(HERE, "..", "..", "spartacus_surface_amd", "csrc")
    deps = [src, os.path.join(HERE, "row16_pack.hpp")] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hpp")]
    if (not force and os.path.exists(LIB)
            and os.path.getmtime(LIB) >= max(os.path.getmtime(d) for d in deps)):
        return LIB
    subprocess.run(["g++"] + CXXFLAGS + list(extra_flags) + ["-o", LIB, src], check=True)
    return LIB


def run_emul(config, cp, sw, lw, want_profiles=True, do_simple_spectrum=False,
             istartcol=1, iendcol=None, out=None):
    from oracle import checker   # only for alloc_outputs / argument packing (test code)
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
    if iendcol is None:
        iendcol = cp.ncol
    if out is None:
        out = checker.alloc_outputs(config, cp, want_profiles)
    keep, args = checker._common_args(config, cp, sw, lw, out, istartcol, iendcol)
    nfail = C.c_longlong(0)
    _lib.spartacus_radsurf_emul.restype = C.c_int
    rc = _lib.spartacus_radsurf_emul(*args, C.c_int32(int(do_simple_spectrum)),
                                     C.c_int32(int(want_profiles)), C.byref(nfail))
    if rc != 0:
        raise RuntimeError("emulated HIP path returned %d" % rc)
    return out, nfail.value


def run_scale_sum_emul(config, cp, sw, lw, top_sw, top_sw_dir, top_lw, want_profiles=True,
                       istartcol=1, iendcol=None):
    """solve + scale_and_sum through the emulated plan; returns (outputs, sw_flux, lw_flux)."""
    import numpy as np
    from oracle import checker
    from spartacus_surface_amd import types
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
    out = checker.alloc_outputs(config, cp, want_profiles)
    keep, args = checker._common_args(config, cp, sw, lw, out, istartcol,
                                      cp.ncol if iendcol is None else iendcol)
    sums = []
    for nspec, direct in ((config.nsw, True), (config.nlw, False)):
        f = types.CanopyFlux()
        f.allocate(config, cp.ncol, cp.ntotlay, nspec, use_direct=direct, do_save_flux_profile=want_profiles)
        sums.append(f)
    cs = [f.to_c() for f in sums]
    dp = C.POINTER(C.c_double)
    arrs = [np.ascontiguousarray(a, dtype=np.float64) if a is not None else None
            for a in (top_sw, top_sw_dir, top_lw)]
    ptrs = [a.ctypes.data_as(dp) if a is not None else None for a in arrs]
    _lib.spartacus_scale_sum_emul.restype = C.c_int
    rc = _lib.spartacus_scale_sum_emul(*args, C.c_int32(int(want_profiles)), *ptrs,
                                       C.byref(cs[0]), C.byref(cs[1]))
    if rc != 0:
        raise RuntimeError("emulated scale_and_sum returned %d" % rc)
    return out, sums[0], sums[1]
