"""Page-locked caller arrays in the host entry (spartacus_radsurf_hip), and the life time of what the
entry caches between calls.

Round 2 lost a process here (gpurun_out/pin_tests.log: SIGABRT inside spartacus_radsurf_hip on the sixth
golden under SPARTACUS_PIN_HOST=1).  Cause, reproduced in isolation by tools/experiments/pin_probe
(scenario munmap_remap_copy): the registry kept registrations of arrays the caller had freed; numpy
arrays above malloc's mmap threshold are unmapped on free and the next test's arrays get NEW pages at
the SAME addresses, "covered" by the stale registration - the copy is then a DMA to pages that no longer
exist, the GPU raises a memory access fault and the runtime ends the process.  The registry now never
keeps a registration it was not explicitly given (ss_host_pin: the caller's contract); registrations made
for SPARTACUS_PIN_HOST=1 are owned by the call and gone when it returns.  These tests walk exactly the
sequences that used to kill the process, once, in one run.
"""
import gc
import os

import numpy as np
import pytest

from tests import fixture_io
from tests.conftest import golden_files

pytestmark = pytest.mark.gpu


def _hip():
    from spartacus_surface_amd import radsurf
    radsurf.load_library()
    assert radsurf.device_count() >= 1
    return radsurf


def _alloc(case, profiles=False):
    from oracle import checker      # test-only helper for output allocation
    return checker.alloc_outputs(case["config"], case["canopy"], profiles)


def _solve(radsurf, case, out):
    cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
    radsurf.radsurf(cfg, cp, sw, lw, out["bc_out"], 1, cp.ncol, out["sw_norm_dir"], out["sw_norm_diff"],
                    out["lw_internal"], out["lw_norm"])
    return {k: np.array(v, copy=True) for k, v in fixture_io.flatten_outputs(out).items()}


def _arrays(radsurf, case, out):
    return radsurf.host_arrays(case["config"], case["canopy"], case["sw"], case["lw"], out["bc_out"],
                               out["sw_norm_dir"], out["sw_norm_diff"], out["lw_internal"], out["lw_norm"])


NCOL = 20000       # per-layer arrays of 1.6 MB: far above malloc's mmap threshold, unmapped when freed


def test_pin_solve_free_reallocate_solve():
    """ss_host_pin -> solve -> ss_host_unpin_all -> free -> re-allocate (the allocator hands out the same
    addresses) -> solve unpinned -> pin the new arrays -> solve: same bits every time, nothing stays locked."""
    from spartacus_surface_amd import workloads
    radsurf = _hip()
    radsurf.host_unpin_all()
    case = workloads.c3_vegetated_urban(ncol=NCOL, seed=3)
    base = _solve(radsurf, case, _alloc(case))                       # pageable
    out = _alloc(case)
    arrs = _arrays(radsurf, case, out)
    addresses = sorted(a.ctypes.data for a in arrs)
    for a in arrs:
        radsurf.host_pin(a)
    assert radsurf.host_pinned_bytes() == sum(a.nbytes for a in arrs)
    pinned = _solve(radsurf, case, out)
    for k, v in base.items():
        assert np.array_equal(v, pinned[k]), k
    radsurf.host_unpin_all()                                          # the caller's side of the contract
    assert radsurf.host_pinned_bytes() == 0
    del out, arrs, case
    gc.collect()
    case = workloads.c3_vegetated_urban(ncol=NCOL, seed=3)           # new pages, (mostly) the old addresses
    out = _alloc(case)
    arrs = _arrays(radsurf, case, out)
    reused = len(set(addresses) & set(a.ctypes.data for a in arrs))
    again = _solve(radsurf, case, out)
    for a in arrs:
        radsurf.host_pin(a)
    again_pinned = _solve(radsurf, case, out)
    radsurf.host_unpin_all()
    for k, v in base.items():
        assert np.array_equal(v, again[k]), k
        assert np.array_equal(v, again_pinned[k]), k
    print("addresses re-used by the second allocation: %d of %d" % (reused, len(addresses)))


def test_env_pinning_is_owned_by_the_call(monkeypatch):
    """SPARTACUS_PIN_HOST=1 with arrays that are freed and re-allocated between calls (what killed the
    process in round 2): every call locks its own arrays and releases them before it returns."""
    from spartacus_surface_amd import workloads
    radsurf = _hip()
    radsurf.host_unpin_all()
    monkeypatch.delenv("SPARTACUS_PIN_HOST", raising=False)
    case = workloads.c3_vegetated_urban(ncol=NCOL, seed=4)
    base = _solve(radsurf, case, _alloc(case))
    del case
    monkeypatch.setenv("SPARTACUS_PIN_HOST", "1")
    for _ in range(3):
        gc.collect()
        case = workloads.c3_vegetated_urban(ncol=NCOL, seed=4)
        got = _solve(radsurf, case, _alloc(case))
        assert radsurf.host_pinned_bytes() == 0
        for k, v in base.items():
            assert np.array_equal(v, got[k]), k
        del case, got
    # block calls below the column threshold do not lock at all; forced (threshold 0) they still work
    monkeypatch.setenv("SPARTACUS_PIN_HOST_MIN_COLUMNS", "0")
    case = workloads.c3_vegetated_urban(ncol=NCOL, seed=4)
    out = _alloc(case)
    cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
    for i0 in range(1, NCOL + 1, NCOL // 4):
        radsurf.radsurf(cfg, cp, sw, lw, out["bc_out"], i0, min(NCOL, i0 + NCOL // 4 - 1), out["sw_norm_dir"],
                        out["sw_norm_diff"], out["lw_internal"], out["lw_norm"])
        assert radsurf.host_pinned_bytes() == 0
    for k, v in fixture_io.flatten_outputs(out).items():
        assert np.array_equal(v, base[k]), k


def test_golden_loop_with_env_pinning(monkeypatch):
    """The loop of gpurun_out/pin_tests.log: every golden case through the host entry under
    SPARTACUS_PIN_HOST=1 (threshold 0: the small cases lock too), fresh arrays per case."""
    radsurf = _hip()
    radsurf.host_unpin_all()
    monkeypatch.setenv("SPARTACUS_PIN_HOST", "1")
    monkeypatch.setenv("SPARTACUS_PIN_HOST_MIN_COLUMNS", "0")
    from tests.test_hip_parity import REFERENCE_SINGULAR
    ran = 0
    for path in golden_files():
        name = os.path.basename(path)[:-4]
        if name in REFERENCE_SINGULAR:
            continue
        case, expected = fixture_io.load_case(path)
        cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
        if case["do_simple_spectrum"]:
            from spartacus_surface_amd.inputs import calc_simple_spectrum_lw
            calc_simple_spectrum_lw(cfg, cp, lw)
        out = _alloc(case, case["want_profiles"])
        radsurf.radsurf(cfg, cp, sw, lw, out["bc_out"], 1, cp.ncol, out["sw_norm_dir"], out["sw_norm_diff"],
                        out["lw_internal"], out["lw_norm"])
        assert radsurf.host_pinned_bytes() == 0, name
        bad = fixture_io.parity_violations(fixture_io.flatten_outputs(out), expected, dict(case["ref_noise"]),
                                           truth=case["ref_truth"], noise_arr=case["ref_noise_arr"])
        assert not bad, (name, bad[:3])
        ran += 1
        del case, expected, out
    assert ran >= 70


def test_partly_pinned_array_is_copied_in_pieces():
    """A copy that starts inside a registration and ends outside it is rejected by the runtime
    (pin_probe: copy_straddles); the library splits it at the end of the registration."""
    from spartacus_surface_amd import workloads
    radsurf = _hip()
    radsurf.host_unpin_all()
    case = workloads.c3_vegetated_urban(ncol=NCOL, seed=5)
    base = _solve(radsurf, case, _alloc(case))
    out = _alloc(case)
    lib = radsurf.load_library()
    import ctypes as C
    # the first half of two inputs and of two outputs only
    halves = [case["canopy"].dz, case["lw"].air_ext, out["lw_internal"].veg_abs, out["sw_norm_dir"].ground_dn]
    for a in halves:
        assert lib.ss_host_pin(C.c_void_p(a.ctypes.data), C.c_int64(a.nbytes // 2)) == 0
    got = _solve(radsurf, case, out)
    # a range that partly overlaps a registration is refused, not re-registered
    assert lib.ss_host_pin(C.c_void_p(halves[0].ctypes.data), C.c_int64(halves[0].nbytes)) != 0
    radsurf.host_unpin_all()
    for k, v in base.items():
        assert np.array_equal(v, got[k]), k


def test_extra_members_and_pointers_toggled_between_calls():
    """(i) members the caller allocated but a plan of this configuration does not have
    (an LW flux object with *_dir members) must come back untouched - they used to be bound to
    uninitialised device memory and copied over the caller's array; (ii) a member passed in one call and
    NULL in the next must not leave a binding behind in the cached sub-plans."""
    from spartacus_surface_amd import workloads
    radsurf = _hip()
    case = workloads.c3_vegetated_urban(ncol=3000, seed=6)
    cfg, cp = case["config"], case["canopy"]
    base = _solve(radsurf, case, _alloc(case))
    sentinel = -4321.5
    out = _alloc(case)
    extra = {}
    for obj in ("lw_internal", "lw_norm"):
        for name, shape in (("ground_dn_dir", (cp.ncol, cfg.nlw)), ("top_dn_dir", (cp.ncol, cfg.nlw)),
                            ("ground_sunlit_frac", (cp.ncol,))):
            a = np.full(shape, sentinel)
            setattr(out[obj], name, a)
            extra[obj + "." + name] = a
    got = _solve(radsurf, case, out)
    for k, a in extra.items():
        assert np.all(a == sentinel), k
    for k, v in base.items():
        assert np.array_equal(v, got[k]), k
    # next call: a larger column set, several members the first call had are NULL now
    big = workloads.c3_vegetated_urban(ncol=9000, seed=6)
    out2 = _alloc(big)
    for obj, name in (("sw_norm_dir", "ground_dn"), ("sw_norm_diff", "top_net"), ("lw_norm", "ground_net"),
                      ("lw_internal", "ground_vertical_diff")):
        setattr(out2[obj], name, None)
    got2 = _solve(radsurf, big, out2)
    ref2 = _solve(radsurf, big, _alloc(big))
    for k, v in got2.items():
        assert np.array_equal(v, ref2[k]), k
    # and back to the first shape with everything present
    got3 = _solve(radsurf, case, _alloc(case))
    for k, v in base.items():
        assert np.array_equal(v, got3[k]), k


def test_arrays_the_caller_registered_itself_keep_their_registration(monkeypatch):
    """A caller that page-locked an array with hipHostRegister outside this library (a host model with its own
    GPU code): the entry's call-owned locking must neither register it a second time nor take the caller's
    registration away when the call returns (probe scenario double_register_single_unregister)."""
    import ctypes as C
    from spartacus_surface_amd import workloads
    radsurf = _hip()
    radsurf.host_unpin_all()
    hip = C.CDLL("libamdhip64.so")
    case = workloads.c3_vegetated_urban(ncol=NCOL, seed=9)
    base = _solve(radsurf, case, _alloc(case))
    dz = case["canopy"].dz
    assert dz.flags["C_CONTIGUOUS"] and dz.nbytes > (1 << 20)
    ptr = C.c_void_p(dz.ctypes.data)
    assert hip.hipHostRegister(ptr, C.c_size_t(dz.nbytes), C.c_uint(0)) == 0
    try:
        monkeypatch.setenv("SPARTACUS_PIN_HOST", "1")
        for _ in range(2):
            got = _solve(radsurf, case, _alloc(case))
            for k in base:
                assert np.array_equal(base[k], got[k], equal_nan=True), k
            # (measured, tools/experiments/reg_attr_probe.py: a second hipHostRegister of the range succeeds and ONE
            # hipHostUnregister then drops the registration altogether - the device pointer is gone)
            dev = C.c_void_p()
            assert hip.hipHostGetDevicePointer(C.byref(dev), ptr, C.c_uint(0)) == 0, "the caller's registration is gone"
        assert radsurf.host_pinned_bytes() == 0       # nothing of the call survives it
    finally:
        assert hip.hipHostUnregister(ptr) == 0        # still the caller's to release - exactly once
    assert hip.hipHostUnregister(ptr) != 0


def test_unpin_of_an_unknown_base_is_an_error():
    radsurf = _hip()
    a = np.zeros(1 << 18)
    with pytest.raises(radsurf.SpartacusHipError):
        radsurf.host_unpin(a)
