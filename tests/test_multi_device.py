"""Several GPUs of a node from ONE process (scope row e, the form a Fortran host can reach): the host entry
cuts a large call into one contiguous column shard per configured device (ss_set_devices / SPARTACUS_DEVICES),
one host thread per shard; ss_plan_create_multi is the device-resident form.  The test box has one GPU, so the
device list names device 0 several times: the shards then run side by side on one card - every piece of the
path except the second card.  Results must equal the unsharded call BIT FOR BIT (columns are independent:
radsurf_interface.F90:105).  Also here: many host threads in the entry's small-call path at once."""
import threading

import numpy as np
import pytest

from tests import fixture_io

pytestmark = pytest.mark.gpu


def _hip():
    from spartacus_surface_amd import radsurf
    radsurf.load_library()
    assert radsurf.device_count() >= 1
    return radsurf


def _alloc(case, profiles=False):
    from oracle import checker      # test-only helper for output allocation
    return checker.alloc_outputs(case["config"], case["canopy"], profiles)


def _call(radsurf, case, out, c0=1, c1=None):
    cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
    radsurf.radsurf(cfg, cp, sw, lw, out["bc_out"], c0, cp.ncol if c1 is None else c1, out["sw_norm_dir"],
                    out["sw_norm_diff"], out["lw_internal"], out["lw_norm"])


def _same(a, b):
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), k


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_host_entry_sharded_over_devices(devices):
    from spartacus_surface_amd import workloads
    radsurf = _hip()
    case = workloads.ragged_all_tiles(ncol=6000, seed=11)          # all tile kinds, ragged layer counts
    def sentinel_outputs():
        out = _alloc(case, True)
        for v in fixture_io.flatten_outputs(out).values():
            v[...] = -7.0           # (elements the reference leaves untouched - bc_out of single-layer urban tiles - stay)
        return out
    base = sentinel_outputs()
    radsurf.set_devices([])
    _call(radsurf, case, base)
    ref = {k: np.array(v, copy=True) for k, v in fixture_io.flatten_outputs(base).items()}
    try:
        radsurf.set_devices(devices)
        assert radsurf.get_devices() == devices
        out = sentinel_outputs()
        _call(radsurf, case, out)
        _same(ref, fixture_io.flatten_outputs(out))
        # a column range inside the objects, sentinel everywhere else
        out2 = sentinel_outputs()
        _call(radsurf, case, out2, 1001, 5000)
        cp = case["canopy"]
        l0, l1 = int(cp.istartlay[1000]) - 1, int(cp.istartlay[4999]) - 1 + int(cp.nlay[4999])
        for k, v in fixture_io.flatten_outputs(out2).items():
            n0 = v.shape[0]
            lo, hi = (1000, 5000) if n0 == cp.ncol else (l0, l1)
            assert np.array_equal(v[lo:hi], ref[k][lo:hi], equal_nan=True), k
            assert np.all(v[:lo] == -7.0) and np.all(v[hi:] == -7.0), k
    finally:
        radsurf.set_devices([])


def test_multi_plan_equals_single_plan():
    from spartacus_surface_amd import workloads
    radsurf = _hip()
    case = workloads.c3_vegetated_urban(ncol=5000, seed=5)
    cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
    one = radsurf.Plan(cfg, cp, sw, lw, want_profiles=False)
    one.solve()
    a = _alloc(case)
    one.download(a["bc_out"], a["sw_norm_dir"], a["sw_norm_diff"], a["lw_internal"], a["lw_norm"])
    one.close()
    mp = radsurf.MultiPlan(cfg, cp, sw, lw, [0, 0, 0], want_profiles=False)
    parts = mp.parts()
    assert len(parts) == 3 and parts[0][1] == 1 and parts[-1][2] == cp.ncol
    assert all(parts[k][2] + 1 == parts[k + 1][1] for k in range(2))
    mp.solve()
    mp.synchronize()
    b = _alloc(case)
    mp.download(b["bc_out"], b["sw_norm_dir"], b["sw_norm_diff"], b["lw_internal"], b["lw_norm"])
    mp.close()
    _same(fixture_io.flatten_outputs(a), fixture_io.flatten_outputs(b))


def test_small_calls_from_many_threads():
    """The driver's default: blocks of 16 columns from OpenMP threads (spartacus_surface_driver.F90:206-234).
    Twelve host threads take blocks from a shared counter - more threads than the pool has slots -; the
    assembled result equals one large call bit for bit."""
    from spartacus_surface_amd import workloads
    radsurf = _hip()
    radsurf.set_devices([])
    case = workloads.ragged_all_tiles(ncol=1600, seed=23)
    big = _alloc(case, True)
    _call(radsurf, case, big)
    out = _alloc(case, True)
    ncol, block = case["canopy"].ncol, 16
    nxt, lock, errors = [0], threading.Lock(), []

    def worker():
        try:
            while True:
                with lock:
                    c0 = nxt[0]
                    nxt[0] += block
                if c0 >= ncol:
                    return
                _call(radsurf, case, out, c0 + 1, min(ncol, c0 + block))
        except Exception as e:      # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker) for _ in range(12)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:2]
    _same(fixture_io.flatten_outputs(big), fixture_io.flatten_outputs(out))
