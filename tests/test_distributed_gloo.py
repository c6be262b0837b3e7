"""The N > 1 path on CPU: two ranks over the gloo backend shard a mixed
forest / urban column set (different layer counts), solve their shards and
gather the results; the assembled arrays must equal the single-process result
bit for bit.  The per-shard solver here is the CPU oracle (test stand-in for
the HIP plan, which needs a GPU); what is under test is the sharding and the
padded all_gather of spartacus_surface_amd.distributed."""
import os
import socket

import numpy as np
import pytest

from tests import fixture_io


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _mixed_case():
    """test/simple surfaces (flat, forest, urban, veg-urban; 0 or 2 layers)
    replicated 6x from the golden fixture -> 24 columns with ragged layers."""
    from tests.conftest import golden_files
    from spartacus_surface_amd import abi
    path = [p for p in golden_files() if p.endswith("simple__test_surfaces.npz")][0]
    case, _ = fixture_io.load_case(path)
    cp, sw, lw = case["canopy"], case["sw"], case["lw"]
    rep = 6
    ncol0 = cp.ncol
    nlay = np.tile(cp.nlay, rep)
    for n in abi.CANOPY_REAL_FIELDS + ("i_representation",):
        a = getattr(cp, n)
        if a is not None:
            setattr(cp, n, np.concatenate([a] * rep, axis=0))
    for obj, names in ((sw, abi.SW_FIELDS), (lw, abi.LW_FIELDS)):
        for n in names:
            a = getattr(obj, n)
            if a is not None:
                setattr(obj, n, np.concatenate([a] * rep, axis=0))
    cp.set_layers(nlay)
    # make the columns differ
    cp.cos_sza = np.linspace(0.2, 0.9, cp.ncol)
    sw.ncol = lw.ncol = cp.ncol
    sw.ntotlay = lw.ntotlay = cp.ntotlay
    cp.normalise(); sw.normalise(); lw.normalise()
    assert cp.ncol == rep * ncol0
    return case


def _worker(rank, world, port, result_dict):
    import torch.distributed as dist
    from oracle import checker
    from spartacus_surface_amd import distributed
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case = _mixed_case()
    cp = case["canopy"]
    ranges = distributed.shard_ranges(cp.nlay, world, cost_per_layer=1.0, cost_per_column=0.5)
    shards = [distributed.extract_shard(case, *r) for r in ranges]
    layer_ranges = [s[1] for s in shards]
    mine = shards[rank][0]
    out, _ = checker.run_oracle(mine["config"], mine["canopy"], mine["sw"], mine["lw"],
                                want_profiles=True, do_simple_spectrum=True)
    local = fixture_io.flatten_outputs(out)
    full = distributed.gather_flat_outputs(local, ranges, layer_ranges, cp.ncol, cp.ntotlay, dist)
    # the device-to-device gather of the HIP path (pack, ONE gather to rank 0, unpack), driven
    # here over host arrays: same result as the per-array padded all_gather
    g = distributed.DeviceGather(distributed.HostArrays(local), dist, mine["canopy"].ncol,
                                 mine["canopy"].ntotlay)
    g()
    if rank == 0:
        result_dict.update({k: v for k, v in full.items()})
        result_dict["__ranges__"] = ranges
        packed = g.to_host()
        result_dict["__packed_equal__"] = bool(
            set(packed) == set(full) and all(np.array_equal(packed[k].reshape(full[k].shape), full[k])
                                             for k in full))
        result_dict["__packed_bytes__"] = int(g.bytes_per_rank)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather_matches_single_process():
    import torch.multiprocessing as mp
    from oracle import checker
    if not checker.have_oracle():
        checker.build(ref=False)
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    result = mgr.dict()
    mp.spawn(_worker, args=(world, port, result), nprocs=world, join=True)
    case = _mixed_case()
    ref, _ = checker.run_oracle(case["config"], case["canopy"], case["sw"], case["lw"],
                                want_profiles=True, do_simple_spectrum=True)
    ref = fixture_io.flatten_outputs(ref)
    ranges = result["__ranges__"]
    assert result["__packed_equal__"] and result["__packed_bytes__"] > 0
    assert ranges[0][0] == 1 and ranges[-1][1] == case["canopy"].ncol
    assert all(ranges[i][1] + 1 == ranges[i + 1][0] for i in range(len(ranges) - 1))
    for k, v in ref.items():
        assert np.array_equal(result[k], v), k


def test_device_gather_with_an_empty_shard():
    """More ranks than columns with layers: a rank with an empty shard takes part in the
    gather (its slot of the packed buffer is all padding)."""
    import torch.multiprocessing as mp
    port = _free_port()
    mgr = mp.Manager()
    result = mgr.dict()
    mp.spawn(_worker_empty, args=(2, port, result), nprocs=2, join=True)
    assert result["ok"], result.get("why")


def _worker_empty(rank, world, port, result):
    import torch.distributed as dist
    from spartacus_surface_amd import distributed
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ncol, nlay, nspec = (3, 7, 2) if rank == 0 else (0, 0, 2)
    rng = np.random.default_rng(rank)
    local = {"bc_out.sw_albedo": rng.random((ncol, nspec)), "sw_norm_dir.ground_dn": rng.random((ncol, nspec)),
             "sw_norm_dir.veg_abs": rng.random((nlay, nspec)), "sw_norm_dir.veg_sunlit_frac": rng.random(nlay)}
    g = distributed.DeviceGather(distributed.HostArrays(local), dist, ncol, nlay)
    g()
    if rank == 0:
        got = g.to_host()
        ok = all(np.array_equal(got[k].reshape(local[k].shape), local[k]) for k in local)
        result["ok"] = bool(ok)
        result["why"] = "" if ok else "assembled arrays differ"
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_balance_and_edges():
    from spartacus_surface_amd import distributed
    nlay = np.array([0, 2, 2, 2, 10, 10, 0, 0, 62, 1])
    for world in (1, 2, 3, 4, 8):
        r = distributed.shard_ranges(nlay, world)
        assert r[0][0] == 1 and r[-1][1] == nlay.size
        assert all(r[i][1] + 1 == r[i + 1][0] for i in range(world - 1))
    # more ranks than columns: empty shards are (i, i-1)
    r = distributed.shard_ranges(np.array([3, 3]), 4)
    assert sum(max(0, b - a + 1) for a, b in r) == 2
    # balanced by layers
    r = distributed.shard_ranges(np.full(1000, 10), 8)
    sizes = [b - a + 1 for a, b in r]
    assert max(sizes) - min(sizes) <= 1
