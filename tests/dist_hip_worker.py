"""Worker of tests/test_distributed_hip.py: one rank of a two-rank rehearsal of the
N > 1 path on ONE GPU (both ranks on cuda:0, gloo rendezvous - RCCL needs a device
per rank).  Every rank shards the case, creates a HIP plan for its shard, solves
it, and the DeviceGather of spartacus_surface_amd.distributed assembles bc_out and
the four flux objects on rank 0 from the plans' device arrays; rank 0 then solves
the whole case with a single plan and compares bit for bit.  Exit code 0 = equal."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    import numpy as np
    import torch
    import torch.distributed as dist
    from spartacus_surface_amd import distributed, radsurf, workloads
    from oracle import checker          # output allocation helper only (test code)
    from tests import fixture_io
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    radsurf.load_library()
    radsurf.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # every tile type, 1..10 layers per column, several bands, day and night
    case = workloads.ragged_all_tiles(ncol=600, seed=31)
    cfg, cp = case["config"], case["canopy"]
    ranges = distributed.shard_ranges(cp.nlay, world, cost_per_layer=1.0, cost_per_column=0.5)
    mine, (l0, l1) = distributed.extract_shard(case, *ranges[rank])
    plan = radsurf.Plan(mine["config"], mine["canopy"], mine["sw"], mine["lw"], want_profiles=True)
    plan.solve()
    plan.synchronize()
    g = distributed.DeviceGather(plan, dist, mine["canopy"].ncol, mine["canopy"].ntotlay, stage_host=True)
    g()
    rc = 0
    if rank == 0:
        got = g.to_host()
        whole = radsurf.Plan(cfg, cp, case["sw"], case["lw"], want_profiles=True)
        whole.solve()
        out = checker.alloc_outputs(cfg, cp, True)
        whole.download(out["bc_out"], out["sw_norm_dir"], out["sw_norm_diff"], out["lw_internal"], out["lw_norm"])
        whole.close()
        ref = fixture_io.flatten_outputs(out)
        assert set(ref) <= set(got), sorted(set(ref) - set(got))
        for k, v in ref.items():
            a = got[k].reshape(v.shape)
            if not np.array_equal(a, v):
                print("MISMATCH", k, float(np.max(np.abs(a - v))), flush=True)
                rc = 1
        print("rank 0: %d arrays, %d bytes per rank gathered, %s" % (len(ref), g.bytes_per_rank,
                                                                       "all equal" if rc == 0 else "DIFFERENT"), flush=True)
    dist.barrier()
    plan.close()
    dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
