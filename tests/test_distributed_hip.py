"""The N > 1 path with the HIP plan under it (gpu): two ranks on the one GPU of the
box (gloo rendezvous, the rehearsal form of bench.py's multi-GPU mode) shard a ragged
all-tile-types column set, solve their shards with their own plans and gather bc_out
and the four flux objects from device memory (spartacus_surface_amd.distributed.
DeviceGather); the assembled arrays equal a single-plan run bit for bit."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_hip_plans_device_gather_bit_equal():
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_hip_worker.py"), str(r), "2", str(port)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "all equal" in outs[0], outs[0]
