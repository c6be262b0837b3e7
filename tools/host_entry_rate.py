"""PCIe-inclusive rate of the host entry (spartacus_radsurf_hip) on the GPU box:
pageable vs page-locked caller arrays, per chunk size.  usage: python tools/host_entry_rate.py [ncol]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
out = {}
for chunk in (None, "12544", "25024", "33344", "100032"):
    if chunk:
        os.environ["SPARTACUS_HOST_CHUNK"] = chunk
    else:
        os.environ.pop("SPARTACUS_HOST_CHUNK", None)
    out["chunk=%s" % (chunk or "default")] = bench.host_entry_rate(ncol)
print(json.dumps(out, indent=1))
