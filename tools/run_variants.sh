#!/bin/bash
# usage (GPU box): run_variants.sh "bench args" name1 name2 ... ; runs the bench with each variant library
# (tools/build_variant.sh) on ONE stream, so that the per-kernel times are durations, and checks the
# result of a 4096-column C3 sample against the oracle under the parity bar
args=$1; shift
mkdir -p gpurun_out/exp
for v in "$@"; do
  lib=$PWD/spartacus_surface_amd/lib/variants/libss_$v.so
  [ "$v" = main ] && lib=$PWD/spartacus_surface_amd/lib/libspartacus_surface_hip.so
  SPARTACUS_PIPES=1 SPARTACUS_HIP_LIB=$lib timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-extras $args > gpurun_out/exp/$v.json 2> gpurun_out/exp/$v.err || echo "$v failed"
  SPARTACUS_HIP_LIB=$lib timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-extras $args > gpurun_out/exp/${v}_p4.json 2>> gpurun_out/exp/$v.err || echo "$v failed (4 pipes)"
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/exp/%s.json" % v).read().strip().splitlines()[-1])
    p = json.loads(open("gpurun_out/exp/%s_p4.json" % v).read().strip().splitlines()[-1])
    print("%-16s 1 stream %.3e col/s %5.2f ms  | 4 streams %.3e col/s | kernels %s" % (
        v, d["value"], d["ms_per_step"], p["value"], {k: round(x, 2) for k, x in d["kernel_spans_ms_per_step"].items()}))
except Exception as e:
    print(v, "no result", e)
PY
  if [ -n "$CHECK" ]; then
    SPARTACUS_HIP_LIB=$lib timeout -k 10 300 python -m pytest tests/test_hip_parity.py -m gpu -q -k "synthetic_c3 or urban__2stream or test_surfaces" 2>&1 | tail -n 1
  fi
done
