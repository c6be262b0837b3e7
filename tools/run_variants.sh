#!/bin/bash
# usage: run_variants.sh "bench args" name1 name2 ... ; runs bench for each variant lib
args=$1; shift
mkdir -p gpurun_out/exp
for v in "$@"; do
  SPARTACUS_HIP_LIB=$PWD/spartacus_surface_amd/lib/variants/libss_$v.so timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-extras $args > gpurun_out/exp/$v.json 2> gpurun_out/exp/$v.err || echo "$v failed"
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/exp/%s.json" % v).read().strip().splitlines()[-1])
    print("%-14s %.3e col/s  ms/step %.1f  kernels %s" % (v, d["value"], d["ms_per_step"], {k: round(x,1) for k,x in d["kernel_ms_per_step"].items()}))
except Exception as e:
    print(v, "no result", e)
PY
done
