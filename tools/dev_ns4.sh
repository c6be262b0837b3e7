#!/bin/bash
# usage (GPU box): tools/dev_ns4.sh TAG [variant] -> parity of the 4-stream goldens and one-stream kernel stats of c3ns4 with lib/variants/libss_<variant>.so
v=${2:-dev}
export SPARTACUS_HIP_LIB=$PWD/spartacus_surface_amd/lib/variants/libss_$v.so
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "urban__4stream or HET07_JPS_SUM-56-direct" 2>&1 | tail -n 3 || exit 1
tools/kstats.sh $1 --workload c3ns4 --columns 200000 || exit 1
python3 -c "
import json
d = json.loads(open('gpurun_out/kstats_$1/bench.json').read().strip().splitlines()[-1])
print('columns/s %.3e  ms/step %.2f' % (d['value'], d['ms_per_step']))"
