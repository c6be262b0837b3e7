#!/bin/bash
# usage (GPU box): tools/coop6.sh -> the experiment "headline order (n = 6) on DPP rows" (lib/variants/libss_coop6.so,
# built with tools/build_variant.sh coop6 "-DSS_COOP_MIN_ORDER=6" 2): parity on C3, one-stream kernel stats, throughput
export SPARTACUS_HIP_LIB=$PWD/spartacus_surface_amd/lib/variants/libss_coop6.so
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "synthetic_c3 or urban__2stream" 2>&1 | tail -n 2 || exit 1
tools/kstats.sh coop6 --columns 500000 || exit 1
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('coop6: %.3e col/s %.2f ms (4 pipes, 1e6 columns)' % (d['value'], d['ms_per_step']))"
