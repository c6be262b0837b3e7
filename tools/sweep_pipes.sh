#!/bin/bash
# usage (GPU box): sweep_pipes.sh -> columns/s of the default bench for streams x scratch budget
for mb in 1024 4096 16384; do
  for p in 2 4 6 8; do
    v=$(SPARTACUS_PIPES=$p SPARTACUS_SCRATCH_MB=$mb timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-extras 2>/dev/null | python -c "import json,sys; print('%.3e' % json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
    echo "scratch ${mb} MB pipes $p: $v"
  done
done
