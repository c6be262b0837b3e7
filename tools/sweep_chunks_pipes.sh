#!/bin/bash
# usage (GPU box): sweep_chunks_pipes.sh [bench args] -> columns/s of the workload for streams x chunks per stream
for p in 2 4 8; do
  for c in 1 2 4; do
    v=$(SPARTACUS_PIPES=$p SPARTACUS_CHUNKS_PER_PIPE=$c timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; print('%.3e' % json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
    echo "pipes $p chunks/pipe $c: $v"
  done
done
