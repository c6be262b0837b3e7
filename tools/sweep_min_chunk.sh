#!/bin/bash
# usage (GPU box): sweep_min_chunk.sh -> columns/s of the BASELINE workloads for the floor of the chunk size (column-units)
for w in "c2 --columns 100000 --steps 20" "c3 --columns 200000 --steps 6" "c3 --columns 1000000 --steps 4" "c3ns4 --columns 200000 --steps 3" "c4 --columns 100000 --steps 3" "rami5 --columns 20000 --steps 3"; do
  for m in 16384 65536 131072 262144; do
    v=$(SPARTACUS_MIN_CHUNK_UNITS=$m timeout -k 10 200 python bench.py --warmup 2 --no-extras --no-cpu-baseline --workload $w 2>/dev/null | python -c "import json,sys; print('%.3e' % json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
    echo "$w min_chunk_units $m: $v"
  done
done
