#!/bin/bash
# usage: sweep_pipes.sh "pipes list" [bench args...]: throughput against the number of HIP streams (scratch budget per stream scaled to keep the total)
for p in $1; do
  mb=$((16384 / p)); [ $mb -gt 4096 ] && mb=4096
  echo "== SPARTACUS_PIPES=$p SPARTACUS_SCRATCH_MB=$mb"
  SPARTACUS_PIPES=$p SPARTACUS_SCRATCH_MB=$mb bash tools/quick_bench.sh --no-extras "${@:2}" || exit 1
done
