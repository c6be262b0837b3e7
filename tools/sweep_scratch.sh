#!/bin/bash
# usage: sweep_scratch.sh "MB list" [bench args...]: bench throughput against the per-pipe scratch budget
mbs="$1"; shift
for mb in $mbs; do
  echo "== SPARTACUS_SCRATCH_MB=$mb"
  SPARTACUS_SCRATCH_MB=$mb bash tools/quick_bench.sh --no-extras "$@" || exit 1
done
