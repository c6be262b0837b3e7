#!/bin/bash
# usage (GPU box): tools/kstats.sh TAG [bench args] -> gpurun_out/kstats_TAG/: rocprofv3 kernel-trace stats of one-stream bench steps
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/kstats_$tag
rm -rf $out; mkdir -p $out
SPARTACUS_PIPES=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- python3 bench.py --steps 3 --warmup 1 --no-extras "$@" > $out/bench.json 2> $out/log.txt || { tail -n 5 $out/log.txt; exit 1; }
f=$(find $out/t -name "*kernel_stats.csv" | head -n 1)
cp $f $out/kernel_stats.csv
python3 - $out/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-90s calls %4s avg %10.1f us  %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
