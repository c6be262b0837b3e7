#!/bin/bash
# usage (GPU box): tools/pmc_icache.sh TAG -> instruction-fetch counters per kernel
tag=${1:-main}
export TMPDIR=/tmp
P="python bench.py --steps 1 --warmup 0 --columns 200000 --no-extras"
mkdir -p gpurun_out/pmc_$tag
timeout -k 10 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d gpurun_out/pmc_$tag/p5 -- $P > gpurun_out/pmc_$tag/p5.log 2>&1 || { tail -5 gpurun_out/pmc_$tag/p5.log; exit 1; }
python - "$tag" <<'PY'
import csv, glob, collections, sys, re
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmc_%s/p5/*/*_counter_collection.csv" % tag):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+<[^>]+>)", r["Kernel_Name"])
        if m: agg[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in sorted(agg.items()):
    w = max(1.0, c["SQ_WAVES"])
    print("%-30s per wave: icache req %8.0f hits %8.0f misses %8.0f | ifetch %8.0f | ifetch_level/wave_cycles %.3f" % (
        k, c["SQC_ICACHE_REQ"]/w, c["SQC_ICACHE_HITS"]/w, c["SQC_ICACHE_MISSES"]/w, c["SQ_IFETCH"]/w,
        c["SQ_IFETCH_LEVEL"]/max(1.0, c["SQ_WAVE_CYCLES"])))
PY
