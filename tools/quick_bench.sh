#!/bin/bash
# usage: quick_bench.sh [bench args...]  -> one summary line
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/bench_quick.json 2>gpurun_out/bench_quick.err || { tail -5 gpurun_out/bench_quick.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/bench_quick.json').read().strip().splitlines()[-1]); print('%.4g col/s  %.2f ms/step ' % (d['value'], d['ms_per_step']), {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()})"
