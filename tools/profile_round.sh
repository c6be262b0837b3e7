#!/bin/bash
# usage (on the GPU box): tools/profile_round.sh TAG [bench args, e.g. --workload c3ns4 --columns 200000]
# -> gpurun_out/prof_TAG/ : kernel-trace stats of the default bench command, its JSON line,
#    and three separate PMC passes (200k columns) summarised by tools/pmc_summary.py
tag=${1:-main}
shift
X="$@"
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench.py --steps 5 --warmup 2 --no-extras $X > $out/bench_under_rocprof.json 2> $out/trace.log || exit 1
timeout -k 10 300 python bench.py --steps 5 --warmup 2 $X > $out/bench.json 2> $out/bench.log || exit 1
mkdir -p gpurun_out/pmc_$tag
P="python bench.py --steps 1 --warmup 0 --columns 1000000 --no-extras $X"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_$tag/p1 -- $P > gpurun_out/pmc_$tag/p1.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/pmc_$tag/p2 -- $P > gpurun_out/pmc_$tag/p2.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE SQ_INSTS_SALU SQ_INSTS_FLAT SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_$tag/p3 -- $P > gpurun_out/pmc_$tag/p3.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d gpurun_out/pmc_$tag/p4 -- $P > gpurun_out/pmc_$tag/p4.log 2>&1 || exit 1
python tools/pmc_summary.py $tag > $out/pmc_summary.txt
cat $out/bench.json
