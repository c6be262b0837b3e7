#!/bin/bash
# usage (GPU box): SPARTACUS_PIPES=1 tools/pmc_only.sh TAG  -> the four PMC passes of profile_round.sh only
tag=${1:-main}
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out gpurun_out/pmc_$tag
rm -rf gpurun_out/pmc_$tag/p*
P="python bench.py --steps 1 --warmup 0 --columns 200000 --no-extras"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_$tag/p1 -- $P > gpurun_out/pmc_$tag/p1.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/pmc_$tag/p2 -- $P > gpurun_out/pmc_$tag/p2.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE SQ_INSTS_SALU SQ_INSTS_FLAT SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_$tag/p3 -- $P > gpurun_out/pmc_$tag/p3.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d gpurun_out/pmc_$tag/p4 -- $P > gpurun_out/pmc_$tag/p4.log 2>&1 || exit 1
python tools/pmc_summary.py $tag > $out/pmc_summary.txt
cat $out/pmc_summary.txt
