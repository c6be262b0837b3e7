#!/bin/bash
# usage (GPU box): tools/host_entry_trace.sh -> gpurun_out/he_trace/: kernel + memory-copy timeline of 3 pinned host-entry calls
export TMPDIR=/tmp
out=$PWD/gpurun_out/he_trace
rm -rf $out; mkdir -p $out
cd /tmp
SPARTACUS_PIN_HOST=1 timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/t -- python3 $GRAFT_REPO_ROOT/tools/host_entry_calls.py > $out/log.txt 2>&1
echo rc=$? >> $out/log.txt
python3 $GRAFT_REPO_ROOT/tools/host_entry_trace_summary.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
