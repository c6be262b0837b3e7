#!/bin/bash
# usage: pmc_variant.sh NAME  -> gpurun_out/pmc_NAME/*.csv (3 counter passes, 200k columns)
v=$1
export SPARTACUS_HIP_LIB=$PWD/spartacus_surface_amd/lib/variants/libss_$v.so
[ "$v" = "main" ] && unset SPARTACUS_HIP_LIB
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_$v
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_$v/p1 -- python bench.py --steps 1 --warmup 0 --columns 200000 --no-extras > gpurun_out/pmc_$v/p1.log 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/pmc_$v/p2 -- python bench.py --steps 1 --warmup 0 --columns 200000 --no-extras > gpurun_out/pmc_$v/p2.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE SQ_INSTS_SALU SQ_INSTS_FLAT SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_$v/p3 -- python bench.py --steps 1 --warmup 0 --columns 200000 --no-extras > gpurun_out/pmc_$v/p3.log 2>&1
python tools/pmc_summary.py "$v"; exit 0
import csv, glob, collections, sys
v = sys.argv[1]
agg = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/pmc_%s/p*/*/*_counter_collection.csv" % v):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ss::" not in k: continue
        k = k.split("::")[-1][:40]
        agg[k][r["Counter_Name"]] = agg[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        agg[k]["_scratch"] = r["Scratch_Size"]; agg[k]["_vgpr"] = r["VGPR_Count"]+"+"+r["Accum_VGPR_Count"]
for k, c in sorted(agg.items()):
    w = c.get("SQ_WAVES", 1)
    print("%-40s regs %s scratch %s | per wave: VALU %.0f VMEM_RD %.0f WR %.0f | wait_any %.0f%% active %.0f%% | FETCH %.2f GB WRITE %.2f GB" % (
        k, c["_vgpr"], c["_scratch"], c.get("SQ_INSTS_VALU",0)/w, c.get("SQ_INSTS_VMEM_RD",0)/w, c.get("SQ_INSTS_VMEM_WR",0)/w,
        100*c.get("SQ_WAIT_ANY",0)/max(1,c.get("SQ_WAVE_CYCLES",1)), 100*c.get("SQ_ACTIVE_INST_ANY",0)/max(1,c.get("SQ_WAVE_CYCLES",1)),
        c.get("FETCH_SIZE",0)/1e6, c.get("WRITE_SIZE",0)/1e6))
PY
