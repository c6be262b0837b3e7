import csv, glob, collections, sys, re

def newest_per_pass(root):
    """The newest counter file of every pass directory (gpurun merges new files
    next to those of earlier runs of the same tag)."""
    import os
    out = []
    for d in sorted(glob.glob(root + "/p*")):
        fs = glob.glob(d + "/*/*_counter_collection.csv")
        if fs:
            out.append(max(fs, key=os.path.getmtime))
    return out


for v in sys.argv[1:]:
    agg = collections.defaultdict(dict)
    for f in newest_per_pass("gpurun_out/pmc_%s" % v):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_\w+<[^>]+>)", r["Kernel_Name"])
            if not m: continue
            k = m.group(1)
            agg[k][r["Counter_Name"]] = agg[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            agg[k]["_scratch"] = r["Scratch_Size"]; agg[k]["_vgpr"] = r["VGPR_Count"]+"+"+r["Accum_VGPR_Count"]
    print("== variant", v)
    for k, c in sorted(agg.items()):
        w = c.get("SQ_WAVES", 1)
        print("%-32s regs %-7s scratch %-5s | per wave: VALU %6.0f VMEM_RD %5.0f WR %5.0f SALU %6.0f | wait_any %2.0f%% wait_inst %2.0f%% active %2.0f%% | FETCH %.2f GB WRITE %.2f GB" % (
            k, c["_vgpr"], c["_scratch"], c.get("SQ_INSTS_VALU",0)/w, c.get("SQ_INSTS_VMEM_RD",0)/w, c.get("SQ_INSTS_VMEM_WR",0)/w, c.get("SQ_INSTS_SALU",0)/w,
            100*c.get("SQ_WAIT_ANY",0)/max(1,c.get("SQ_WAVE_CYCLES",1)), 100*c.get("SQ_WAIT_INST_ANY",0)/max(1,c.get("SQ_WAVE_CYCLES",1)), 100*c.get("SQ_ACTIVE_INST_ANY",0)/max(1,c.get("SQ_WAVE_CYCLES",1)),
            c.get("FETCH_SIZE",0)/1e6, c.get("WRITE_SIZE",0)/1e6))
