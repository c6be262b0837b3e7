#!/usr/bin/env python3
"""Condense the rocprofv3 --pmc passes of tools/profile_round.sh into
profiles/<tag>_pmc.json (per kernel: registers, scratch, counters, derived
per-column HBM traffic and executed fp64 flops) and refresh profiles/traffic.json,
which bench.py reads to fill roofline.traffic.

usage: tools/pmc_to_profile.py TAG NCOLUMNS [TRAFFIC_FILE [WORKLOAD]]
       (NCOLUMNS = columns of the PMC runs; TRAFFIC_FILE defaults to profiles/traffic.json, which is
        the C3 file bench.py reads - pass another name for other workloads)
"""
import collections
import csv
import glob
import json
import re
import sys

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



tag, ncol = sys.argv[1], int(sys.argv[2])
traffic_file = sys.argv[3] if len(sys.argv) > 3 else "profiles/traffic.json"
workload = sys.argv[4] if len(sys.argv) > 4 else "C3"
FAMILY = {("k_layer", "false"): "sw_layer", ("k_column", "false"): "sw_column",
          ("k_down", "false"): "sw_down", ("k_down", "true"): "lw_down",
          ("k_layer", "true"): "lw_layer", ("k_column", "true"): "lw_column",
          # the fused layer + upward kernel is timed as the "layer" family of its spectral region
          ("k_layer_up", "false"): "sw_layer", ("k_layer_up", "true"): "lw_layer"}


def family_of(kernel, args):
    # template arguments: <URBAN, NREG, NS, LW[, STAGE]>; the row-of-16 kernels of the orders 7..16:
    # k_layer16<N, D, LW>, k_column16 / k_down16 / k_pre16 as the one-lane kernels, k_eig_coop<N> serves
    # both spectral regions ("layer_both": half to each)
    a = [x.strip() for x in args.split(",")]
    if kernel == "k_layer16":       # <URBAN, NREG, NS, LW> since the second session of round 4, <N, D, LW> before
        return "lw_layer" if a[3 if len(a) > 3 else 2] == "true" else "sw_layer"
    if kernel == "k_pre16" and len(a) > 3:          # k_pre16<URBAN, NREG, NS, LW>
        return "lw_layer" if a[3] == "true" else "sw_layer"
    if kernel in ("k_pre16", "k_eig_coop"):
        return "layer_both"
    kernel = {"k_column16": "k_column", "k_down16": "k_down"}.get(kernel, kernel)
    return FAMILY.get((kernel, a[3] if len(a) > 3 else ""), kernel)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
meta = {}
for f in newest_per_pass("gpurun_out/pmc_%s" % tag):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+)<([^>]+)>", r["Kernel_Name"])
        if not m:
            continue
        k = "%s<%s>" % (m.group(1), m.group(2))
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        meta[k] = {"vgpr": int(r["VGPR_Count"]), "agpr": int(r["Accum_VGPR_Count"]),
                   "sgpr": int(r["SGPR_Count"]), "scratch_bytes_per_lane": int(r["Scratch_Size"]),
                   "family": family_of(m.group(1), m.group(2))}
out = {"command": "rocprofv3 --pmc <one group per pass> -- python bench.py --steps 1 --warmup 0 "
                  "--columns %d --no-cpu-baseline (tools/profile_round.sh)" % ncol,
       "columns": ncol, "kernels": {}}
traffic = {}
for k, c in sorted(agg.items()):
    d = dict(meta[k])
    d["counters"] = dict(c)
    waves = max(1.0, c.get("SQ_WAVES", 1.0))
    # FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B?  rocprofiler-sdk derives them as
    # (requests * bytes) / 1024; see MI355X_MICROARCH.md "HBM": FETCH_SIZE counts 128-B requests
    # of 16-B/lane reads at 64 B, so reads are doubled before being compared with a byte count.
    fetch = c.get("FETCH_SIZE", 0.0) * 1024.0
    write = c.get("WRITE_SIZE", 0.0) * 1024.0
    d["derived"] = {
        "fetch_bytes_per_column_raw": fetch / ncol, "fetch_bytes_per_column_x2": 2 * fetch / ncol,
        "write_bytes_per_column": write / ncol,
        "valu_insts_per_wave": c.get("SQ_INSTS_VALU", 0.0) / waves,
        "wait_any_frac": c.get("SQ_WAIT_ANY", 0.0) / max(1.0, c.get("SQ_WAVE_CYCLES", 1.0)),
        "active_frac": c.get("SQ_ACTIVE_INST_ANY", 0.0) / max(1.0, c.get("SQ_WAVE_CYCLES", 1.0)),
    }
    if "SQ_INSTS_VALU_FMA_F64" in c:
        fl = 64.0 * (2 * c["SQ_INSTS_VALU_FMA_F64"] + c.get("SQ_INSTS_VALU_ADD_F64", 0.0)
                     + c.get("SQ_INSTS_VALU_MUL_F64", 0.0) + c.get("SQ_INSTS_VALU_TRANS_F64", 0.0))
        d["derived"]["fp64_flops_per_column_executed"] = fl / ncol
    out["kernels"][k] = d
    fam = d["family"]
    t = traffic.setdefault(fam, {"fetch_x2": 0.0, "write": 0.0, "flops": 0.0})
    t["fetch_x2"] += 2 * fetch / ncol
    t["write"] += write / ncol
    t["flops"] += d["derived"].get("fp64_flops_per_column_executed", 0.0)
# kernels that serve both spectral regions in launches of their own (k_pre16, k_eig_coop): half to each layer family
if "layer_both" in traffic:
    both = traffic.pop("layer_both")
    for fam in ("sw_layer", "lw_layer"):
        t = traffic.setdefault(fam, {"fetch_x2": 0.0, "write": 0.0, "flops": 0.0})
        for k in both:
            t[k] += 0.5 * both[k]
def library_stamp():
    """build id (hash of its sources, ss_build_id) of the HIP library the counters were measured on = the one in
    the tree that travelled to the GPU box; bench.py refuses to price a DIFFERENT build with these counters."""
    import ctypes
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spartacus_surface_amd", "lib",
                        "libspartacus_surface_hip.so")
    lib = ctypes.CDLL(path)
    lib.ss_build_id.restype = ctypes.c_char_p
    return lib.ss_build_id().decode()


json.dump(out, open("profiles/%s_pmc.json" % tag, "w"), indent=1)
json.dump({"source": "profiles/%s_pmc.json" % tag, "workload": "%s, %d columns" % (workload, ncol),
           "library_build_id": library_stamp(),
           "note": "bytes per column per kernel family between the L2s and the fabric: FETCH_SIZE x 2 + WRITE_SIZE, "
                   "separate --pmc passes.  The x 2 is the gfx950 correction of MI355X_MICROARCH.md for 16-byte-per-lane "
                   "streaming reads, which is what these kernels issue for their records, input packs and spill "
                   "reloads (global_load_dwordx4 / scratch_load_dwordx4); the few narrower reads (8-byte inputs "
                   "of the pack kernel, 4-byte index loads) are uncalibrated there, so the read part is an upper "
                   "bound.  WRITE_SIZE is exact for the 16-byte record stores; the 8-byte per-layer output stores "
                   "of the downward kernels are uncalibrated.  Infinity-Cache hits are included: HBM itself may see less.",
           "bytes_per_column": {f: t["fetch_x2"] + t["write"] for f, t in traffic.items()},
           "fp64_flops_per_column_executed": {f: t["flops"] for f, t in traffic.items()}},
          open(traffic_file, "w"), indent=1)
print(json.dumps(json.load(open(traffic_file)), indent=1))
