#!/usr/bin/env python3
"""Opcode histogram of the kernels in a gfx950 assembly file (hipcc --cuda-device-only -S), no GPU needed.

usage: isa_hist.py FILE.s [kernel-substring ...]
Per kernel: instruction count by class (fp64 arithmetic split into fma / mul / add / other, moves, selects,
AGPR moves = VGPR spills of the first level, readlane / writelane = SGPR spills, scratch = spills to memory,
global / LDS / scalar memory, SALU, branches, waits), plus the top opcodes.
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_fma_f64") or op.startswith("v_fmac_f64"):
        return "f64_fma"
    if op.startswith("v_mul_f64"):
        return "f64_mul"
    if op.startswith("v_add_f64"):
        return "f64_add"
    if re.match(r"v_(rcp|rsq|sqrt|div_scale|div_fmas|div_fixup|frexp|ldexp|trig|fract|floor|rndne|min|max|cmp.*)_f64", op) or op.endswith("_f64"):
        return "f64_other"
    if op.startswith("v_accvgpr"):
        return "agpr_move"
    if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"):
        return "lane_move"
    if op.startswith("v_mov") or op.startswith("v_pk_mov"):
        return "v_mov"
    if op.startswith("v_cndmask"):
        return "v_cndmask"
    if op.startswith("v_cmp"):
        return "v_cmp"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("global_load") or op.startswith("flat_load") or op.startswith("buffer_load"):
        return "vmem_load"
    if op.startswith("global_store") or op.startswith("flat_store") or op.startswith("buffer_store") or op.startswith("global_atomic"):
        return "vmem_store"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch") or op.startswith("s_setpc") or op.startswith("s_swappc"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu_other"
    return "other"


def kernels(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            if name:
                yield name, body
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            yield name, body
            name, body = None, []
            continue
        t = line.strip()
        if not t or t.startswith(".") or t.startswith(";") or t.endswith(":"):
            continue
        body.append(t.split()[0])
    if name:
        yield name, body


def main():
    path, subs = sys.argv[1], sys.argv[2:]
    for name, body in kernels(path):
        short = re.sub(r"^_ZN2ss\d+_GLOBAL__N_1\d+", "", name)
        if subs and not any(s in name for s in subs):
            continue
        if len(body) < 50:
            continue
        cls = collections.Counter(classify(op) for op in body)
        ops = collections.Counter(body)
        f64 = cls["f64_fma"] + cls["f64_mul"] + cls["f64_add"] + cls["f64_other"]
        spill = cls["agpr_move"] + cls["lane_move"] + cls["scratch"]
        print("== %s" % short[:110])
        print("   %d instructions; fp64 arithmetic %d (fma %d, mul %d, add %d, other %d); spill moves %d = %.0f %% of fp64"
              % (len(body), f64, cls["f64_fma"], cls["f64_mul"], cls["f64_add"], cls["f64_other"], spill,
                 100.0 * spill / max(1, f64)))
        print("   " + "  ".join("%s %d" % (k, v) for k, v in sorted(cls.items(), key=lambda kv: -kv[1])))
        print("   top: " + "  ".join("%s %d" % (k, v) for k, v in ops.most_common(14)))


if __name__ == "__main__":
    main()
