#!/usr/bin/env python3
"""Which SOURCE function owns the instructions of a kernel: opcode classes per (file, function), static, no GPU.

usage: isa_by_source.py FILE.s KERNEL-SUBSTRING [SRCDIR]
FILE.s: hipcc --cuda-device-only -S -gline-tables-only of an instantiation unit (line tables do not change the code).
Every instruction is charged to the innermost source location the compiler recorded for it (.loc), i.e. after
inlining to the helper it came from; the function is the nearest definition above that line in the source file.
Spill code and constant materialisation carry the location of the instruction they serve.
"""
import collections
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_hist import classify  # noqa: E402

path, want = sys.argv[1], sys.argv[2]
srcdir = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                                            "spartacus_surface_amd", "csrc")
files, defs = {}, {}
DEF = re.compile(r"^\s*(?:template\s*<[^>]*>\s*)?(?:SS_HD|static|inline|__device__|__forceinline__|constexpr|\s)*"
                 r"[\w:<>,\s\*&]+?\b(\w+)\s*\([^;]*$")


def function_at(fname, line):
    if fname not in defs:
        d = []
        p = os.path.join(srcdir, fname)
        if os.path.exists(p):
            for n, text in enumerate(open(p, errors="replace"), 1):
                if text.startswith((" ", "\t", "#", "/", "}")) or "(" not in text:
                    # definitions of this code base start in column 0; lambdas and members are charged to their owner
                    if not re.match(r"^\s{0,2}SS_HD ", text):
                        continue
                m = DEF.match(text)
                if m and m.group(1) not in ("if", "for", "while", "switch", "return", "static_assert"):
                    d.append((n, m.group(1)))
        defs[fname] = d
    best = "?"
    for n, name in defs[fname]:
        if n <= line:
            best = name
        else:
            break
    return best


per = collections.defaultdict(collections.Counter)
inside, cur = False, ("?", "?")
total = 0
for text in open(path):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', text)
    if m:
        files[int(m.group(1))] = os.path.basename(m.group(3))
        continue
    m = re.match(r"^(_Z\w+):", text)
    if m:
        inside = want in m.group(1)
        continue
    if not inside:
        continue
    if "s_endpgm" in text:
        inside = False
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", text)
    if m:
        f = files.get(int(m.group(1)), "?")
        cur = (f, function_at(f, int(m.group(2))))
        continue
    t = text.strip()
    if not t or t.startswith((";", ".", "/")) or t.endswith(":"):
        continue
    per[cur][classify(t.split()[0])] += 1
    total += 1

print("%s: %d instructions of kernels matching '%s'" % (os.path.basename(path), total, want))
cols = ["f64_fma", "f64_mul", "f64_add", "f64_other", "v_mov", "v_cndmask", "v_cmp", "agpr_move", "lane_move", "valu_other",
        "salu", "branch", "wait", "lds", "scratch", "vmem_load", "vmem_store", "smem"]
print("%-22s %-28s %6s %5s | " % ("file", "function", "instr", "%") + " ".join("%9s" % c[:9] for c in cols))
for key, c in sorted(per.items(), key=lambda kv: -sum(kv[1].values())):
    n = sum(c.values())
    if n < max(20, total // 400):
        continue
    print("%-22s %-28s %6d %5.1f | " % (key[0][:22], key[1][:28], n, 100.0 * n / total) + " ".join("%9d" % c.get(k, 0) for k in cols))
tot = collections.Counter()
for c in per.values():
    tot.update(c)
print("%-22s %-28s %6d %5.1f | " % ("all", "", total, 100.0) + " ".join("%9d" % tot.get(k, 0) for k in cols))
