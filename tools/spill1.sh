#!/bin/bash
# usage: spill1.sh UNIT [extra flags] -> register / scratch metadata + ISA histogram of the kernels of one instantiation unit
# (default stream count 2: -DSS_ONLY_NS=${NSV:-2}); output assembly in /tmp/spillcheck/UNIT$TAG.s
u=$1; shift
mkdir -p /tmp/spillcheck; cd "$(dirname "$0")/../spartacus_surface_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fapprox-func -I../../include -DSS_ONLY_NS=${NSV:-2} "$@" --cuda-device-only -S $u.hip -o /tmp/spillcheck/$u$TAG.s 2>/tmp/spillcheck/$u$TAG.err || { head -30 /tmp/spillcheck/$u$TAG.err; exit 1; }
python3 - /tmp/spillcheck/$u$TAG.s <<'PY'
import re, sys
s = open(sys.argv[1]).read()
for m in re.finditer(r"\.group_segment_fixed_size: (\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size: (\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count: (\d+)", s, re.S):
    lds, name, priv, vg, sp = m.groups()
    k = re.search(r"\d+(k_\w+?)I", name).group(1)
    print("%-12s lds %6s scratch %5s vgpr %4s spill %4s" % (k, lds, priv, vg, sp))
PY
