#!/bin/bash
# usage: spillcheck.sh [extra flags]  -> register/scratch metadata of the C3 kernels (urban, 3 regions, 2 streams)
mkdir -p /tmp/spillcheck; cd "$(dirname "$0")/../spartacus_surface_amd/csrc"
for f in inst_sw_urban_r3 inst_lw_urban_r3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fapprox-func -I../../include -DSS_ONLY_NS=${NSV:-2} "$@" --cuda-device-only -S $f.hip -o /tmp/spillcheck/$f.s 2>/dev/null &
done
wait
for f in inst_sw_urban_r3 inst_lw_urban_r3; do
  python3 - /tmp/spillcheck/$f.s <<'PY'
import re, sys
s = open(sys.argv[1]).read()
for m in re.finditer(r"\.group_segment_fixed_size: (\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size: (\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count: (\d+)", s, re.S):
    lds, name, priv, vg, sp = m.groups()
    k = re.search(r"\d+(k_\w+?)I", name).group(1)
    print("%-40s %-9s lds %6s scratch %5s vgpr %4s spill %4s" % (sys.argv[1].split('/')[-1], k, lds, priv, vg, sp))
PY
done
