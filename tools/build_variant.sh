#!/bin/bash
# usage: build_variant.sh NAME "FLAGS"  -> <repo>/spartacus_surface_amd/lib/variants/libss_NAME.so
set -e
NAME=$1; FLAGS=$2; NSV=${3:-2}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT/spartacus_surface_amd/csrc"
OD="${TMPDIR:-/tmp}/ss_variants/$NAME"; mkdir -p $OD ../lib/variants
# UNITS: the instantiation units rebuilt with the flags (default: the two of the C3 workload)
UNITS=${UNITS:-"inst_sw_urban_r3 inst_lw_urban_r3 inst_fuse_sw_urban_r3 inst_fuse_lw_urban_r3"}
for u in $UNITS; do
  X=""; case $u in inst_fuse_*) X="-mllvm -disable-machine-licm";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-cuda-compat -fapprox-func -DSS_ONLY_NS=$NSV $X $FLAGS -c $u.hip -o $OD/$u.o &
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-cuda-compat $FLAGS -c ss_capi.hip -o $OD/ss_capi.o &
wait
OBJS="$OD/ss_capi.o ../lib/obj/ss_units_capi.o"
for u in inst_sw_forest_r2 inst_sw_forest_r3 inst_sw_urban_r1 inst_sw_urban_r2 inst_sw_urban_r3 inst_lw_forest_r2 inst_lw_forest_r3 inst_lw_urban_r1 inst_lw_urban_r2 inst_lw_urban_r3 inst_fuse_sw_forest_r2 inst_fuse_sw_forest_r3 inst_fuse_sw_urban_r1 inst_fuse_sw_urban_r2 inst_fuse_sw_urban_r3 inst_fuse_lw_forest_r2 inst_fuse_lw_forest_r3 inst_fuse_lw_urban_r1 inst_fuse_lw_urban_r2 inst_fuse_lw_urban_r3; do
  case " $UNITS " in *" $u "*) OBJS="$OBJS $OD/$u.o";; *) OBJS="$OBJS ../lib/obj/$u.o";; esac
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libss_$NAME.so $OBJS
echo built $NAME
