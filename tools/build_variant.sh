#!/bin/bash
# usage: tools/build_variant.sh <name> [extra hipcc flags, e.g. -DGS_BLEND_BATCH=128]  -> csrc/libgs_variant_<name>.so
# (A/B aid: run bench.py / tools with GS_SPLAT_LIB=<that file>)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/aframe-gaussian-splatting_amd/csrc; N=$1; shift
T=$(mktemp -d)
for f in $(cd $C && ls *.hip | sed s/.hip//); do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -fno-fast-math -Wno-unused-function "$@" -x hip -c $C/$f.hip -o $T/$f.o &
done
hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -fno-fast-math -x hip -c $C/gs_host.cpp -o $T/gs_host.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libgs_variant_$N.so $T/*.o
rm -rf $T; echo built $C/libgs_variant_$N.so
