#!/bin/bash
# usage: tools/build_render_variant.sh <name> [extra hipcc flags]  -> csrc/libgs_variant_<name>.so with ONLY gs_render.hip rebuilt under the flags
# (the other objects are the product build's: run aframe-gaussian-splatting_amd/build.py first).  A/B aid: GS_SPLAT_LIB=<that file>.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/aframe-gaussian-splatting_amd/csrc; N=$1; shift
T=$(mktemp -d)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -fno-fast-math -Wno-unused-function "$@" -x hip -c $C/gs_render.hip -o $T/gs_render.o
OBJS=$(ls $C/*.o | grep -v gs_render.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libgs_variant_$N.so $OBJS $T/gs_render.o
rm -rf $T; echo built $C/libgs_variant_$N.so
