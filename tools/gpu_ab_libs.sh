#!/bin/bash
# usage (GPU box, repo root): tools/gpu_ab_libs.sh <tag> <variant names...>  -- A/B of library builds (csrc/libgs_variant_<name>.so,
# tools/build_variant.sh; "main" = the committed build) on the pipelined headline loop, pinned and adaptive share, interleaved
# twice so that box drift shows
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/ab_$TAG.txt; cd $R; : > $O
C=$R/aframe-gaussian-splatting_amd/csrc
for rep in 1 2; do
for v in "$@"; do
  L=""; [ $v != main ] && L=$C/libgs_variant_$v.so
  for args in "--near 0 --depths 3 --batch 2 --frames 480" "--near 0 --depths 1 --frames 240" ${AB_EXTRA:+"$AB_EXTRA"}; do
    echo "== $v (rep $rep): $args" >> $O
    GS_SPLAT_LIB=$L timeout 300 python tools/stage_bench.py $args 2>&1 | grep "frames/s" >> $O
  done
done
done
cat $O
