#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r5o.sh -- which of the things that precede bench.py's 20-step region makes its first repetition slower (tools/region_probe.py), both libraries
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for f in "--profile --pre5" "--profile --presync" "--profile --pre6" "--profile --presync --pre6 --pre5" "--presync --pre6 --pre5"; do
  for w in old new; do
    [ $w = old ] && export GS_SPLAT_LIB=$R/aframe-gaussian-splatting_amd/csrc/libgs_variant_r04.so || unset GS_SPLAT_LIB
    echo "[$f]" $(timeout 300 python tools/region_probe.py $f 2>&1 | tail -1)
  done
done
