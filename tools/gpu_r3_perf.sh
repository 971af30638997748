#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r3_perf.sh <tag>  -- round-3 measurements in one call: hipGraph vs stream launches of a
# frame-like chain, A/B of the blend against csrc/libgs_variant_base.so (if present), kernel traces of the unsaturated scene
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/perf_$TAG; mkdir -p $O; cd $R
C=$R/aframe-gaussian-splatting_amd/csrc
( cd tools/micro && hipcc --offload-arch=gfx950 -O3 -o graph_rate graph_rate.hip 2>/dev/null; timeout 120 ./graph_rate ) > $O/graph_rate.txt 2>&1; cat $O/graph_rate.txt
for lib in base new; do
  L=""; [ $lib = base ] && L=$C/libgs_variant_base.so
  [ $lib = base ] && [ ! -f $L ] && continue
  for args in "--near 180 --depths 1,3" "--near 180 --depths 3 --batch 2" "--near 180 --depths 1 --no-early-out --frames 60"; do
    echo "== $lib: $args" | tee -a $O/ab.txt
    GS_SPLAT_LIB=$L timeout 300 python tools/stage_bench.py $args 2>&1 | grep "frames/s" | tee -a $O/ab.txt
  done
done
cd /tmp && export TMPDIR=/tmp
for mode in "adaptive --near 0" "pinned150 --near 150"; do
  set -- $mode; name=$1; shift
  timeout 600 rocprofv3 --kernel-trace -d $O/tr_$name -o tr -- python $R/tools/stage_bench.py --opacity-div 10 "$@" --depths 3 --batch 2 --frames 120 > $O/unsat_$name.log 2>&1
  grep "frames/s" $O/unsat_$name.log
  python $R/tools/prof_tail.py $O/tr_$name/tr_results.db 1600 > $O/unsat_${name}_kernels.txt 2>&1; head -30 $O/unsat_${name}_kernels.txt
  rm -rf $O/tr_$name
done
cd $R
