#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r4j.sh -- k_lists testing a dense batch's runs per column instead of marking them through LDS atomics:
# the GPU tier, then the previous build (libgs_variant_head.so) against it: pipelined C2 / unsaturated / outside / depth 1, and k_lists alone
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|Error" | tail -3
AB_EXTRA="--near 0 --depths 3 --batch 2 --opacity-div 10 --frames 120" tools/gpu_ab_libs.sh r4j main head > /dev/null 2>&1
grep -A1 "==" gpurun_out/ab_r4j.txt | grep -v "^--" | paste - - | cut -c1-200
for v in main head; do
  L=""; [ $v != main ] && L=$R/aframe-gaussian-splatting_amd/csrc/libgs_variant_$v.so
  for a in "--near 0" "--near 0 --opacity-div 10 --frames 120" "--near 0 --outside" "--near 0 --splats 20971520 --size 3840x2160 --frames 60"; do
    ( cd /tmp && export TMPDIR=/tmp && GS_SPLAT_LIB=$L timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r4j -o st -- python $R/tools/stage_bench.py --depths 1 $a > $R/gpurun_out/r4j.log 2>&1 )
    echo "== $v $a: $(grep depth gpurun_out/r4j.log | cut -c1-100)"
    python tools/prof_tail.py gpurun_out/r4j/st_results.db 800 2>/dev/null | grep -E "k_lists<0|k_seg_count<0" | cut -c1-100
    rm -rf gpurun_out/r4j
  done
done
for v in main head main head; do
  L=""; [ $v != main ] && L=$R/aframe-gaussian-splatting_amd/csrc/libgs_variant_$v.so
  GS_SPLAT_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v c2', d['value'])"
  GS_SPLAT_LIB=$L timeout 600 python bench.py --splats 20971520 --size 3840x2160 --steps 120 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v c5', d['value'])"
done
