#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r4k.sh -- the depth pass with unconditional (clamped) loads, f64 min / max and the affine cut-out path:
# the GPU tier, then against the previous build (libgs_variant_head.so): the depth kernels alone at C2 / C3 / C5 and the frame rates
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|Error" | tail -3
for v in main head; do
  L=""; [ $v != main ] && L=$R/aframe-gaussian-splatting_amd/csrc/libgs_variant_$v.so
  for a in "--near 0" "--near 0 --splats 6291456 --cutout --split 1 --frames 120" "--near 0 --splats 20971520 --size 3840x2160 --frames 60"; do
    ( cd /tmp && export TMPDIR=/tmp && GS_SPLAT_LIB=$L timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r4k -o st -- python $R/tools/stage_bench.py --depths 1 $a > $R/gpurun_out/r4k.log 2>&1 )
    echo "== $v $a: $(grep depth gpurun_out/r4k.log | cut -c1-110)"
    python tools/prof_tail.py gpurun_out/r4k/st_results.db 800 2>/dev/null | grep -E "k_sort_depth" | cut -c1-100
    rm -rf gpurun_out/r4k
  done
done
for v in main head main head; do
  L=""; [ $v != main ] && L=$R/aframe-gaussian-splatting_amd/csrc/libgs_variant_$v.so
  GS_SPLAT_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v c2', d['value'])"
  GS_SPLAT_LIB=$L timeout 600 python bench.py --splats 6291456 --cutout --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v c3', d['value'])"
  GS_SPLAT_LIB=$L timeout 600 python bench.py --splats 20971520 --size 3840x2160 --steps 120 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v c5', d['value'])"
done
