#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r4d.sh -- near-only sorts whose depth pass stashes the candidates (k_sort_depth<.., SPEC> / k_near_filter):
# parity tests, then C5 (20 M @ 4K) and C3 with the path on / off (GS_SPEC_STASH=0) and the sort kernels of a C5 frame
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "depth_pass_stashes or near_only or c5_twenty or overflowed" 2>&1 | grep -E "speculative|near-only strip|passed|failed|Error|assert" | cut -c1-200 | tail -20
for e in 1 0; do
  for i in 1 2; do
    GS_SPEC_STASH=$e timeout 600 python bench.py --splats 20971520 --size 3840x2160 --steps 120 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5 spec=$e', d['value'], d['config'].get('frames_redrawn_by_sync'), d['config'].get('near_only_sorts_from_the_depth_pass_stash'))"
  done
done
for e in 1 0; do
  GS_SPEC_STASH=$e timeout 600 python bench.py --splats 6291456 --cutout --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3 spec=$e', d['value'])"
done
for e in 1 0; do
  ( cd /tmp && export TMPDIR=/tmp && GS_SPEC_STASH=$e timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r4d -o st -- python $R/tools/stage_bench.py --depths 3 --near 0 --splats 20971520 --size 3840x2160 --frames 60 > $R/gpurun_out/r4d_$e.log 2>&1 )
  echo "== spec=$e"; grep "depth" gpurun_out/r4d_$e.log | cut -c1-150
  python tools/prof_tail.py gpurun_out/r4d/st_results.db 2000 2>/dev/null | head -24 | cut -c1-120
  rm -rf gpurun_out/r4d
done
timeout 600 python -m pytest tests/test_stress_gpu.py -m gpu -q -x 2>&1 | tail -2
