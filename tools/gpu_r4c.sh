#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r4c.sh -- span-list parity tests, then depth-1/3 stage lines + k_lists / k_seg_count alone for the regimes that differ
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "span_lists or two_round or huge_splats or c4_xr or scene_depth" 2>&1 | tail -2
for a in "--near 0" "--near 0 --splats 6291456 --cutout --split 1 --frames 120" "--near 0 --opacity-div 10 --frames 120" "--near 0 --outside" "--near 0 --splats 20971520 --size 3840x2160 --frames 60"; do
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r4c -o st -- python $R/tools/stage_bench.py --depths 1,3 $a > $R/gpurun_out/r4c.log 2>&1 )
  echo "== $a"; grep "depth" gpurun_out/r4c.log | cut -c1-150
  python tools/prof_tail.py gpurun_out/r4c/st_results.db 2000 2>/dev/null | grep -E "k_lists<0>|k_seg_count<0>|k_emit_runs<0>|k_project<0|k_row_scan<0>|F_lists|F_seg" | cut -c1-100
  rm -rf gpurun_out/r4c
done
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps20', d['value'], d['config'].get('steady_state_fps'))"; done
timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default', d['value'])"
timeout 600 python bench.py --splats 6291456 --cutout --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3', d['value'])"
timeout 600 python bench.py --xr --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4', d['value'])"
