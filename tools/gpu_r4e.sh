#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r4e.sh -- the speculative stash after the rebuild: its parity test in full, the stress tier, and the sort
# kernels of a C5 frame one frame at a time (isolated durations) with the path on / off
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "depth_pass_stashes or c5_twenty or overflowed" 2>&1 | grep -E "speculative|near-only strip|passed|failed|Error|assert" | cut -c1-200 | tail -20
timeout 900 python -m pytest tests/test_stress_gpu.py -m gpu -q -x 2>&1 | tail -2
for e in 1 0; do
  ( cd /tmp && export TMPDIR=/tmp && GS_SPEC_STASH=$e timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r4e -o st -- python $R/tools/stage_bench.py --depths 1 --near 0 --splats 20971520 --size 3840x2160 --frames 60 > $R/gpurun_out/r4e_$e.log 2>&1 )
  echo "== spec=$e"; grep "depth" gpurun_out/r4e_$e.log | cut -c1-150
  python tools/prof_tail.py gpurun_out/r4e/st_results.db 1000 2>/dev/null | head -24 | cut -c1-120
  rm -rf gpurun_out/r4e
done
