#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r5v.sh -- per-kernel times of the poses outside the headline regime (outside the cloud; opacity / 10), one frame at a time (kernels alone) and pipelined
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp && export TMPDIR=/tmp
for w in outside unsat; do
  [ $w = outside ] && A="--outside" || A="--opacity-div 10"
  for d in 1 3; do
    B=1; [ $d = 3 ] && B=2
    timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/kt_$w$d -o k -- python $R/tools/stage_bench.py $A --near 0 --depths $d --batch $B --frames 120 > $R/gpurun_out/kt_$w$d.log 2>&1
    (cd $R; grep "frames/s" gpurun_out/kt_$w$d.log; python tools/prof_tail.py gpurun_out/kt_$w$d/k_results.db 1800 > gpurun_out/r05_kernels_${w}_depth$d.txt 2>&1; rm -rf gpurun_out/kt_$w$d; head -24 gpurun_out/r05_kernels_${w}_depth$d.txt)
  done
done
