#!/bin/bash
# usage (GPU box, repo root): LIBS="r05c r05d new" tools/gpu_r5aa.sh -- the sort's kernels in the pipelined loop and alone, per build of the library (saved variants / the working tree)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp && export TMPDIR=/tmp
C=$R/aframe-gaussian-splatting_amd/csrc
for w in ${LIBS:-r05c new}; do
  [ $w = new ] && unset GS_SPLAT_LIB || export GS_SPLAT_LIB=$C/libgs_variant_$w.so
  for d in 3 1; do
    B=2; [ $d = 1 ] && B=1
    timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/kt_$w$d -o k -- python $R/tools/stage_bench.py --near 0 --depths $d --batch $B --frames 480 > $R/gpurun_out/kt_$w$d.log 2>&1
    (cd $R; echo "== $w depth $d: $(grep 'frames/s' gpurun_out/kt_$w$d.log | cut -c1-140)"; python tools/prof_tail.py gpurun_out/kt_$w$d/k_results.db 2400 2>&1 | grep -i "msd\|seg_sort\|sort_bucket\|sort_depth\|kernel time" ; rm -rf gpurun_out/kt_$w$d)
  done
done
