#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r5w.sh -- the concurrency of the steady pipelined loop (tools/prof_overlap.py) once more, the share settled over whole laps first
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/pov -o b -- python $R/tools/stage_bench.py --depths 3 --frames 3000 --batch 2 --near 0 > $O/pov.log 2>&1
cd $R; grep "frames/s" $O/pov.log; python tools/prof_overlap.py $O/pov/b_results.db 2 0.3 0.8 > $O/r05_overlap_c2.txt 2>&1; rm -rf $O/pov; head -14 $O/r05_overlap_c2.txt
