#!/bin/bash
# usage (GPU box, repo root): tools/gpu_p20.sh -- kernel trace of the driver form of the bench (20 steps), one line per dispatch and queue
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/p20 -o b -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/p20.log 2>&1
cd $R; python tools/prof_lanes.py gpurun_out/p20/b_results.db 260 > gpurun_out/p20_timeline.txt 2>&1; rm -rf gpurun_out/p20; tail -1 gpurun_out/p20.log | cut -c1-200
