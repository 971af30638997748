#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r4i.sh -- timeline of the driver's 20-step region (tools/prof_lanes.py over its last dispatches)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/r4i -o b -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/r4i.log 2>&1
cd $R; python tools/prof_lanes.py gpurun_out/r4i/b_results.db 150 > gpurun_out/r4i_lanes.txt 2>&1; rm -rf gpurun_out/r4i
tail -1 gpurun_out/r4i.log | cut -c1-300
