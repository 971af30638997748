#!/bin/bash
# usage (GPU box, repo root): tools/gpu_stage.sh <tag> [stage_bench args]  -- un-profiled run, then a depth-1 kernel trace
TAG=${1:-st}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
timeout 900 python tools/stage_bench.py "$@" 2>&1 | tee gpurun_out/stage_$TAG.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $R/gpurun_out/st_$TAG -o st -- python $R/tools/stage_bench.py "$@" --depths 1 --frames 60 > $R/gpurun_out/st_$TAG.log 2>&1
cd $R
python tools/prof_tail.py gpurun_out/st_$TAG/st_results.db ${TAIL:-1200} ${TRACE:+--trace $TRACE} | tee -a gpurun_out/stage_$TAG.txt
rm -rf gpurun_out/st_$TAG
