#!/bin/bash
# usage (GPU box, repo root): tools/gpu_tail.sh <tag> <ndispatch> [bench args]  -- kernel trace of bench.py, averages over the last dispatches
TAG=${1:-t}; N=${2:-1900}; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $R/gpurun_out/tl_$TAG -o tl -- python $R/bench.py --steps 120 --warmup 5 --no-cpu-baseline "$@" > $R/gpurun_out/tl_$TAG.log 2>&1
cd $R
python tools/prof_tail.py gpurun_out/tl_$TAG/tl_results.db $N | tee gpurun_out/tail_$TAG.txt
rm -rf gpurun_out/tl_$TAG
