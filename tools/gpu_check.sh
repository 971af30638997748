#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_check.sh <tag> [bench args...]
# runs the gpu test tier, an un-profiled bench, and a rocprofv3 kernel-trace of the same bench command.
TAG=${1:-run}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_$TAG.log
tail -4 gpurun_out/pytest_$TAG.log
timeout 600 python bench.py "$@" > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"
cat gpurun_out/bench_$TAG.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o bench -- python $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline > $R/gpurun_out/bench_prof_$TAG.log 2>&1
cd $R
python tools/prof_summary.py gpurun_out/prof_$TAG/bench_results.db 125 > gpurun_out/prof_$TAG.md 2>&1
rm -f gpurun_out/prof_$TAG/bench_results.db
head -30 gpurun_out/prof_$TAG.md
