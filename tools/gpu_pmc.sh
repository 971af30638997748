#!/bin/bash
# usage (GPU box, repo root): tools/gpu_pmc.sh <tag>  -- two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over the
# pipelined frame loop of the headline configuration (tools/stage_bench.py: adaptive share, 3 lanes x 2 frames per launch; its few
# synchronous warm-up frames are < 10 % of the launches of a kernel)
TAG=${1:-pmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_${TAG}_$c -o pmc -- python $R/tools/stage_bench.py --near 0 --depths 3 --frames 240 --batch 2 > $R/gpurun_out/pmc_${TAG}_$c.log 2>&1
  echo "$c rc=$?"
done
cd $R
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_FETCH_SIZE/pmc_results.db gpurun_out/pmc_${TAG}_WRITE_SIZE/pmc_results.db gpurun_out/pmc_$TAG.md gpurun_out/pmc_$TAG.json
rm -rf gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE
head -12 gpurun_out/pmc_$TAG.md
