#!/bin/bash
# usage (GPU box, repo root): tools/gpu_steps20.sh [runs] -- the driver's form of the bench (--steps 20 --warmup 5) over and over: its spread
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/steps20
for i in $(seq 1 ${1:-12}); do
  timeout 300 python bench.py --steps 20 --warmup 5 $([ $i != 1 ] && echo --no-cpu-baseline --no-extras) > gpurun_out/steps20/run$i.json 2>/dev/null
  python -c "import json; d=json.load(open('gpurun_out/steps20/run$i.json')); c=d['config']; print($i, d['value'], c['region_ms'], c['steady_state_fps'], c['fill_drain_share'])"
done
