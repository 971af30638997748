#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r4f.sh -- pipeline depth 3 (default) vs 4 on C5, C3, C2 (GS_BENCH_DEPTH)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
P="import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d['value'], d['config'].get('steady_state_fps'), d['config'].get('near_only_sorts_from_the_depth_pass_stash'))"
for dpt in 3 4 3 4; do
  GS_BENCH_DEPTH=$dpt timeout 600 python bench.py --splats 20971520 --size 3840x2160 --steps 120 --no-cpu-baseline --no-extras 2>/dev/null | python -c "$P" "c5 depth=$dpt"
done
for dpt in 3 4; do
  GS_BENCH_DEPTH=$dpt timeout 600 python bench.py --splats 6291456 --cutout --no-cpu-baseline --no-extras 2>/dev/null | python -c "$P" "c3 depth=$dpt"
  GS_BENCH_DEPTH=$dpt timeout 600 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "$P" "c2 depth=$dpt"
done
