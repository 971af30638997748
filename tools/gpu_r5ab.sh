#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r5ab.sh -- four pipeline lanes once the runtime may use more than its default four hardware queues (GPU_MAX_HW_QUEUES)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5ab; mkdir -p $O; cd $R
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); print(sys.argv[1], "value", d["value"], "steady", d["config"].get("steady_state_fps"), "region_ms", d["config"]["region_ms"])
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
for q in 4 8; do
  for dp in 3 4; do
    export GPU_MAX_HW_QUEUES=$q GS_BENCH_DEPTH=$dp
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-extras > $O/q${q}_d${dp}.json 2>$O/q${q}_d${dp}.err; show "queues $q depth $dp 20" $O/q${q}_d${dp}.json
    timeout 600 python bench.py --steps 480 --warmup 5 --no-cpu-baseline --no-configs --no-extras > $O/q${q}_d${dp}_480.json 2>$O/q${q}_d${dp}_480.err; show "queues $q depth $dp 480" $O/q${q}_d${dp}_480.json
  done
done
