#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r5u.sh -- pipeline depth 3 against 4 with this round's shorter chains (480 steps and the driver's form)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5u; mkdir -p $O; cd $R
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); pf=d["per_frame"]; print(sys.argv[1], "value", d["value"], "steady", d["config"].get("steady_state_fps"), "region_ms", d["config"]["region_ms"], "sort/proj/bin/blend", pf["ms_sort"], pf["ms_project"], pf["ms_bin"], pf["ms_blend"], "share", d["occlusion_binning"]["near_permille"], "redrawn", d["config"].get("frames_redrawn_by_sync"))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
for i in 1 2; do
  for dp in 3 4 2; do
    export GS_BENCH_DEPTH=$dp
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-extras > $O/d${dp}_$i.json 2>$O/d${dp}_$i.err; show "20 depth $dp" $O/d${dp}_$i.json
    timeout 600 python bench.py --steps 480 --warmup 5 --no-cpu-baseline --no-configs --no-extras > $O/d${dp}_480_$i.json 2>$O/d${dp}_480_$i.err; show "480 depth $dp" $O/d${dp}_480_$i.json
  done
done
