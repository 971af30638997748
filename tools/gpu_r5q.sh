#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r5q.sh -- the one-pass pre-roll against the 96-frame one: the driver's form x3 each, and every configuration once
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5q; mkdir -p $O; cd $R
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); print(sys.argv[1], "value", d["value"], "steady", d["config"].get("steady_state_fps"), "region_ms", d["config"]["region_ms"], "retries", d["occlusion_binning"].get("timed_region_retries"), "redrawn", d["config"].get("frames_redrawn_by_sync"), "share", d["occlusion_binning"]["near_permille"], "preroll", d["config"].get("preroll_frames"))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
for i in 1 2 3; do
  for w in one 96; do
    [ $w = 96 ] && export GS_BENCH_PREROLL_FRAMES=96 || unset GS_BENCH_PREROLL_FRAMES
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-extras > $O/${w}_$i.json 2>$O/${w}_$i.err; show "$w $i" $O/${w}_$i.json
  done
done
unset GS_BENCH_PREROLL_FRAMES
for c in C1 C3 C4 C5; do
  timeout 900 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-extras > $O/$c.json 2>$O/$c.err; show "$c" $O/$c.json
done
timeout 900 python bench.py --steps 480 --warmup 5 --no-cpu-baseline --no-configs --no-extras > $O/s480.json 2>$O/s480.err; show "480" $O/s480.json
