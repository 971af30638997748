#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r5z.sh -- the first frame after a gs_sync waiting for a partner (GS_PAIR_FIRST=1: ten pairs in the twenty-frame region) against going out alone (one + nine pairs + one)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5z; mkdir -p $O; cd $R
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); print(sys.argv[1], "value", d["value"], "steady", d["config"].get("steady_state_fps"), "region_ms", d["config"]["region_ms"], "share", d["occlusion_binning"]["near_permille"])
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
for i in 1 2 3 4; do
  for w in alone paired; do
    [ $w = paired ] && export GS_PAIR_FIRST=1 || unset GS_PAIR_FIRST
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-extras > $O/${w}_$i.json 2>$O/${w}_$i.err; show "20 $w $i" $O/${w}_$i.json
  done
done
for w in alone paired; do
  [ $w = paired ] && export GS_PAIR_FIRST=1 || unset GS_PAIR_FIRST
  python tools/cold_laps.py 2>&1 | grep ratio | tail -2
done
