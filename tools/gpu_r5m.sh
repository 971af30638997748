#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r5m.sh -- the driver's 20-step form, interleaved x4: this round's library with the MSD sort and with the two LSD passes
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5m; mkdir -p $O; cd $R
for i in 1 2 3 4; do
  for w in msd lsd; do
    [ $w = lsd ] && export GS_SORT_MSD=0 || unset GS_SORT_MSD
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-extras > $O/${w}_$i.json 2>$O/${w}_$i.err
    python - <<PY
import json
try:
    d=json.load(open("$O/${w}_$i.json")); print("$w $i: value", d["value"], "steady", d["config"].get("steady_state_fps"), "region_ms", d["config"]["region_ms"], "sort/proj/bin/blend", d["per_frame"]["ms_sort"], d["per_frame"]["ms_project"], d["per_frame"]["ms_bin"], d["per_frame"]["ms_blend"], "share", d["occlusion_binning"]["near_permille"], "I", d["per_frame"]["I_pairs"])
except Exception as e: print("$w $i FAILED", e)
PY
  done
done
