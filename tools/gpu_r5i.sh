#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r5i.sh -- round 5: every BASELINE configuration as its own bench run (default steps), like profiles/r04_config_c*.json
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5i; mkdir -p $O; cd $R
for c in C1 C3 C4 C5; do
  timeout 1200 python bench.py --config $c $([ $c = C5 ] && echo --steps 120) --no-cpu-baseline $([ $c != C5 ] && echo --no-extras) --no-configs > $O/config_$c.json 2>$O/config_$c.err
  python - <<PY
import json
try:
    d=json.load(open("$O/config_$c.json")); print("$c", d["value"], d["occlusion_binning"], d["per_frame"], "redrawn", d["config"]["frames_redrawn_by_sync"], "cold", (d.get("cold_orbit") or {}).get("fps_first_lap"), (d.get("cold_orbit") or {}).get("fps_second_lap"), d.get("extras_failed"))
except Exception as e: print("$c FAILED", e)
PY
done
