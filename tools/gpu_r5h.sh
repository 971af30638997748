#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r5h.sh -- round 5: bench lines (driver form x3, default form) of the measured-share build, all configs in the line
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5h; mkdir -p $O; cd $R
show() { python - <<PY
import json
try:
    d=json.load(open("$1")); c=d.get("configs",{})
    print("$2: value", d["value"], "steady", d["config"]["steady_state_fps"], "depth1", d.get("latency",{}).get("fps_depth1"), "sort/proj/bin/blend", d["per_frame"]["ms_sort"], d["per_frame"]["ms_project"], d["per_frame"]["ms_bin"], d["per_frame"]["ms_blend"], "share", d["occlusion_binning"]["near_permille"], "I", d["per_frame"]["I_pairs"], "redrawn", d["config"]["frames_redrawn_by_sync"], "outside", d.get("outside_cloud",{}).get("fps"), "cold", d.get("cold_orbit",{}).get("fps_first_lap"), d.get("cold_orbit",{}).get("fps_second_lap"), d.get("cold_orbit",{}).get("near_permille_after_first_lap"), d.get("cold_orbit",{}).get("frames_redrawn_by_sync"), "unsat", d.get("unsaturated_scene",{}).get("fps"), "failed", d.get("extras_failed"))
    for k,v in c.items(): print("   ", k, v.get("frames_per_s"), v.get("near_permille"), v.get("dominant_stage"), v.get("error"))
except Exception as e: print("$2 FAILED", e)
PY
}
for i in 1 2 3; do timeout 900 python bench.py --steps 20 --warmup 5 $([ $i != 1 ] && echo --no-cpu-baseline --no-configs) > $O/bench20_$i.json 2>$O/bench20_$i.err; show $O/bench20_$i.json steps20_$i; done
timeout 900 python bench.py --no-cpu-baseline --no-configs > $O/bench480.json 2>$O/bench480.err; show $O/bench480.json steps480
tail -3 $O/bench20_1.err
