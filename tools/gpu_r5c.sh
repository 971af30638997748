#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r5c.sh -- round 5: the whole GPU tier on the build with the MSD sort and the measured share, then
# interleaved A/B bench lines (GS_SORT_MSD=0 / 1) and the depth-1 trace
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5d; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -25 $O/gpu_tests.log | cut -c1-400
for i in 1 2; do for m in 0 1; do
  GS_SORT_MSD=$m timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $O/bench20_msd${m}_$i.json 2>$O/bench20_msd${m}_$i.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench20_msd${m}_$i.json")); print("msd=$m run $i: value", d["value"], "steady", d["config"]["steady_state_fps"], "depth1", d["latency"]["fps_depth1"], "per_frame", d["per_frame"], "share", d["occlusion_binning"], "outside", d["outside_cloud"]["fps"], d["outside_cloud"]["stages"]["ms_sort"], "cold", d["cold_orbit"]["fps_first_lap"], d["cold_orbit"]["fps_second_lap"], d["cold_orbit"]["near_permille_after_first_lap"], "unsat", d["unsaturated_scene"]["fps"])
except Exception as e: print("msd=$m run $i FAILED", e)
PY
done; done
GS_SORT_MSD=1 TRACE=14 tools/gpu_stage.sh r5d_msd1 --near 0 --depths 1,3 > /dev/null 2>&1; cp gpurun_out/stage_r5d_msd1.txt $O/ 2>/dev/null
grep -E "k_sort|k_msd|k_seg|radix|frames/s" $O/stage_r5d_msd1.txt | head -20
