#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r5e.sh -- round 5: three depth sorts side by side, interleaved: LSD (two radix passes, 7 launches), MSD (4 launches),
# MSD with near-only sorts at 1 M splats; sort parity tests first
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5g; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_as_benched.py -m gpu -q -k "sort or fuzz or golden or c3 or c5 or near or pixels or strips or paired or pipelined or as_bench or async" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log | cut -c1-300
run() { # tag env...
  tag=$1; shift
  env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $O/b_$tag.json 2>$O/b_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$O/b_$tag.json")); print("$tag: value", d["value"], "steady", d["config"]["steady_state_fps"], "depth1", d["latency"]["fps_depth1"], "sort/proj/bin/blend", d["per_frame"]["ms_sort"], d["per_frame"]["ms_project"], d["per_frame"]["ms_bin"], d["per_frame"]["ms_blend"], "share", d["occlusion_binning"]["near_permille"], "redrawn", d["config"]["frames_redrawn_by_sync"], "outside", d["outside_cloud"]["fps"], d["outside_cloud"]["stages"]["ms_sort"], "cold", d["cold_orbit"]["fps_first_lap"], d["cold_orbit"]["fps_second_lap"], d["cold_orbit"]["near_permille_after_first_lap"], d["cold_orbit"]["frames_redrawn_by_sync"], "unsat", d["unsaturated_scene"]["fps"])
except Exception as e: print("$tag FAILED", e)
PY
}
for i in 1 2; do
  run lsd_$i GS_SORT_MSD=0
  run msd_$i GS_SORT_MSD=1 GS_SORT_NEAR_SHORT_OFF=1
  run msdnear_$i GS_SORT_MSD=1
done
GS_SORT_NEAR_SHORT_OFF=1 TRACE=14 tools/gpu_stage.sh r5g_msd --near 0 --depths 1,3 > /dev/null 2>&1; cp gpurun_out/stage_r5g_msd.txt $O/ 2>/dev/null
TRACE=14 tools/gpu_stage.sh r5g_msdnear --near 0 --depths 1,3 > /dev/null 2>&1; cp gpurun_out/stage_r5g_msdnear.txt $O/ 2>/dev/null
for t in msd msdnear; do echo == $t; grep -E "k_sort|k_msd|k_seg|radix|frames/s" $O/stage_r5e_$t.txt | head -12; done
