#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r5b.sh -- round 5: the MSD depth sort (4 launches) against the two LSD passes (7): every sort parity test,
# then interleaved A/B bench lines (GS_SORT_MSD=0 / 1), driver form and steady state, depth-1 stage traces
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5b; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sort or fuzz or golden or c3 or c5 or near or pixels or strips or paired or pipelined" > $O/sort_tests.log 2>&1; echo "sort tests rc=$?"; tail -4 $O/sort_tests.log | cut -c1-300
for i in 1 2; do for m in 0 1; do
  GS_SORT_MSD=$m timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $O/bench20_msd${m}_$i.json 2>/dev/null
  python - <<PY
import json
try:
    d=json.load(open("$O/bench20_msd${m}_$i.json")); print("msd=$m run $i: value", d["value"], "steady", d["config"]["steady_state_fps"], "depth1", d["latency"]["fps_depth1"], "per_frame", d["per_frame"], "outside", d["outside_cloud"]["fps"], d["outside_cloud"]["stages"]["ms_sort"], "cold", d["cold_orbit"]["fps_first_lap"])
except Exception as e: print("msd=$m run $i FAILED", e)
PY
done; done
for m in 0 1; do GS_SORT_MSD=$m TRACE=14 tools/gpu_stage.sh r5b_msd$m --near 0 --depths 1,3 > /dev/null 2>&1; cp gpurun_out/stage_r5b_msd$m.txt $O/ 2>/dev/null; done
grep -E "k_sort|k_msd|k_seg|radix" $O/stage_r5b_msd1.txt | head -20
