#!/bin/bash
# usage (GPU box, repo root): CONFIGS="C3 C1" STEPS=120 tools/gpu_r5s.sh -- round 4's library against this round's on other configurations, interleaved x2
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5s; mkdir -p $O; cd $R
C=$R/aframe-gaussian-splatting_amd/csrc
for cfg in ${CONFIGS:-C3}; do
for i in 1 2; do
  for w in old new; do
    [ $w = old ] && export GS_SPLAT_LIB=$C/libgs_variant_r04.so || unset GS_SPLAT_LIB
    timeout 900 python bench.py --config $cfg --steps ${STEPS:-120} --warmup 5 --no-cpu-baseline --no-configs --no-extras > $O/${cfg}_${w}_$i.json 2>$O/${cfg}_${w}_$i.err
    python - <<PY
import json
try:
    d=json.load(open("$O/${cfg}_${w}_$i.json")); print("$cfg $w $i: value", d["value"], "steady", d["config"].get("steady_state_fps"), "region_ms", d["config"]["region_ms"], "sort/proj/bin/blend", d["per_frame"]["ms_sort"], d["per_frame"]["ms_project"], d["per_frame"]["ms_bin"], d["per_frame"]["ms_blend"], "share", d["occlusion_binning"]["near_permille"], "I", d["per_frame"]["I_pairs"], "V", d["per_frame"]["V_sorted"])
except Exception as e: print("$cfg $w $i FAILED", e)
PY
  done
done
done
