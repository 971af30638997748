#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r5l.sh -- the driver's 20-step form, interleaved x4: round 4's library, this round's, this round's without the blend's need recording
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5l; mkdir -p $O; cd $R
C=$R/aframe-gaussian-splatting_amd/csrc
for i in 1 2 3; do
  for w in old new; do
    case $w in old) export GS_SPLAT_LIB=$C/libgs_variant_r04.so;; new) unset GS_SPLAT_LIB;; noneed) export GS_SPLAT_LIB=$C/libgs_variant_noneed.so;; esac
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-extras > $O/${w}_$i.json 2>$O/${w}_$i.err
    python - <<PY
import json
try:
    d=json.load(open("$O/${w}_$i.json")); print("$w $i: value", d["value"], "steady", d["config"].get("steady_state_fps"), "region_ms", d["config"]["region_ms"], "sort/proj/bin/blend", d["per_frame"]["ms_sort"], d["per_frame"]["ms_project"], d["per_frame"]["ms_bin"], d["per_frame"]["ms_blend"], "share", d["occlusion_binning"]["near_permille"], "I", d["per_frame"]["I_pairs"])
except Exception as e: print("$w $i FAILED", e)
PY
  done
done
unset GS_SPLAT_LIB
