#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r5y.sh -- the margin on the measured share shrinking to 1.01 instead of 1.04 (csrc/libgs_variant_m101.so): the driver's form, 480 steps, and a fresh context's laps (misses?)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5y; mkdir -p $O; cd $R
C=$R/aframe-gaussian-splatting_amd/csrc
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); pf=d["per_frame"]; print(sys.argv[1], "value", d["value"], "steady", d["config"].get("steady_state_fps"), "region_ms", d["config"]["region_ms"], "share", d["occlusion_binning"]["near_permille"], "redrawn", d["config"].get("frames_redrawn_by_sync"), "retries", d["occlusion_binning"].get("timed_region_retries"))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
for i in 1 2 3; do
  for w in m104 m101; do
    [ $w = m101 ] && export GS_SPLAT_LIB=$C/libgs_variant_m101.so || unset GS_SPLAT_LIB
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-extras > $O/${w}_$i.json 2>$O/${w}_$i.err; show "20 $w $i" $O/${w}_$i.json
  done
done
for w in m104 m101; do
  [ $w = m101 ] && export GS_SPLAT_LIB=$C/libgs_variant_m101.so || unset GS_SPLAT_LIB
  timeout 600 python bench.py --steps 480 --warmup 5 --no-cpu-baseline --no-configs --no-extras > $O/${w}_480.json 2>$O/${w}_480.err; show "480 $w" $O/${w}_480.json
  echo "laps $w:"; python tools/cold_laps.py 2>&1 | grep -v amdgpu.ids | grep "lap\|ratio" | tail -6
done
unset GS_SPLAT_LIB
