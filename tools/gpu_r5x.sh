#!/bin/bash
# usage (GPU box, repo root): OLD=<variant name> tools/gpu_r5x.sh -- a saved build of this round's library (csrc/libgs_variant_<OLD>.so) against the working tree's: sort parity tests, then the driver's form x3 and 480 steps, interleaved
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5x; mkdir -p $O; cd $R
C=$R/aframe-gaussian-splatting_amd/csrc
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_as_benched.py -x -q -k "sort or near or as_bench" 2>&1 | tail -3
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); pf=d["per_frame"]; print(sys.argv[1], "value", d["value"], "steady", d["config"].get("steady_state_fps"), "region_ms", d["config"]["region_ms"], "sort/proj/bin/blend", pf["ms_sort"], pf["ms_project"], pf["ms_bin"], pf["ms_blend"], "share", d["occlusion_binning"]["near_permille"])
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
for i in 1 2 3; do
  for w in old new; do
    [ $w = old ] && export GS_SPLAT_LIB=$C/libgs_variant_${OLD:-r05a}.so || unset GS_SPLAT_LIB
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-extras > $O/${w}_$i.json 2>$O/${w}_$i.err; show "20 $w $i" $O/${w}_$i.json
  done
done
for i in 1 2; do
for w in old new; do
  [ $w = old ] && export GS_SPLAT_LIB=$C/libgs_variant_${OLD:-r05a}.so || unset GS_SPLAT_LIB
  timeout 600 python bench.py --steps 480 --warmup 5 --no-cpu-baseline --no-configs --no-extras > $O/${w}_480_$i.json 2>$O/${w}_480_$i.err; show "480 $w" $O/${w}_480_$i.json
done
done
unset GS_SPLAT_LIB
