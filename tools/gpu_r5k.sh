#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r5k.sh -- round 5 against round 4 ON ONE BOX, interleaved: the library of commit a1a8f72 (built here as
# csrc/libgs_variant_r04.so) and this round's, under this round's bench.py: C2 (driver form and default), C1, C3, C4, C5
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5k; mkdir -p $O; cd $R
OLD=$R/aframe-gaussian-splatting_amd/csrc/libgs_variant_r04.so
show() { python - <<PY
import json
try:
    d=json.load(open("$1")); print("$2: value", d["value"], "steady", d["config"].get("steady_state_fps"), "sort/proj/bin/blend", d["per_frame"]["ms_sort"], d["per_frame"]["ms_project"], d["per_frame"]["ms_bin"], d["per_frame"]["ms_blend"], "share", d["occlusion_binning"]["near_permille"], "I", d["per_frame"]["I_pairs"], "redrawn", d["config"]["frames_redrawn_by_sync"], "depth1", (d.get("latency") or {}).get("fps_depth1"), "outside", (d.get("outside_cloud") or {}).get("fps"), "cold", (d.get("cold_orbit") or {}).get("fps_first_lap"), (d.get("cold_orbit") or {}).get("fps_second_lap"), "unsat", (d.get("unsaturated_scene") or {}).get("fps"))
except Exception as e: print("$2 FAILED", e)
PY
}
for i in 1 2; do
  for w in old new; do
    [ $w = old ] && export GS_SPLAT_LIB=$OLD || unset GS_SPLAT_LIB
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $O/c2s20_${w}_$i.json 2>$O/c2s20_${w}_$i.err; show $O/c2s20_${w}_$i.json C2-steps20-$w-$i
  done
done
for c in C2 C1 C3 C4 C5; do
  for w in old new; do
    [ $w = old ] && export GS_SPLAT_LIB=$OLD || unset GS_SPLAT_LIB
    timeout 900 python bench.py --config $c $([ $c = C5 ] && echo --steps 120) --no-cpu-baseline --no-extras --no-configs > $O/${c}_$w.json 2>$O/${c}_$w.err; show $O/${c}_$w.json $c-$w
  done
done
unset GS_SPLAT_LIB
