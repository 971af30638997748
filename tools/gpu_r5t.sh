#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r5t.sh -- tail sorts (near-only sorts on the MSD path) against whole sorts: parity tests, then the driver's form and 480 steps, interleaved
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5t; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_as_benched.py -x -q -k "sort or near or as_bench" 2>&1 | tail -5
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); pf=d["per_frame"]; print(sys.argv[1], "value", d["value"], "steady", d["config"].get("steady_state_fps"), "region_ms", d["config"]["region_ms"], "sort/proj/bin/blend", pf["ms_sort"], pf["ms_project"], pf["ms_bin"], pf["ms_blend"], "share", d["occlusion_binning"]["near_permille"], "redrawn", d["config"].get("frames_redrawn_by_sync"), "depth1", (d.get("latency") or {}).get("fps_depth1"))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
for i in 1 2 3; do
  for w in whole tail; do
    [ $w = whole ] && export GS_BENCH_SORT_NEAR=0 || unset GS_BENCH_SORT_NEAR
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-extras > $O/${w}_$i.json 2>$O/${w}_$i.err; show "20 $w $i" $O/${w}_$i.json
  done
done
for w in whole tail; do
  [ $w = whole ] && export GS_BENCH_SORT_NEAR=0 || unset GS_BENCH_SORT_NEAR
  timeout 600 python bench.py --steps 480 --warmup 5 --no-cpu-baseline --no-configs --no-extras > $O/${w}_480.json 2>$O/${w}_480.err; show "480 $w" $O/${w}_480.json
done
