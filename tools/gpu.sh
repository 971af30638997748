#!/bin/bash
# tools/gpu.sh <job> [args] -- everything that is sent to the MI355X box (`gpurun -- tools/gpu.sh <job> ...`), one script, one job per
# call or several joined with `+` (tools/gpu.sh tests+bench20).  Outputs go to gpurun_out/ (merged back by gpurun); what a round keeps is
# copied to profiles/ by hand or by tools/copy_final.sh.  Jobs:
#   tests [pytest args]        the GPU test tier (default: tests -m gpu -q -x)
#   test <expr>                pytest -m gpu -k <expr> -s
#   bench20 [n]                the driver's form of the bench (--steps 20 --warmup 5), n times (default 3), then the default run once
#   bench [bench.py args]      one bench.py run, line to gpurun_out/bench_<tag>.json (TAG=...)
#   stage <tag> [args]         tools/stage_bench.py un-profiled, then a depth-1 kernel trace of 60 frames summarised by tools/prof_tail.py
#   trace <tag> [args]         rocprofv3 --kernel-trace --stats of `python <args>`; per-kernel table to gpurun_out/trace_<tag>.txt
#   regimes <tag>              the three regimes (headline, outside the cloud, opacity / 10) x GS_OPT_SUBTILE 0 / 2: frames/s and stage times
#   cumask <tag>               A/B of CU-masked lane streams (GS_LANE_CUS / GS_BLEND_CUS; library built with tools/experiments/cumask_streams.patch)
#   variants <tag> [args]      tools/stage_bench.py [args] for the product and every csrc/libgs_variant_*.so
#   pmcx <tag> "<ctrs>" <kernel> [args]   one --pmc pass over tools/stage_bench.py [args], per-kernel averages of the counters
#   pmc <tag>                  tools/gpu_pmc.sh (counter passes) for the configurations bench.py reports traffic for
#   stress                     tools/stress_lanes.py + tools/stress_ranks.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${TAG:-r6}

job_tests() { if [ $# -eq 0 ]; then set -- tests -m gpu -q -x; fi; timeout ${T:-1500} python -m pytest "$@" 2>&1 | tee gpurun_out/pytest_$TAG.log | tail -25; }
job_test() { timeout ${T:-900} python -m pytest tests -m gpu -q -x -s -k "$1" 2>&1 | tee gpurun_out/pytest_k_$TAG.log | tail -40; }
job_bench20() {
    local n=${1:-3}
    for i in $(seq 1 "$n"); do
        timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>gpurun_out/bench20_$TAG.$i.err > gpurun_out/bench20_$TAG.$i.json
        python - "$i" <<'EOF'
import json, sys, os
t = os.environ.get("TAG", "r6")
d = json.load(open("gpurun_out/bench20_%s.%s.json" % (t, sys.argv[1])))
print("steps20", d["value"], "steady", d["config"].get("steady_state_fps"), "region_ms", d["config"].get("region_ms"), "sort_mode", d["config"].get("sort_mode"))
EOF
    done
}
job_bench() { timeout ${T:-1500} python bench.py "$@" 2>gpurun_out/bench_$TAG.err > gpurun_out/bench_$TAG.json; tail -c 600 gpurun_out/bench_$TAG.json; }
job_stage() {
    local tag=$1; shift
    timeout 900 python tools/stage_bench.py "$@" 2>&1 | tee gpurun_out/stage_$tag.txt
    ( cd /tmp && timeout 900 rocprofv3 --kernel-trace -d "$R/gpurun_out/st_$tag" -o st -- python "$R/tools/stage_bench.py" "$@" --depths 1 --frames 60 > "$R/gpurun_out/st_$tag.log" 2>&1 )
    python tools/prof_tail.py gpurun_out/st_$tag/st_results.db ${TAIL:-1200} | tee -a gpurun_out/stage_$tag.txt
    rm -rf gpurun_out/st_$tag
}
job_trace() {
    local tag=$1; shift
    ( cd /tmp && timeout ${T:-900} rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/tr_$tag" -o tr -- python "$@" > "$R/gpurun_out/tr_$tag.log" 2>&1 )
    python tools/prof_tail.py gpurun_out/tr_$tag/tr_results.db ${TAIL:-100000} | tee gpurun_out/trace_$tag.txt | head -40
    rm -rf gpurun_out/tr_$tag
}
job_regimes() {
    local tag=${1:-$TAG}
    for sub in 0 2; do
        for reg in "headline --near 0" "outside --outside --near 0" "unsat --opacity-div 10 --near 0"; do
            set -- $reg; local name=$1; shift
            echo "== $name subtile=$sub" | tee -a gpurun_out/regimes_$tag.txt
            timeout 600 python tools/stage_bench.py "$@" --batch 2 --subtile $sub --depths 1,3 --frames 240 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/regimes_$tag.txt
        done
    done
}
job_cumask() {   # CU-masked lane streams: needs the library built with tools/experiments/cumask_streams.patch (measured in round 6 and dropped:
                 # profiles/r06_ab_cumask.txt) -- per-lane ranges of every XCD's CUs, or a second masked stream for the blends
    local tag=${1:-$TAG}
    for v in "" "GS_LANE_CUS=8" "GS_LANE_CUS=11,11,10" "GS_LANE_CUS=16,16" "GS_BLEND_CUS=24" "GS_BLEND_CUS=20"; do
        echo "== ${v:-no mask}" | tee -a gpurun_out/cumask_$tag.txt
        env $v timeout 120 python tools/stage_bench.py --near 0 --batch 2 --depths 3 --frames 480 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/cumask_$tag.txt
        env $v timeout 120 python tools/stage_bench.py --near 0 --batch 2 --outside --depths 3 --frames 240 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/cumask_$tag.txt
    done
    # the C2 driver form with the two best of them, and the per-queue timeline of one
    for v in "" "GS_LANE_CUS=11,11,10" "GS_BLEND_CUS=24"; do
        echo "== bench.py --steps 20 --warmup 5 :: ${v:-no mask}" | tee -a gpurun_out/cumask_$tag.txt
        for i in 1 2 3; do env $v timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps20', d['value'], 'steady', d['config'].get('steady_state_fps'))" | tee -a gpurun_out/cumask_$tag.txt; done
    done
}
job_variants() {   # every csrc/libgs_variant_*.so (tools/build_render_variant.sh) next to the product: stage_bench with the given args
    local tag=$1; shift
    for lib in "" $(ls aframe-gaussian-splatting_amd/csrc/libgs_variant_*.so 2>/dev/null); do
        echo "== ${lib:-product} :: $*" | tee -a gpurun_out/variants_$tag.txt
        GS_SPLAT_LIB=${lib:+$R/$lib} timeout 600 python tools/stage_bench.py "$@" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/variants_$tag.txt
    done
}
job_pmcx() {   # pmcx <tag> "<counters>" <kernel substring> [stage_bench args]: one rocprofv3 --pmc pass (counters alone, no other trace domain)
    local tag=$1 ctr=$2 flt=$3; shift 3
    ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $ctr -d "$R/gpurun_out/px_$tag" -o px -- python "$R/tools/stage_bench.py" "$@" > "$R/gpurun_out/px_$tag.log" 2>&1 )
    echo "== $tag :: $ctr :: $*" | tee -a gpurun_out/pmcx_$TAG.txt
    python tools/pmc_generic.py gpurun_out/px_$tag/px_results.db "$flt" | tee -a gpurun_out/pmcx_$TAG.txt
    rm -rf gpurun_out/px_$tag
}
job_pmc() { tools/gpu_pmc.sh "${1:-$TAG}"; }
job_stress() { timeout 300 python tools/stress_lanes.py 31 2>&1 | tail -2; timeout 300 python tools/stress_ranks.py 32 3 2>&1 | tail -2; }

jobs=$1; shift
IFS='+' read -ra JL <<< "$jobs"
for j in "${JL[@]}"; do
    if declare -f "job_$j" > /dev/null; then "job_$j" "$@"; else echo "unknown job $j"; exit 2; fi
done
