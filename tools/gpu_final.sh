#!/bin/bash
# usage (GPU box, repo root): tools/gpu_final.sh <tag>  -- everything profiles/ keeps for a round, from ONE build of the code:
# the GPU test tier, the default bench line, rocprofv3 kernel stats of the same bench command, the two PMC passes, bench lines of
# the other BASELINE.json configurations, depth-1 kernel traces.
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final_$TAG; mkdir -p $O; cd $R
# PART=profiles: only the bench lines, the kernel traces and the counter passes (no test tier, no other configurations)
if [ "$PART" != "profiles" ]; then
rm -f gpurun_out/pixel_parity.jsonl
timeout 1200 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; grep -E "passed|failed" $O/pytest.log | tail -1
cp gpurun_out/pixel_parity.jsonl $O/ 2>/dev/null; cp gpurun_out/js_visible_fps.txt $O/ 2>/dev/null
fi
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c 1-240 $O/bench.json
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2>/dev/null
if [ "$PART" != "profiles" ]; then
timeout 600 python bench.py --size 1280x720 --no-cpu-baseline --no-extras > $O/config_c1.json 2>/dev/null
timeout 900 python bench.py --splats 6291456 --cutout --no-cpu-baseline --no-extras > $O/config_c3.json 2>/dev/null
timeout 600 python bench.py --xr --no-cpu-baseline > $O/config_c4.json 2>/dev/null
timeout 900 python bench.py --splats 20971520 --size 3840x2160 --steps 120 --no-cpu-baseline --no-extras > $O/config_c5.json 2>/dev/null
GS_BENCH_COMM=1 timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_comm_world1.json 2>/dev/null
# ONE host process driving several "devices" (gs_create_multi; on a one-GPU box they share the GPU: the path, not the scaling)
for n in 1 2 8; do
  timeout 300 python bench.py --gpus $n --single-process > $O/single_process_device_$n.json 2>/dev/null
  timeout 300 python bench.py --gpus $n --single-process --host-direct > $O/single_process_host_$n.json 2>/dev/null
done
for f in $O/single_process_*.json; do python -c "
import json,sys
try:
    d=json.load(open('$f')); print('$f'.split('/')[-1], d['value'], d['config']['frame_equals_single_context_render'], d.get('host_frame_GBps'))
except Exception as e: print('$f FAILED', e)"; done
fi
timeout 120 python tools/pcie_probe.py > $O/pcie_probe.txt 2>&1
( cd tools/micro && hipcc --offload-arch=gfx950 -O3 -o graph_rate graph_rate.hip 2>/dev/null; timeout 120 ./graph_rate ) > $O/graph_rate.txt 2>&1
for f in config_c1 config_c3 config_c4 config_c5 bench_comm_world1 bench_steps20; do python -c "
import json,sys
try:
    d=json.load(open('$O/$f.json')); print('$f', d['value'], d.get('latency'))
except Exception as e: print('$f FAILED', e)"; done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --steps 120 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_prof.log 2>&1
cd $R
python tools/prof_summary.py $O/prof/bench_results.db > $O/kernel_stats.md 2>&1
python tools/prof_tail.py $O/prof/bench_results.db 1680 > $O/timed_frames_c2.txt 2>&1
rm -rf $O/prof
tools/gpu_pmc.sh $TAG > $O/pmc.log 2>&1; cp gpurun_out/pmc_$TAG.md gpurun_out/pmc_$TAG.json $O/ 2>/dev/null
TRACE=18 tools/gpu_stage.sh ${TAG}_c2 --near 0 --depths 1,3 > /dev/null 2>&1; cp gpurun_out/stage_${TAG}_c2.txt $O/
TAIL=600 TRACE=19 tools/gpu_stage.sh ${TAG}_c3 --splats 6291456 --cutout --near 0 --depths 1,3 --frames 120 --split 1 > /dev/null 2>&1; cp gpurun_out/stage_${TAG}_c3.txt $O/
TAIL=330 TRACE=18 tools/gpu_stage.sh ${TAG}_c5 --splats 20971520 --size 3840x2160 --frames 60 --depths 1,3 --near 0 > /dev/null 2>&1; cp gpurun_out/stage_${TAG}_c5.txt $O/
ls $O
