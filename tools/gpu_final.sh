#!/bin/bash
# usage (GPU box, repo root): tools/gpu_final.sh <tag>  -- everything profiles/ keeps for a round, from ONE build of the code:
# the GPU test tier, the counter passes (-> profiles/pmc_counters.json, which the bench lines then read), the bench lines of every
# BASELINE.json configuration, rocprofv3 kernel stats of the bench command, depth-1 kernel traces (headline pose, outside the cloud,
# unsaturated scene), the concurrency of the pipelined loop.   PART=profiles skips the test tier and the other configurations.
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final_$TAG; mkdir -p $O; cd $R
if [ "$PART" != "profiles" ]; then
rm -f gpurun_out/pixel_parity.jsonl gpurun_out/ply_load.txt
timeout 1500 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; grep -E "passed|failed" $O/pytest.log | tail -1
cp gpurun_out/pixel_parity.jsonl gpurun_out/js_visible_fps.txt gpurun_out/ply_load.txt $O/ 2>/dev/null
grep -E "^stress_|stress ok" $O/pytest.log > $O/stress_and_tsan.txt
fi
# counters first: the bench lines report them only from the sources they were taken from
CONFIGS="${CONFIGS:-c2 c1 c3 c5 outside unsat}" tools/gpu_pmc.sh $TAG > $O/pmc.log 2>&1; tail -3 $O/pmc.log | cut -c1-160
cp gpurun_out/pmc_$TAG.md $O/pmc_counters.md; cp gpurun_out/pmc_$TAG.json $O/pmc_counters.json; cp gpurun_out/pmc_$TAG.json profiles/pmc_counters.json
# the driver's form first, exactly as the driver runs it (all configurations, all extras, the CPU baseline), then twice without the extras
for i in 1 2 3; do timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 $([ $i != 1 ] && echo --no-cpu-baseline --no-extras --no-configs) > $O/bench_steps20_$i.json 2>$O/bench_steps20_$i.err; done
timeout 900 python bench.py --no-configs > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c 1-200 $O/bench.json
if [ "$PART" != "profiles" ]; then
timeout 600 python bench.py --config C1 --no-cpu-baseline --no-extras > $O/config_c1.json 2>/dev/null
timeout 900 python bench.py --config C3 --no-cpu-baseline --no-extras > $O/config_c3.json 2>/dev/null
timeout 600 python bench.py --config C4 --no-cpu-baseline > $O/config_c4.json 2>/dev/null
timeout 1200 python bench.py --config C5 --steps 120 --no-cpu-baseline > $O/config_c5.json 2>/dev/null
timeout 600 python bench.py --config R_outside --no-cpu-baseline --no-extras > $O/config_r_outside.json 2>/dev/null
timeout 600 python bench.py --config R_unsat --no-cpu-baseline --no-extras > $O/config_r_unsat.json 2>/dev/null
GS_BENCH_COMM=1 timeout 600 python bench.py --no-cpu-baseline --no-extras --no-configs > $O/bench_comm_world1.json 2>/dev/null
GS_BENCH_BINNING=1 timeout 600 python bench.py --no-cpu-baseline --no-extras --no-configs > $O/bench_pair_records.json 2>/dev/null
GS_BENCH_SORT_NEAR=0 timeout 600 python bench.py --no-cpu-baseline --no-extras --no-configs > $O/bench_whole_sorts.json 2>/dev/null
GS_BENCH_SORT_NEAR=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-configs > $O/bench_whole_sorts_steps20.json 2>/dev/null
for n in 1 2 8; do
  timeout 300 python bench.py --gpus $n --single-process > $O/single_process_device_$n.json 2>/dev/null
  timeout 300 python bench.py --gpus $n --single-process --host-direct > $O/single_process_host_$n.json 2>/dev/null
done
fi
for f in bench bench_steps20_1 bench_steps20_2 bench_steps20_3 config_c1 config_c3 config_c4 config_c5 config_r_outside config_r_unsat bench_comm_world1 bench_pair_records bench_whole_sorts bench_whole_sorts_steps20; do python -c "
import json,sys
try:
    d=json.load(open('$O/$f.json')); print('$f', d['value'], d.get('latency',{}).get('fps_depth1'), (d.get('roofline') or {}).get('traffic'), (d.get('frame_hbm') or {}).get('traffic'))
except Exception as e: print('$f FAILED', e)"; done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --steps 120 --warmup 5 --no-cpu-baseline --no-extras --no-configs > $O/bench_prof.log 2>&1
cd $R
python tools/prof_summary.py $O/prof/bench_results.db > $O/kernel_stats.md 2>&1
python tools/prof_tail.py $O/prof/bench_results.db 1440 > $O/timed_frames_c2.txt 2>&1
rm -rf $O/prof
( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $O/pov -o b -- python $R/tools/stage_bench.py --depths 3 --frames 3000 --batch 2 --near 0 > $O/pov.log 2>&1 ); python tools/prof_overlap.py $O/pov/b_results.db 2 0.3 0.8 > $O/overlap_c2.txt 2>&1; rm -rf $O/pov
tools/gpu.sh stage ${TAG}_c2 --near 0 --depths 1,3 > /dev/null 2>&1; cp gpurun_out/stage_${TAG}_c2.txt $O/stage_c2.txt
tools/gpu.sh stage ${TAG}_c2_outside --near 0 --depths 1,3 --outside > /dev/null 2>&1; cp gpurun_out/stage_${TAG}_c2_outside.txt $O/stage_c2_outside_cloud.txt
tools/gpu.sh stage ${TAG}_unsat --near 0 --depths 1,3 --opacity-div 10 --frames 120 > /dev/null 2>&1; cp gpurun_out/stage_${TAG}_unsat.txt $O/stage_unsaturated.txt
if [ "$PART" != "profiles" ]; then
TAIL=600 tools/gpu.sh stage ${TAG}_c3 --splats 6291456 --cutout --near 0 --depths 1,3 --frames 120 --split 1 > /dev/null 2>&1; cp gpurun_out/stage_${TAG}_c3.txt $O/stage_c3.txt
TAIL=330 tools/gpu.sh stage ${TAG}_c5 --splats 20971520 --size 3840x2160 --frames 60 --depths 1,3 --near 0 > /dev/null 2>&1; cp gpurun_out/stage_${TAG}_c5.txt $O/stage_c5.txt
TAIL=330 tools/gpu.sh stage ${TAG}_c5_outside --splats 20971520 --size 3840x2160 --frames 60 --depths 1,3 --near 0 --outside > /dev/null 2>&1; cp gpurun_out/stage_${TAG}_c5_outside.txt $O/stage_c5_outside_cloud.txt
fi
timeout 120 python tools/pcie_probe.py > $O/pcie_probe.txt 2>&1
ls $O
