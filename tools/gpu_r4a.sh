#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r4a.sh -- round 4, first contact of the span-list binning: its parity test, the GPU tier,
# bench A/B against the pair records, depth-1 kernel traces of both
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4a; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "span_lists" > $O/t_span.log 2>&1; echo "span test rc=$?"; tail -15 $O/t_span.log
if [ "$QUICK" != "1" ]; then
timeout 1500 python -m pytest tests -m gpu -q -x > $O/t_all.log 2>&1; echo "gpu tier rc=$?"; tail -5 $O/t_all.log
fi
for b in 0 1 0 1; do
  GS_BENCH_BINNING=$b timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('binning $b steps20', d['value'], d['config'].get('steady_state_fps'), d.get('per_frame'))"
done
for b in 0 1; do
  GS_BENCH_BINNING=$b timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('binning $b default', d['value'], d.get('per_frame'))"
done
for b in 0 1; do
  TRACE=20 tools/gpu_stage.sh r4a_c2_b$b --near 0 --depths 1,3 --binning $b > /dev/null 2>&1; cp gpurun_out/stage_r4a_c2_b$b.txt $O/
  head -4 $O/stage_r4a_c2_b$b.txt
done
TRACE=20 tools/gpu_stage.sh r4a_unsat --near 0 --depths 1,3 --opacity-div 10 --frames 120 > /dev/null 2>&1; cp gpurun_out/stage_r4a_unsat.txt $O/; head -4 $O/stage_r4a_unsat.txt
tools/gpu_stage.sh r4a_unsat_b1 --near 0 --depths 3 --opacity-div 10 --frames 120 --binning 1 > /dev/null 2>&1; head -3 gpurun_out/stage_r4a_unsat_b1.txt
TAIL=330 TRACE=20 tools/gpu_stage.sh r4a_c5 --splats 20971520 --size 3840x2160 --frames 60 --depths 1,3 --near 0 > /dev/null 2>&1; cp gpurun_out/stage_r4a_c5.txt $O/; head -4 $O/stage_r4a_c5.txt
