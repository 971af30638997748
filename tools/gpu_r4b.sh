#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r4b.sh -- span-list parity test, then the binning kernels alone (depth 1) for C2 / unsaturated / C5, with the timing variants
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4b; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "span_lists" > $O/t_span.log 2>&1; echo "span test rc=$?"; tail -3 $O/t_span.log
echo C2; tools/gpu_variants.sh --near 157
echo UNSAT; tools/gpu_variants.sh --near 731 --opacity-div 10
echo C5; tools/gpu_variants.sh --near 15 --splats 20971520 --size 3840x2160
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps20', d['value'], d['config'].get('steady_state_fps'))"; done
timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default', d['value'])"
