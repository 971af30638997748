#!/bin/bash
# usage (GPU box, repo root): tools/gpu_quick.sh -- the GPU test tier, one stress run, the driver form of the bench three times and the default once
timeout 400 python -m pytest tests -m gpu -q -x > gpurun_out/tq.log 2>&1; grep -E "passed|failed" gpurun_out/tq.log | tail -1
timeout 120 python tools/stress_lanes.py 31 2>&1 | tail -1
timeout 120 python tools/stress_ranks.py 32 3 2>&1 | tail -1
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps20', d['value'], d['config'].get('steady_state_fps'), d['config'].get('region_ms'))"; done
timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default', d['value'])"
