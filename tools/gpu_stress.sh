#!/bin/bash
# usage (GPU box, repo root): tools/gpu_stress.sh -- the random-program stress tools against the built library (20 s each)
for s in 11 12; do timeout 120 python tools/stress_lanes.py $s 2>&1 | tail -1; done
STRESS_NEAR=1 timeout 120 python tools/stress_lanes.py 13 2>&1 | tail -1
timeout 120 python tools/stress_ranks.py 21 2 2>&1 | tail -1
timeout 120 python tools/stress_ranks.py 22 3 2>&1 | tail -1
timeout 160 python tools/stress_ranks.py 23 8 2>&1 | tail -1
