#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r5a.sh -- round 5, first contact: the as-benched parity test of every BASELINE configuration
# and the driver-form bench line of the refactored bench.py (one table of configurations)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5a; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_as_benched.py -m gpu -q -s -x > $O/as_benched.log 2>&1; echo "as_benched rc=$?"; tail -15 $O/as_benched.log | cut -c1-300
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench20.json 2> $O/bench20.err; echo "bench rc=$?"; cut -c1-400 $O/bench20.json; tail -3 $O/bench20.err
