#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r5j.sh -- the whole GPU tier, then C3 as its own bench run
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5j; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -30 $O/gpu_tests.log | cut -c1-300
timeout 1200 python bench.py --config C3 --no-cpu-baseline --no-extras --no-configs > $O/config_C3.json 2>$O/config_C3.err
python - <<PY
import json
d=json.load(open("$O/config_C3.json")); print("C3", d["value"], d["occlusion_binning"], d["per_frame"])
PY
