#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r5r.sh -- does a deeper warm-up of every new stream's queue (GS_WARM_QUEUE launches) replace the long pre-roll? (tools/region_probe.py: first repetition against the later ones)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for q in 1 32 128 512; do
  for i in 1 2 3; do echo "[warm $q]" $(GS_WARM_QUEUE=$q python tools/region_probe.py --profile --reps 3 | tail -1); done
done
