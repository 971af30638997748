#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r5p.sh -- HIP API calls next to the dispatches of the FIRST 20-step region after the pre-roll (tools/region_probe.py --reps 1)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp && export TMPDIR=/tmp
for w in ${LIBS:-new}; do
  [ $w = old ] && export GS_SPLAT_LIB=$R/aframe-gaussian-splatting_amd/csrc/libgs_variant_r04.so || unset GS_SPLAT_LIB
  timeout 600 rocprofv3 --kernel-trace --hip-trace -d $R/gpurun_out/p20a$w -o b -- python $R/tools/region_probe.py --profile --reps 1 > $R/gpurun_out/p20a$w.log 2>&1
  (cd $R; python tools/prof_api.py gpurun_out/p20a$w/b_results.db 1700 > gpurun_out/p20_api_$w.txt 2>&1; rm -rf gpurun_out/p20a$w; grep enqueue gpurun_out/p20a$w.log)
done
