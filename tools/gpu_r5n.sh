#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r5n.sh -- per-queue dispatch timeline of the driver's 20-step region (tools/region_probe.py, last repetition), round 4's library and this round's
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp && export TMPDIR=/tmp
for w in old new; do
  [ $w = old ] && export GS_SPLAT_LIB=$R/aframe-gaussian-splatting_amd/csrc/libgs_variant_r04.so || unset GS_SPLAT_LIB
  timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/p20$w -o b -- python $R/tools/region_probe.py --profile ${PROBE_ARGS} > $R/gpurun_out/p20$w.log 2>&1
  (cd $R; python tools/prof_lanes.py gpurun_out/p20$w/b_results.db 230 > gpurun_out/p20_timeline_$w.txt 2>&1; rm -rf gpurun_out/p20$w; grep enqueue gpurun_out/p20$w.log)
done
