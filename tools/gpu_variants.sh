#!/bin/bash
# usage (GPU box, repo root): tools/gpu_variants.sh <stage_bench args> -- k_lists / k_emit_runs / k_project alone (depth 1) for every libgs_variant_*.so next to the library
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/variants; mkdir -p $O; cd $R
for lib in "" $(ls aframe-gaussian-splatting_amd/csrc/libgs_variant_*.so 2>/dev/null); do
  tag=$(basename "${lib:-base}" .so)
  ( cd /tmp && export TMPDIR=/tmp && GS_SPLAT_LIB=${lib:+$R/$lib} timeout 300 rocprofv3 --kernel-trace -d $O/p_$tag -o st -- python $R/tools/stage_bench.py --depths 1 --frames 60 "$@" > $O/$tag.log 2>&1 )
  echo "== $tag: $(grep 'depth 1' $O/$tag.log | cut -c1-150)"
  python tools/prof_tail.py $O/p_$tag/st_results.db 600 2>/dev/null | grep -E "k_lists<0>|k_emit_runs<0>|k_project<0|k_row_scan<0>|k_blend<false, 0" | cut -c1-100
  rm -rf $O/p_$tag
done
