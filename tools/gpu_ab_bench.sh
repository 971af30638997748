#!/bin/bash
# usage (GPU box, repo root): tools/gpu_ab_bench.sh -- the bench line (default and the driver's 20-step form) of library variants built
# by tools/build_variant.sh ("main" = the committed build) x frames in flight, interleaved twice (edit the lists)
C=aframe-gaussian-splatting_amd/csrc
for rep in 1 2; do for v in ${AB_VARIANTS:-main}; do for d in ${AB_DEPTHS:-3}; do L=""; [ $v != main ] && L=$C/libgs_variant_$v.so
for st in "--steps 20 --warmup 5" ""; do
GS_BENCH_DEPTH=$d GS_SPLAT_LIB=$L timeout 600 python bench.py --no-cpu-baseline --no-extras $st 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v','depth $d','$st',d['value'], d['occlusion_binning']['near_permille'], d['per_frame']['I_pairs'], d['config']['frames_redrawn_by_sync'], d['config'].get('steady_state_fps'))"
done; done; done; done
