#!/bin/bash
# usage (GPU box, repo root): tools/gpu_ab_bench.sh -- the bench line (default and the 20-step form) of library variants built by
# tools/build_variant.sh, interleaved twice (edit the list of names)
C=aframe-gaussian-splatting_amd/csrc
for rep in 1 2; do for v in main fl120 fl115; do L=""; [ $v != main ] && L=$C/libgs_variant_$v.so
for st in "" "--steps 20"; do
GS_SPLAT_LIB=$L timeout 600 python bench.py --no-cpu-baseline --no-extras $st 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v','$st',d['value'], d['occlusion_binning'], d['per_frame']['I_pairs'], d['config']['frames_redrawn_by_sync'], d['config'].get('steady_state_fps'))"
done; done; done
