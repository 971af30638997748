#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r4h.sh -- the whole GPU tier on the re-shaped blend loops (k_blend: LDS address in a vector register, packed
# dy products, op_sel written out; k_blend_px: colour record (r, -alpha, g, b)), then C3 / C2 / C5 against the previous build (libgs_variant_head.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|Error" | tail -3
for v in main head main head; do
  L=""; [ $v != main ] && L=$R/aframe-gaussian-splatting_amd/csrc/libgs_variant_$v.so
  GS_SPLAT_LIB=$L timeout 600 python bench.py --splats 6291456 --cutout --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v c3', d['value'])"
  GS_SPLAT_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v c2', d['value'])"
  GS_SPLAT_LIB=$L timeout 600 python bench.py --splats 20971520 --size 3840x2160 --steps 120 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v c5', d['value'])"
done
