#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r4g.sh -- the blend's inner loop with the LDS address in a vector register and the colour record's
# fourth word selected by op_sel: pixel parity (oracle + GL goldens + scene + two rounds), then A/B against the previous build
# (csrc/libgs_variant_head.so, tools/build_variant.sh) on the pipelined loops of C2 / unsaturated / C5
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gl_pin.py -m gpu -q -x -k "not c5_twenty and not six_million and not c3_full" 2>&1 | grep -E "passed|failed|Error" | tail -3
AB_EXTRA="--near 0 --depths 3 --batch 2 --opacity-div 10 --frames 120" tools/gpu_ab_libs.sh r4g main head > /dev/null 2>&1
grep -A1 "==" gpurun_out/ab_r4g.txt | grep -v "^--" | paste - - | cut -c1-220
for v in main head main head; do
  L=""; [ $v != main ] && L=$R/aframe-gaussian-splatting_amd/csrc/libgs_variant_$v.so
  GS_SPLAT_LIB=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v steps20', d['value'], d['config'].get('steady_state_fps'))"
  GS_SPLAT_LIB=$L timeout 600 python bench.py --splats 20971520 --size 3840x2160 --steps 120 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v c5', d['value'])"
done
