#!/bin/bash
# usage (GPU box, repo root): [CONFIGS="c2 c1 c3 c5 outside"] tools/gpu_pmc.sh <tag>  -- separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE and, for
# c2, the VALU group) over the pipelined frame loop of each BASELINE configuration (tools/stage_bench.py --pmc-run: adaptive share, 3 lanes
# x 2 frames per launch; `outside`: the 1 M scene seen from outside the cloud, bench.py's outside_cloud) -> gpurun_out/pmc_<tag>.md / .json (copied to profiles/ as r05_pmc_counters.md / pmc_counters.json)
TAG=${1:-pmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=$R/gpurun_out/pmc_$TAG; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
VALU="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS"
for cfg in ${CONFIGS:-c2 c1 c3 c5 outside unsat}; do
  case $cfg in
    c2) A="--frames 240"; G="FETCH_SIZE WRITE_SIZE VALU";;
    c1) A="--size 1280x720 --frames 240"; G="FETCH_SIZE WRITE_SIZE";;
    c3) A="--splats 6291456 --seed 0x5EED0003 --cutout --split 1 --frames 120"; G="FETCH_SIZE WRITE_SIZE";;
    outside) A="--outside --frames 240"; G="FETCH_SIZE WRITE_SIZE VALU";;
    unsat) A="--opacity-div 10 --frames 120"; G="FETCH_SIZE WRITE_SIZE VALU";;
    c5) A="--splats 20971520 --size 3840x2160 --frames 96"; G="FETCH_SIZE WRITE_SIZE VALU";;
  esac
  SF=""; case $cfg in c2|c1|outside|unsat) SF="--sort-for";; esac    # (as bench.py: frames of up to 2 M splats are sorted for their frustum)
  for g in $G; do
    if [ $g = VALU ]; then C="$VALU"; else C=$g; fi
    timeout 900 rocprofv3 --kernel-trace --pmc $C -d $D/${cfg}_$g -o pmc -- python $R/tools/stage_bench.py --pmc-run --near 0 --depths 3 --batch 2 $SF $A > $D/${cfg}_$g.log 2>&1
    echo "$cfg $g rc=$? $(grep PMCRUN $D/${cfg}_$g.log | cut -c1-160)"
  done
done
cd $R
python tools/pmc_summary.py $D gpurun_out/pmc_$TAG.md gpurun_out/pmc_$TAG.json
find $D -name "*.db" -delete
head -30 gpurun_out/pmc_$TAG.md | cut -c1-220
