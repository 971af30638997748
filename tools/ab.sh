#!/bin/bash
# usage: tools/ab.sh "<env assignments>" ...   -- one profiled bench per variant (e.g. different build flags or
# environment), prints the per-kernel averages
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for V in "$@"; do
  i=$((i+1))
  env $V timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ab_$i -o ab -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline > $R/gpurun_out/ab_$i.log 2>&1
  echo "=== variant $i: $V"
  grep -o '"value": [0-9.]*' $R/gpurun_out/ab_$i.log | head -1
  python $R/tools/prof_summary.py $R/gpurun_out/ab_$i/ab_results.db | grep -E "${AB_FILTER:-k_}" | awk -F'|' '{printf "   %-34s calls %6s avg %9s us\n", $2, $3, $5}'
  rm -rf $R/gpurun_out/ab_$i
done
