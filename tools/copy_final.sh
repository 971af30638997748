#!/bin/bash
# usage (repo root, after tools/gpu_final.sh <tag> has been merged back): tools/copy_final.sh <tag>  -- gpurun_out/final_<tag>/* -> profiles/<tag>_*
T=${1:-r05}; F=gpurun_out/final_$T; P=profiles
cp $F/bench.json $P/${T}_final_bench.json
cp $F/bench_steps20_1.json $P/${T}_bench_steps20.json; cp $F/bench_steps20_2.json $P/${T}_bench_steps20_run2.json; cp $F/bench_steps20_3.json $P/${T}_bench_steps20_run3.json
cp $F/bench_pair_records.json $P/${T}_bench_pair_records.json
cp $F/bench_whole_sorts.json $P/${T}_bench_whole_sorts.json; cp $F/bench_whole_sorts_steps20.json $P/${T}_bench_whole_sorts_steps20.json
for c in c1 c3 c4 c5 r_outside r_unsat; do cp $F/config_$c.json $P/${T}_config_$c.json; done
cp $F/bench_comm_world1.json $P/${T}_bench_comm_world1.json
for n in 1 2 8; do cp $F/single_process_device_$n.json $P/${T}_single_process_device_$n.json; cp $F/single_process_host_$n.json $P/${T}_single_process_host_$n.json; done
cp $F/pmc_counters.md $P/${T}_pmc_counters.md; cp $F/pmc_counters.json $P/pmc_counters.json
cp $F/kernel_stats.md $P/${T}_final_kernel_stats.md; cp $F/timed_frames_c2.txt $P/${T}_final_timed_frames_c2.txt
cp $F/stage_c2.txt $P/${T}_stage_c2.txt; cp $F/stage_c3.txt $P/${T}_stage_c3.txt; cp $F/stage_c5.txt $P/${T}_stage_c5.txt
cp $F/stage_c2_outside_cloud.txt $P/${T}_stage_c2_outside_cloud.txt; cp $F/stage_c5_outside_cloud.txt $P/${T}_stage_c5_outside_cloud.txt; cp $F/stage_unsaturated.txt $P/${T}_stage_unsaturated.txt
cp $F/overlap_c2.txt $P/${T}_overlap_c2.txt; cp $F/pixel_parity.jsonl $P/${T}_pixel_parity.jsonl; cp $F/ply_load.txt $P/${T}_ply_load.txt
cp $F/js_visible_fps.txt $P/${T}_js_visible_fps.txt; cp $F/stress_and_tsan.txt $P/${T}_stress_and_tsan.txt; cp $F/pcie_probe.txt $P/${T}_pcie_probe.txt
tail -60 $F/pytest.log > $P/${T}_pytest_gpu.log
python - <<'PY'
import json,hashlib,os
pmc=json.load(open("profiles/pmc_counters.json")); h=hashlib.sha1()
c="aframe-gaussian-splatting_amd/csrc"
for f in sorted(os.listdir(c)):
    if f.endswith((".hip",".h",".cpp")): h.update(open(os.path.join(c,f),"rb").read())
print("pmc_counters.json matches csrc:", pmc["_csrc_sha1"]==h.hexdigest())
PY
