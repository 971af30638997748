#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r1.sh -- the scene that does not saturate (opacity / 10), share adapted and pinned at 15 % (every
# tile through both binning rounds), and the headline scene with the share pinned at 5 % (most tiles through both rounds)
timeout 300 python tools/stage_bench.py --opacity-div 10 --near 0 --depths 3 --batch 2 --frames 120 2>&1 | grep frames/s | cut -c1-210
timeout 300 python tools/stage_bench.py --opacity-div 10 --near 150 --depths 3 --batch 2 --frames 120 2>&1 | grep frames/s | cut -c1-210
timeout 300 python tools/stage_bench.py --near 50 --depths 3 --batch 2 --frames 240 2>&1 | grep frames/s | cut -c1-210
timeout 300 python tools/stage_bench.py --near 0 --depths 3 --batch 2 --frames 480 2>&1 | grep frames/s | cut -c1-210
timeout 300 python -m pytest tests -m gpu -q -x -k "two_round or unsat or near or retry" 2>&1 | tail -1
