#!/usr/bin/env python3
"""What share of the nearest splats does a pose NEED (round 5: GsControl::need_near against the truth)?  For the poses of bench.py's
driver region: the need the blend measures in one round over everything (stats need_splats), and -- by bisection over a pinned share
(GS_OPT_NEAR_PERMILLE, both rounds) -- the smallest share whose first round leaves no tile unsaturated."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("aframe-gaussian-splatting_amd.capi")
synth = importlib.import_module("aframe-gaussian-splatting_amd.synth")
BC = importlib.import_module("aframe-gaussian-splatting_amd.bench_configs")

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
cfg = BC.CONFIGS[name]
rows = BC.make_rows(cfg, synth)
cams, views, W, H = BC.poses(cfg, synth, capi)
N = cfg["splats"]
with capi.Context(0) as c:
    BC.push_rows(c, rows)
    for k in range(5, 25, 3):
        cam, p = cams[k], views[k][0]
        c.set_option(capi.OPT_NEAR_PERMILLE, 1000)
        c.sort(cam["view"], cam["cutout"], want_indices=False); p.flags = 0; c.render_device(p, None)
        s = c.stats()
        need = s["need_splats"]
        lo, hi = 1, 999
        while lo < hi:
            mid = (lo + hi) // 2
            c.set_option(capi.OPT_NEAR_PERMILLE, mid)
            c.sort(cam["view"], cam["cutout"], want_indices=False); c.render_device(p, None)
            if c.stats()["unsat_tiles"] == 0:
                hi = mid
            else:
                lo = mid + 1
        print("pose %3d: V %d  measured need %d splats = %.1f permille of N; smallest pinned share with no unsaturated tile: %d permille" % (
            k, s["n_sorted"], need, 1000.0 * need / N if need != 0xFFFFFFFF else -1, lo))
