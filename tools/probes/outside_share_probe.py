#!/usr/bin/env python3
"""round 5: the cloud seen from outside (sky tiles never saturate) with the share of splats binned first PINNED: is one round over
everything (what the controller chooses when a tile never saturates) really the cheapest?"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("aframe-gaussian-splatting_amd.capi")
synth = importlib.import_module("aframe-gaussian-splatting_amd.synth")
W, H = 1920, 1080
rows = synth.make_splat_rows(synth.N_TRAIN)
div = int(sys.argv[1]) if len(sys.argv) > 1 else 1
if div > 1:
    r = rows.reshape(-1, 32).copy(); r[:, 27] = r[:, 27] // div; rows = r.reshape(-1)
fn = synth.index_html_camera if div > 1 else synth.outside_cloud_camera
cams = [fn(W, H, 3.0 * i, capi=capi) for i in range(120)]
ps = [capi.make_params(c["gs_mv"], c["gs_proj"], W, H, focal_=c["focal"]) for c in cams]
with capi.Context(0) as c:
    c.push_splat(rows)
    c.set_option(capi.OPT_FRAME_BATCH, 2)
    for pm in (0, 1000, 700, 500, 350, 250, 150, 80):
        c.set_option(capi.OPT_NEAR_PERMILLE, pm)
        best = 0
        for lap in range(3):
            t0 = time.perf_counter()
            for i in range(120):
                c.sort(cams[i]["view"], None, want_indices=False)
                ps[i].flags = capi.RENDER_ASYNC
                c.render_device(ps[i], None)
                if i % 48 == 47:
                    c.sync()
            c.sync()
            best = max(best, 120 / (time.perf_counter() - t0))
        s = c.stats()
        print("near_permille %4d: %.0f frames/s  (share now %d, unsat tiles %d of %d, pairs %d, visible %d)" % (pm, best, s["near_permille"], s["unsat_tiles"], s["n_tiles"], s["n_pairs"], s["n_visible"]))
