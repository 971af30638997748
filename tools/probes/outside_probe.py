#!/usr/bin/env python3
"""round 5 diagnostic: bench.py's cold_orbit + outside_cloud sequence on one context, per sync interval"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("aframe-gaussian-splatting_amd.capi")
synth = importlib.import_module("aframe-gaussian-splatting_amd.synth")
W, H = 1920, 1080
rows = synth.make_splat_rows(synth.N_TRAIN)
def poses(fn, off):
    cams = [fn(W, H, off + 3.0 * i, capi=capi) for i in range(120)]
    return cams, [capi.make_params(c["gs_mv"], c["gs_proj"], W, H, focal_=c["focal"]) for c in cams]
with capi.Context(0) as c:
    c.push_splat(rows)
    c.set_option(capi.OPT_FRAME_BATCH, 2)
    for name, fn, off, every, laps in (("inside", synth.index_html_camera, 1.5, 24, 2), ("outside", synth.outside_cloud_camera, 0.0, 24, 1), ("outside48", synth.outside_cloud_camera, 0.0, 48, 2)):
        cams, ps = poses(fn, off)
        for lapno in range(laps):
            tl = time.perf_counter(); t0 = tl; marks = []
            for i in range(120):
                c.sort(cams[i]["view"], None, want_indices=False)
                ps[i].flags = capi.RENDER_ASYNC
                c.render_device(ps[i], None)
                if i % every == every - 1 or i == 119:
                    ta = time.perf_counter(); c.sync(); tb = time.perf_counter()
                    s = c.stats()
                    marks.append("%.2f+%.2fms share %d need %s redrawn %d" % ((ta - t0) * 1e3, (tb - ta) * 1e3, s["near_permille"], s["need_splats"], s["retried_frames"]))
                    t0 = time.perf_counter()
            print(name, "lap", lapno, "%.0f fps" % (120 / (time.perf_counter() - tl)), " | ".join(marks))
