#!/usr/bin/env python3
"""Development aid: per-tile list length and entries evaluated before saturation (GS_OPT_RECORD_STAGED = 2) for one frame."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
capi = importlib.import_module("aframe-gaussian-splatting_amd.capi"); synth = importlib.import_module("aframe-gaussian-splatting_amd.synth")
W, H = 1920, 1080
near = int(sys.argv[1]) if len(sys.argv) > 1 else 160
rows = synth.make_splat_rows(synth.N_TRAIN)
ctx = capi.Context(0)
ctx.push_splat(rows.reshape(-1, 32))
ctx.set_option(capi.OPT_NEAR_PERMILLE, near)
ctx.set_option(capi.OPT_RECORD_STAGED, 2)
ntl = ((W + 15) // 16) * ((H + 15) // 16)
for k in (0, 40, 80):
    cam = synth.index_html_camera(W, H, 3.0 * k, capi=capi)
    ctx.sort(cam["view"], cam["cutout"], want_indices=False)
    ctx.render_device(capi.make_params(cam["gs_mv"], cam["gs_proj"], W, H, focal_=cam["focal"]), None)
    ts = ctx.download(capi.BUF_TILE_STATS, ntl, np.uint32, 2).astype(np.int64)
    ev, ln = ts[:, 0], ts[:, 1]
    print("frame", k, "tiles", ntl, "evaluated: mean %.1f p50 %d p90 %d p99 %d max %d | list: mean %.1f p99 %d max %d" %
          (ev.mean(), np.percentile(ev, 50), np.percentile(ev, 90), np.percentile(ev, 99), ev.max(), ln.mean(), np.percentile(ln, 99), ln.max()))
    order = np.argsort(-ev)[:10]
    print("   slowest tiles (evaluated, list length):", [(int(ev[i]), int(ln[i])) for i in order])
    for thr in (256, 384, 512, 768, 1024):
        sel = ln >= thr
        print("   list >= %4d: %5d tiles, evaluated max %d mean %.0f; the others' max evaluated %d" % (thr, sel.sum(), ev[sel].max() if sel.any() else 0, ev[sel].mean() if sel.any() else 0, ev[~sel].max()))
