"""How much do independent kernel chains overlap on one MI355X?  Two contexts (own streams), same scene, frames enqueued
alternately vs one context alone.  Upper bound for pipelining the sort of frame k+1 under the render of frame k."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
capi = importlib.import_module("aframe-gaussian-splatting_amd.capi"); synth = importlib.import_module("aframe-gaussian-splatting_amd.synth")
rows = synth.make_splat_rows(synth.N_TRAIN)
W, H, K = 1920, 1080, 240
cams = [synth.index_html_camera(W, H, 3.0 * i, capi=capi) for i in range(120)]
params = [capi.make_params(c["gs_mv"], c["gs_proj"], W, H, focal_=c["focal"]) for c in cams]
def run(ctxs, frames):
    for c in ctxs:                                   # settle the adaptive share synchronously
        for k in range(0, 120, 4):
            c.sort(cams[k]["view"], None, want_indices=False); params[k].flags = 0; c.render_device(params[k], None)
    def go(n):
        for i in range(n):
            c = ctxs[i % len(ctxs)]; k = i % 120
            c.sort(cams[k]["view"], None, want_indices=False); params[k].flags = capi.RENDER_ASYNC; c.render_device(params[k], None)
        for c in ctxs:
            try: c.sync()
            except capi.GsError as e:
                if e.code != capi.E_RETRY: raise
    go(60)
    t = time.perf_counter(); go(frames); return frames / (time.perf_counter() - t)
a = capi.Context(0); a.push_splat(rows)
print("one context : %.0f frames/s" % run([a], K))
b = capi.Context(0); b.push_splat(rows)
print("two contexts: %.0f frames/s" % run([a, b], K))
c = capi.Context(0); c.push_splat(rows)
print("three       : %.0f frames/s" % run([a, b, c], K))
