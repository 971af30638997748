"""Is the asynchronous frame loop host-bound?  Time to ENQUEUE K frames vs time until they have all completed, for
pipeline depths 1..4, with and without the HIP-event profiling."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
capi = importlib.import_module("aframe-gaussian-splatting_amd.capi"); synth = importlib.import_module("aframe-gaussian-splatting_amd.synth")
rows = synth.make_splat_rows(synth.N_TRAIN)
W, H, K = 1920, 1080, 360
cams = [synth.index_html_camera(W, H, 3.0 * i, capi=capi) for i in range(120)]
params = [capi.make_params(c["gs_mv"], c["gs_proj"], W, H, focal_=c["focal"]) for c in cams]
ctx = capi.Context(0); ctx.push_splat(rows)
def go(n):
    t0 = time.perf_counter()
    for i in range(n):
        k = i % 120
        ctx.sort(cams[k]["view"], None, want_indices=False); params[k].flags = capi.RENDER_ASYNC; ctx.render_device(params[k], None)
    t1 = time.perf_counter()
    try: ctx.sync()
    except capi.GsError as e:
        if e.code != capi.E_RETRY: raise
    return t1 - t0, time.perf_counter() - t0
for k in range(0, 120, 2):
    ctx.sort(cams[k]["view"], None, want_indices=False); params[k].flags = 0; ctx.render_device(params[k], None)
for prof in (0, 1):
    for depth in (1, 2, 3, 4):
        ctx.set_option(capi.OPT_PIPELINE_DEPTH, depth); ctx.set_option(capi.OPT_PROFILE, 0); ctx.set_option(capi.OPT_PROFILE, prof)
        go(60); go(60)
        enq, tot = go(K)
        print("profile %d depth %d: enqueue %.1f us/frame, complete %.1f us/frame -> %.0f frames/s" % (prof, depth, enq / K * 1e6, tot / K * 1e6, K / tot))
