"""Ceiling with the host out of the way: T Python threads, each driving its own context (depth D) -- ctypes releases the
GIL inside the library, so the enqueue work of the threads runs in parallel."""
import importlib, os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
capi = importlib.import_module("aframe-gaussian-splatting_amd.capi"); synth = importlib.import_module("aframe-gaussian-splatting_amd.synth")
rows = synth.make_splat_rows(synth.N_TRAIN)
W, H, K = 1920, 1080, 360
cams = [synth.index_html_camera(W, H, 3.0 * i, capi=capi) for i in range(120)]
def make(depth):
    c = capi.Context(0); c.push_splat(rows); c.set_option(capi.OPT_PIPELINE_DEPTH, depth)
    ps = [capi.make_params(cm["gs_mv"], cm["gs_proj"], W, H, focal_=cm["focal"]) for cm in cams]
    for k in range(0, 120, 2):
        c.sort(cams[k]["view"], None, want_indices=False); ps[k].flags = 0; c.render_device(ps[k], None)
    return c, ps
def drive(c, ps, n):
    for i in range(n):
        k = i % 120
        c.sort(cams[k]["view"], None, want_indices=False); ps[k].flags = capi.RENDER_ASYNC; c.render_device(ps[k], None)
    try: c.sync()
    except capi.GsError as e:
        if e.code != capi.E_RETRY: raise
for T, D in ((1, 3), (2, 2), (3, 1), (3, 2), (2, 3)):
    ctxs = [make(D) for _ in range(T)]
    for c, ps in ctxs: drive(c, ps, 60); drive(c, ps, 60)
    th = [threading.Thread(target=drive, args=(c, ps, K)) for c, ps in ctxs]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print("threads %d x depth %d: %.0f frames/s" % (T, D, T * K / dt))
    for c, _ in ctxs: c.close()
