#!/usr/bin/env python3
"""round 5 diagnostic: the cut-out configuration (C3) with the first-round share pinned to several values against the adaptive one:
frames/s of a 60-frame queued region and the per-stage times.  usage: tools/c3_probe.py [C3|C1|...] [shares...]"""
import importlib, os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("aframe-gaussian-splatting_amd.capi")
synth = importlib.import_module("aframe-gaussian-splatting_amd.synth")
BC = importlib.import_module("aframe-gaussian-splatting_amd.bench_configs")
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
shares = [int(a) for a in sys.argv[2:]] or [0, 44, 100, 1000]
cfg = BC.CONFIGS[name]
rows = BC.make_rows(cfg, synth)
cams, views, W, H = BC.poses(cfg, synth, capi)
with capi.Context(0) as c:
    BC.push_rows(c, rows)
    BC.apply_options(c, capi, BC.options_for(cfg, env={}))
    def frame(i, flags=0):
        k = i % 120
        c.sort(cams[k]["view"], cams[k]["cutout"], want_indices=False); views[k][0].flags = flags; c.render_device(views[k][0], None)
    def sync():
        try:
            c.sync(); return False
        except capi.GsError as e:
            if e.code != capi.E_RETRY: raise
            return True
    seq, used = BC.region_frames(5, 60)
    for sh in shares:
        c.set_option(capi.OPT_NEAR_PERMILLE, sh)
        BC.preroll(frame, sync, used, 5, capi.RENDER_ASYNC)
        best = None
        for rep in range(3):
            sync(); gc.collect(); gc.disable()
            t0 = time.perf_counter()
            for i in range(60):
                frame(5 + i, capi.RENDER_ASYNC)
            again = sync(); dt = time.perf_counter() - t0; gc.enable()
            best = dt if best is None or dt < best else best
        c.set_option(capi.OPT_PROFILE, 1); sync()
        for i in range(60):
            frame(5 + i, capi.RENDER_ASYNC)
        sync(); s = c.stats(); c.set_option(capi.OPT_PROFILE, 0)
        k = max(1, s["prof_frames"]) * 2.0
        print(name, "share asked", sh, "used", s["near_permille"], "fps %.0f" % (60 / best), "again", again, "sort/proj/bin/blend us %.1f %.1f %.1f %.1f" % (
            s["sum_ms_sort"] / k * 1e3, s["sum_ms_project"] / k * 1e3, s["sum_ms_bin"] / k * 1e3, s["sum_ms_blend"] / k * 1e3),
            "V", s["n_sorted"], "Vp", s["n_visible"], "I", s["n_pairs"], "unsat tiles", s["unsat_tiles"], "need", s["need_splats"], "redrawn", s["retried_frames"])
