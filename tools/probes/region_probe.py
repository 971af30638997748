#!/usr/bin/env python3
"""round 5 diagnostic: the driver's 20-step region (bench.py --steps 20 --warmup 5) by hand: enqueue time and closing gs_sync separately,
ten repetitions (GS_SPLAT_LIB picks the library)"""
import importlib, os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("aframe-gaussian-splatting_amd.capi")
synth = importlib.import_module("aframe-gaussian-splatting_amd.synth")
BC = importlib.import_module("aframe-gaussian-splatting_amd.bench_configs")
cfg = BC.CONFIGS["C2"]
rows = BC.make_rows(cfg, synth)
cams, views, W, H = BC.poses(cfg, synth, capi)
with capi.Context(0) as c:
    BC.push_rows(c, rows)
    BC.apply_options(c, capi, BC.options_for(cfg, env={}))
    def frame(i, flags=0):
        k = i % 120
        c.sort(cams[k]["view"], None, want_indices=False); views[k][0].flags = flags; c.render_device(views[k][0], None)
    def sync():
        c.sync(); return False
    seq, used = BC.region_frames(5, 20)
    if "--count" in sys.argv:                                  # bench.py's fragment-counting renders
        for k in used:
            c.sort(cams[k]["view"], None, want_indices=False); views[k][0].flags = capi.RENDER_COUNT_FRAGS; c.render_device(views[k][0], None)
        for k in used[::2][:8]:
            c.sort(cams[k]["view"], None, want_indices=False); views[k][0].flags = capi.RENDER_COUNT_FRAGS | capi.RENDER_COUNT_EVALUATED; c.render_device(views[k][0], None)
    if "--staged" in sys.argv:                                 # ... and its staged-entries measurement
        import numpy as np
        c.set_option(capi.OPT_RECORD_STAGED, 1); c.set_option(capi.OPT_NEAR_PERMILLE, 1000)
        for k in used[::2][:8]:
            c.sort(cams[k]["view"], None, want_indices=False); views[k][0].flags = 0; c.render_device(views[k][0], None)
            c.download(capi.BUF_TILE_STATS, ((W + 15) // 16) * ((H + 15) // 16), np.uint32, 2)
        c.set_option(capi.OPT_RECORD_STAGED, 0); c.set_option(capi.OPT_NEAR_PERMILLE, 0)
    BC.preroll(frame, sync, used, 5, capi.RENDER_ASYNC)
    def arg(name, d=0):
        return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else d
    for j in range(arg("--extra-async")):                      # what a longer pre-roll may have warmed: the queued path ...
        frame(used[j % len(used)], capi.RENDER_ASYNC)
    sync()
    for j in range(arg("--extra-sync")):                       # ... the synchronous one (one pose) ...
        frame(used[0], 0)
    if arg("--sleep-ms"):                                      # ... or nothing but time
        time.sleep(arg("--sleep-ms") / 1e3)
    out = []
    for rep in range(int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 10):
        if "--profile" in sys.argv:
            c.set_option(capi.OPT_PROFILE, 0); c.set_option(capi.OPT_PROFILE, 3)
        if "--presync" in sys.argv:                             # what precedes bench.py's region: synchronous frames ...
            for k in used[:8]:
                frame(k, 0)
        if "--pre6" in sys.argv:                                # ... six queued ones, a sync ...
            for j in range(6):
                frame(used[j], capi.RENDER_ASYNC)
            sync()
        if "--pre5" in sys.argv:                                # ... and the five warm-up steps
            for j in range(5):
                frame(j, capi.RENDER_ASYNC)
        sync()
        if "--stats" in sys.argv:
            c.stats()
        gc.collect(); gc.disable()
        t0 = time.perf_counter()
        if "--calls" in sys.argv:                               # every call of the enqueuing thread timed on its own
            ts = [t0]
            for i in range(20):
                k = (5 + i) % 120
                c.sort(cams[k]["view"], None, want_indices=False); ts.append(time.perf_counter())
                views[k][0].flags = capi.RENDER_ASYNC; c.render_device(views[k][0], None); ts.append(time.perf_counter())
            print("rep", rep, "calls us:", " ".join("%.0f" % ((b - a) * 1e6) for a, b in zip(ts, ts[1:])))
        else:
            for i in range(20):
                frame(5 + i, capi.RENDER_ASYNC)
        t1 = time.perf_counter(); sync(); t2 = time.perf_counter(); gc.enable()
        out.append("%.0f+%.0f" % ((t1 - t0) * 1e6, (t2 - t1) * 1e6))
    s = c.stats()
    print(os.path.basename(os.environ.get("GS_SPLAT_LIB", "this round")), "enqueue+sync us:", " ".join(out), "share", s["near_permille"])
