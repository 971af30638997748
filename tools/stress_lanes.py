"""Stress: random interleaving of asynchronous frames, synchronous frames, syncs, stat reads, option changes and pushes on
one context with pipeline lanes and enqueue threads; every synchronous frame is checked against a fresh context."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
capi = importlib.import_module("aframe-gaussian-splatting_amd.capi"); synth = importlib.import_module("aframe-gaussian-splatting_amd.synth")
g = np.random.Generator(np.random.PCG64(int(sys.argv[1]) if len(sys.argv) > 1 else 1))
NEAR = os.environ.get("STRESS_NEAR") == "1"                # a dense scene + GS_OPT_SORT_NEAR = 2: near-only sorts and their fall-backs
BIG = os.environ.get("STRESS_BIG") == "1"                  # 4 M splats, pixels stopping at T < 1/4: near-only sorts through the chunk stashes and through the
                                                           # depth pass' own candidate stash (the hint, its misses, the back-off), an orbit with jumps
if BIG:
    NEAR = True
rows = (synth.make_splat_rows(1 << 22, seed=synth.SEED_BASE + 11) if BIG else synth.make_splat_rows(synth.N_TRAIN) if NEAR else synth.make_splat_rows(60000, seed=5)).reshape(-1, 32)
W, H = (1280, 720) if BIG else (640, 360)
cams = [synth.index_html_camera(W, H, 3.0 * i, capi=capi) for i in range(120)] if BIG else [synth.index_html_camera(W, H, 15.0 * i, capi=capi) for i in range(24)]
P = lambda cam, **kw: capi.make_params(cam["gs_mv"], cam["gs_proj"], W, H, focal_=cam["focal"], **kw)
ref = capi.Context(0); ref.set_option(capi.OPT_PIPELINE_DEPTH, 1)
c = capi.Context(0)
c.set_option(capi.OPT_FRAME_BATCH, 2)                       # frames pair from the start; toggled at random below
n = N0 = (1 << 22) if BIG else 700000 if NEAR else 20000
if NEAR:
    c.set_option(capi.OPT_SORT_NEAR, 2); ref.set_option(capi.OPT_SORT_NEAR, 0)
if BIG:
    c.set_option(capi.OPT_TERMINATION, 4); ref.set_option(capi.OPT_TERMINATION, 4)
kk = 0
c.push_splat(rows[:n]); ref.push_splat(rows[:n])
near_sorts = 0
t0 = time.time(); ops = 0; checked = 0
while time.time() - t0 < float(os.environ.get("STRESS_SECONDS", "20")):
    r = g.random(); k = int(g.integers(0, len(cams))); ops += 1
    if BIG:                                                  # mostly the next pose of the orbit (the hint follows), a jump one time in twelve
        kk = k if g.random() < 0.08 else (kk + 1) % len(cams); k = kk
        if r >= 0.86: r = 0.0 if g.random() < 0.9 else r     # (fewer option changes: the share has to settle for near-only sorts to start)
    try:
        if r < 0.62:
            c.sort(cams[k]["view"], None, want_indices=False); c.render_device(P(cams[k], flags=capi.RENDER_ASYNC), None)
        elif r < 0.68:                                       # (round 6) the frame sorted for its frustum: gs_sort_for over the whole frame, queued
            c.sort_for(cams[k]["view"], None, P(cams[k]), want_indices=False); c.render_device(P(cams[k], flags=capi.RENDER_ASYNC), None)
        elif r < 0.70:                                       # (round 6) a posted sort: draws meanwhile use the last completed order (index.js:201-207)
            k0 = int(g.integers(0, len(cams)))
            c.sort(cams[k0]["view"], None, want_indices=False)                   # the completed order: pose k0's
            c.sort_begin(cams[k]["view"])
            stale = c.render(P(cams[k]))
            if g.random() < 0.5:
                c.render_device(P(cams[k], flags=capi.RENDER_ASYNC), None)       # a queued frame next to the posted sort
            got = c.sort_poll(wait=True)
            fresh = c.render(P(cams[k]))
            ref.sort(cams[k0]["view"], None, want_indices=False); rstale = ref.render(P(cams[k]))
            ridx = ref.sort(cams[k]["view"]); rfresh = ref.render(P(cams[k]))
            assert np.array_equal(got, ridx) and np.array_equal(stale, rstale) and np.array_equal(fresh, rfresh), "posted sort mismatch at op %d" % ops
            checked += 1
        elif r < 0.80:
            idx = c.sort(cams[k]["view"]); img = c.render(P(cams[k]))
            ridx = ref.sort(cams[k]["view"]); rimg = ref.render(P(cams[k]))
            assert np.array_equal(idx, ridx) and np.array_equal(img, rimg), "mismatch at op %d" % ops
            checked += 1
        elif r < 0.83:                                       # a synchronous frame on a sort that hands nothing back (possibly near-only)
            c.sort(cams[k]["view"], None, want_indices=False); img = c.render(P(cams[k]))
            st = c.stats(); near_sorts += 1 if 0 < st["sort_records"] < st["n_sorted"] else 0
            ref.sort(cams[k]["view"], None, want_indices=False); rimg = ref.render(P(cams[k]))
            assert np.array_equal(img, rimg), "mismatch (no indices) at op %d" % ops
            checked += 1
        elif r < 0.86:
            try: c.sync()
            except capi.GsError as e:
                if e.code != capi.E_RETRY: raise
        elif r < 0.90: c.stats()
        elif r < 0.93: c.set_option(capi.OPT_PIPELINE_DEPTH, int(g.integers(1, 5)))
        elif r < 0.95: c.set_option(capi.OPT_ENQUEUE_THREADS, int(g.integers(0, 2)))
        elif r < 0.96: c.set_option(capi.OPT_PROFILE, int(g.integers(0, 3)))
        elif r < 0.966: c.set_option(capi.OPT_FRAME_BATCH, int(g.integers(1, 3)))
        elif r < 0.968: c.set_option(capi.OPT_BINNING, int(g.integers(0, 2)))     # span lists <-> pair records (the reference context keeps the default)
        elif r < 0.97: c.set_option(capi.OPT_SUBTILE, int(g.integers(0, 3)))     # (round 6) sub-tile lists off / automatic / always: same pixels
        elif r < 0.972 and NEAR: c.set_option(capi.OPT_SORT_NEAR, int(g.integers(0, 3)) or 2)
        elif r < 0.985 and n < rows.shape[0]:
            m = min(rows.shape[0], n + int(g.integers(1, 9000)))
            c.push_splat(rows[n:m]); ref.push_splat(rows[n:m]); n = m
        elif r < 0.992: c.frame_stream()
        else:
            c.clear(); ref.clear(); n = N0; c.push_splat(rows[:n]); ref.push_splat(rows[:n])
    except capi.GsError as e:
        if e.code != capi.E_RETRY: raise
try: c.sync()
except capi.GsError: pass
st = c.stats()
c.close(); ref.close()
print("stress ok: %d operations, %d checked frames (%d on near-only sorts), %d splats%s" % (
    ops, checked, near_sorts, n, "; %d sorts from the depth pass' own stash, %d missed, %d frames drawn again" % (st["spec_sorts"], st["spec_misses"], st["retried_frames"]) if BIG else ""))
