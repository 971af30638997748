#!/usr/bin/env python3
"""Measurement aid: how evenly is the blend's work spread over the chip?  For a few poses of the headline scene the list entries
every tile's wavefront evaluates before it saturates are read back (GS_OPT_RECORD_STAGED = 2) and the kernel is replayed on paper:
1024 SIMDs, waves placed in dispatch order on the SIMD with a free slot (6 slots each), a wave's work = its staged batches and its
evaluated entries.  Printed: the distribution of the per-tile work, the sum over 1024 SIMDs (the kernel at perfect balance) against
the busiest SIMD in tile order and with the long tiles dispatched first."""
import heapq, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
capi = importlib.import_module("aframe-gaussian-splatting_amd.capi"); synth = importlib.import_module("aframe-gaussian-splatting_amd.synth")
W, H = 1920, 1080
tx, ty = (W + 15) // 16, (H + 15) // 16
rows = synth.make_splat_rows(synth.N_TRAIN)
NEAR = int(os.environ.get("BAL_NEAR", "164"))


def replay(work, slots=6, simds=1024):
    """waves are dispatched in order; each goes to the SIMD slot that frees first; a SIMD runs its resident waves round-robin, i.e.
    a wave's duration = its work x the number of waves resident (approximated by `slots` while the queue is not empty)"""
    free = [(0.0, s) for s in range(simds * slots)]
    heapq.heapify(free)
    end = 0.0
    for w in work:
        t, s = heapq.heappop(free)
        t2 = t + w * slots
        end = max(end, t2)
        heapq.heappush(free, (t2, s))
    return end


with capi.Context(0) as c:
    c.push_splat(rows)
    c.set_option(capi.OPT_RECORD_STAGED, 2)
    c.set_option(capi.OPT_NEAR_PERMILLE, NEAR)
    for yaw in (21.0, 120.0, 200.0, 300.0):
        cam = synth.index_html_camera(W, H, yaw, capi=capi)
        prm = capi.make_params(cam["gs_mv"], cam["gs_proj"], W, H, focal_=cam["focal"])
        c.sort(cam["view"], None, want_indices=False)
        c.render_device(prm, None)
        st = c.download(capi.BUF_TILE_STATS, tx * ty, np.uint32, 2).astype(np.int64)
        ev, ln = st[:, 0], st[:, 1]
        # cycles of one wave alone on its SIMD: ~180 issue cycles per evaluated entry, ~1500 per staged batch of 64 (two dependent gathers)
        batches = (np.minimum(ln, np.maximum(ev, 1)) + 63) // 64
        work = ev * 180.0 + batches * 1500.0 + 800.0
        tot = work.sum() / 1024.0
        q = np.percentile(ev, [50, 90, 99, 100]).astype(int)
        order_len = np.argsort(-ln, kind="stable")
        order_ev = np.argsort(-ev, kind="stable")
        print("yaw %5.1f: evaluated entries per tile: mean %.1f, p50 %d p90 %d p99 %d max %d | list length mean %.0f max %d | corr(len, evaluated) %.2f" % (
            yaw, ev.mean(), q[0], q[1], q[2], q[3], ln.mean(), ln.max(), float(np.corrcoef(ln, ev)[0, 1])))
        for name, w in (("tile order", work), ("longest lists first", work[order_len]), ("most evaluated first (oracle)", work[order_ev])):
            e = replay(list(w))
            print("    %-30s kernel %.1f us at 2.4 GHz (perfect balance %.1f us; the longest wave alone %.1f us)" % (name, e / 2400.0, tot / 2400.0, work.max() * 6 / 2400.0))
