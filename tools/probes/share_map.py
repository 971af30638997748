#!/usr/bin/env python3
"""Measurement aid: which tiles need how much of the order?  For a few poses of the headline scene the frame is drawn with the
share of splats binned first pinned at 1, 2, 4, 8, 12, 16, 24 % and the mask of tiles that share left unsaturated is downloaded:
the share a tile needs is the first one that saturates it.  Printed: the distribution over tiles, and per screen region (a 4 x 4
grid) the largest share any of its tiles needs -- what a per-region share would have to be against the one global share."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
capi = importlib.import_module("aframe-gaussian-splatting_amd.capi"); synth = importlib.import_module("aframe-gaussian-splatting_amd.synth")
W, H = 1920, 1080
tx, ty = (W + 15) // 16, (H + 15) // 16
mw = (tx + 31) // 32
rows = synth.make_splat_rows(synth.N_TRAIN)
shares = [10, 20, 40, 80, 120, 160, 240, 400]
with capi.Context(0) as c:
    c.push_splat(rows)
    for yaw in (21.0, 120.0, 200.0, 300.0):
        cam = synth.index_html_camera(W, H, yaw, capi=capi)
        prm = capi.make_params(cam["gs_mv"], cam["gs_proj"], W, H, focal_=cam["focal"])
        need = np.full((ty, tx), 1000, np.int32)
        pairs = {}
        for s in reversed(shares):
            c.set_option(capi.OPT_NEAR_PERMILLE, s)
            c.sort(cam["view"], want_indices=False); c.render_device(prm, None)
            st = c.stats()
            m = c.download(capi.BUF_UNSAT_MASK, ty, np.uint32, mw)
            bits = ((m[:, :, None] >> np.arange(32, dtype=np.uint32)[None, None, :]) & 1).reshape(ty, mw * 32)[:, :tx].astype(bool)
            need[~bits] = s                                          # saturated with this share (shares descend: the smallest wins)
            pairs[s] = st["n_pairs"]
        c.set_option(capi.OPT_NEAR_PERMILLE, 1000)
        c.sort(cam["view"], want_indices=False); c.render_device(prm, None)
        allp = c.stats()["n_pairs"]
        hist = {s: int((need == s).sum()) for s in shares + [1000]}
        reg = np.zeros((4, 4), np.int32)
        for ry in range(4):
            for rx in range(4):
                reg[ry, rx] = need[ry * ty // 4:(ry + 1) * ty // 4, rx * tx // 4:(rx + 1) * tx // 4].max()
        rows8 = [int(need[r * ty // 8:(r + 1) * ty // 8].max()) for r in range(8)]
        print("yaw %5.1f: tiles by the share (permille) that saturates them: %s | pairs binned at that share: %s (all: %d)" % (yaw, hist, pairs, allp))
        print("           largest need per region (4 x 4):", reg.tolist(), "| per band of tile rows (8):", rows8)
