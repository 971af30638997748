#!/usr/bin/env python3
"""Development aid: distribution of tiles-per-splat in the first binning round (what k_emit expands) for the headline scene."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
capi = importlib.import_module("aframe-gaussian-splatting_amd.capi"); synth = importlib.import_module("aframe-gaussian-splatting_amd.synth")
W, H = 1920, 1080
near = int(sys.argv[1]) if len(sys.argv) > 1 else 164
rows = synth.make_splat_rows(synth.N_TRAIN)
ctx = capi.Context(0)
ctx.push_splat(rows.reshape(-1, 32))
ctx.set_option(capi.OPT_NEAR_PERMILLE, near)
for k in (0, 30, 60):
    cam = synth.index_html_camera(W, H, 3.0 * k, capi=capi)
    ctx.sort(cam["view"], cam["cutout"], want_indices=False)
    ctx.render_device(capi.make_params(cam["gs_mv"], cam["gs_proj"], W, H, focal_=cam["focal"]), None)
    s = ctx.stats()
    V = s["n_sorted"] if "n_sorted" in s else None
    tc = ctx.download(capi.BUF_TILE_COUNT, s.get("n_sorted", 0) or 800000, np.uint32, 1).ravel()
    nz = tc[tc > 0]
    print("frame", k, {kk: s[kk] for kk in s if kk in ("n_sorted", "n_visible", "n_pairs", "near_permille")}, "len", len(tc), "nonzero", len(nz), "sum", int(nz.sum()))
    edges = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192]
    h, _ = np.histogram(nz, bins=edges + [1 << 20])
    pairs, _ = np.histogram(nz, bins=edges + [1 << 20], weights=nz)
    for e, a, b in zip(edges, h, pairs):
        print("  >=%5d: %7d splats %9d pairs" % (e, a, int(b)))
    idx = np.nonzero(tc)[0]
    if len(idx):
        print("  positions of nonzero: min %d max %d; per 256-chunk nonzero count: mean %.1f max %d" % (idx.min(), idx.max(), np.bincount(idx // 256).mean(), np.bincount(idx // 256).max()))
        per_chunk = np.bincount(idx // 256, weights=tc[idx])
        print("  pairs per chunk: mean %.0f max %d; chunks %d" % (per_chunk.mean(), per_chunk.max(), len(per_chunk)))
