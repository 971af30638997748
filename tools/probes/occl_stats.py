"""How deep (in sorted order, from the near end) does each tile actually read?  Decides whether progressive
front-to-back binning would pay."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
capi = importlib.import_module("aframe-gaussian-splatting_amd.capi"); synth = importlib.import_module("aframe-gaussian-splatting_amd.synth")
rows = synth.make_splat_rows(synth.N_TRAIN)
with capi.Context(0) as ctx:
    ctx.push_splat(rows)
    for yaw in (0.0, 200.0):
        cam = synth.index_html_camera(1920, 1080, yaw, capi=capi)
        idx = ctx.sort(cam["view"]); V = idx.size
        ctx.render(capi.make_params(cam["gs_mv"], cam["gs_proj"], 1920, 1080, focal_=cam["focal"]))
        cnt = ctx.download(capi.BUF_TILE_COUNT, V, np.uint32, 1).reshape(-1).astype(np.int64)
        cum = np.cumsum(cnt[::-1])            # pairs carried by the nearest k splats
        tot = cum[-1]
        print("yaw", yaw, "V", V, "pairs", tot)
        for frac in (1 / 64, 1 / 32, 1 / 16, 1 / 8, 1 / 4, 1 / 2):
            k = int(V * frac)
            print("  nearest %6.2f%% of splats (%7d) carry %5.1f%% of the pairs" % (100 * frac, k, 100.0 * cum[k - 1] / tot))
