"""Per-tile work of the blend kernel: list length vs entries actually staged before early termination."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
capi = importlib.import_module("aframe-gaussian-splatting_amd.capi"); synth = importlib.import_module("aframe-gaussian-splatting_amd.synth")
N = int(os.environ.get("GS_STATS_N", synth.N_TRAIN)); CUT = os.environ.get("GS_STATS_CUTOUT") == "1"
rows = synth.make_splat_rows(N)
with capi.Context(0) as ctx:
    ctx.push_splat(rows)
    ctx.set_option(capi.OPT_RECORD_STAGED, int(os.environ.get("GS_RECORD", "1"))); ctx.set_option(capi.OPT_NEAR_PERMILLE, 1000)
    cam = (synth.cutout_demo_camera if CUT else synth.index_html_camera)(1920, 1080, float(os.environ.get("GS_STATS_YAW", "0")), capi=capi)
    ctx.sort(cam["view"], cam["cutout"])
    ctx.render(capi.make_params(cam["gs_mv"], cam["gs_proj"], 1920, 1080, focal_=cam["focal"]))
    t = ctx.download(capi.BUF_TILE_STATS, 8160, np.uint32, 2)
steps, ln = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
if os.environ.get("GS_STATS_OUT"):
    np.save(os.environ["GS_STATS_OUT"], t)
print("tiles", len(ln), "pairs", ln.sum(), "staged", steps.sum(), "(%.1f%%)" % (100.0 * steps.sum() / ln.sum()))
for q in (50, 90, 99, 99.9, 100):
    print("  p%-5s list len %6d   staged %6d" % (q, np.percentile(ln, q), np.percentile(steps, q)))
print("  mean staged", steps.mean(), " sum/1024 SIMDs =", steps.sum() / 1024.0, " max =", steps.max())
