#!/usr/bin/env python3
"""Development aid: frames/s of the frame loop for one configuration with the adaptive knobs PINNED, so two builds can be
compared (bench.py lets the library adapt the share of splats binned first, which moves between runs).

  tools/stage_bench.py [--splats N] [--size WxH] [--cutout] [--frames K] [--depths 1,3] [--near PERMILLE] [--sort-only]

Prints one line per pipeline depth: frames/s, and the HIP-event stage times of a profiled pass.  Run under
`rocprofv3 --kernel-trace --stats` + tools/prof_tail.py for per-kernel times (depth 1 = kernels alone on the GPU)."""
import argparse, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
capi = importlib.import_module("aframe-gaussian-splatting_amd.capi"); synth = importlib.import_module("aframe-gaussian-splatting_amd.synth")
ap = argparse.ArgumentParser()
ap.add_argument("--splats", type=int, default=synth.N_TRAIN)
ap.add_argument("--size", default="1920x1080")
ap.add_argument("--cutout", action="store_true")
ap.add_argument("--frames", type=int, default=240)
ap.add_argument("--seed", default=None, help="seed of the synthetic scene (bench_configs: C3 = 0x5EED0003; default: the generator's own)")
ap.add_argument("--depths", default="1,3")
ap.add_argument("--near", type=int, default=180, help="pinned share (permille) of the splats binned in the first round; 0 = adaptive")
ap.add_argument("--sort-only", action="store_true")
ap.add_argument("--split", type=int, default=0, help="GS_OPT_BLEND_SPLIT")
ap.add_argument("--term", type=int, default=0, help="GS_OPT_TERMINATION (1/eps)")
ap.add_argument("--batch", type=int, default=1, help="GS_OPT_FRAME_BATCH")
ap.add_argument("--sort-near", type=int, default=None, help="GS_OPT_SORT_NEAR (0 off, 1 auto = the library's default, 2 always)")
ap.add_argument("--opacity-div", type=int, default=1, help="divide every splat's opacity byte by this (10: the scene whose tiles do not saturate)")
ap.add_argument("--no-early-out", action="store_true", help="GS_RENDER_NO_EARLY_OUT: every fragment blended")
ap.add_argument("--strip", default=None, help="k/G: render only strip k of G tile-aligned column strips (what one of G GPUs does)")
ap.add_argument("--sort-for", action="store_true", help="gs_sort_for the strip (--strip) or the whole frame instead of the full gs_sort")
ap.add_argument("--outside", action="store_true", help="the camera OUTSIDE the cloud, 3 sigma from its centre (synth.outside_cloud_camera)")
ap.add_argument("--subtile", type=int, default=None, help="GS_OPT_SUBTILE (0 off, 1 auto = the library's default, 2 always)")
ap.add_argument("--opt", action="append", default=[], help="NAME=VALUE: any capi.OPT_<NAME> (repeatable)")
ap.add_argument("--binning", type=int, default=None, help="GS_OPT_BINNING (0 span lists, 1 pair records + radix passes)")
ap.add_argument("--pmc-run", action="store_true", help="the run rocprofv3 --pmc passes profile (tools/gpu_pmc.sh): settle the share with synchronous frames, "
                                                        "then ONLY queued frames of the orbit at the first depth; prints the frames queued and, untimed and "
                                                        "synchronously afterwards, the list entries the blend evaluates per frame (GS_OPT_RECORD_STAGED = 2)")
a = ap.parse_args()
W, H = (int(v) for v in a.size.lower().split("x"))
rows = synth.make_splat_rows_fast(a.splats) if a.splats >= (8 << 20) else (synth.make_splat_rows(a.splats, seed=int(a.seed, 0)) if a.seed else synth.make_splat_rows(a.splats))
if a.opacity_div > 1:
    rows = rows.reshape(-1, 32).copy(); rows[:, 27] = rows[:, 27] // a.opacity_div; rows = rows.reshape(-1)
pose = synth.cutout_demo_camera if a.cutout else (synth.outside_cloud_camera if a.outside else synth.index_html_camera)
cams = [pose(W, H, 3.0 * i, capi=capi) for i in range(120)]
x0, x1 = 0, W
if a.strip:
    k_, g_ = (int(v) for v in a.strip.split("/"))
    x0, x1 = [(p[1], p[2]) for p in capi.partition([W], g_) if p[3] == k_][0]
params = [capi.make_params(c["gs_mv"], c["gs_proj"], W, H, x0=x0, x1=x1, focal_=c["focal"]) for c in cams]
ctx = capi.Context(0)
r = rows.reshape(-1, 32)
for o in range(0, a.splats, 1 << 22):
    ctx.push_splat(r[o:o + (1 << 22)])
if a.binning is not None:
    ctx.set_option(capi.OPT_BINNING, a.binning)
if a.split:
    ctx.set_option(capi.OPT_BLEND_SPLIT, a.split)
if a.term:
    ctx.set_option(capi.OPT_TERMINATION, a.term)
if a.near:
    ctx.set_option(capi.OPT_NEAR_PERMILLE, a.near)
if a.batch != 1:
    ctx.set_option(capi.OPT_FRAME_BATCH, a.batch)
if a.sort_near is not None:
    ctx.set_option(capi.OPT_SORT_NEAR, a.sort_near)
if a.subtile is not None:
    ctx.set_option(capi.OPT_SUBTILE, a.subtile)
for kv in a.opt:
    ctx.set_option(getattr(capi, "OPT_" + kv.split("=")[0].upper()), int(kv.split("=")[1]))


def sort_k(k):
    if a.sort_for:
        ctx.sort_for(cams[k]["view"], cams[k]["cutout"], params[k], want_indices=False)
    else:
        ctx.sort(cams[k]["view"], cams[k]["cutout"], want_indices=False)


def go(n):
    t0 = time.perf_counter()
    for i in range(n):
        k = i % 120
        sort_k(k)
        if not a.sort_only:
            params[k].flags = capi.RENDER_ASYNC | (capi.RENDER_NO_EARLY_OUT if a.no_early_out else 0)
            ctx.render_device(params[k], None)
    try:
        ctx.sync()
    except capi.GsError as e:
        if e.code != capi.E_RETRY:
            raise
        return None
    return time.perf_counter() - t0


for k in range(0, 120, 4):                                   # buffers sized, share settled (if adaptive)
    sort_k(k)
    if not a.sort_only:
        params[k].flags = 0
        ctx.render_device(params[k], None)
if a.pmc_run:
    import numpy as np
    ctx.set_option(capi.OPT_PIPELINE_DEPTH, int(a.depths.split(",")[0]))
    for rep in range(4):                                     # the second binning round is switched off after 16 clean frames
        for k in range(0, 120, 3):
            sort_k(k); params[k].flags = 0; ctx.render_device(params[k], None)
    go(24)
    t = go(a.frames) or go(a.frames)
    s = ctx.stats()
    ctx.set_option(capi.OPT_RECORD_STAGED, 2)
    ntl = ((x1 - x0 + 15) // 16) * ((H + 15) // 16)
    ev = []
    for k in range(0, 120, 5):
        sort_k(k); params[k].flags = 0; ctx.render_device(params[k], None)
        ev.append(int(ctx.download(capi.BUF_TILE_STATS, ntl, np.uint32, 2)[:, 0].astype(np.int64).sum()))
    print("PMCRUN frames_queued=%d frames_per_launch=%d fps=%.1f entries_evaluated_per_frame=%.1f pairs_last_frame=%d visible_last_frame=%d near_permille=%d" % (
        24 + a.frames, a.batch, a.frames / t, float(np.mean(ev)), s["n_pairs"], s["n_visible"], s["near_permille"]), flush=True)
    ctx.close()
    sys.exit(0)
for depth in (int(v) for v in a.depths.split(",")):
    ctx.set_option(capi.OPT_PIPELINE_DEPTH, depth)
    ctx.set_option(capi.OPT_PROFILE, 0)
    for _ in range(3 if a.near == 0 else 2):                 # (adaptive share: the loop below should run in the settled state -- the share is what
        go(120 if a.near == 0 else 24)                       # the poses SEEN needed: whole laps, or the first new pose of the timed loop misses and
                                                             # gs_sync draws every frame queued since the last sync again, one by one)
    t = go(a.frames) or go(a.frames)
    ctx.set_option(capi.OPT_PROFILE, 1)
    go(24); go(min(a.frames, 120))
    s = ctx.stats()
    ctx.set_option(capi.OPT_PROFILE, 0)
    k = max(1, s["prof_frames"])
    print("N=%d %dx%d%s depth %d: %.0f frames/s (%.1f us/frame) | events: sort %.1f project %.1f bin %.1f blend %.1f us | V=%d Vp=%d I=%d near=%d sortrec=%d" % (
        a.splats, W, H, " cutout" if a.cutout else "", depth, a.frames / t, t / a.frames * 1e6, s["sum_ms_sort"] / k * 1e3,
        s["sum_ms_project"] / k * 1e3, s["sum_ms_bin"] / k * 1e3, s["sum_ms_blend"] / k * 1e3, s["n_sorted"], s["n_visible"], s["n_pairs"],
        s["near_permille"], s["sort_records"]), flush=True)
ctx.close()
