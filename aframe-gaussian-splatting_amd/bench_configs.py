"""The five BASELINE.json configurations as `bench.py` runs them -- ONE table, read by `bench.py` (what is measured) and by
`tests/test_as_benched.py` (what is checked), so that the path a number is quoted on is the path a parity test has drawn.

  C1  train.splat-shaped 1 M splats, 1280x720                  (BASELINE.json configs[0]; index.html:13 pose)
  C2  the same scene, 1920x1080: the headline                  (configs[1])
  C3  bicycle.ply-shaped 6 M splats, 1920x1080, cutoutEntity   (configs[2]; cutout-demo.html:22-24 pose and box)
  C4  XR stereo 2 x (2064x2208 x xrPixelRatio 0.5)             (configs[3]; one shared head-camera sort, index.js:441)
  C5  20 M splats, 3840x2160                                   (configs[4])

Per configuration: the scene's recipe (generator + seed, SURVEY.md 8(d): seed = 0x5EED0000 + config -- C1 / C2 / C4 share the
1 M scene, which is also what the GL goldens of tests/golden were drawn from), the viewport, the pose family, and the
library options `bench.py` switches on for it.  Host-side plumbing only: nothing here is on the hot path.
"""
import os

ORBIT_FRAMES = 120          # the benchmark orbit: entity yaw 360 i / 120 degrees
LANES = 3                   # the library's default pipeline depth (GS_OPT_PIPELINE_DEPTH), what bench.py measures with
PUSH_ROWS = 1 << 22         # progressive ingest (index.js:279-298): rows per gs_push_splat
ASYNC_WARM = 6              # queued frames per lane before the warm-up steps (pre-roll, untimed): see preroll()
FRUSTUM_SORT = True         # frames whose order stays on the GPU are sorted with gs_sort_for over the WHOLE frame (bench.py, tests/test_as_benched.py):
                            # only the splats whose fragments can reach the viewport enter the order -- a sub-sequence of the reference's.
                            # Up to 2 M splats (frustum_sort(cfg)): +5.6 % at the headline, +2.5 % at 720p; at 20 M the reach test makes the
                            # depth pass -- the frame's longest kernel there -- a third slower (5 399 -> 3 634 frames/s): whole sorts


def frustum_sort(cfg):
    return bool(FRUSTUM_SORT and cfg["splats"] <= (2 << 20) and not cfg["xr"])

_SEED_BASE = 0x5EED0000

# option names are resolved against capi (OPT_*) when they are applied: this module imports nothing of the product
CONFIGS = {
    "C1": {"splats": 1 << 20, "rows": ("make_splat_rows", {}), "size": (1280, 720), "pose": "index", "xr": False,
           "options": {"OPT_FRAME_BATCH": 2}},
    "C2": {"splats": 1 << 20, "rows": ("make_splat_rows", {}), "size": (1920, 1080), "pose": "index", "xr": False,
           "options": {"OPT_FRAME_BATCH": 2}},
    # the cut-out scene fills a tenth of the screen: ~600 active tiles with lists of thousands of entries.  Tiles with a list are
    # blended by four wavefronts, one pixel per lane (GS_OPT_BLEND_SPLIT = 1: k_blend_px; images within the same 1 LSB tolerance)
    "C3": {"splats": 6 * (1 << 20), "rows": ("make_splat_rows", {"seed": _SEED_BASE + 3}), "size": (1920, 1080), "pose": "cutout", "xr": False,
           "options": {"OPT_FRAME_BATCH": 2, "OPT_BLEND_SPLIT": 1}},
    # both eyes on one GPU: frames go through gs_sort_gathered / gs_render_gathered (world 1), the two views of a frame share launches
    "C4": {"splats": 1 << 20, "rows": ("make_splat_rows", {}), "size": None, "pose": "xr", "xr": True,
           "options": {"OPT_FRAME_BATCH": 2}},
    # 20 M splats: near-only sorts (GS_OPT_SORT_NEAR's default from 4 M splats) through the depth pass' own candidate stash
    "C5": {"splats": 20 * (1 << 20), "rows": ("make_splat_rows_fast", {}), "size": (3840, 2160), "pose": "index", "xr": False,
           "options": {"OPT_FRAME_BATCH": 2}},
}

# Regimes of the headline scene that the headline pose does not show: measured, checked (tests/test_as_benched.py) and
# reported like the configurations.  Same 1 M splats, same viewport, same options.
#   R_outside  the camera OUTSIDE the cloud, 3 sigma from its centre, looking in (synth.outside_cloud_camera): 912 K visible splats of
#              3.5 tiles each, sky around them; nothing like the 16 K screen-filling splats of the pose inside the cloud
#   R_unsat    every opacity byte divided by ten at the headline pose: tiles need 5-10 x longer lists before T < 1/1024, most of the
#              order is binned and blended
REGIMES = {
    "R_outside": {"splats": 1 << 20, "rows": ("make_splat_rows", {}), "size": (1920, 1080), "pose": "outside", "xr": False,
                  "options": {"OPT_FRAME_BATCH": 2}},
    "R_unsat": {"splats": 1 << 20, "rows": ("make_splat_rows", {}), "opacity_div": 10, "size": (1920, 1080), "pose": "index", "xr": False,
                "options": {"OPT_FRAME_BATCH": 2}},
}
ALL = dict(CONFIGS, **REGIMES)

DESCRIPTION = {
    "R_outside": "the 1 M scene seen from outside the cloud (entity 7.5 units = 3 sigma in front of the camera) @1920x1080",
    "R_unsat": "the 1 M scene with every opacity / 10 (tiles do not saturate) @1920x1080, headline pose",
    "C1": "train.splat-shaped 1 M splats @1280x720 (configs[0])",
    "C2": "train.splat-shaped 1 M splats @1920x1080 (configs[1], the headline)",
    "C3": "bicycle.ply-shaped 6 M splats @1920x1080 + cutoutEntity box (configs[2])",
    "C4": "XR stereo 2 x 1032x1104, one shared head-camera sort, both eyes on this GPU (configs[3])",
    "C5": "20 M splats @3840x2160 on this one GPU (configs[4])",
}


def name_of(splats, size, cutout, xr):
    """The table's name for what bench.py's flags describe, or None (a shape of the caller's own)."""
    size = tuple(size) if size else None
    for name, c in CONFIGS.items():
        if bool(xr) != c["xr"]:
            continue
        if xr:
            if splats in (None, c["splats"]) and not cutout:
                return name
            continue
        if (splats or (1 << 20)) == c["splats"] and (size or (1920, 1080)) == c["size"] and bool(cutout) == (c["pose"] == "cutout"):
            return name
    return None


def custom(splats, size, cutout, xr):
    """A configuration of the caller's own shape (bench.py --splats / --size / --cutout), with the options its nearest table entry gets."""
    return {"splats": splats or (1 << 20), "rows": ("make_splat_rows_fast" if (splats or 0) >= (8 << 20) else "make_splat_rows", {}),
            "size": tuple(size) if size else (1920, 1080), "pose": "xr" if xr else ("cutout" if cutout else "index"), "xr": bool(xr),
            "options": {"OPT_FRAME_BATCH": 2, **({"OPT_BLEND_SPLIT": 1} if cutout else {})}}


def make_rows(cfg, synth, cache=None):
    fn, kw = cfg["rows"]
    rows = cache(fn, cfg["splats"], **kw) if cache is not None else getattr(synth, fn)(cfg["splats"], **kw)
    if cfg.get("opacity_div", 1) > 1:                            # (a copy: the cache's scene is shared)
        import numpy as np
        r = np.asarray(rows).reshape(-1, 32).copy()
        r[:, 27] = r[:, 27] // cfg["opacity_div"]
        rows = r.reshape(-1)
    return rows


def options_for(cfg, env=None, pieces_of_rank=1, gathered=False):
    """{capi option name: value} bench.py sets for cfg.  The GS_BENCH_* environment knobs are experiment overrides (A/B runs);
    a test passes env = {} and gets the table's own values."""
    env = os.environ if env is None else env
    o = dict(cfg["options"])
    if env.get("GS_BENCH_BATCH"):
        o["OPT_FRAME_BATCH"] = int(env["GS_BENCH_BATCH"])
    if gathered and pieces_of_rank not in (1, 2):
        o["OPT_FRAME_BATCH"] = 1                                  # (pairs: two frames of one piece each, or the two views of one frame)
    if env.get("GS_BENCH_SPLIT") is not None and env.get("GS_BENCH_SPLIT") != "":
        o["OPT_BLEND_SPLIT"] = int(env["GS_BENCH_SPLIT"])
    if env.get("GS_BENCH_BINNING"):
        o["OPT_BINNING"] = int(env["GS_BENCH_BINNING"])           # 1 = pair records + radix passes (rounds 1-3)
    if env.get("GS_BENCH_SORT_NEAR"):
        o["OPT_SORT_NEAR"] = int(env["GS_BENCH_SORT_NEAR"])       # 0 = whole sorts only (library default 1: near-only sorts where they pay)
    if env.get("GS_BENCH_DEPTH"):
        o["OPT_PIPELINE_DEPTH"] = int(env["GS_BENCH_DEPTH"])      # frames in flight (library default 3)
    if o.get("OPT_FRAME_BATCH") == 1:
        o.pop("OPT_FRAME_BATCH")
    if not o.get("OPT_BLEND_SPLIT"):
        o.pop("OPT_BLEND_SPLIT", None)
    if env.get("GS_BENCH_SORT_NEAR") == "0":
        o["OPT_SORT_NEAR"] = 0
    return o


def apply_options(ctx, capi, opts):
    for k in sorted(opts):
        ctx.set_option(getattr(capi, k), int(opts[k]))


def push_rows(ctx, rows):
    r32 = rows.reshape(-1, 32)
    for o in range(0, r32.shape[0], PUSH_ROWS):
        ctx.push_splat(r32[o:o + PUSH_ROWS])


def poses(cfg, synth, capi, frames=None):
    """The orbit of cfg: (cams, views, W, H).  cams[k]: the camera the SORT uses (the head camera for XR, index.js:441); views[k]:
    the gs_render_params of the frame's view(s) -- one, or the two eyes."""
    frames = range(ORBIT_FRAMES) if frames is None else frames
    if cfg["xr"]:
        rigs = {k: synth.xr_eye_cameras(360.0 * k / ORBIT_FRAMES, 0.5, capi=capi) for k in frames}
        any_rig = next(iter(rigs.values()))
        W, H = any_rig[0]["vw"], any_rig[0]["vh"]
        cams = {k: r[2] for k, r in rigs.items()}
        views = {k: [capi.make_params(e["gs_mv"], e["gs_proj"], W, H, focal_=e["focal"]) for e in r[:2]] for k, r in rigs.items()}
    else:
        W, H = cfg["size"]
        pose = {"cutout": synth.cutout_demo_camera, "outside": synth.outside_cloud_camera}.get(cfg["pose"], synth.index_html_camera)
        cams = {k: pose(W, H, 360.0 * k / ORBIT_FRAMES, capi=capi) for k in frames}
        views = {k: [capi.make_params(c["gs_mv"], c["gs_proj"], W, H, focal_=c["focal"])] for k, c in cams.items()}
    if isinstance(frames, range) and frames == range(ORBIT_FRAMES):
        cams = [cams[k] for k in frames]
        views = [views[k] for k in frames]
    return cams, views, W, H


def region_frames(warmup, steps):
    """orbit frames of the timed region, in the order they are drawn, and the sorted set of them"""
    seq = [(warmup + i) % ORBIT_FRAMES for i in range(steps)]
    return seq, sorted(set(seq))


def preroll(frame, sync, frames_used, warmup, async_flag, lanes=LANES, warmup_first=0):
    """What bench.py does before its timed region (untimed): ONE pass of synchronous frames over the region's own poses -- the library
    sets the share of splats it bins first from what the blend measures (one frame is enough; the window keeps the largest need of
    the poses it has seen) and after four clean frames stops launching the second binning round --, then ASYNC_WARM x lanes queued
    frames in one batch so that every pipeline lane is allocated and its enqueue thread awake (the threads sleep while frames are
    drawn synchronously, and a thread's first launches after a sleep take twice as long: 55 us per pair of frames against 27 --
    GS_DEBUG_WORKER; with 2 x lanes frames the twenty-frame region came out 7 % slower), a sync, the warm-up steps, a sync.  frame(k, flags) draws orbit frame
    k; returns the number of synchronous frames.  (Until round 4 the share was WALKED down 2 % per frame: 96 frames and every pose
    twice.  GS_BENCH_PREROLL_FRAMES asks for at least that many again.)"""
    n = 0
    at_least = int(os.environ.get("GS_BENCH_PREROLL_FRAMES", "0") or 0)
    while n < max(at_least, len(frames_used)):
        for k in frames_used:
            frame(k, 0)
            n += 1
    for j in range(ASYNC_WARM * lanes):
        frame(frames_used[j % len(frames_used)], async_flag)
    sync()
    for i in range(warmup):
        frame((warmup_first + i) % ORBIT_FRAMES, async_flag)
    sync()
    return n


# ---- what a frame of a timed region runs, in words (bench.py's `config.timed_work`; tests/test_bench_configs.py holds the sentences
# to the option set and the statistics they are made from)
SORT_MODES = {0: "whole", 1: "histogram", 2: "stash", 3: "tail"}      # gs_stats.sort_mode


def sort_mode_name(opts, stats):
    if opts.get("OPT_SORT_NEAR", 1) == 0:
        return "whole"
    return SORT_MODES.get(int(stats.get("sort_mode", 0)), "whole")


def timed_work(opts, stats, lanes=LANES, frustum=False):
    """{sort_mode, near_permille, frames_in_flight, text}: the reference sorts EVERY splat it keeps (index.js:507-570) and shades every
    fragment (index.js:166-176); what the timed frames do instead -- with bit-identical pixels -- is said here, not implied."""
    fb = int(opts.get("OPT_FRAME_BATCH", 1))
    depth = int(opts.get("OPT_PIPELINE_DEPTH", lanes))
    mode = sort_mode_name(opts, stats)
    share = int(stats.get("near_permille", 1000))
    parts = ["%d frames in flight (%d pipeline lanes%s)" % (depth * fb, depth, " x 2 frames per launch, GS_OPT_FRAME_BATCH: the two frames of a pair share each "
                                                            "kernel launch, grid (x, 2), on separate scratch" if fb == 2 else "")]
    parts.append("every frame runs its own depth pass over all N resident splats (key, culls, bucket: index.js:517-561)")
    if frustum:
        parts.append("which hands on only the splats whose fragments can reach the viewport (gs_sort_for over the whole frame: a conservative bound on "
                     "the quad's reach in x and y; the order is the reference's with the splats the frame cannot show left out, the bucket scale still that of "
                     "every kept splat)")
    if mode == "whole":
        parts.append("and sorts every kept splat (the reference's whole order)")
    elif mode == "tail":
        parts.append("and a TAIL sort: of the 256 depth segments only those the frame will read are scattered and sorted -- the positions it reads "
                     "hold what the whole order holds there, the rest of the order is not produced (GS_OPT_SORT_NEAR)")
    elif mode == "stash":
        parts.append("and a near-only sort of the candidates its depth pass stashed (threshold bin from the frames before; a frame whose stash cannot "
                     "be vouched for is drawn again from a whole sort) -- the positions it reads hold what the whole order holds there (GS_OPT_SORT_NEAR)")
    else:
        parts.append("and a near-only sort behind an exact threshold from a depth histogram -- the positions it reads hold what the whole order "
                     "holds there (GS_OPT_SORT_NEAR)")
    if share >= 1000:
        parts.append("projects, bins and blends the whole order in one round")
    else:
        parts.append("projects, bins and blends the nearest %d permille of the order (the share the blends of the frames before measured, plus a "
                     "margin); the second binning round over the rest is not launched once four collected frames needed none -- a tile that "
                     "turns out unsaturated flags its frame, which gs_sync() draws again in full" % share)
    if stats.get("subtile"):
        parts.append("the blend walks sub-tile lists (GS_OPT_SUBTILE: sixteen 4x4-pixel blocks per tile)")
    return {"sort_mode": mode, "near_permille": share, "frames_in_flight": depth * fb, "text": "; ".join(parts),
            "sort_call": "gs_sort_for(whole frame)" if frustum else "gs_sort"}
