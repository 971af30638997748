#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (oracle side).  Scenes for oracle/gen_golden_gl.js: .splat rows from the benchmark's generator and
the three.js-style camera inputs (camera / entity / cutout world matrices, projection matrix) of the demo poses, written to
oracle/_ref/gl_scenes/ (scratch, git-ignored).  The uniforms are NOT computed here: the reference computes them itself."""
import importlib, json, math, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
synth = importlib.import_module("aframe-gaussian-splatting_amd.synth")
OUT = os.path.join(ROOT, "oracle", "_ref", "gl_scenes"); os.makedirs(OUT, exist_ok=True)
scenes = {}


def scene(name, rows, w, h, cam, obj, proj, cutout=None, note="", depth=None, rgba=None, strip=None, recipe=None, eye=None):
    fn = name + ".splat"
    open(os.path.join(OUT, fn), "wb").write(np.ascontiguousarray(rows).tobytes())
    scenes[name] = {"rows": fn, "width": w, "height": h, "cam_world": list(map(float, cam)), "obj_world": list(map(float, obj)),
                    "proj": list(map(float, proj)), "cutout_world": (list(map(float, cutout)) if cutout is not None else None), "note": note}
    if eye is not None:
        scenes[name].update({"eye_cam_world": list(map(float, eye[0])), "eye_proj": list(map(float, eye[1]))})
    if strip is not None:
        scenes[name].update({"strip": list(strip), "store_rows": False, "rows_recipe": recipe})
    if depth is not None:
        open(os.path.join(OUT, name + ".depth"), "wb").write(np.ascontiguousarray(depth, "<f4").tobytes()); scenes[name]["scene_depth"] = name + ".depth"
    if rgba is not None:
        open(os.path.join(OUT, name + ".rgba"), "wb").write(np.ascontiguousarray(rgba, np.uint8).tobytes()); scenes[name]["scene_rgba"] = name + ".rgba"


# index.html:13 pose, entity yaw 40 deg: the benchmark's scene generator at a size the software rasteriser finishes in seconds
scene("index_yaw40", synth.make_splat_rows(4000, seed=401), 320, 180, synth.compose((0.0, 1.6, 0.0)), synth.compose((0.0, 1.5, -2.0), 40.0),
      synth.perspective(80.0, 320 / 180), note="index.html:13 pose, entity yaw 40 deg, 4000 splats, 320x180")
# cutout-demo.html:22-24 pose with the cutout box
scene("cutout_demo", synth.make_splat_rows(8000, seed=402), 320, 180, synth.compose((5.132, 1.6, 7.237)), synth.compose((0.0, 0.8, -2.0), 15.0, (2.0, 2.0, 2.0)),
      synth.perspective(80.0, 320 / 180), cutout=synth.compose((0.8145, 1.73322, -2.35981), 0.0, (4.17, 2.95, 3.89)),
      note="cutout-demo.html:22-24 pose with its cutout box, 8000 splats, 320x180")
# XR-like eye: asymmetric frustum, odd viewport (1032x1104 / 4)
near, far = 0.005, 10000.0
proj = synth.frustum(-math.tan(math.radians(54)) * near, math.tan(math.radians(40)) * near, math.tan(math.radians(44)) * near,
                     -math.tan(math.radians(55)) * near, near, far)
scene("xr_left_eye", synth.make_splat_rows(3000, seed=403), 258, 276, synth.compose((-0.032, 1.6, 0.0)), synth.compose((0.0, 1.5, -2.0), 200.0), proj,
      note="left XR eye (asymmetric frustum, SURVEY 8d), 3000 splats, 258x276 (the sort uses this camera: one context per eye)")
# the splats inside a three.js scene: an opaque surface drawn first (its window-space depth and colour are in the framebuffer
# when the transparent splat mesh is drawn with depthTest on, depthWrite off: index.js:179-180)
w, h = 256, 144
yy, xx = np.mgrid[0:h, 0:w]
dist = 2.2 + 3.5 * (xx / (w - 1.0)) + 0.8 * np.sin(yy / 17.0)             # metres in front of the camera, varies over the image
zndc = (far + near) / (far - near) - 2.0 * far * near / ((far - near) * dist)
depth = (0.5 * zndc + 0.5).astype(np.float32)
depth[(xx // 32 + yy // 24) % 5 == 0] = 1.0                               # holes: nothing opaque there
rgba = np.zeros((h, w, 4), np.uint8)
rgba[..., 0] = (xx * 255 // (w - 1)).astype(np.uint8); rgba[..., 1] = (yy * 255 // (h - 1)).astype(np.uint8)
rgba[..., 2] = (((xx // 16 + yy // 16) % 2) * 180 + 40).astype(np.uint8); rgba[..., 3] = 255
scene("index_yaw130_scene", synth.make_splat_rows(4000, seed=404), w, h, synth.compose((0.0, 1.6, 0.0)), synth.compose((0.0, 1.5, -2.0), 130.0),
      synth.perspective(80.0, w / h), depth=depth, rgba=rgba,
      note="index.html pose, yaw 130 deg, drawn over an opaque scene (depth LEQUAL, no depth write; colour = destination), 4000 splats, 256x144")
# BASELINE configs[1] itself: the benchmark's 1,048,576-splat scene at 1920x1080, orbit frame 7 (entity yaw 21 deg); the rows are
# stored by recipe + SHA-1, the frame as a 64-pixel column strip
# the dropped-bucket pathology (golden sort_oob_bucket): a cloud 1e-5 wide 500 m away -- the f32 rounding of the stored depths
# exceeds their range, buckets fall outside the table, the reference's typed-array writes drop them and its index list ends in
# zeros: splat 0 is drawn once more for every dropped splat (index.js:561-567, 201-207)
g = np.random.Generator(np.random.PCG64(405))
rows = synth.make_splat_rows(2048, seed=405).reshape(-1, 32).copy()
pos = (np.array([0.4, 0.3, 500.0]) + 2e-6 * g.standard_normal((2048, 3))).astype("<f4")
rows[:, 0:12] = pos.view(np.uint8).reshape(2048, 12)
rows[:, 12:24] = (rows[:, 12:24].copy().view("<f4") * np.float32(60.0)).view(np.uint8)
scene("oob_bucket", rows, 192, 108, synth.compose((0.0, 0.0, 0.0)), synth.compose((0.0, 0.0, 0.0), 1.0), synth.perspective(80.0, 192 / 108),
      note="dropped-bucket pathology: 2048 splats within 1e-5 of (0.4, 0.3, 500) in file coordinates (the loader negates z); the index list ends in zeros and splat 0 is drawn again for each")
if "--big" in sys.argv:
    rows_1m = synth.make_splat_rows(synth.N_TRAIN)
    rec_1m = {"fn": "make_splat_rows", "n": int(synth.N_TRAIN)}
    scene("c2_1m_1080p_strip", rows_1m, 1920, 1080, synth.compose((0.0, 1.6, 0.0)), synth.compose((0.0, 1.5, -2.0), 21.0),
          synth.perspective(80.0, 1920 / 1080), strip=(928, 992), recipe=rec_1m,
          note="BASELINE configs[1]: the benchmark scene (1,048,576 splats) at 1920x1080, orbit frame 7, columns 928..991")
    scene("c1_1m_720p_strip", rows_1m, 1280, 720, synth.compose((0.0, 1.6, 0.0)), synth.compose((0.0, 1.5, -2.0), 300.0),
          synth.perspective(80.0, 1280 / 720), strip=(400, 464), recipe=rec_1m,
          note="BASELINE configs[0]: the same scene at 1280x720, entity yaw 300 deg, columns 400..463")
    # BASELINE configs[4]'s frame size with the 1 M scene (its 20 M rows are beyond what node packs in reasonable time): 3840x2160 has
    # 32 400 tiles -- 15 bits of tile id, so the binning takes its 8-byte pair records (GS_OPT_WIDE_PAIRS' path) -- one 64-pixel strip
    scene("c5_size_4k_strip", rows_1m, 3840, 2160, synth.compose((0.0, 1.6, 0.0)), synth.compose((0.0, 1.5, -2.0), 222.0),
          synth.perspective(80.0, 3840 / 2160), strip=(2112, 2176), recipe=rec_1m,
          note="3840x2160 (BASELINE configs[4]'s size) with the 1 M scene, entity yaw 222 deg, columns 2112..2175")
    # BASELINE configs[3]: XR, 2064x2208 x 0.5 per eye; ONE sort from the head camera (index.js:441), drawn with the right eye's camera
    wx, hx = 1032, 1104
    ro, lo = math.tan(math.radians(54)), math.tan(math.radians(40))
    eye_proj = synth.frustum(-lo * near, ro * near, math.tan(math.radians(44)) * near, -math.tan(math.radians(55)) * near, near, far)
    scene("c4_xr_right_eye_strip", rows_1m, wx, hx, synth.compose((0.0, 1.6, 0.0)), synth.compose((0.0, 1.5, -2.0), 10.0), synth.perspective(80.0, wx / hx),
          strip=(480, 544), recipe=rec_1m, eye=(synth.compose((0.032, 1.6, 0.0)), eye_proj),
          note="BASELINE configs[3]: right XR eye 1032x1104 (asymmetric frustum), order from the HEAD camera's sort, columns 480..543")
    # BASELINE configs[2]: 6 M splats + the cutout-demo box at 1920x1080
    scene("c3_6m_cutout_strip", synth.make_splat_rows(synth.N_BICYCLE, seed=synth.SEED_BASE + 3), 1920, 1080, synth.compose((5.132, 1.6, 7.237)),
          synth.compose((0.0, 0.8, -2.0), 75.0, (2.0, 2.0, 2.0)), synth.perspective(80.0, 1920 / 1080),
          cutout=synth.compose((0.8145, 1.73322, -2.35981), 0.0, (4.17, 2.95, 3.89)), strip=(640, 704),
          recipe={"fn": "make_splat_rows", "n": int(synth.N_BICYCLE), "seed": int(synth.SEED_BASE + 3)},
          note="BASELINE configs[2]: 6,291,456 splats + the cutout-demo.html box at 1920x1080, entity yaw 75 deg, columns 640..703")
if "--big" in sys.argv or "--full-frame" in sys.argv:
    # the WHOLE headline frame (BASELINE configs[1], orbit frame 40: entity yaw 120 deg); only the float-buffer image is kept
    scene("c2_1m_1080p_frame", synth.make_splat_rows(synth.N_TRAIN), 1920, 1080, synth.compose((0.0, 1.6, 0.0)), synth.compose((0.0, 1.5, -2.0), 120.0),
          synth.perspective(80.0, 1920 / 1080), strip=(0, 1920), recipe={"fn": "make_splat_rows", "n": int(synth.N_TRAIN)},
          note="BASELINE configs[1], the whole 1920x1080 frame, orbit frame 40 (also times the reference's GPU half on this host's cores)")
    scenes["c2_1m_1080p_frame"]["store_rgba8"] = False
if "--big" in sys.argv or "--frames" in sys.argv:
    # whole frames of the other single-GPU configurations (float-buffer image only): C1, and BOTH eyes of C4 -- one sort from the
    # head camera (index.js:441), each eye drawn with its own camera
    rows_1m = synth.make_splat_rows(synth.N_TRAIN)
    rec_1m = {"fn": "make_splat_rows", "n": int(synth.N_TRAIN)}
    scene("c1_1m_720p_frame", rows_1m, 1280, 720, synth.compose((0.0, 1.6, 0.0)), synth.compose((0.0, 1.5, -2.0), 63.0),
          synth.perspective(80.0, 1280 / 720), strip=(0, 1280), recipe=rec_1m,
          note="BASELINE configs[0], the whole 1280x720 frame, entity yaw 63 deg")
    scenes["c1_1m_720p_frame"]["store_rgba8"] = False
    wx, hx = 1032, 1104
    for tag, sx in (("left", -1.0), ("right", 1.0)):
        lo, ro = (math.tan(math.radians(54)), math.tan(math.radians(40))) if sx < 0 else (math.tan(math.radians(40)), math.tan(math.radians(54)))
        eye_proj = synth.frustum(-lo * near, ro * near, math.tan(math.radians(44)) * near, -math.tan(math.radians(55)) * near, near, far)
        nm = "c4_xr_%s_eye_frame" % tag
        scene(nm, rows_1m, wx, hx, synth.compose((0.0, 1.6, 0.0)), synth.compose((0.0, 1.5, -2.0), 155.0), synth.perspective(80.0, wx / hx),
              strip=(0, wx), recipe=rec_1m, eye=(synth.compose((0.032 * sx, 1.6, 0.0)), eye_proj),
              note="BASELINE configs[3]: the whole %s XR eye 1032x1104, order from the HEAD camera's sort, entity yaw 155 deg" % tag)
        scenes[nm]["store_rgba8"] = False
if "--big" in sys.argv or "--c3frame" in sys.argv:
    # the whole C3 frame: 6 M splats + the cutout-demo box (the reference's worker culls the rows itself), float-buffer image only
    scene("c3_6m_cutout_frame", synth.make_splat_rows(synth.N_BICYCLE, seed=synth.SEED_BASE + 3), 1920, 1080, synth.compose((5.132, 1.6, 7.237)),
          synth.compose((0.0, 0.8, -2.0), 200.0, (2.0, 2.0, 2.0)), synth.perspective(80.0, 1920 / 1080),
          cutout=synth.compose((0.8145, 1.73322, -2.35981), 0.0, (4.17, 2.95, 3.89)), strip=(0, 1920),
          recipe={"fn": "make_splat_rows", "n": int(synth.N_BICYCLE), "seed": int(synth.SEED_BASE + 3)},
          note="BASELINE configs[2], the whole 1920x1080 frame: 6,291,456 splats + the cutout-demo.html box, entity yaw 200 deg")
    scenes["c3_6m_cutout_frame"]["store_rgba8"] = False
if "--c5" in sys.argv:
    # BASELINE configs[4]'s OWN scene: 20,971,520 splats at 3840x2160 -- more than 4096^2 vertices, so the "renderer" has to report
    # MAX_TEXTURE_SIZE 8192 (index.js:30-36 would clamp at 16.7 M otherwise); pushed in chunks like a progressive load; one
    # 64-pixel strip.  ~1.3 GB of worker rows in node, a few minutes.
    n5 = 20 * (1 << 20)
    scene("c5_20m_4k_strip", synth.make_splat_rows_fast(n5), 3840, 2160, synth.compose((0.0, 1.6, 0.0)), synth.compose((0.0, 1.5, -2.0), 33.0),
          synth.perspective(80.0, 3840 / 2160), strip=(1888, 1952), recipe={"fn": "make_splat_rows_fast", "n": n5},
          note="BASELINE configs[4]: 20,971,520 splats at 3840x2160, entity yaw 33 deg, columns 1888..1951 (MAX_TEXTURE_SIZE 8192)")
    scenes["c5_20m_4k_strip"]["push_chunk"] = 1 << 22
if "--only-new" in sys.argv:                                  # (side runs: GS_GL_MERGE=1 node oracle/gen_golden_gl.js keeps the other cases)
    keep = [k for k in scenes if (k.endswith("_frame") and not k.startswith("c2_") and ("--c3frame" not in sys.argv or k.startswith("c3_"))) or k == "c5_20m_4k_strip"]
    for k in list(scenes):
        if k not in keep:
            os.unlink(os.path.join(OUT, scenes[k]["rows"]))
            for e in ("scene_depth", "scene_rgba"):
                if e in scenes[k]: os.unlink(os.path.join(OUT, scenes[k][e]))
            del scenes[k]
json.dump(scenes, open(os.path.join(OUT, "scenes.json"), "w"), indent=1)
print("wrote", len(scenes), "scenes ->", OUT)
