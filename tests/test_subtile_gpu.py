"""GPU tier: GS_OPT_SUBTILE (round 6) -- the blend walking, per batch of 64 list entries, the lists of a tile's sixteen 4x4-pixel
blocks instead of the whole batch with all 256 pixels.  The reference's rasteriser shades only the fragments a quad covers
(index.js:52-66) and its fragment shader discards on |p|^2 > 4 (index.js:171-172); an entry left out of a block's list is one
that would have been discarded in every pixel of the block, so the frames must be the SAME frames, bit for bit, as the whole-tile
walk's, with the same fragment counts -- and therefore stand in the same relation to the oracle (<= 1 LSB, counts exactly equal).

Compared here, sub-tile lists forced on (2) against off (0): scenes of small, medium and huge splats, the camera inside and
outside the cloud; one binning round and two (tiny, medium and adaptive first-round shares: round 1 resumes per-pixel state);
column strips at odd offsets; flipped rows; the opaque scene's depth buffer and colour image; counting renders with and without
early termination; queued frames alone and in pairs (two frames per launch); the automatic setting (1) on a scene it switches
itself on for."""
import numpy as np
import pytest

from conftest import pkg
from oracle import oracle

pytestmark = pytest.mark.gpu
capi = pkg("capi")
synth = pkg("synth")


def _params(cam, **kw):
    return capi.make_params(cam["gs_mv"], cam["gs_proj"], cam["vw"], cam["vh"], focal_=cam["focal"], **kw)


def _scene(n, seed, fat):
    rows = synth.make_splat_rows(n, seed=seed).reshape(-1, 32).copy()
    if fat != 1.0:
        rows[:, 12:24] = (rows[:, 12:24].copy().view("<f4") * np.float32(fat)).view(np.uint8)
    return rows


CASES = [  # w, h, n, fat, camera, yaw
    (640, 360, 30000, 1.0, "outside", 17.0),
    (1920, 1080, 300000, 1.0, "outside", 211.0),
    (1920, 1080, 300000, 1.0, "index", 40.0),
    (333, 211, 5000, 0.3, "outside", 95.0),
    (1280, 720, 6000, 25.0, "index", 211.0),
    (3840, 2160, 200000, 0.5, "outside", 300.0),
]


@pytest.mark.parametrize("w,h,n,fat,pose,yaw", CASES)
def test_subtile_lists_change_no_pixel_and_no_fragment_count(w, h, n, fat, pose, yaw):
    rows = _scene(n, 4343, fat)
    cam = (synth.outside_cloud_camera if pose == "outside" else synth.index_html_camera)(w, h, yaw, capi=capi)
    x0 = (w // 3) & ~3
    strips = [(0, w), (x0, min(w, x0 + 16)), (x0 + 4, min(w, x0 + 207))]
    depth = np.full((h, w), 0.9996, np.float32); depth[:, : w // 2] = 1.0
    rgba = np.zeros((h, w, 4), np.uint8); rgba[..., 1] = 90; rgba[..., 3] = 255
    tiles = ((w + 15) // 16) * ((h + 15) // 16)
    out = {}
    for mode in (0, 2):
        with capi.Context(0) as c:
            c.set_option(capi.OPT_SUBTILE, mode)
            c.push_splat(rows)
            res = []
            for permille in (1000, 3, 400, 0):
                c.set_option(capi.OPT_NEAR_PERMILLE, permille)
                c.sort(cam["view"])
                for a, b in strips:
                    res.append((permille, a, b, c.render(_params(cam, x0=a, x1=b))))
                if permille == 1000:
                    res.append(("flip", c.render(_params(cam, flags=capi.RENDER_FLIP_Y))))
                    c.render(_params(cam, flags=capi.RENDER_COUNT_FRAGS))
                    res.append(("frags", c.stats()["n_frags"]))
                    c.render(_params(cam, flags=capi.RENDER_COUNT_FRAGS | capi.RENDER_COUNT_EVALUATED))
                    res.append(("frags evaluated", c.stats()["n_frags"]))
                    res.append(("no early out", c.render(_params(cam, flags=capi.RENDER_NO_EARLY_OUT))))
                    c.set_scene(depth, rgba)
                    c.sort(cam["view"])
                    res.append(("scene", c.render(_params(cam))))
                    c.render(_params(cam, flags=capi.RENDER_COUNT_FRAGS))
                    res.append(("scene frags", c.stats()["n_frags"]))
                    c.set_scene(None, None)
                    # what the walk costs: entries the tiles' wavefronts step through (GS_OPT_RECORD_STAGED = 2) -- reported, and smaller
                    # with the lists wherever splats are small against a tile
                    c.set_option(capi.OPT_RECORD_STAGED, 2)
                    c.sort(cam["view"])
                    c.render(_params(cam))
                    out[("walked", mode)] = int(c.download(capi.BUF_TILE_STATS, tiles, np.uint32, 2)[:, 0].astype(np.int64).sum())
                    c.set_option(capi.OPT_RECORD_STAGED, 0)
            c.set_option(capi.OPT_NEAR_PERMILLE, 250)
            for batch in (1, 2):
                c.set_option(capi.OPT_FRAME_BATCH, batch)
                bufs = [capi.host_frame(h, w) for _ in range(5)]
                for b, _ in bufs:
                    c.sort(cam["view"], want_indices=False)
                    c.render_into(_params(cam, flags=capi.RENDER_ASYNC), b)
                c.sync()
                for b, o in bufs:
                    res.append(("queued", batch, b.copy()))
                    o.free()
            out[mode] = res
    assert len(out[0]) == len(out[2])
    for a, b in zip(out[0], out[2]):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            if isinstance(x, np.ndarray):
                assert np.array_equal(x, y), (a[:3], int(np.abs(x.astype(np.int16) - y.astype(np.int16)).max()))
            else:
                assert x == y, (a[:3], x, y)
    frags = [r for r in out[2] if r[0] == "frags"][0][1]
    assert frags > 10000
    print("subtile %dx%d n=%d fat=%g %s: entries walked %d -> %d (%.2f), %d fragments" % (
        w, h, n, fat, pose, out[("walked", 0)], out[("walked", 2)], out[("walked", 2)] / max(1, out[("walked", 0)]), frags))
    assert out[("walked", 2)] <= out[("walked", 0)]
    if pose == "outside" and fat <= 1.0:
        assert out[("walked", 2)] < 0.8 * out[("walked", 0)], "small splats: the lists must shorten the walk"


@pytest.mark.parametrize("w,h,yaw", [(320, 180, 10.0), (640, 360, 200.0)])
def test_subtile_lists_against_the_oracle(w, h, yaw):
    """the same bar as test_gpu_parity.test_pixels_match_oracle, with the lists forced on: <= 1 LSB, fragment counts exactly the oracle's"""
    rows = synth.make_splat_rows(30000, seed=77)
    cs, cc, mats = oracle.pack(rows)
    cam = synth.outside_cloud_camera(w, h, yaw, capi=capi, distance=5.0)
    with capi.Context(0) as c:
        c.set_option(capi.OPT_SUBTILE, 2)
        c.push_splat(rows)
        idx = c.sort(cam["view"])
        assert np.array_equal(idx, oracle.sort(mats, cam["view"]))
        want_u8, _, want_frags = oracle.render(cs, cc, idx, cam["gs_mv"].astype(np.float32), cam["gs_proj"].astype(np.float32), np.float32(cam["focal"]), w, h)
        got = c.render(_params(cam))
        assert int(np.abs(got.astype(np.int16) - want_u8.astype(np.int16)).max()) <= 1
        c.render(_params(cam, flags=capi.RENDER_COUNT_FRAGS))
        assert c.stats()["n_frags"] == want_frags and want_frags > 50000


def test_subtile_lists_switch_themselves_on_where_splats_are_small():
    """GS_OPT_SUBTILE = 1 (the default): decided from the last collected frame's pairs per visible splat.  Outside the cloud (3-4 tiles
    per splat, the rule: fewer than 16) the walk shrinks after the first frame; at the headline pose (dozens of tiles per splat) it stays whole."""
    w, h = 1280, 720
    rows = synth.make_splat_rows(200000, seed=99)
    tiles = ((w + 15) // 16) * ((h + 15) // 16)

    def walked(c, cam):
        c.set_option(capi.OPT_RECORD_STAGED, 2)
        c.sort(cam["view"])
        c.render(_params(cam))
        n = int(c.download(capi.BUF_TILE_STATS, tiles, np.uint32, 2)[:, 0].astype(np.int64).sum())
        c.set_option(capi.OPT_RECORD_STAGED, 0)
        return n

    rows = rows.reshape(-1, 32).copy()
    big = rows.copy(); big[:, 12:24] = (big[:, 12:24].copy().view("<f4") * np.float32(4.0)).view(np.uint8)
    for pose, scene, expect_on in ((synth.outside_cloud_camera, rows, True), (synth.index_html_camera, big, False)):
        cam = pose(w, h, 30.0, capi=capi)
        ref = {}
        for mode in (0, 2, 1):
            with capi.Context(0) as c:
                c.set_option(capi.OPT_SUBTILE, mode)
                c.set_option(capi.OPT_NEAR_PERMILLE, 1000)
                c.push_splat(scene)
                c.sort(cam["view"]); img0 = c.render(_params(cam))              # (the frame the automatic setting decides from)
                st = c.stats()
                ref[mode] = (walked(c, cam), img0)
        assert np.array_equal(ref[0][1], ref[2][1]) and np.array_equal(ref[0][1], ref[1][1])
        ratio = st["n_pairs"] / max(1, st["n_visible"])
        print("pose %s: %.1f tiles per visible splat; entries walked off %d, on %d, auto %d" % (pose.__name__, ratio, ref[0][0], ref[2][0], ref[1][0]))
        assert (ratio < 16.0) == expect_on
        assert ref[1][0] == (ref[2][0] if expect_on else ref[0][0])
