"""CPU tier: the product's per-splat arithmetic header (csrc/gs_device_math.h) compiled for the host
by tests/host_check/ and compared with golden vectors / the oracle.  This validates the formulas the HIP
kernels execute before any GPU time is spent; it is not a product path (the library has no CPU fallback)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, cases_of, load_case, pkg
from oracle import oracle

HC_DIR = os.path.join(ROOT, "tests", "host_check")
CSRC = os.path.join(ROOT, "aframe-gaussian-splatting_amd", "csrc")


@pytest.fixture(scope="module")
def hc():
    so = os.path.join(HC_DIR, "libhost_check.so")
    srcs = [os.path.join(HC_DIR, "host_check.cpp"), os.path.join(CSRC, "gs_device_math.h"), os.path.join(CSRC, "gs_host_tables.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fvisibility=hidden",
                               "-I", CSRC, "-o", so, srcs[0]])
    L = C.CDLL(so)
    L.hc_sort.restype = C.c_size_t
    L.hc_sort.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    L.hc_pack.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    L.hc_project.restype = C.c_int
    L.hc_project.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p]
    L.hc_frag_power.restype = C.c_float
    L.hc_frag_power.argtypes = [C.c_float] * 6
    L.hc_toint32.restype = C.c_int32
    L.hc_toint32.argtypes = [C.c_double]
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


@pytest.mark.parametrize("name", cases_of("sort"))
def test_sort_keys_match_reference(hc, name):
    c = load_case(name)
    rows = np.ascontiguousarray(c["rows4"], np.float32)
    n = rows.size // 4
    out = np.zeros(max(n, 1), np.uint32)
    cut = c.get("cutout")
    v = hc.hc_sort(_p(rows), n, _p(c["view"]), _p(cut), _p(out))
    assert v == c["sorted"].size
    assert np.array_equal(out[:v], c["sorted"])


@pytest.mark.parametrize("name", cases_of("pack"))
def test_pack_matches_reference(hc, name):
    c = load_case(name)
    rows = np.ascontiguousarray(c["rows"])
    n = rows.size // 32
    cs = np.zeros(n * 4, np.float32); cc = np.zeros(n * 4, np.uint32); sr = np.zeros(n * 4, np.float32)
    hc.hc_pack(_p(rows), n, _p(cs), _p(cc), _p(sr))
    assert np.array_equal(cs.view(np.uint32), c["center_scale"].view(np.uint32))
    assert np.array_equal(cc, c["cov_color"])
    ref_rows = c["matrices"].reshape(-1, 16)[:, 12:16].reshape(-1)
    assert np.array_equal(sr.view(np.uint32), np.ascontiguousarray(ref_rows).view(np.uint32))


def test_affine_cutout_path_equals_the_general_one(hc):
    """round 4: a cut-out matrix whose last row is (0, 0, 0, 1) skips the perspective division (w is exactly 1 for finite
    positions; a position that is not finite takes the general path).  Same answer for random boxes and positions incl. IEEE specials."""
    hc.hc_cutout_affine_mismatches.restype = C.c_size_t
    hc.hc_cutout_affine_mismatches.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    g = np.random.Generator(np.random.PCG64(20260927))
    specials = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 3.4e38, -3.4e38, 0.5, -0.5, 1.0], np.float32)
    for trial in range(40):
        m = np.eye(4)
        m[:3, :3] = g.normal(size=(3, 3)) * g.choice([0.05, 0.4, 1.0, 7.0])
        m[:3, 3] = g.normal(size=3) * g.choice([0.0, 0.3, 3.0])
        c16 = np.ascontiguousarray(m.T.reshape(-1), np.float64)        # column-major, last ROW (elements 3, 7, 11, 15) = 0 0 0 1
        assert c16[3] == 0 and c16[7] == 0 and c16[11] == 0 and c16[15] == 1
        pos = (g.normal(size=(20000, 3)) * g.choice([0.1, 1.0, 30.0])).astype(np.float32)
        # positions on the faces of the box (|q| = 0.5 up to rounding): the boundary decides
        inv = np.linalg.inv(m)
        face = g.uniform(-0.5, 0.5, size=(4000, 3)); face[np.arange(4000), g.integers(0, 3, 4000)] = g.choice([-0.5, 0.5], 4000)
        onface = (inv[:3, :3] @ face.T).T + inv[:3, 3]
        onface[:, 1] *= -1.0                                           # (the test negates y: index.js:528)
        sp = specials[g.integers(0, len(specials), size=(3000, 3))]
        allpos = np.ascontiguousarray(np.concatenate([pos, onface.astype(np.float32), sp]), np.float32)
        assert hc.hc_cutout_affine_mismatches(_p(allpos), allpos.shape[0], _p(c16)) == 0, trial


def test_toint32(hc):
    for d, want in [(0.0, 0), (-0.9, 0), (65535.99, 65535), (-1.0, -1), (-393.7, -393), (2.0**31, -2**31), (2.0**32 + 5, 5),
                    (-(2.0**32) - 7, -7), (float("inf"), 0), (float("nan"), 0), (1e300, 0), (2.0**53 + 2, 2), (4294967295.0, -1)]:
        assert hc.hc_toint32(d) == want, d


def test_project_matches_oracle_bit_exact(hc):
    synth = pkg("synth")
    rows = synth.make_splat_rows(3000, seed=5)
    cs, cc, _ = oracle.pack(rows)
    cam = synth.index_html_camera(1920, 1080, yaw_deg=23.0)
    mv = cam["gs_mv"].astype(np.float32); P = cam["gs_proj"].astype(np.float32)
    focal = np.float32(cam["focal"])
    out = np.zeros(17, np.float32)
    nvis = 0
    for i in range(cs.shape[0]):
        o = oracle.project(cs, cc, i, mv, P, focal, 1920, 1080)
        vis = hc.hc_project(_p(cs), _p(cc), i, _p(mv), _p(P), focal, 1920.0, 1080.0, _p(out))
        assert vis == o.visible, i
        if vis:
            nvis += 1
            want = np.array([o.cx, o.cy, o.ax, o.ay, o.bx, o.by, o.v1x, o.v1y, o.v2x, o.v2y, o.zndc, o.alpha], np.float32)
            assert np.array_equal(out[:12].view(np.uint32), want.view(np.uint32)), i
            # conservative bounds really contain the ellipse extent
            hw = 2 * np.sqrt(np.float64(o.v1x) ** 2 + np.float64(o.v2x) ** 2)
            assert out[13] <= np.ceil(o.cx - hw - 0.5) and out[14] >= np.floor(o.cx + hw - 0.5)
    assert nvis > 500


def test_exact_tile_rows_are_a_superset_of_pixel_coverage_and_tighter_than_the_rect(hc):
    """Per-row ellipse/tile coverage (gsm::splat_tile_row): never misses a tile that holds a covered pixel centre
    (|p|^2 <= 4 by the exact fragment formula), and prunes the bounding rectangle."""
    hc.hc_tile_rows.restype = C.c_int
    hc.hc_tile_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    synth = pkg("synth")
    W, H = 640, 360
    rows = synth.make_splat_rows(6000, seed=8)
    cs, cc, _ = oracle.pack(rows)
    tot_rect = tot_exact = checked = 0
    for yaw, (x0, x1) in [(0.0, (0, W)), (140.0, (0, W)), (250.0, (160, 331))]:
        cam = synth.index_html_camera(W, H, yaw_deg=yaw)
        mv = cam["gs_mv"].astype(np.float32); P = cam["gs_proj"].astype(np.float32); focal = np.float32(cam["focal"])
        out = np.zeros(17, np.float32); tr = np.zeros(3 * 64, np.int32); rect = np.zeros(4, np.int32)
        for i in range(cs.shape[0]):
            if not hc.hc_project(_p(cs), _p(cc), i, _p(mv), _p(P), focal, float(W), float(H), _p(out)):
                continue
            rec = np.concatenate([out[0:6], out[12:13], out[11:12], out[6:11]]).astype(np.float32)
            k = hc.hc_tile_rows(_p(rec), W, H, x0, x1, _p(tr), 64, _p(rect))
            if k == 0:
                continue
            t = tr[:3 * k].reshape(k, 3)
            tot_rect += (rect[2] - rect[0] + 1) * (rect[3] - rect[1] + 1)
            tot_exact += int(t[:, 2].sum())
            nz = t[t[:, 2] > 0]
            # rows use a slightly larger safety pad than the bounding box: at most one tile beyond it
            assert np.all(nz[:, 1] >= rect[0] - 1) and np.all(nz[:, 1] + nz[:, 2] - 1 <= rect[2] + 1)
            # exact pixel coverage inside the (clamped) bounding box
            bx0, bx1 = int(max(out[13], x0)), int(min(out[14], x1 - 1))
            by0, by1 = int(max(out[15], 0)), int(min(out[16], H - 1))
            if (bx1 - bx0 + 1) * (by1 - by0 + 1) > 40000:
                continue
            xs = (np.arange(bx0, bx1 + 1, dtype=np.float32) + np.float32(0.5)) - out[0]
            ys = (np.arange(by0, by1 + 1, dtype=np.float32) + np.float32(0.5)) - out[1]
            dx, dy = np.meshgrid(xs, ys)
            ppx = dx * out[2] + dy * out[3]; ppy = dx * out[4] + dy * out[5]
            cov = (ppx * ppx + ppy * ppy) <= np.float32(4.0000005)
            jj, ii = np.nonzero(cov)
            if jj.size == 0:
                continue
            checked += 1
            trow = (H - 1 - (jj + by0)) // 16; tcol = (ii + bx0 - x0) // 16
            lut = {int(r[0]): (int(r[1]), int(r[1] + r[2] - 1)) for r in t}
            for ty_, tx_ in set(zip(trow.tolist(), tcol.tolist())):
                lo, hi = lut[ty_]
                assert lo <= tx_ <= hi, (i, ty_, tx_, lo, hi)
    assert checked > 1500
    assert tot_exact < 0.9 * tot_rect, (tot_exact, tot_rect)


def test_product_js_exp_matches_the_engine(hc):
    """gsm::js_exp (csrc/gs_ply.h, shared by the host and the HIP .ply converters) against Math.exp's own outputs."""
    c = load_case("math_exp")
    x = np.ascontiguousarray(c["x"])
    out = np.zeros_like(x)
    hc.hc_js_exp.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    hc.hc_js_exp.restype = None
    hc.hc_js_exp(_p(x), x.size, _p(out))
    both_nan = np.isnan(out) & np.isnan(c["exp"])
    assert np.array_equal(out.view(np.uint64)[~both_nan], c["exp"].view(np.uint64)[~both_nan])


def test_xcd_chunk_order_visits_every_chunk_once(hc):
    """gsm::xcd_chunk (the radix kernels' chunk order): over the virtual indices [0, 8*ceil(n/8)) every chunk appears exactly
    once, and the indices with the same v % 8 (one XCD) form one contiguous range of chunks."""
    hc.hc_xcd_chunk.restype = C.c_int
    hc.hc_xcd_chunk.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    for n in (1, 2, 7, 8, 9, 15, 16, 17, 63, 64, 65, 512, 937, 1000, 9766):
        seen, by_xcd = [], {}
        out = C.c_uint32(0)
        for v in range(8 * ((n + 7) // 8)):
            if hc.hc_xcd_chunk(v, n, C.byref(out)):
                seen.append(out.value); by_xcd.setdefault(v % 8, []).append(out.value)
        assert sorted(seen) == list(range(n)), n
        for x, cs in by_xcd.items():
            assert cs == list(range(cs[0], cs[0] + len(cs))), (n, x)
