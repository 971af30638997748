"""Pins the CPU oracle (oracle/gs_oracle.c) against golden vectors produced by
the reference's own JavaScript (oracle/gen_golden.js ran /root/reference/index.js
under node).  Bit-exact for sort / pack / ply; camera matrices exact in f64."""
import numpy as np
import pytest

from conftest import cases_of, load_case
from oracle import oracle


@pytest.mark.parametrize("name", cases_of("sort"))
def test_sort_matches_reference_worker(name):
    c = load_case(name)
    rows = c["rows4"].reshape(-1, 4)
    got = oracle.sort(rows, c["view"], c.get("cutout"))
    assert got.dtype == np.uint32
    assert got.size == c["sorted"].size
    assert np.array_equal(got, c["sorted"])


def test_js_worker_restatement_matches_reference_worker():
    """oracle/worker_sort.js (the JS-worker CPU baseline bench.py times under node) against the same golden vectors."""
    import json, os, shutil, subprocess
    from conftest import GOLDEN, ROOT
    node = shutil.which("node")
    if not node:
        pytest.skip("node not installed")
    r = subprocess.run([node, os.path.join(ROOT, "oracle", "worker_sort.js"), "check", GOLDEN], capture_output=True, text=True, timeout=120)
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert r.returncode == 0 and rep["failures"] == [] and rep["cases"] >= 12, (r.stdout, r.stderr)


def test_js_worker_bench_mode_agrees_with_c_oracle(tmp_path):
    import json, os, shutil, subprocess
    from conftest import ROOT
    node = shutil.which("node")
    if not node:
        pytest.skip("node not installed")
    c = load_case("sort_n4096_cutout")
    rows = c["rows4"].reshape(-1, 4)
    rows.astype("<f4").tofile(tmp_path / "rows.f32")
    np.concatenate([c["view"], c["cutout"]]).astype("<f4").tofile(tmp_path / "un.f32")
    r = subprocess.run([node, os.path.join(ROOT, "oracle", "worker_sort.js"), "bench", str(tmp_path / "rows.f32"), str(tmp_path / "un.f32"), "2"],
                       capture_output=True, text=True, timeout=120)
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    want = oracle.sort(rows, c["view"], c["cutout"])
    assert rep["n"] == rows.shape[0] and rep["kept"] == want.size
    assert rep["order_sum"] == oracle.order_sum(want)


def test_sort_full_matrix_stride():
    c = load_case("sort_n4096")
    rows = c["rows4"].reshape(-1, 4)
    m = np.zeros((rows.shape[0], 16), np.float32)
    m[:, 12:16] = rows
    m[:, :12] = 7.0  # garbage that must be ignored (index.js:520-548 read only 12..15)
    assert np.array_equal(oracle.sort(m, c["view"]), c["sorted"])


@pytest.mark.parametrize("name", cases_of("pack"))
def test_pack_matches_reference_pushDataBuffer(name):
    c = load_case(name)
    cs, cc, mats = oracle.pack(c["rows"])
    assert np.array_equal(cs.reshape(-1).view(np.uint32), c["center_scale"].view(np.uint32))
    assert np.array_equal(cc.reshape(-1), c["cov_color"])
    assert np.array_equal(mats.reshape(-1).view(np.uint32), c["matrices"].view(np.uint32))


def test_pack_parseint_quirk_present():
    """Row 0 of pack_n300 is a needle with identity rotation: the reference's
    parseInt(Number) returns the leading digit of the exponent-form string."""
    c = load_case("pack_n300")
    q = c["cov_color"].reshape(-1, 4)[0]
    m22 = int(q[1] >> 16)
    m33 = int(q[2] >> 16)
    assert (m22, m33) != (0, 0)


@pytest.mark.parametrize("name", cases_of("ply"))
def test_ply_matches_reference_processPlyBuffer(name):
    c = load_case(name)
    got = oracle.ply_to_splat(c["ply"])
    assert np.array_equal(got, c["rows"])


def test_js_exp_matches_the_engine_bit_for_bit():
    """Math.exp is third-party arithmetic (V8's fdlibm port): the oracle's restatement must reproduce the 8200 values
    node returned here, bit for bit -- libm's exp does not (it differs in the last bit for ~10 % of them)."""
    c = load_case("math_exp")
    got = oracle.js_exp(c["x"])
    both_nan = np.isnan(got) & np.isnan(c["exp"])
    assert c["x"].size >= 8000
    assert np.array_equal(got.view(np.uint64)[~both_nan], c["exp"].view(np.uint64)[~both_nan])


def _ply(props, nbytes, end=True):
    return (b"ply\nformat binary_little_endian 1.0\nelement vertex 1\n" +
            b"".join(b"property float %s\n" % n for n in props) + (b"end_header\n" if end else b"") + b"\0" * nbytes)


def test_ply_errors(manifest):
    e = manifest["ply_errors"]["meta"]
    assert e["no_end_header"] == "Unable to read .ply file header"
    with pytest.raises(oracle.PlyError) as ei:
        oracle.ply_to_splat(_ply([b"x"], 8, end=False))
    assert str(ei.value) == e["no_end_header"]
    with pytest.raises(oracle.PlyError) as ei:
        oracle.ply_to_splat(_ply([b"x", b"y", b"z", b"scale_0", b"scale_1", b"scale_2", b"opacity", b"rot_0", b"rot_1",
                                  b"rot_2"], 40))
    assert str(ei.value) == e["missing_rot_3"]
    with pytest.raises(oracle.PlyError) as ei:
        oracle.ply_to_splat(_ply([b"x", b"y", b"z"], 12))
    assert str(ei.value) == e["missing_red"]


@pytest.mark.parametrize("name", cases_of("camera"))
def test_camera_matches_reference(name):
    c = load_case(name)
    mv = oracle.model_view(c["cam_world"], c["obj_world"])
    assert np.array_equal(mv, c["gs_mv"])
    pr = oracle.projection(c["proj"])
    assert np.array_equal(pr, c["gs_proj"])
    view, cut = oracle.tick(c["cam_world"], c["obj_world"], c.get("cutout_world"))
    assert np.array_equal(view.view(np.uint32), c["view"].view(np.uint32))
    if "cutout" in c:
        assert np.array_equal(cut.view(np.uint32), c["cutout"].view(np.uint32))
    assert oracle.focal(pr, c["viewport"][1]) == c["focal"][0]


def test_oracle_scene_depth_semantics():
    """depthTest LEQUAL / depthWrite off (index.js:179-180) in the pixel oracle: a depth buffer at the far plane changes
    nothing, one at the near plane rejects every fragment and leaves the scene colour."""
    from conftest import pkg
    synth = pkg("synth")
    rows = synth.make_splat_rows(2000, seed=4)
    cs, cc, mats = oracle.pack(rows)
    cam = synth.index_html_camera(160, 90, 20.0)
    idx = oracle.sort(mats, cam["view"])
    mv, P = cam["gs_mv"].astype(np.float32), cam["gs_proj"].astype(np.float32)
    base, _, n0 = oracle.render(cs, cc, idx, mv, P, cam["focal"], 160, 90)
    far, _, n1 = oracle.render(cs, cc, idx, mv, P, cam["focal"], 160, 90, scene_depth=np.ones((90, 160), np.float32))
    assert n0 == n1 > 0 and np.array_equal(base, far)
    col = np.random.default_rng(1).integers(0, 256, (90, 160, 4)).astype(np.uint8)
    near, _, n2 = oracle.render(cs, cc, idx, mv, P, cam["focal"], 160, 90, scene_depth=np.zeros((90, 160), np.float32), scene_rgba=col)
    assert n2 == 0 and np.array_equal(near, col)
