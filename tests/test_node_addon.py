"""The reference-language host side: N-API addon + component shim, driven by node (v12 here and on the GPU box)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, pkg

JS = os.path.join(ROOT, "tests", "js")                      # the node-side tests (the package's js/ holds the addon and the shim only)
NODE = shutil.which("node")


def _addon():
    b = pkg("build")
    b.build_lib()
    return b.build_addon()


@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_addon_surface_and_host_helpers_cpu():
    assert _addon() is not None
    r = subprocess.run([NODE, os.path.join(JS, "test_addon.js"), "cpu", GOLDEN], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "addon cpu checks ok" in r.stdout


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_addon_worker_protocol_and_render_gpu(tmp_path):
    from oracle import oracle
    capi, synth = pkg("capi"), pkg("synth")
    assert _addon() is not None
    rows = synth.make_splat_rows(20000, seed=99)
    scene = tmp_path / "scene.splat"
    scene.write_bytes(rows.tobytes())
    out = tmp_path / "frame.rgba"
    w, h, yaw = 320, 180, 35.0
    r = subprocess.run([NODE, os.path.join(JS, "test_addon.js"), "gpu", GOLDEN, str(scene), str(out), str(w), str(h), str(yaw)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "asynchronous" not in r.stderr
    fps = [l for l in r.stdout.splitlines() if l.startswith("js-visible frames/s")]
    assert fps, r.stdout
    print("\n".join(fps))                                    # the rates a JavaScript caller sees (-s shows them)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        open(os.path.join(ROOT, "gpurun_out", "js_visible_fps.txt"), "w").write("\n".join(fps) + "\n")
    except OSError:
        pass
    img = np.frombuffer(out.read_bytes(), np.uint8).reshape(h, w, 4)
    idx = np.frombuffer((tmp_path / "frame.rgba.idx").read_bytes(), np.uint32)
    cam = synth.index_html_camera(w, h, yaw, capi=capi)
    _, _, mats = oracle.pack(rows)
    assert np.array_equal(idx, oracle.sort(mats, cam["view"]))              # JS tick -> same order as the reference worker
    with capi.Context(0) as ctx:                                              # same frame through the ctypes binding
        ctx.push_splat(rows); ctx.sort(cam["view"])
        want = ctx.render(capi.make_params(cam["gs_mv"], cam["gs_proj"], w, h, focal_=cam["focal"]))
    assert np.array_equal(img, want)
    # ... and against the oracle itself (the node path's pixels were only HIP-vs-HIP)
    cs, cc, _ = oracle.pack(rows)
    ref, _, _ = oracle.render(cs, cc, idx, cam["gs_mv"].astype(np.float32), cam["gs_proj"].astype(np.float32), cam["focal"], w, h, want_f32=False)
    assert int(np.abs(img.astype(int) - ref.astype(int)).max()) <= 1
