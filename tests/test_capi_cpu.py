"""CPU tier for the C ABI: the library loads and exports every symbol include/gs_splat.h declares, the host-side
uniform producers and the PLY converter match the golden vectors captured from the reference JS, and -- because
there is NO CPU fallback -- creating a context without a GPU fails loudly."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, cases_of, load_case, pkg

capi = pkg("capi")


def _has_gpu():
    return os.path.exists("/dev/kfd")


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "gs_splat.h")).read()
    declared = sorted(set(re.findall(r"GS_API\s+[\w\s\*]+?\b(gs_\w+)\s*\(", hdr)))
    assert len(declared) >= 20
    L = capi.load()
    for name in declared:
        assert hasattr(L, name), name
    assert sorted(declared) == sorted(capi.EXPORTS)
    assert L.gs_version() == 0x000500


def test_no_cpu_fallback_without_gpu():
    if _has_gpu():
        pytest.skip("GPU present")
    with pytest.raises(capi.GsError) as ei:
        capi.Context(0)
    assert ei.value.code in (capi.E_NODEVICE, capi.E_HIP)
    assert "no CPU fallback" in ei.value.message or "HIP" in ei.value.message


@pytest.mark.parametrize("name", cases_of("camera"))
def test_uniform_producers_match_reference(name):
    c = load_case(name)
    mv = capi.model_view_matrix(c["cam_world"], c["obj_world"])
    assert np.array_equal(mv, c["gs_mv"])
    pr = capi.projection_matrix(c["proj"])
    assert np.array_equal(pr, c["gs_proj"])
    view, cut = capi.tick_uniforms(c["cam_world"], c["obj_world"], c.get("cutout_world"))
    assert np.array_equal(view.view(np.uint32), c["view"].view(np.uint32))
    if "cutout" in c:
        assert np.array_equal(cut.view(np.uint32), c["cutout"].view(np.uint32))
    assert capi.focal(pr, c["viewport"][1]) == c["focal"][0]


@pytest.mark.parametrize("name", cases_of("ply"))
def test_ply_to_splat_matches_reference(name):
    c = load_case(name)
    assert np.array_equal(capi.ply_to_splat(c["ply"]), c["rows"])


def _ply(props, nbytes, end=True):
    return (b"ply\nformat binary_little_endian 1.0\nelement vertex 1\n" +
            b"".join(b"property float %s\n" % n for n in props) + (b"end_header\n" if end else b"") + b"\0" * nbytes)


def test_ply_errors_carry_the_reference_messages(manifest):
    e = manifest["ply_errors"]["meta"]
    with pytest.raises(capi.GsError) as ei:
        capi.ply_to_splat(_ply([b"x"], 8, end=False))
    assert (ei.value.code, ei.value.message) == (capi.E_PLY_HEADER, e["no_end_header"])
    with pytest.raises(capi.GsError) as ei:
        capi.ply_to_splat(_ply([b"x", b"y", b"z", b"scale_0", b"scale_1", b"scale_2", b"opacity", b"rot_0", b"rot_1", b"rot_2"], 40))
    assert (ei.value.code, ei.value.message) == (capi.E_PLY_PROP, e["missing_rot_3"])
    with pytest.raises(capi.GsError) as ei:
        capi.ply_to_splat(_ply([b"x", b"y", b"z"], 12))
    assert (ei.value.code, ei.value.message) == (capi.E_PLY_PROP, e["missing_red"])
    with pytest.raises(capi.GsError) as ei:      # truncated body: DataView RangeError in the reference
        capi.ply_to_splat(_ply([b"x", b"y", b"z", b"red"], 3))
    assert ei.value.code == capi.E_PLY_DATA


def test_scaled_size_follows_pixel_ratio_rule():
    # pixelRatio / xrPixelRatio only apply when > 0 (index.js:10-15)
    assert capi.scaled_size(2064, 2208, 0.5) == (1032, 1104)
    assert capi.scaled_size(1920, 1080, 1.0) == (1920, 1080)
    assert capi.scaled_size(1920, 1080, 0.0) == (1920, 1080)
    assert capi.scaled_size(1920, 1080, -1.0) == (1920, 1080)
    assert capi.scaled_size(1001, 701, 0.75) == (750, 525)


def test_synth_inria_ply_roundtrip_via_loader():
    synth = pkg("synth")
    rows = synth.make_splat_rows(500, seed=9)
    back = capi.ply_to_splat(synth.rows_to_inria_ply(rows)).reshape(-1, 32)
    r = rows.reshape(-1, 32)
    # the loader re-orders by importance and re-normalises the quaternion; compare as sets keyed by position
    assert back.shape == r.shape
    kb = np.lexsort(back[:, 0:12].copy().view("<f4").T); kr = np.lexsort(r[:, 0:12].copy().view("<f4").T)
    b, rr = back[kb], r[kr]
    assert np.array_equal(b[:, 0:12], rr[:, 0:12])                       # positions exact
    assert np.array_equal(b[:, 24:27], rr[:, 24:27])                     # SH-DC -> RGB bytes exact
    assert np.max(np.abs(b[:, 27].astype(int) - rr[:, 27].astype(int))) <= 1
    assert np.allclose(b[:, 12:24].copy().view("<f4"), rr[:, 12:24].copy().view("<f4"), rtol=1e-6)


def test_partition_is_host_side_and_validates():
    """gs_partition needs no GPU: strips / XR eyes, and bad arguments are refused."""
    assert capi.partition([1920], 8)[3] == (0, 720, 960, 3)
    assert capi.partition([1032, 1032], 2) == [(0, 0, 1032, 0), (1, 0, 1032, 1)]
    for bad in ([0], [100, 100, 100], []):
        with pytest.raises(capi.GsError):
            capi.partition(bad, 2)
    with pytest.raises(capi.GsError):
        capi.partition([100], 0)
