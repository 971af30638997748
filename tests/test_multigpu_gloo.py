"""CPU tier for the N > 1 path: world_size 2 and 3 over gloo.  Each rank produces its column strip with the CPU
oracle (test infrastructure standing in for the HIP renderer, which needs a GPU), the strips are gathered with the
product's multigpu.gather_strips and the assembled frame must equal the single-process full-frame render bit for
bit -- the partition (tile-aligned, ragged last strip), padding and assembly logic are what is under test."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, pkg

W, H = 200, 72


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _scene():
    from oracle import oracle
    synth = pkg("synth")
    rows = synth.make_splat_rows(1500, seed=21)
    cs, cc, mats = oracle.pack(rows)
    cam = synth.index_html_camera(W, H, 30.0)
    idx = oracle.sort(mats, cam["view"])
    return oracle, cs, cc, idx, cam


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mg = pkg("multigpu")
    oracle, cs, cc, idx, cam = _scene()
    x0, x1 = mg.strip_bounds(W, world, rank)
    flat = torch.zeros(mg.strip_buffer_bytes(W, H, world), dtype=torch.uint8)
    if x1 > x0:
        u8, _, _ = oracle.render(cs, cc, idx, cam["gs_mv"].astype(np.float32), cam["gs_proj"].astype(np.float32), cam["focal"],
                                 W, H, x0=x0, x1=x1, want_f32=False)
        flat[: u8.size] = torch.from_numpy(u8.reshape(-1))
    frame = mg.gather_strips(flat, W, H, dist)
    if rank == 0:
        q.put(frame.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_strip_gather_reassembles_the_frame(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    frame = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    oracle, cs, cc, idx, cam = _scene()
    full, _, _ = oracle.render(cs, cc, idx, cam["gs_mv"].astype(np.float32), cam["gs_proj"].astype(np.float32), cam["focal"], W, H,
                               want_f32=False)
    assert frame.shape == (H, W, 4)
    assert np.array_equal(frame, full)


def test_strip_bounds_are_tile_aligned_and_cover():
    mg = pkg("multigpu")
    for width in (1920, 3840, 1032, 200, 17):
        for world in (1, 2, 3, 4, 8):
            b = [s for s in (mg.strip_bounds(width, world, r) for r in range(world)) if s[1] > s[0]]   # (idle ranks: empty strip)
            assert b[0][0] == 0 and b[-1][1] == width
            for (a0, a1), (b0, b1) in zip(b[:-1], b[1:]):
                assert a1 == b0 and a1 % 16 == 0
            assert sum(mg.strip_widths(width, world)) == width
    assert mg.strip_widths(1920, 8) == [240] * 8 and mg.strip_widths(3840, 8) == [480] * 8


# ---------------------------------------------------------------- XR: one eye per GPU (BASELINE C4)

XW, XH = 129, 138                      # the C4 eye shape (1032 x 1104) scaled by 1/8: a ragged last tile column


def _xr_scene():
    from oracle import oracle
    synth = pkg("synth")
    rows = synth.make_splat_rows(1500, seed=23)
    cs, cc, mats = oracle.pack(rows)
    l, r, head = synth.xr_eye_cameras(25.0, 0.5)
    for cam in (l, r, head):           # same frusta, small framebuffer
        cam["vw"], cam["vh"] = XW, XH
        cam["focal"] = float((XH / 2.0) * abs(cam["gs_proj"][5]))
    idx = oracle.sort(mats, head["view"])              # ONE sort from the head camera, shared by both eyes (index.js:441)
    return oracle, cs, cc, idx, (l, r)


def _xr_worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mg = pkg("multigpu")
    oracle, cs, cc, idx, eyes = _xr_scene()
    flat = torch.zeros(mg.piece_buffer_bytes([XW, XW], [XH, XH], world), dtype=torch.uint8)
    o = 0
    for v, x0, x1, owner in mg.partition([XW, XW], world):
        if owner != rank:
            continue
        cam = eyes[v]
        u8, _, _ = oracle.render(cs, cc, idx, cam["gs_mv"].astype(np.float32), cam["gs_proj"].astype(np.float32), cam["focal"],
                                 XW, XH, x0=x0, x1=x1, want_f32=False)
        flat[o: o + u8.size] = torch.from_numpy(u8.reshape(-1)); o += u8.size
    frames = mg.gather_views(flat, [XW, XW], [XH, XH], dist)
    if rank == 0:
        q.put([f.numpy().copy() for f in frames])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_xr_one_eye_per_rank_gathers_two_images(world):
    """C4: eye k -> rank k (world 2), the shared head-camera order on every rank, two images back on rank 0 -- bit-equal to
    the two eyes rendered by one process.  World 3: eye 0 split in two strips over ranks 0-1, eye 1 on rank 2."""
    mg = pkg("multigpu")
    parts = mg.partition([XW, XW], world)
    if world == 2:
        assert parts == [(0, 0, XW, 0), (1, 0, XW, 1)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_xr_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    frames = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    oracle, cs, cc, idx, eyes = _xr_scene()
    for v in range(2):
        cam = eyes[v]
        full, _, _ = oracle.render(cs, cc, idx, cam["gs_mv"].astype(np.float32), cam["gs_proj"].astype(np.float32), cam["focal"],
                                   XW, XH, want_f32=False)
        assert frames[v].shape == (XH, XW, 4) and np.array_equal(frames[v], full)
    assert not np.array_equal(frames[0], frames[1])


def test_partition_rules():
    mg = pkg("multigpu")
    # one view: tile-aligned strips that cover the width; more ranks than tile columns leaves ranks idle
    for width in (1920, 3840, 1032, 200, 17):
        for world in (1, 2, 3, 4, 8):
            p = mg.partition([width], world)
            assert p[0][1] == 0 and p[-1][2] == width and all(v == 0 for v, _, _, _ in p)
            for (_, _, a1, _), (_, b0, _, _) in zip(p[:-1], p[1:]):
                assert a1 == b0 and a1 % 16 == 0
            assert [o for _, _, _, o in p] == sorted(set(o for _, _, _, o in p))
    assert len(mg.partition([17], 8)) == 2 and mg.strip_bounds(17, 8, 5) == (17, 17)
    # two views (XR): world 1 renders both; world 2 one eye per rank; world 8 four strips per eye
    assert mg.partition([1032, 1032], 1) == [(0, 0, 1032, 0), (1, 0, 1032, 0)]
    assert mg.partition([1032, 1032], 2) == [(0, 0, 1032, 0), (1, 0, 1032, 1)]
    p8 = mg.partition([1032, 1032], 8)
    assert [o for _, _, _, o in p8] == list(range(8)) and [v for v, _, _, _ in p8] == [0] * 4 + [1] * 4
    assert p8[3][2] == 1032 and p8[4][1] == 0 and p8[7][2] == 1032
