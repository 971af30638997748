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
            b = [mg.strip_bounds(width, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == width
            for (a0, a1), (b0, b1) in zip(b[:-1], b[1:]):
                assert a1 == b0 and a1 % 16 == 0
            assert sum(mg.strip_widths(width, world)) == width
    assert mg.strip_widths(1920, 8) == [240] * 8 and mg.strip_widths(3840, 8) == [480] * 8
