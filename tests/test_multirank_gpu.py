"""GPU tier: the multi-rank path of the C library (csrc/gs_comm.hip) run with MORE THAN ONE RANK on a single GPU.

`north_star` splits the viewport into column strips over the GPUs of a node (XR: the eyes over the GPUs) and gathers them on a
root.  The driver's GPU box has one MI355X, so the ranks here are contexts of one process on that one device -- one thread
each, as separate processes would be -- joined through the library's in-process transport (GS_OPT_COMM_TRANSPORT = 1:
ncclSend / ncclRecv semantics behind the same five calls RCCL serves).  Everything else is the code that runs on a node:
gs_partition, gs_sort_gathered (the sort of a rank's strip), gs_render_gathered with its tickets, staging offsets, one receive
per foreign piece, k_assemble with several sources, the lanes' enqueue threads, frame pairing, XR eye -> rank.

Bar: the frame assembled on the root is BIT-IDENTICAL to gs_render's frame of one context (which the other GPU tests hold to
the oracle and to the reference's GLSL goldens), at world 2 / 3 / 8, for C2 (1 M @ 1920x1080), C4 (XR, world 2, also canted
eyes) and C5's size (20 M @ 3840x2160, world 8)."""
import ctypes
import threading

import numpy as np
import pytest

from conftest import cached_rows, pkg

pytestmark = pytest.mark.gpu
capi = pkg("capi")
synth = pkg("synth")


def _params(cam, **kw):
    return capi.make_params(cam["gs_mv"], cam["gs_proj"], cam["vw"], cam["vh"], focal_=cam["focal"], **kw)


class Ranks:
    """`world` contexts on device 0 joined by an in-process communicator; run(fn) calls fn(rank, ctx, self) on one thread per
    rank and re-raises the first failure.  agree(flag) = logical OR over the ranks (what a launcher does with an all-reduce)."""

    def __init__(self, world, rows, depth=None, batch=None):
        self.world = world
        self.ctx = [capi.Context(0) for _ in range(world)]
        self.barrier = threading.Barrier(world)
        self._flags = [False] * world
        uid = self.ctx[0].comm_unique_id(capi.TRANSPORT_INPROC)
        r32 = np.asarray(rows).reshape(-1, 32)

        def setup(rank, c, _):
            for o in range(0, r32.shape[0], 1 << 22):
                c.push_splat(r32[o:o + (1 << 22)])
            if depth:
                c.set_option(capi.OPT_PIPELINE_DEPTH, depth)
            if batch:
                c.set_option(capi.OPT_FRAME_BATCH, batch)
            c.comm_init(uid, rank, world)
        self.run(setup)

    def run(self, fn):
        errs = [None] * self.world

        def body(rank):
            try:
                fn(rank, self.ctx[rank], self)
            except BaseException as e:          # noqa: BLE001 -- reported below, on the test's thread
                errs[rank] = e
                self.barrier.abort()
        ts = [threading.Thread(target=body, args=(r,)) for r in range(self.world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        self.barrier.reset()
        real = [e for e in errs if e is not None and not isinstance(e, threading.BrokenBarrierError)]
        if real:
            raise real[0]
        if any(errs):
            raise [e for e in errs if e is not None][0]

    def agree(self, rank, flag):
        self._flags[rank] = bool(flag)
        self.barrier.wait()
        out = any(self._flags)
        self.barrier.wait()
        return out

    def sync_all(self, rank, c):
        """gs_sync on this rank; True if ANY rank was asked to render its frames again (GS_E_RETRY)"""
        need = False
        try:
            c.sync()
        except capi.GsError as e:
            if e.code != capi.E_RETRY:
                raise
            need = True
        return self.agree(rank, need)

    def close(self):
        for c in self.ctx:
            c.close()


class DevBuf:
    def __init__(self, nbytes):
        self.hip = capi.hip_runtime()
        self.p = ctypes.c_void_p()
        self.n = nbytes
        assert self.hip.hipMalloc(ctypes.byref(self.p), ctypes.c_size_t(nbytes)) == 0

    def zero(self):
        assert self.hip.hipMemset(self.p, 0, ctypes.c_size_t(self.n)) == 0

    def read(self, shape):
        out = np.empty(shape, np.uint8)
        assert out.nbytes == self.n
        assert self.hip.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), self.p, ctypes.c_size_t(self.n), 2) == 0
        return out

    def free(self):
        self.hip.hipFree(self.p)


def _single_frames(rows, cams, views_of):
    """what one context draws: the reference images of the test"""
    out = []
    with capi.Context(0) as c:
        r32 = np.asarray(rows).reshape(-1, 32)
        for o in range(0, r32.shape[0], 1 << 22):
            c.push_splat(r32[o:o + (1 << 22)])
        for cam in cams:
            c.sort(cam["view"], cam.get("cutout"))
            out.append([c.render(p) for p in views_of(cam)])
    return out


@pytest.fixture(scope="module")
def rows_small():
    return synth.make_splat_rows(30000, seed=77)


@pytest.fixture(scope="module")
def rows_1m():
    return synth.make_splat_rows(synth.N_TRAIN)


def _gathered_mono(world, rows, w, h, yaws, roots, n_async, depth, batch, strip_sort=True):
    cams = [synth.index_html_camera(w, h, y, capi=capi) for y in yaws]
    want = [f[0] for f in _single_frames(rows, cams, lambda cam: [_params(cam)])]
    R = Ranks(world, rows, depth=depth, batch=batch)
    try:
        for root in roots:
            bufs = [DevBuf(w * h * 4) for _ in cams]
            got_sync = {}

            def body(rank, c, R):
                # synchronous frames: the assembled image is the context's own, read with gs_read_gathered
                for k, cam in enumerate(cams[:2]):
                    flip = capi.RENDER_FLIP_Y if k == 1 else 0
                    c.sort_gathered(cam["view"], None, _params(cam))
                    c.render_gathered(_params(cam, flags=flip), root=root, flags=flip)
                    if rank == root:
                        got_sync[k] = c.read_gathered(0, w, h)
                    R.barrier.wait()
                # asynchronous frames, several in flight on the lanes (paired when batch == 2), into the root's device buffers
                for attempt in range(6):
                    if rank == root:
                        for b in bufs:
                            b.zero()
                    R.barrier.wait()
                    for i in range(n_async):
                        k = i % len(cams)
                        if strip_sort:
                            c.sort_gathered(cams[k]["view"], None, _params(cams[k]))
                        else:
                            c.sort(cams[k]["view"], want_indices=False)
                        c.render_gathered(_params(cams[k]), root=root, device_frames=[bufs[k].p.value] if rank == root else None,
                                          flags=capi.RENDER_ASYNC)
                    if not R.sync_all(rank, c):
                        break
                    assert attempt < 5, "frames kept asking for a re-render"
            R.run(body)
            assert np.array_equal(got_sync[0], want[0]), (world, root, "sync")
            assert np.array_equal(got_sync[1], want[1][::-1]), (world, root, "sync flipped")
            for k, b in enumerate(bufs):
                assert np.array_equal(b.read((h, w, 4)), want[k]), (world, root, "async", k)
                b.free()
    finally:
        R.close()


@pytest.mark.parametrize("world,depth,batch", [(2, 3, 1), (2, 3, 2), (3, 3, 2), (8, 2, 1), (8, 3, 2)])
def test_strips_of_several_ranks_assemble_to_the_single_context_frame(rows_small, world, depth, batch):
    """small scene, 640x360: world 2 / 3 / 8, root 0 and the last rank, synchronous + flipped + pipelined (three lanes: the
    gathers of frames enqueued by different threads are issued in ticket order) + paired frames"""
    _gathered_mono(world, rows_small, 640, 360, (0.0, 75.0, 150.0, 225.0, 300.0), (0, world - 1), n_async=20, depth=depth, batch=batch)


@pytest.mark.parametrize("world,roots", [(2, (0, 1)), (3, (1,)), (8, (0,))])
def test_paired_gathered_frames_with_reordered_delivery(rows_small, world, roots, monkeypatch):
    """RCCL orders the messages of one PAIR of ranks and nothing else.  With GS_COMM_TEST_JITTER_US the in-process transport holds every
    post back by a pseudo-random time (up to 1.5 ms, several frames' worth): the pieces of a frame reach the root in any order across its
    peers, a peer's piece of the next frame before another peer's piece of this one -- with frames PAIRED (two gathers behind every shared
    chain of launches, GS_OPT_FRAME_BATCH = 2) and three lanes deep, i.e. six frames and their gathers in flight per rank (VERDICT r5
    "next" #9).  The assembled frames must still be the single-context frames, bit for bit."""
    monkeypatch.setenv("GS_COMM_TEST_JITTER_US", "1500")
    _gathered_mono(world, rows_small, 640, 360, (0.0, 75.0, 150.0, 225.0, 300.0), roots, n_async=24, depth=3, batch=2)


def test_more_ranks_than_tile_columns_and_whole_sorts(rows_small):
    """48 pixels = 3 tile columns over 8 ranks: five ranks own nothing, send nothing, and still keep the ticket sequence;
    plain gs_sort instead of the strip sort"""
    _gathered_mono(8, rows_small, 48, 64, (10.0, 200.0), (0, 5), n_async=12, depth=3, batch=1, strip_sort=False)


@pytest.mark.parametrize("world", [2, 8])
def test_c2_headline_frame_over_ranks(rows_1m, world):
    """C2: 1 M splats @ 1920x1080 over 2 and 8 ranks, frames paired and three lanes deep"""
    _gathered_mono(world, rows_1m, 1920, 1080, (21.0, 180.0, 303.0), (0,), n_async=18, depth=3, batch=2)


@pytest.mark.parametrize("world,cant", [(2, 0.0), (2, 12.0), (3, 12.0), (8, 0.0)])
def test_c4_xr_eyes_over_ranks(rows_1m, world, cant):
    """C4: XR stereo 2 x 1032x1104, ONE shared sort from the head camera (index.js:441), eye k -> rank k at world 2, the eyes
    split in strips over more ranks; with canted eyes an eye's own view direction differs from the head camera's that sorts
    (the strip test of gs_sort_for must use the eye's matrix)"""
    rigs = [synth.xr_eye_cameras(y, 0.5, capi=capi, cant_deg=cant) for y in (40.0, 200.0)]
    want = []
    with capi.Context(0) as c:
        c.push_splat(rows_1m)
        for l, r, head in rigs:
            c.sort(head["view"])
            want.append(c.render_stereo(_params(l), _params(r)))
    w, h = rigs[0][0]["vw"], rigs[0][0]["vh"]
    R = Ranks(world, rows_1m, depth=3, batch=2)
    try:
        bufs = [[DevBuf(w * h * 4), DevBuf(w * h * 4)] for _ in rigs]
        got_sync = {}

        def body(rank, c, R):
            l, r, head = rigs[0]
            c.sort_gathered(head["view"], None, [_params(l), _params(r)])
            c.render_gathered([_params(l), _params(r)], root=0)
            if rank == 0:
                got_sync[0] = (c.read_gathered(0), c.read_gathered(1))
            R.barrier.wait()
            for attempt in range(6):
                for i in range(12):
                    k = i % len(rigs)
                    l, r, head = rigs[k]
                    c.sort_gathered(head["view"], None, [_params(l), _params(r)])
                    c.render_gathered([_params(l), _params(r)], root=0, flags=capi.RENDER_ASYNC,
                                      device_frames=[bufs[k][0].p.value, bufs[k][1].p.value] if rank == 0 else None)
                if not R.sync_all(rank, c):
                    break
                assert attempt < 5
        R.run(body)
        assert np.array_equal(got_sync[0][0], want[0][0]) and np.array_equal(got_sync[0][1], want[0][1])
        for k in range(len(rigs)):
            for e in range(2):
                assert np.array_equal(bufs[k][e].read((h, w, 4)), want[k][e]), (world, cant, k, e)
                bufs[k][e].free()
    finally:
        R.close()


def test_c5_size_twenty_million_splats_4k_over_eight_ranks():
    """C5's size: 20 M splats @ 3840x2160 in eight column strips of 480 pixels (8-byte pair records, the long radix geometry,
    near-only strip sorts once the share has settled), gathered on rank 0 -- against one context's frame"""
    n = 20 * (1 << 20)
    rows = cached_rows("make_splat_rows_fast", n)
    w, h = 3840, 2160
    cams = [synth.index_html_camera(w, h, y, capi=capi) for y in (33.0, 35.0)]
    want = [f[0] for f in _single_frames(rows, cams, lambda cam: [_params(cam)])]
    R = Ranks(8, rows, depth=1)
    try:
        bufs = [DevBuf(w * h * 4) for _ in cams]
        stats = {}

        def body(rank, c, R):
            for attempt in range(6):
                for i in range(40):                                  # (16 clean frames switch round 1 off and near-only sorts on)
                    k = i % len(cams)
                    c.sort_gathered(cams[k]["view"], None, _params(cams[k]))
                    c.render_gathered(_params(cams[k]), root=0, device_frames=[bufs[k].p.value] if rank == 0 else None, flags=capi.RENDER_ASYNC)
                    if i % 4 == 3 and R.sync_all(rank, c):
                        break
                else:
                    stats[rank] = c.stats()
                    break
                assert attempt < 5
        R.run(body)
        for k, b in enumerate(bufs):
            assert np.array_equal(b.read((h, w, 4)), want[k]), k
            b.free()
        print("C5 world 8: sort records / kept splats per rank:", [(s["sort_records"], s["n_sorted"]) for _, s in sorted(stats.items())])
    finally:
        R.close()


def _shared_sort_frames(world, rows, w, h, yaws, permille, n_async, depth=3, batch=1, sync_every=6):
    """GS_OPT_SORT_SHARE: frame f is sorted by rank f mod world alone and its order sent to the others; the assembled frames
    must still equal one context's, whether a frame is covered by the exchanged part of the order or falls back to a local sort"""
    cams = [synth.index_html_camera(w, h, y, capi=capi) for y in yaws]
    want = [f[0] for f in _single_frames(rows, cams, lambda cam: [_params(cam)])]
    R = Ranks(world, rows, depth=depth, batch=batch)
    try:
        bufs = [DevBuf(w * h * 4) for _ in cams]
        got_sync, stats = {}, {}

        def body(rank, c, R):
            c.set_option(capi.OPT_SORT_SHARE, permille)
            for k, cam in enumerate(cams[:2]):                         # synchronous frames (rank 0 owns the first, rank 1 the second)
                c.sort_gathered(cam["view"], None, _params(cam))
                c.render_gathered(_params(cam), root=0)
                if rank == 0:
                    got_sync[k] = c.read_gathered(0, w, h)
                R.barrier.wait()
            for attempt in range(8):
                for i in range(n_async):
                    k = i % len(cams)
                    c.sort_gathered(cams[k]["view"], None, _params(cams[k]))
                    c.render_gathered(_params(cams[k]), root=0, device_frames=[bufs[k].p.value] if rank == 0 else None, flags=capi.RENDER_ASYNC)
                    if i % sync_every == sync_every - 1:
                        if R.sync_all(rank, c):
                            break
                        st = c.stats()                                 # (was the last collected frame drawn from a partial order?)
                        if 0 < st["sort_records"] < st["n_sorted"]:
                            stats[rank] = stats.get(rank, 0) + 1
                else:
                    if not R.sync_all(rank, c):
                        break
                assert attempt < 7, "frames kept asking for a re-render"
        R.run(body)
        for k in (0, 1):
            assert np.array_equal(got_sync[k], want[k]), (world, "sync", k)
        for k, b in enumerate(bufs):
            assert np.array_equal(b.read((h, w, 4)), want[k]), (world, permille, "async", k)
            b.free()
        return stats
    finally:
        R.close()


@pytest.mark.parametrize("world,permille,batch", [(2, 1000, 1), (3, 400, 2), (8, 1000, 1)])
def test_ranks_take_turns_sorting_small_scene(rows_small, world, permille, batch):
    """the small scene never saturates its tiles (one binning round over everything): with 1000 permille the whole order is
    exchanged and covers every frame; with 400 the exchanged part does not cover a frame and every rank sorts again locally"""
    _shared_sort_frames(world, rows_small, 640, 360, (0.0, 75.0, 150.0, 225.0, 300.0), permille, n_async=18, batch=batch)


@pytest.mark.parametrize("world", [2, 8])
def test_ranks_take_turns_sorting_c2(rows_1m, world):
    """C2: once the share of splats binned first has settled (~16 %), the exchanged 30 % of the order covers every frame"""
    st = _shared_sort_frames(world, rows_1m, 1920, 1080, (21.0, 22.0, 23.0), 300, n_async=216)   # (round 1 stays on until the share has failed once)
    print("C2 shared sort, world %d: syncs whose last frame was drawn from the exchanged (partial) order, per rank:" % world, sorted(st.items()))
    assert len(st) == world and all(v > 0 for v in st.values()), st              # every rank drew frames from the exchanged part alone


def test_ranks_take_turns_sorting_c5_size():
    """C5's size over eight ranks: 20 M splats @ 3840x2160, 3 % of the order exchanged (629 146 splats, 5.3 MB per peer)"""
    n = 20 * (1 << 20)
    rows = cached_rows("make_splat_rows_fast", n)
    st = _shared_sort_frames(8, rows, 3840, 2160, (33.0, 35.0), 30, n_async=180, depth=1, sync_every=2)   # (the share shrinks 10 % per collected sync
                                                                                                            # until it fails once; then 32 frames of hold)
    print("C5 shared sort, world 8: syncs whose last frame was drawn from the exchanged (partial) order, per rank:", sorted(st.items()))
    assert len(st) == 8 and all(v > 0 for v in st.values()), st


def test_a_rank_that_never_sends_fails_the_frame_instead_of_hanging(rows_small, monkeypatch):
    """in-process transport: a receive whose sender never shows up gives up (GS_COMM_TIMEOUT_S) and the frame fails with
    GS_E_HIP; the ticket sequence moves on, so the next frame -- with the peer present -- is complete again"""
    monkeypatch.setenv("GS_COMM_TIMEOUT_S", "2")
    cam = synth.index_html_camera(320, 180, 30.0, capi=capi)
    a, b = capi.Context(0), capi.Context(0)
    try:
        for c in (a, b):
            c.push_splat(rows_small)
        uid = a.comm_unique_id(capi.TRANSPORT_INPROC)
        a.comm_init(uid, 0, 2); b.comm_init(uid, 1, 2)
        a.sort_gathered(cam["view"], None, _params(cam))
        with pytest.raises(capi.GsError) as ei:
            a.render_gathered(_params(cam), root=0)                  # rank 1 never renders this frame
        assert ei.value.code == capi.E_HIP and "posted nothing" in ei.value.message
    finally:
        a.close(); b.close()


# ---------------------------------------------------------------- one host process, several "devices" (gs_create_multi)

@pytest.mark.parametrize("ndev", [1, 2, 3, 8])
def test_multi_one_process_drives_all_devices(rows_small, ndev):
    """gs_create_multi over `ndev` contexts on the one GPU, ONE caller thread: host-direct frames (every context copies its strip
    into the caller's page-locked image), device frames gathered on devices[0] through the in-process transport, synchronous
    and pipelined, mono and XR -- all bit-identical to one context's gs_render"""
    w, h = 640, 360
    cams = [synth.index_html_camera(w, h, y, capi=capi) for y in (0.0, 90.0, 180.0, 270.0)]
    want = [f[0] for f in _single_frames(rows_small, cams, lambda cam: [_params(cam)])]
    l, r, head = synth.xr_eye_cameras(40.0, 0.25, capi=capi)
    with capi.Context(0) as c:
        c.push_splat(rows_small)
        c.sort(head["view"]); wl, wr = c.render_stereo(_params(l), _params(r))
    frames = [capi.host_frame(h, w) for _ in cams]
    with capi.Multi([0] * ndev) as m:
        assert m.count() == 0
        half = (rows_small.size // 64) * 32
        m.push_splat(rows_small[:half]); m.push_splat(rows_small[half:])     # progressive ingest (index.js:279-298)
        assert m.count() == rows_small.size // 32
        # synchronous host-direct frames
        for cam, (fr, _), wnt in zip(cams, frames, want):
            fr[:] = 0
            m.sort(cam["view"], None, _params(cam))
            m.render(_params(cam), fr)
            assert np.array_equal(fr, wnt)
        # flipped rows
        m.sort(cams[1]["view"], None, _params(cams[1]))
        m.render(_params(cams[1], flags=capi.RENDER_FLIP_Y), frames[1][0], flags=capi.RENDER_FLIP_Y)
        assert np.array_equal(frames[1][0], want[1][::-1])
        # asynchronous host-direct frames: four in flight, paired on the lanes
        m.set_option(capi.OPT_FRAME_BATCH, 2)
        for attempt in range(6):
            for fr, _ in frames:
                fr[:] = 0
            for rep in range(3):
                for cam, (fr, _) in zip(cams, frames):
                    m.sort(cam["view"], None, _params(cam))
                    m.render(_params(cam), fr, flags=capi.RENDER_ASYNC)
            try:
                m.sync()
                break
            except capi.GsError as e:
                assert e.code == capi.E_RETRY and attempt < 5
        for (fr, _), wnt in zip(frames, want):
            assert np.array_equal(fr, wnt)
        # device frames, gathered on devices[0]
        m.sort(cams[2]["view"], None, _params(cams[2]))
        m.render_device(_params(cams[2]))
        assert np.array_equal(m.read(0, w, h), want[2])
        # XR: two eyes, the head camera's sort, eye k on device k from two devices on
        el, er = capi.host_frame(l["vh"], l["vw"]), capi.host_frame(r["vh"], r["vw"])
        m.sort(head["view"], None, [_params(l), _params(r)])
        m.render([_params(l), _params(r)], [el[0], er[0]])
        assert np.array_equal(el[0], wl) and np.array_equal(er[0], wr)
        m.sort(head["view"], None, [_params(l), _params(r)])
        m.render_device([_params(l), _params(r)])
        assert np.array_equal(m.read(0, l["vw"], l["vh"]), wl) and np.array_equal(m.read(1, r["vw"], r["vh"]), wr)
        m.sync()
        st = [m.ctx_stats(i) for i in range(ndev)]
        assert all(s["n_splats"] == rows_small.size // 32 for s in st)
        el[1].free(); er[1].free()
    for _, o in frames:
        o.free()


def test_multi_rejects_bad_arguments():
    with pytest.raises(capi.GsError):
        capi.Multi([])
    with pytest.raises(capi.GsError):
        capi.Multi([99])
    cam = synth.index_html_camera(64, 64, 0.0, capi=capi)
    with capi.Multi([0, 0]) as m:
        fr = np.zeros((64, 64, 4), np.uint8)
        with pytest.raises(capi.GsError):
            m.render([_params(cam)] * 3, [fr] * 3)
        with pytest.raises(capi.GsError):
            m.render(_params(cam), fr, flags=capi.RENDER_COUNT_FRAGS)
        m.sort(cam["view"], None, _params(cam))                 # nothing pushed: every context answers like the reference's empty worker
        m.render(_params(cam), fr)
        assert (fr[..., :3] == 0).all() and (fr[..., 3] == 255).all()
