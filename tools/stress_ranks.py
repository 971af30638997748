"""Stress of the multi-rank path on ONE GPU: `world` contexts joined by the in-process transport, one thread each (as separate
processes would be), all drawing the same random program -- gathered frames of random poses and sizes, synchronous and queued,
random roots, XR frames, option changes (lanes, pairing, enqueue threads, shared sorts), pushes, syncs -- while every frame the
root assembles synchronously is compared with what ONE reference context draws for the same pose.
usage: python tools/stress_ranks.py [seed] [world]      (STRESS_SECONDS, default 20)"""
import importlib, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
capi = importlib.import_module("aframe-gaussian-splatting_amd.capi"); synth = importlib.import_module("aframe-gaussian-splatting_amd.synth")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
world = int(sys.argv[2]) if len(sys.argv) > 2 else 3
SECONDS = float(os.environ.get("STRESS_SECONDS", "20"))
rows = synth.make_splat_rows(60000, seed=5).reshape(-1, 32)
SIZES = [(640, 360), (333, 190), (48, 64)]
cams = {s: [synth.index_html_camera(s[0], s[1], 15.0 * i, capi=capi) for i in range(24)] for s in SIZES}
rigs = [synth.xr_eye_cameras(30.0 * i, 0.25, capi=capi, cant_deg=float(i % 3) * 6.0) for i in range(12)]
P = lambda cam, **kw: capi.make_params(cam["gs_mv"], cam["gs_proj"], cam["vw"], cam["vh"], focal_=cam["focal"], **kw)

# the program: generated once, executed by every rank (a communicator wants the same calls in the same order everywhere)
g = np.random.Generator(np.random.PCG64(seed))
prog = []
n = N0 = 20000
for _ in range(int(os.environ.get("STRESS_OPS", "4000"))):
    r = g.random()
    if r < 0.55: prog.append(("async", SIZES[int(g.integers(0, 3))], int(g.integers(0, 24)), int(g.integers(0, world))))
    elif r < 0.70: prog.append(("sync", SIZES[int(g.integers(0, 3))], int(g.integers(0, 24)), int(g.integers(0, world)), bool(g.integers(0, 2))))
    elif r < 0.76: prog.append(("xr", int(g.integers(0, 12)), bool(g.integers(0, 2))))
    elif r < 0.82: prog.append(("gsync",))
    elif r < 0.86: prog.append(("opt", capi.OPT_PIPELINE_DEPTH, int(g.integers(1, 4))))
    elif r < 0.89: prog.append(("opt", capi.OPT_FRAME_BATCH, int(g.integers(1, 3))))
    elif r < 0.91: prog.append(("opt", capi.OPT_ENQUEUE_THREADS, int(g.integers(0, 2))))
    elif r < 0.95: prog.append(("opt", capi.OPT_SORT_SHARE, [0, 0, 300, 1000][int(g.integers(0, 4))]))
    elif r < 0.985 and n < rows.shape[0]:
        m = min(rows.shape[0], n + int(g.integers(1, 9000))); prog.append(("push", n, m)); n = m
    else:
        prog.append(("clear",)); n = N0

ctx = [capi.Context(0) for _ in range(world)]
ref = capi.Context(0); ref.set_option(capi.OPT_PIPELINE_DEPTH, 1)
uid = ctx[0].comm_unique_id(capi.TRANSPORT_INPROC)
barrier = threading.Barrier(world)
flags = [False] * world
errors, checked, frames = [], [0], [0]
deadline = time.time() + SECONDS


def agree(rank, f):
    flags[rank] = bool(f); barrier.wait(); out = any(flags); barrier.wait(); return out


def sync_all(rank, c):
    need = False
    try: c.sync()
    except capi.GsError as e:
        if e.code != capi.E_RETRY: raise
        need = True
    return agree(rank, need)


def body(rank):
    c = ctx[rank]
    try:
        c.push_splat(rows[:N0]); c.comm_init(uid, rank, world)
        if rank == 0: ref.push_splat(rows[:N0])
        barrier.wait()
        for op in prog:
            if agree(rank, time.time() > deadline): break             # (every rank stops at the same operation)
            if op[0] == "async":
                _, size, k, root = op; cam = cams[size][k]
                c.sort_gathered(cam["view"], None, P(cam)); c.render_gathered(P(cam), root=root, flags=capi.RENDER_ASYNC)
                if rank == 0: frames[0] += 1
            elif op[0] == "sync":
                _, size, k, root, flip = op; cam = cams[size][k]
                sync_all(rank, c)                                      # (a synchronous gathered frame behind queued ones: drain first, all ranks)
                fl = capi.RENDER_FLIP_Y if flip else 0
                c.sort_gathered(cam["view"], None, P(cam)); c.render_gathered(P(cam, flags=fl), root=root, flags=fl)
                if rank == root:
                    got = c.read_gathered(0)
                    ref.sort(cam["view"], None, want_indices=False); want = ref.render(P(cam))
                    assert np.array_equal(got, want[::-1] if flip else want), ("mono frame differs", size, k, root, flip)
                    checked[0] += 1
                barrier.wait()
            elif op[0] == "xr":
                _, k, asyn = op; l, r, head = rigs[k]
                if not asyn: sync_all(rank, c)
                c.sort_gathered(head["view"], None, [P(l), P(r)])
                c.render_gathered([P(l), P(r)], root=0, flags=capi.RENDER_ASYNC if asyn else 0)
                if not asyn:
                    if rank == 0:
                        ref.sort(head["view"], None, want_indices=False); wl, wr = ref.render_stereo(P(l), P(r))
                        assert np.array_equal(c.read_gathered(0), wl) and np.array_equal(c.read_gathered(1), wr), ("XR frame differs", k)
                        checked[0] += 1
                    barrier.wait()
            elif op[0] == "gsync": sync_all(rank, c)
            elif op[0] == "opt":
                sync_all(rank, c); c.set_option(op[1], op[2])
            elif op[0] == "push":
                sync_all(rank, c); c.push_splat(rows[op[1]:op[2]])
                if rank == 0: ref.push_splat(rows[op[1]:op[2]])
                barrier.wait()
            elif op[0] == "clear":
                sync_all(rank, c); c.clear(); c.push_splat(rows[:N0])
                if rank == 0: ref.clear(); ref.push_splat(rows[:N0])
                barrier.wait()
        sync_all(rank, c)
    except BaseException as e:      # noqa: BLE001
        errors.append((rank, repr(e))); barrier.abort()


ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
for t in ts: t.start()
for t in ts: t.join()
real = [e for e in errors if "BrokenBarrier" not in e[1]]
for c in ctx: c.close()
ref.close()
if real or errors:
    print("stress FAILED:", real or errors); sys.exit(1)
print("stress ok: world %d, seed %d: %d queued frames, %d synchronous frames checked against one context" % (world, seed, frames[0], checked[0]))
