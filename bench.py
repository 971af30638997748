#!/usr/bin/env python3
"""bench.py -- frames/s of the hot path (sort -> project -> bin -> blend [-> strip gather]) on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one frame of the 120-frame benchmark orbit (entity yaw 0..360 deg, index.html:13 pose) over the
synthetic train.splat-shaped scene (N = 1,048,576 splats, 1920x1080; BASELINE.json configs[1]): a bit-exact
gs_sort for the frame's view vector followed by a full render, splat data already resident in HBM.  With N > 1 GPUs
the viewport is split into tile-aligned column strips (splat buffer replicated, sort + project replicated on every
GPU), each rank renders its strip into a device tensor and the strips are gathered to rank 0 over RCCL.

Rank 0 prints ONE JSON line.  `value` = frames/s of the whole job; extra keys give Msplat-frags/s (reference-
equivalent fragments, counted untimed by the GS_RENDER_COUNT_FRAGS variant), the per-stage GPU times from HIP
events on the library's stream, the roofline of the dominant kernel and the CPU baseline.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "aframe-gaussian-splatting_amd"
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6290 GB/s measured copy peak
ORBIT_FRAMES = 120
W, H = 1920, 1080


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=480)
    ap.add_argument("--warmup", type=int, default=24)
    ap.add_argument("--splats", type=int, default=None, help="override N (default: train.splat-shaped 1,048,576)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--size", default=None, help="override the viewport, e.g. 3840x2160 (default 1920x1080)")
    ap.add_argument("--cutout", action="store_true", help="cutout-demo.html pose with the cutoutEntity box (config C3)")
    args = ap.parse_args()
    # stdout carries exactly ONE JSON line: whatever native libraries (RCCL's version banner, HIP warnings) write to file
    # descriptor 1 during the run goes to stderr instead; the line is written to the real stdout at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    global W, H
    if args.size:
        W, H = (int(v) for v in args.size.lower().split("x"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = torch = None
    # N > 1: one process per GPU over RCCL.  GS_BENCH_TORCH=1 forces the same code path (torch stream, device strip
    # tensor, RCCL gather) in a single process so it can be exercised on a 1-GPU box.
    multi = world > 1 or os.environ.get("GS_BENCH_TORCH") == "1"
    if multi:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("NCCL_DEBUG", "WARN")        # keep RCCL's version banner off stdout (ONE JSON line)
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    capi = importlib.import_module(PKG + ".capi")
    synth = importlib.import_module(PKG + ".synth")
    mg = importlib.import_module(PKG + ".multigpu")

    n_splats = args.splats or synth.N_TRAIN
    rows = synth.make_splat_rows(n_splats)
    ctx = capi.Context(local_rank)
    ctx.push_splat(rows)
    depth = int(os.environ.get("GS_BENCH_DEPTH", "0"))       # experiment knob: frames in flight (library default 3)
    if depth and not multi:
        ctx.set_option(capi.OPT_PIPELINE_DEPTH, depth)

    # tile-aligned column strips
    x0, x1 = mg.strip_bounds(W, world, rank)

    pose = synth.cutout_demo_camera if args.cutout else synth.index_html_camera
    cams = [pose(W, H, 360.0 * i / ORBIT_FRAMES, capi=capi) for i in range(ORBIT_FRAMES)]
    params = [capi.make_params(c["gs_mv"], c["gs_proj"], W, H, x0=x0, x1=x1, focal_=c["focal"]) for c in cams]
    strip = None
    LANES = 3                                                # frames in flight (the library's default pipeline depth)
    LAG = 2                                                  # a frame's gather is queued after LAG more frames were handed over
    strips, gathereds, lane_streams, owed = [], [], {}, []
    if multi:
        # One strip buffer per frame in flight.  The RCCL gather of a frame is queued on the SAME stream as the frame's
        # kernels (the library's pipeline lane, wrapped as a torch ExternalStream; c10d runs a blocking-style collective on
        # the current stream), so it is ordered after the blend and before the frame that reuses the lane and the buffer --
        # no cross-stream event anywhere (each one stalls the pipeline for ~50 us here), and the gather of frame k overlaps
        # the sort/render of frames k+1, k+2 on the other lanes.  The lane's worker thread enqueues the frame; this thread
        # queues the gather of frame k-LAG after handing over frame k, when that worker has long finished (LAG < LANES, so
        # the gather still precedes the next frame of its lane).
        for _ in range(LANES):
            strips.append(torch.zeros(mg.strip_buffer_bytes(W, H, world), dtype=torch.uint8, device="cuda"))   # tight H x sw x 4 rows
            gathereds.append([torch.zeros_like(strips[-1]) for _ in range(world)] if rank == 0 else None)
        strip = strips[0]
    last_frame = [None]

    def gather_owed(keep):
        while len(owed) > keep:
            lane, b = owed.pop(0)
            sp = ctx.lane_stream(lane)
            if sp not in lane_streams:
                lane_streams[sp] = torch.cuda.ExternalStream(sp)
            with torch.cuda.stream(lane_streams[sp]):
                last_frame[0] = mg.gather_strips(strips[b], W, H, dist, gathereds[b])   # RCCL gather + row-major frame on rank 0

    def frame(i, flags=0):
        k = i % ORBIT_FRAMES
        p = params[k]
        p.flags = flags
        ctx.sort(cams[k]["view"], cams[k]["cutout"], want_indices=False)
        if not multi:
            ctx.render_device(p, None)
            return
        b = ctx.frame_lane()                                 # the strip buffer belongs to the lane: its stream orders gather and reuse
        ctx.render_device(p, strips[b].data_ptr())
        owed.append((b, b))
        gather_owed(LAG if (flags & capi.RENDER_ASYNC) else 0)

    def sync():
        """Drain the stream; True if the library asks for the frames since the last sync to be rendered again
        (GS_E_RETRY) -- agreed on by all ranks so that their control flow stays identical."""
        need = 0
        if multi:
            gather_owed(0)
        try:
            ctx.sync()                                       # collects status/statistics of the asynchronous frames
        except capi.GsError as e:
            if e.code != capi.E_RETRY:
                raise
            need = 1
        if multi:
            torch.cuda.synchronize()
            t = torch.tensor([need], dtype=torch.int32, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            need = int(t.item())
            dist.barrier()
            torch.cuda.synchronize()
        return bool(need)

    # reference-equivalent fragments per orbit frame (untimed; no early termination)
    frames_used = sorted(set((args.warmup + i) % ORBIT_FRAMES for i in range(args.steps)))
    frags = {}
    for k in frames_used:
        frame(k, capi.RENDER_COUNT_FRAGS)
        frags[k] = ctx.stats()["n_frags"]                    # (counting renders are synchronous)
    if multi:
        t = torch.tensor([frags[k] for k in frames_used], dtype=torch.int64, device="cuda")
        dist.all_reduce(t)
        frags = dict(zip(frames_used, t.tolist()))

    # frames are enqueued back to back like the reference's render loop (GS_RENDER_ASYNC); gs_sync() at the end of
    # the region collects their status (an overflowing pair buffer would surface there as GS_E_RETRY)
    # list entries the blend really stages before its tiles saturate (untimed measurement aid, sampled over 8 poses of the
    # region, rank 0's strip): the byte count behind `roofline.achieved_touched`
    staged_per_frame = None
    if rank == 0:
        if multi:                                            # no gather may still be reading the strip buffers
            gather_owed(0)
            torch.cuda.synchronize()
        ctx.set_option(capi.OPT_RECORD_STAGED, 1)
        ctx.set_option(capi.OPT_NEAR_PERMILLE, 1000)
        tot = []
        ntl = ((x1 - x0 + 15) // 16) * ((H + 15) // 16)
        for k in frames_used[:: max(1, len(frames_used) // 8)][:8]:
            ctx.sort(cams[k]["view"], cams[k]["cutout"], want_indices=False)
            params[k].flags = 0
            ctx.render_device(params[k], strip.data_ptr() if multi else None)
            tot.append(int(ctx.download(capi.BUF_TILE_STATS, ntl, np.uint32, 2)[:, 0].astype(np.int64).sum()))
        staged_per_frame = float(np.mean(tot))
        ctx.set_option(capi.OPT_RECORD_STAGED, 0)
        ctx.set_option(capi.OPT_NEAR_PERMILLE, 0)

    # adaptation pre-roll (untimed, like the fragment counting above): one synchronous pass over the poses of the timed
    # region lets the library settle the share of splats it bins in its first, nearest-splats round for every pose
    retries = 0
    preroll = 0
    while preroll < 96:                                      # (short runs cycle through their poses until the share has settled)
        for k in frames_used:
            frame(k)
            preroll += 1
    for j in range(2 * max(LANES, depth)):                   # every pipeline lane allocated and warm, whatever W is
        frame(frames_used[j % len(frames_used)], capi.RENDER_ASYNC)
    sync()
    for i in range(args.warmup):
        frame(i, capi.RENDER_ASYNC)
    sync()
    # ---- timed region: exactly K steps, barrier + synchronize on both sides.  A retry request from the closing sync
    # (an asynchronous frame outgrew a buffer or needed the skipped second binning round) invalidates the region: the
    # library has adapted, the region is measured again from scratch.
    for attempt in range(4):
        ctx.set_option(capi.OPT_PROFILE, 0)
        ctx.set_option(capi.OPT_PROFILE, 3)                  # HIP events around the dominant kernel (blend) on its stream, every 4th frame
        sync()
        t_start = time.perf_counter()
        for i in range(args.steps):
            frame(args.warmup + i, capi.RENDER_ASYNC)
        again = sync()
        elapsed = time.perf_counter() - t_start
        if not again:
            break
        if attempt == 3:
            raise RuntimeError("timed region kept asking for a re-render")
        retries += 1
        if attempt >= 1:
            # still not settled: pin the first-round share (twice the current one); with a pinned share the second
            # binning round is always launched, so no frame can come back incomplete
            ctx.set_option(capi.OPT_NEAR_PERMILLE, min(1000, 2 * max(ctx.stats()["near_permille"], 50)))
        for k in frames_used:                                # synchronous frames let the library re-adapt
            frame(k)
    s = ctx.stats()
    ctx.set_option(capi.OPT_PROFILE, 0)
    assert s["acc_frames"] == args.steps and s["prof_frames"] >= max(1, args.steps // 4 - 3), (s["acc_frames"], s["prof_frames"])
    blend_frames = s["prof_frames"]                          # frames of the timed region whose blend was bracketed by HIP events
    # per-stage breakdown: a second, UNTIMED pass over the same frames with events around every stage (7 per frame
    # instead of 2; they cost ~4 % of the frame rate, so the timed region carries only the blend's)
    ctx.set_option(capi.OPT_PROFILE, 1)
    sync()
    for i in range(args.steps):
        frame(args.warmup + i, capi.RENDER_ASYNC)
    sync()
    s2 = ctx.stats()
    ctx.set_option(capi.OPT_PROFILE, 0)
    k2 = max(1, s2["prof_frames"])
    stage = {"ms_sort": s2["sum_ms_sort"] * args.steps / k2, "ms_project": s2["sum_ms_project"] * args.steps / k2,
             "ms_bin": s2["sum_ms_bin"] * args.steps / k2, "ms_blend": s["sum_ms_blend"] * args.steps / blend_frames}
    pairs, visible, sorted_n = s["acc_pairs"], s["acc_visible"], s["acc_sorted"]
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # self-check of the N > 1 path: the frame assembled from the gathered strips of the LAST frame must equal, bit for bit,
    # the full frame rank 0 renders alone for the same pose (the splat buffer is replicated, strips are tile-aligned)
    frame_check = None
    if multi and rank == 0 and last_frame[0] is not None:
        k_last = (args.warmup + args.steps - 1) % ORBIT_FRAMES
        ctx.sort(cams[k_last]["view"], cams[k_last]["cutout"], want_indices=False)
        full = ctx.render(capi.make_params(cams[k_last]["gs_mv"], cams[k_last]["gs_proj"], W, H, focal_=cams[k_last]["focal"]))
        frame_check = bool(np.array_equal(last_frame[0].cpu().numpy(), full))
    copy_peak = measured_copy_peak(ctx, capi) if rank == 0 else None
    total_frags = sum(frags[(args.warmup + i) % ORBIT_FRAMES] for i in range(args.steps))
    if rank == 0:
        K = args.steps
        fps = K / elapsed
        sw = x1 - x0
        # dominant kernel = the per-tile blend.  Algorithmic bytes per launch:
        #   B_blend = I*(4 + 32) (pair list entry + projected record, read once per tile) + 4*fb (RGBA8 write)
        blend_bytes = (pairs / K) * 36.0 + 4.0 * sw * H
        blend_s = stage["ms_blend"] / K * 1e-3
        achieved = blend_bytes / blend_s / 1e9 if blend_s > 0 else 0.0
        touched_bytes = staged_per_frame * 36.0 + 4.0 * sw * H
        achieved_touched = touched_bytes / blend_s / 1e9 if blend_s > 0 else 0.0
        # whole-frame algorithmic bytes (SURVEY.md 8d formula)
        V, Vp, I = sorted_n / K, visible / K, pairs / K
        frame_bytes = (16 * n_splats + 4 * V) + (Vp * 36 + V * 4 + Vp * 32) + (I * 20) + (I * 36 + 4 * sw * H)
        traffic = None
        try:                                                 # HBM bytes/launch of k_blend from the committed PMC passes
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_hbm_traffic.json")))
            if world == 1 and n_splats == synth.N_TRAIN:
                key = [k for k in pmc if k.startswith("k_blend<false, 0")][0]     # the timed configuration's first-round blend
                traffic = pmc[key]["hbm_bytes"]
        except Exception:
            traffic = None
        out = {
            "metric": "frames/sec @1920x1080 (sort+project+bin+blend per frame, 1M-splat train.splat-shaped scene)",
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "train.splat-shaped synthetic, N=%d splats, %dx%d, 120-frame orbit (%s)" % (
                           n_splats, W, H, "cutout-demo.html:22-24 pose + cutoutEntity box" if args.cutout else "index.html:13 pose"),
                       "parallelism": "column strips x%d, splat buffer replicated, RCCL gather" % world if world > 1 else "single GPU",
                       "strip_px": sw, "gathered_frame_equals_single_gpu_render": frame_check,
                       "frames_in_flight": "3 (the library's pipeline lanes: every frame still runs its own full sort, projection, "
                                           "binning and blend; consecutive frames overlap on the GPU)"},
            "occlusion_binning": {"near_permille": s["near_permille"], "unsat_tiles_last_frame": s["unsat_tiles"],
                                  "timed_region_retries": retries},
            "msplat_frags_per_s": round(total_frags / 1e6 / elapsed, 1),
            "frags_per_frame": round(total_frags / K),
            "per_frame": {"V_sorted": round(V), "Vp_visible": round(Vp), "I_pairs": round(I),
                          "ms_sort": round(stage["ms_sort"] / K, 4), "ms_project": round(stage["ms_project"] / K, 4),
                          "ms_bin": round(stage["ms_bin"] / K, 4), "ms_blend": round(stage["ms_blend"] / K, 4)},
            "frame_hbm": {"algorithmic_bytes": round(frame_bytes), "achieved_GBps": round(frame_bytes * fps / 1e9, 1),
                          "frac_of_peak": round(frame_bytes * fps / 1e9 / HBM_PEAK_GBS, 5)},
            "roofline": {"kernel": "k_blend", "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "measured_copy_GBps": copy_peak,
                         "traffic_note": "HBM bytes/launch = 2*FETCH_SIZE + WRITE_SIZE from separate rocprofv3 --pmc passes "
                                         "(profiles/r01_pmc_hbm_traffic.md); early termination reads far less than the algorithmic 36*I",
                         "bytes_per_launch": round(blend_bytes), "avg_launch_ms": round(blend_s * 1e3, 4),
                         "launches_timed": int(blend_frames),
                         "achieved_touched": round(achieved_touched, 2), "touched_bytes_per_launch": round(touched_bytes),
                         "touched_note": "36 B x the list entries the kernel actually stages before its tiles saturate (early "
                                         "termination) + the RGBA8 write; `achieved` uses the full algorithmic 36*I of the contract"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(rows, cams[args.warmup % ORBIT_FRAMES], synth)
        if args.size or args.cutout or args.splats:
            out["metric"] = out["metric"].replace("@1920x1080", "@%dx%d" % (W, H)).replace("1M-splat", "%d-splat" % n_splats)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    ctx.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(rows, cam, synth):
    """The oracle (C restatement of the reference's CPU sort + WebGL path, single thread like the reference's one
    Worker) timed on this host: 5 sorts of the full scene + ONE whole frame (all W x H pixels of pose `warmup`) rendered
    back-to-front, about 10 s of CPU work.  frames/s = 1 / (t_sort + t_frame)."""
    from oracle import oracle
    cs, cc, mats = oracle.pack(rows)
    rows4 = np.ascontiguousarray(mats[:, 12:16])
    ts = []
    for _ in range(5):
        t = time.perf_counter(); idx = oracle.sort(rows4, cam["view"], cam["cutout"]); ts.append(time.perf_counter() - t)
    t_sort = float(np.median(ts))
    t = time.perf_counter()
    _, _, fr = oracle.render(cs, cc, idx, cam["gs_mv"].astype(np.float32), cam["gs_proj"].astype(np.float32), cam["focal"], W, H,
                             want_f32=False)
    t_frame = time.perf_counter() - t
    js = js_worker_sort(rows4, cam, idx, oracle)
    # the same frame on all host cores (NOT what the reference does -- it has one worker and one GL context): column strips
    # of the oracle's renderer on a thread pool (ctypes releases the GIL), the sort stays single-threaded
    threads = max(1, min(64, (os.cpu_count() or 1) // 2))
    strips_x = [W * k // threads for k in range(threads + 1)]
    from concurrent.futures import ThreadPoolExecutor
    mvf, prf = cam["gs_mv"].astype(np.float32), cam["gs_proj"].astype(np.float32)
    t = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(lambda k: oracle.render(cs, cc, idx, mvf, prf, cam["focal"], W, H, x0=strips_x[k], x1=strips_x[k + 1], want_f32=False)[2],
                    [k for k in range(threads) if strips_x[k + 1] > strips_x[k]]))
    t_par = time.perf_counter() - t
    return {"value": round(1.0 / (t_sort + t_frame), 5), "unit": "frames/s", "cores": 1, "kind": "port",
            "all_cores": {"value": round(1.0 / (t_sort + t_par), 4), "unit": "frames/s", "threads": threads,
                          "note": "non-reference variant: the oracle's renderer on %d threads (column strips), sort on one" % threads},
            "sample": "oracle/gs_oracle.c, 1 thread: median of 5 sorts of all %d splats (%.1f ms, %.1f Msplat/s) + one whole %dx%d "
                      "frame (%.2f s, %d frags, %.1f Mfrag/s)" % (rows4.shape[0], t_sort * 1e3, rows4.shape[0] / t_sort / 1e6, W, H,
                                                                   t_frame, fr, fr / t_frame / 1e6),
            "sort_msplat_per_s": round(rows4.shape[0] / t_sort / 1e6, 2), "msplat_frags_per_s": round(fr / t_frame / 1e6, 1),
            "host_cpus": os.cpu_count(), "cpu_model": cpu_model(), "js_worker_sort": js}


def measured_copy_peak(ctx, capi, nbytes=1 << 30, reps=5):
    """Device-to-device copy rate of this GPU in the same run: GB/s of bytes MOVED (read + write) by
    hipMemcpyDtoDAsync of a 1 GiB buffer, HIP events around `reps` copies.  The quoted HBM peak stays the 8 TB/s spec."""
    import ctypes as C
    try:
        try:
            hip = C.CDLL("libamdhip64.so")
        except OSError:
            hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
        a, b, e0, e1 = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        if hip.hipMalloc(C.byref(a), C.c_size_t(nbytes)) or hip.hipMalloc(C.byref(b), C.c_size_t(nbytes)):
            return None
        hip.hipMemsetAsync(a, 1, C.c_size_t(nbytes), None); hip.hipMemsetAsync(b, 2, C.c_size_t(nbytes), None)
        hip.hipEventCreate(C.byref(e0)); hip.hipEventCreate(C.byref(e1))
        hip.hipMemcpyDtoDAsync(b, a, C.c_size_t(nbytes), None)
        hip.hipEventRecord(e0, None)
        for _ in range(reps):
            hip.hipMemcpyDtoDAsync(b, a, C.c_size_t(nbytes), None)
        hip.hipEventRecord(e1, None)
        hip.hipEventSynchronize(e1)
        ms = C.c_float(0)
        hip.hipEventElapsedTime(C.byref(ms), e0, e1)
        hip.hipEventDestroy(e0); hip.hipEventDestroy(e1); hip.hipFree(a); hip.hipFree(b)
        return round(2.0 * nbytes * reps / (ms.value * 1e-3) / 1e9, 1) if ms.value > 0 else None
    except Exception:
        return None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def js_worker_sort(rows4, cam, idx, oracle):
    """The reference's sort runs in ONE JavaScript Worker: time oracle/worker_sort.js (a JS restatement of that worker
    loop, pinned against the reference's golden vectors in tests/) under node on this host, same scene and pose, and check
    its order against the C oracle's through a position-sensitive checksum.  None if node is not installed."""
    import shutil
    import subprocess
    import tempfile
    node = shutil.which("node")
    if not node:
        return None
    with tempfile.TemporaryDirectory() as d:
        rows4.astype("<f4").tofile(os.path.join(d, "rows.f32"))
        un = [np.asarray(cam["view"], np.float32)] + ([np.asarray(cam["cutout"], np.float32)] if cam["cutout"] is not None else [])
        np.concatenate(un).astype("<f4").tofile(os.path.join(d, "un.f32"))
        try:
            r = subprocess.run([node, os.path.join(ROOT, "oracle", "worker_sort.js"), "bench", os.path.join(d, "rows.f32"),
                                os.path.join(d, "un.f32"), "5"], capture_output=True, text=True, timeout=300)
            rep = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:                                # the baseline is reported, never required
            return {"error": str(e)[:200]}
    ok = rep["kept"] == int(idx.size) and rep["order_sum"] == oracle.order_sum(idx)
    return {"ms_per_sort": round(rep["ms_median"], 3), "msplat_per_s": round(rows4.shape[0] / rep["ms_median"] / 1e3, 2),
            "cores": 1, "runtime": "node " + subprocess.run([node, "--version"], capture_output=True, text=True).stdout.strip(),
            "matches_c_oracle": bool(ok), "sample": "median of 5 sorts of all %d splats, 64-byte worker rows" % rows4.shape[0]}


if __name__ == "__main__":
    main()
