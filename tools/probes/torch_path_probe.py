"""Host time per call of the multi-GPU frame loop (world = 1 stand-in): where do the ~300 us/frame go?"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544"); os.environ.setdefault("NCCL_DEBUG", "WARN")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
capi = importlib.import_module("aframe-gaussian-splatting_amd.capi"); synth = importlib.import_module("aframe-gaussian-splatting_amd.synth")
mg = importlib.import_module("aframe-gaussian-splatting_amd.multigpu")
rows = synth.make_splat_rows(synth.N_TRAIN)
W, H, K, LANES = 1920, 1080, 240, 3
cams = [synth.index_html_camera(W, H, 3.0 * i, capi=capi) for i in range(120)]
params = [capi.make_params(c["gs_mv"], c["gs_proj"], W, H, focal_=c["focal"]) for c in cams]
ctx = capi.Context(0); ctx.push_splat(rows)
strips = [torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda") for _ in range(LANES)]
gathereds = [[torch.zeros_like(strips[0])] for _ in range(LANES)]
ts = [torch.cuda.Stream() for _ in range(LANES)]
works = [None] * LANES
acc = {}
def T(name, t0):
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
mode = sys.argv[1] if len(sys.argv) > 1 else "gather"
ext = {}
groups = [dist.new_group(ranks=[0]) for _ in range(LANES)] if mode == "ext3" else [None] * LANES
def frame_ext(i, flags):
    k = i % 120; b = i % LANES; p = params[k]; p.flags = flags
    t = time.perf_counter()
    ctx.sort(cams[k]["view"], None, want_indices=False); T("sort", t); t = time.perf_counter()
    ctx.render_device(p, strips[b].data_ptr()); T("render", t); t = time.perf_counter()
    sp = ctx.frame_stream()
    if sp not in ext: ext[sp] = torch.cuda.ExternalStream(sp)
    with torch.cuda.stream(ext[sp]):
        T("stream_ctx", t); t = time.perf_counter()
        dist.gather(strips[b], gathereds[b], dst=0, group=groups[b]); T("collective", t); t = time.perf_counter()
        if mode in ("ext", "ext3"): None
        T("assemble", t)
def frame(i, flags):
    if mode.startswith("ext"): return frame_ext(i, flags)
    k = i % 120; b = i % LANES; p = params[k]; p.flags = flags
    t = time.perf_counter()
    if works[b] is not None:
        with torch.cuda.stream(ts[b]):
            works[b].wait()
            if mode == "gather": None
        works[b] = None
    T("finish", t); t = time.perf_counter()
    ctx.wait_stream(ts[b].cuda_stream); T("wait_stream", t); t = time.perf_counter()
    ctx.sort(cams[k]["view"], None, want_indices=False); T("sort", t); t = time.perf_counter()
    ctx.render_device(p, strips[b].data_ptr()); T("render", t); t = time.perf_counter()
    ctx.stream_wait_frame(ts[b].cuda_stream); T("wait_frame", t); t = time.perf_counter()
    with torch.cuda.stream(ts[b]):
        if mode == "gather": works[b] = dist.gather(strips[b], gathereds[b], dst=0, async_op=True)
        elif mode == "allgather": works[b] = dist.all_gather_into_tensor(gathereds[b][0], strips[b], async_op=True)
    T("collective", t)
for k in range(0, 120, 2): frame(k, 0)
torch.cuda.synchronize()
for i in range(60): frame(i, capi.RENDER_ASYNC)
torch.cuda.synchronize()
try: ctx.sync()
except capi.GsError: pass
acc.clear()
t0 = time.perf_counter()
for i in range(K): frame(i, capi.RENDER_ASYNC)
enq = time.perf_counter() - t0
torch.cuda.synchronize()
try: ctx.sync()
except capi.GsError: pass
tot = time.perf_counter() - t0
print(mode, "enqueue %.1f us/frame, complete %.1f us/frame (%.0f fps)" % (enq / K * 1e6, tot / K * 1e6, K / tot), {k: round(v / K * 1e6, 1) for k, v in acc.items()})
dist.destroy_process_group()
