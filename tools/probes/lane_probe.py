"""Which ingredient of the multi-GPU frame loop costs the lane overlap?  variants: plain | torch (import + init only) |
userbuf (render into a torch tensor) | waits (gs_wait_stream / gs_stream_wait_frame against torch side streams)"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
variant = sys.argv[1]
torch = None
if variant != "plain":
    import torch
    torch.cuda.set_device(0); torch.zeros(1, device="cuda")
capi = importlib.import_module("aframe-gaussian-splatting_amd.capi"); synth = importlib.import_module("aframe-gaussian-splatting_amd.synth")
rows = synth.make_splat_rows(synth.N_TRAIN)
W, H, K, LANES = 1920, 1080, 360, 3
cams = [synth.index_html_camera(W, H, 3.0 * i, capi=capi) for i in range(120)]
params = [capi.make_params(c["gs_mv"], c["gs_proj"], W, H, focal_=c["focal"]) for c in cams]
ctx = capi.Context(0); ctx.push_splat(rows)
bufs = ts = None
if variant in ("userbuf", "waits", "waits_only", "gate_only", "signal_only"):
    bufs = [torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda") for _ in range(LANES)]
    ts = [torch.cuda.Stream() for _ in range(LANES)]
def go(n, flags):
    t0 = time.perf_counter()
    for i in range(n):
        k = i % 120; b = i % LANES
        if variant in ("waits", "waits_only", "gate_only"): ctx.wait_stream(ts[b].cuda_stream)
        ctx.sort(cams[k]["view"], None, want_indices=False); params[k].flags = flags
        ctx.render_device(params[k], bufs[b].data_ptr() if variant in ("userbuf", "waits") else None)
        if variant in ("waits", "waits_only", "signal_only"): ctx.stream_wait_frame(ts[b].cuda_stream)
    t1 = time.perf_counter()
    try: ctx.sync()
    except capi.GsError as e:
        if e.code != capi.E_RETRY: raise
    if torch: torch.cuda.synchronize()
    return t1 - t0, time.perf_counter() - t0
go(120, 0); go(60, capi.RENDER_ASYNC); go(60, capi.RENDER_ASYNC)
enq, tot = go(K, capi.RENDER_ASYNC)
print("%-10s enqueue %.1f us/frame, complete %.1f us/frame -> %.0f frames/s" % (variant, enq / K * 1e6, tot / K * 1e6, K / tot))
