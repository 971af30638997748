"""Load-path timing: processPlyBuffer on the host (one thread) vs on the GPU, INRIA-layout PLY (248 B/row)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
capi = importlib.import_module("aframe-gaussian-splatting_amd.capi"); synth = importlib.import_module("aframe-gaussian-splatting_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
rows = synth.make_splat_rows(n, seed=synth.SEED_BASE + 3)
ply = synth.rows_to_inria_ply(rows)
with capi.Context(0) as ctx:
    ctx.ply_to_splat(ply[: bytes(ply).index(b"end_header\n") + 11 + 248 * 1000].replace(b"element vertex %d" % n, b"element vertex 1000"))  # warm-up
    t = time.perf_counter(); g = ctx.ply_to_splat(ply); tg = time.perf_counter() - t
    t = time.perf_counter(); ctx.load_ply(ply); tl = time.perf_counter() - t
t = time.perf_counter(); h = capi.ply_to_splat(ply); th = time.perf_counter() - t
print("rows %d  ply %.1f MB  host %.3f s  gpu convert (H2D+kernels+D2H) %.3f s  gpu load_ply (convert+pack, rows stay in HBM) %.3f s  equal %s" % (
    n, len(ply) / 1e6, th, tg, tl, bool(np.array_equal(g, h))))
