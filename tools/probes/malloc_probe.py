import importlib, sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
capi = importlib.import_module("aframe-gaussian-splatting_amd.capi")
hip = capi.hip_runtime()
hip.hipFree(None)
for sz in (4096, 1 << 20, 8 << 20, 64 << 20, 256 << 20, 1 << 30):
    ts = []
    ptrs = []
    for i in range(6):
        p = C.c_void_p()
        t = time.perf_counter(); rc = hip.hipMalloc(C.byref(p), C.c_size_t(sz)); ts.append((time.perf_counter() - t) * 1e6); ptrs.append(p)
    tf = []
    for p in ptrs:
        t = time.perf_counter(); hip.hipFree(p); tf.append((time.perf_counter() - t) * 1e6)
    print("hipMalloc %10d B: %s us; hipFree %s us" % (sz, " ".join("%.0f" % x for x in ts), " ".join("%.0f" % x for x in tf)))
s = C.c_void_p()
t = time.perf_counter(); hip.hipStreamCreateWithFlags(C.byref(s), 1); print("stream create %.0f us" % ((time.perf_counter() - t) * 1e6))
