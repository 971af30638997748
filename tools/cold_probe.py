import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
capi = importlib.import_module("aframe-gaussian-splatting_amd.capi"); synth = importlib.import_module("aframe-gaussian-splatting_amd.synth")
W, H = 1920, 1080
rows = synth.make_splat_rows(synth.N_TRAIN)
cams = [synth.index_html_camera(W, H, 1.5 + 3.0 * i, capi=capi) for i in range(120)]
ps = [capi.make_params(c["gs_mv"], c["gs_proj"], W, H, focal_=c["focal"]) for c in cams]
with capi.Context(0) as c:
    c.push_splat(rows)
    c.set_option(capi.OPT_FRAME_BATCH, 2)
    for i in range(240):
        k = i % 120
        c.sort(cams[k]["view"], None, want_indices=False)
        ps[k].flags = capi.RENDER_ASYNC
        c.render_device(ps[k], None)
        if i % 24 == 23:
            try:
                c.sync()
            except capi.GsError as e:
                print("retry asked", e)
            s = c.stats()
            print(i, "near", s["near_permille"], "unsat", s["unsat_tiles"], "pairs", s["n_pairs"], "retried", s["retried_frames"], "vis", s["n_visible"], flush=True)
