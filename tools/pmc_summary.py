#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/gpu_pmc.sh (separate runs per counter group, as MI355X_MICROARCH.md prescribes).

usage: tools/pmc_summary.py <dir> <out.md> <out.json>
  <dir>/<config>_<GROUP>/pmc_results.db + <dir>/<config>_<GROUP>.log for config in c1 c2 c3 c5 outside (whatever is there) and GROUP in
  FETCH_SIZE, WRITE_SIZE, VALU (= SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS)

FETCH_SIZE / WRITE_SIZE are in KiB per dispatch.  On gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane)
coalesced streaming read (MI355X_MICROARCH.md, HBM section) -- verified in round 1 on k_pack (reads 32 B/splat: FETCH = 16.1 MiB
for 32 MiB) and k_sort_depth (16 B/splat: 8.05 MiB for 16 MiB); WRITE_SIZE matched known byte counts exactly (k_pack 48 MiB,
framebuffer 8100 KiB).  `hbm_bytes` = 2*FETCH + WRITE per launch; for kernels that GATHER (k_project: a 32-byte record per
128-byte line) the x2 is an upper bound, and Infinity-Cache hits are counted (they are requests the L2 sent out).
SQ_ACTIVE_INST_VALU counts quad-cycles (MI355X_MICROARCH.md, latency table): x4 = SIMD cycles spent issuing VALU instructions.

The JSON is what bench.py reads (profiles/pmc_counters.json): per configuration the per-kernel averages, the frames the profiled
run queued and the list entries its blend evaluates per frame -- stamped with the SHA-1 of csrc/* so that a bench run from other
kernel sources does not report it."""
import hashlib
import json
import os
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:<>, ]+?)\(", name)
    return (m.group(1) if m else name)[:64]


def load(path):
    out = {}
    if not os.path.exists(path):
        return out
    db = sqlite3.connect(path)
    for name, cname, n, avg in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        out.setdefault(short(name), {})[cname] = (n, avg)
    return out


def runinfo(path):
    try:
        for line in open(path):
            if line.startswith("PMCRUN "):
                return {k: float(v) for k, v in (kv.split("=") for kv in line.split()[1:])}
    except OSError:
        pass
    return {}


def main():
    d, out_md, out_json = sys.argv[1:4]
    res = {"configs": {}}
    md = []
    for cfg in ("c2", "c1", "c3", "c5", "outside", "unsat"):
        f = load(os.path.join(d, cfg + "_FETCH_SIZE", "pmc_results.db")); w = load(os.path.join(d, cfg + "_WRITE_SIZE", "pmc_results.db"))
        v = load(os.path.join(d, cfg + "_VALU", "pmc_results.db"))
        if not f and not w and not v:
            continue
        info = runinfo(os.path.join(d, cfg + "_FETCH_SIZE.log")) or runinfo(os.path.join(d, cfg + "_VALU.log"))
        info_v = runinfo(os.path.join(d, cfg + "_VALU.log"))
        kern = {}
        for k in set(f) | set(w) | set(v):
            fk = f.get(k, {}).get("FETCH_SIZE", (0, 0.0)); wk = w.get(k, {}).get("WRITE_SIZE", (0, 0.0))
            row = {"launches": fk[0] or wk[0] or max([x[0] for x in v.get(k, {}).values()] or [0]),
                   "fetch_KiB": round(fk[1], 1), "write_KiB": round(wk[1], 1), "hbm_bytes": int((2 * fk[1] + wk[1]) * 1024)}
            if k in v:
                row["valu"] = {c: round(x[1], 1) for c, x in v[k].items()}
                row["valu"]["launches"] = max(x[0] for x in v[k].values())
            kern[k] = row
        frames = info.get("frames_queued")
        tot = sum(r["hbm_bytes"] * r["launches"] for r in kern.values())
        # the pipelined, paired frames alone (k_twin<...> / k_sort_depth_pair launches cover two frames each): what a frame of the
        # steady loop moves -- the process' synchronous settling frames (whole sorts, both binning rounds) are not in it
        paired = {k: r for k, r in kern.items() if k.startswith("k_twin") or "_pair<" in k}
        bl = [r["launches"] for k, r in paired.items() if "F_blend<0" in k]
        frame_paired = round(sum(r["hbm_bytes"] * r["launches"] for r in paired.values()) / (2.0 * max(bl))) if bl else None
        # the profiled process also runs synchronous frames first (buffers, share): the per-frame figure counts every launch of the
        # process over every frame it drew; frames_total = queued + the synchronous ones (stage_bench --pmc-run: 30 + 160 + 24)
        frames_total = (frames or 0) + 30 + 160 + 24
        res["configs"][cfg] = {"kernels": kern, "run": info, "run_valu": info_v,
                               "frame_hbm_bytes": frame_paired, "frame_hbm_bytes_whole_process": round(tot / frames_total) if frames else None,
                               "frames_total": frames_total if frames else None}
        md.append("## %s  (%s)\n" % (cfg, " ".join("%s=%g" % kv for kv in sorted(info.items()))))
        md.append("| kernel | launches | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM bytes/launch (2F+W) | SQ_INSTS_VALU | SQ_ACTIVE_INST_VALU x4 (cycles) | cycles / VALU instr | SQ_BUSY_CYCLES | SQ_WAVES | SQ_INSTS_LDS |\n|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
        for k, r in sorted(kern.items(), key=lambda kv: -kv[1]["hbm_bytes"] * max(1, kv[1]["launches"])):
            vv = r.get("valu", {})
            iv, av = vv.get("SQ_INSTS_VALU", 0), vv.get("SQ_ACTIVE_INST_VALU", 0) * 4
            md.append("| %s | %d | %.1f | %.1f | %d | %s | %s | %s | %s | %s | %s |" % (
                k, r["launches"], r["fetch_KiB"], r["write_KiB"], r["hbm_bytes"], "%.0f" % iv if iv else "", "%.0f" % av if av else "",
                "%.2f" % (av / iv) if iv else "", "%.0f" % vv["SQ_BUSY_CYCLES"] if "SQ_BUSY_CYCLES" in vv else "",
                "%.0f" % vv["SQ_WAVES"] if "SQ_WAVES" in vv else "", "%.0f" % vv["SQ_INSTS_LDS"] if "SQ_INSTS_LDS" in vv else ""))
        if frames and frame_paired:
            md.append("\nHBM traffic per frame of the pipelined loop (the paired launches, two frames each): **%.1f MB**; every kernel of the process "
                      "over every frame it drew, its synchronous settling frames included: %.1f MB\n" % (frame_paired / 1e6, tot / frames_total / 1e6))
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aframe-gaussian-splatting_amd", "csrc")
    h = hashlib.sha1()
    for fn in sorted(os.listdir(csrc)):
        if fn.endswith((".hip", ".h", ".cpp")):
            h.update(open(os.path.join(csrc, fn), "rb").read())
    res["_csrc_sha1"] = h.hexdigest()
    open(out_md, "w").write("# rocprofv3 --pmc passes (tools/gpu_pmc.sh), per-launch averages; csrc sha1 %s\n\n" % res["_csrc_sha1"] + "\n".join(md) + "\n")
    json.dump(res, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
