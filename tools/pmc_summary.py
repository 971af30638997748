#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as MI355X_MICROARCH.md prescribes).
usage: tools/pmc_summary.py <fetch.db> <write.db> <out.md> <out.json>
FETCH_SIZE / WRITE_SIZE are in KiB per dispatch.  On gfx950 FETCH_SIZE reports exactly half of the bytes of a wide
(16 B/lane) coalesced streaming read (MI355X_MICROARCH.md, HBM section) -- verified here on k_pack (reads 32 B/splat:
FETCH = 16.1 MiB for 32 MiB) and k_sort_depth (16 B/splat: 8.05 MiB for 16 MiB); WRITE_SIZE matched known byte counts
exactly (k_pack 48 MiB, framebuffer 8100 KiB).  `hbm_bytes` = 2*FETCH + WRITE per launch."""
import json
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:<>, ]+?)\(", name)
    return (m.group(1) if m else name)[:48]


def load(path):
    db = sqlite3.connect(path)
    out = {}
    for name, n, avg in db.execute("select kernel_name, count(*), avg(value) from counters_collection group by kernel_name"):
        out[short(name)] = (n, avg)
    return out


def main():
    f, w = load(sys.argv[1]), load(sys.argv[2])
    rows = []
    for k in sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, (0, 0))[1] + w.get(k, (0, 0))[1])):
        fk, wk = f.get(k, (0, 0.0)), w.get(k, (0, 0.0))
        rows.append({"kernel": k, "launches": fk[0] or wk[0], "fetch_KiB": round(fk[1], 1), "write_KiB": round(wk[1], 1),
                     "hbm_bytes": int((2 * fk[1] + wk[1]) * 1024)})
    with open(sys.argv[3], "w") as o:
        o.write("| kernel | launches | FETCH_SIZE KiB/launch | WRITE_SIZE KiB/launch | HBM bytes/launch (2*F+W) |\n|---|---:|---:|---:|---:|\n")
        for r in rows:
            o.write("| %s | %d | %.1f | %.1f | %d |\n" % (r["kernel"], r["launches"], r["fetch_KiB"], r["write_KiB"], r["hbm_bytes"]))
    # the kernel sources these counters belong to: bench.py reports `roofline.traffic` only while they are unchanged
    import hashlib, os
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aframe-gaussian-splatting_amd", "csrc")
    h = hashlib.sha1()
    for fn in sorted(os.listdir(csrc)):
        if fn.endswith((".hip", ".h", ".cpp")):
            h.update(open(os.path.join(csrc, fn), "rb").read())
    d = {r["kernel"]: r for r in rows}
    d["_csrc_sha1"] = h.hexdigest()
    json.dump(d, open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
