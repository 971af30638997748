#!/usr/bin/env python3
"""Concurrency of a pipelined run from a rocprofv3 (rocpd sqlite) kernel trace: how busy is the GPU, how many kernels run
at once, per-queue share.  usage: tools/prof_timeline.py <results.db> [last_n_dispatches]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    disp = [t for t in tables if "kernel_dispatch" in t]
    if not disp:
        print("tables:", tables); return
    t = sorted(disp, key=len)[0]
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % t)]
    if "--schema" in sys.argv:
        print(t, cols); return
    start, end = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = list(db.execute("select %s, %s, %s from %s order by %s" % (start, end, qcol or "0", t, start)))
    n = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else len(rows)
    rows = rows[-n:]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    ev = []
    for s, e, q in rows:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    busy = [0.0] * 8
    depth, last = 0, t0
    for ts, d in ev:
        busy[min(depth, 7)] += ts - last
        last = ts; depth += d
    wall = t1 - t0
    print("dispatches %d, wall %.1f us, sum of kernel durations %.1f us (x%.2f of wall)" % (
        len(rows), wall / 1e3, sum(e - s for s, e, _ in rows) / 1e3, sum(e - s for s, e, _ in rows) / wall))
    for k in range(8):
        if busy[k] > 0:
            print("  %d kernel(s) in flight: %5.1f %% of the wall time" % (k, 100.0 * busy[k] / wall))
    qs = {}
    for s, e, q in rows:
        qs[q] = qs.get(q, 0) + (e - s)
    print("  per queue/stream busy share:", {k: round(v / wall, 2) for k, v in qs.items()})


if __name__ == "__main__":
    main()
