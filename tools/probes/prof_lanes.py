#!/usr/bin/env python3
"""Timeline of the LAST n dispatches of a rocprofv3 (rocpd sqlite) kernel trace, one line per dispatch with its queue: where a short
timed region (the driver's 20 steps) spends its time filling and draining the lanes.  usage: tools/prof_lanes.py <results.db> [n]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
disp = sorted([t for t in tabs if "kernel_dispatch" in t], key=len)[0]
sym = sorted([t for t in tabs if "kernel_symbol" in t], key=len)[0]
cols = [r[1] for r in db.execute("pragma table_info(%s)" % disp)]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = list(db.execute("select s.kernel_name, d.start, d.end, d.%s from %s d join %s s on d.kernel_id = s.id order by d.start desc limit %d" % (q, disp, sym, n)))
rows.sort(key=lambda r: r[1])
t0 = rows[0][1]
queues = sorted(set(r[3] for r in rows))
print("columns:", cols)
for name, s, e, qu in rows:
    short = re.sub(r"^.*?(F_\w+|k_(?!twin)\w+).*$", r"\1", name)[:22]
    print("%9.1f %8.1f  q%-2d %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, queues.index(qu), "                         " * queues.index(qu), short))
