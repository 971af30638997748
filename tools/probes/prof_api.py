#!/usr/bin/env python3
"""HIP API calls (rocprofv3 --hip-trace, rocpd sqlite) next to the kernel dispatches of the LAST w microseconds of a trace: which thread
issued which launch when, and when the kernel ran.  usage: tools/prof_api.py <results.db> [w_us]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
w = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 2.0e6
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
def pick(sub):
    c = sorted([t for t in tabs if sub in t], key=len)
    return c[0] if c else None
disp, sym, reg, strs = pick("kernel_dispatch"), pick("kernel_symbol"), pick("region"), pick("string")
cols = lambda t: [r[1] for r in db.execute("pragma table_info(%s)" % t)]
print("tables:", disp, cols(disp)); print(reg, cols(reg)); print(strs, cols(strs))
rc = cols(reg)
kd = list(db.execute("select s.kernel_name, d.start, d.end, d.queue_id from %s d join %s s on d.kernel_id = s.id order by d.start" % (disp, sym)))
t_end = max(r[2] for r in kd); t0 = t_end - w
ev = []
for name, s, e, q in kd:
    if s >= t0:
        ev.append((s, "GPU  q%-14d %-28s %7.1f us" % (q, re.sub(r"^.*?(F_\w+|k_(?!twin)\w+).*$", r"\1", name)[:28], (e - s) / 1e3)))
tid_col = "tid" if "tid" in rc else ("thread_id" if "thread_id" in rc else "id")
if "name" in rc:
    q = "select r.name, r.start, r.end, r.%s from %s r where r.start >= ? order by r.start" % (tid_col, reg)
else:
    q = "select st.string, r.start, r.end, r.%s from %s r join %s st on r.name_id = st.id where r.start >= ? order by r.start" % (tid_col, reg, strs)
if True:
    for name, s, e, tid in db.execute(q, (t0,)):
        if (e - s) < 15e3 and not re.search(r"Launch|Memcpy|Memset|Malloc|Free|Synchronize|Event", name):
            continue
        ev.append((s, "API  t%-14s %-28s %7.1f us" % (str(tid)[-6:], name[:28], (e - s) / 1e3)))
ev.sort()
for s, line in ev:
    print("%9.1f  %s" % ((s - t0) / 1e3, line))
