#!/usr/bin/env python3
"""Per-kernel averages of arbitrary rocprofv3 --pmc counters (rocpd sqlite).  usage: tools/pmc_generic.py <results.db> [kernel substring]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
if "counters_collection" not in tabs:
    print("tables:", tabs); sys.exit(0)
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
rows = list(db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"))
out = {}
for k, c, n, v in rows:
    k = re.sub(r"\(anonymous namespace\)::", "", k); k = re.sub(r"^void ", "", k)
    m = re.match(r"([\w:<>, ]+?)\(", k); k = (m.group(1) if m else k)[:40]
    if flt in k: out.setdefault(k, {})[c] = (n, v)
for k, d in sorted(out.items()):
    print(k, {c: (n, round(v, 1)) for c, (n, v) in sorted(d.items())})
