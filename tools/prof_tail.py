#!/usr/bin/env python3
"""Per-kernel averages over the LAST n dispatches of a rocprofv3 (rocpd sqlite) kernel trace -- i.e. the timed, pipelined
frames of bench.py, without the synchronous counting / pre-roll frames.  usage: tools/prof_tail.py <results.db> [n]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
disp = sorted([t for t in tabs if "kernel_dispatch" in t], key=len)[0]
sym = sorted([t for t in tabs if "kernel_symbol" in t], key=len)[0]
rows = list(db.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start desc limit %d" % (disp, sym, n)))
def demangle(names):
    import shutil, subprocess
    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    try:
        out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, timeout=30).stdout.split("\n")
        return dict(zip(names, out)) if len(out) >= len(names) else {}
    except Exception:
        return {}
dm = demangle(sorted(set(r[0].replace(".kd", "") for r in rows)))
agg = {}
for name, s, e in rows:
    name = dm.get(name.replace(".kd", ""), name)
    name = re.sub(r"\(anonymous namespace\)::", "", name); name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:<>, ]+?)\(", name); name = (m.group(1) if m else name)[:44]
    a = agg.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
wall = (max(r[2] for r in rows) - min(r[1] for r in rows)) / 1e3
print("last %d dispatches, wall %.1f us, kernel time %.1f us" % (len(rows), wall, sum(a[1] for a in agg.values())))
for name, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-46s calls %5d  avg %9.2f us  share %5.1f %%" % (name, c, t / c, 100.0 * t / sum(a[1] for a in agg.values())))
if "--trace" in sys.argv:                                    # the last K dispatches in order: start offset, duration, gap to the previous end
    k = int(sys.argv[sys.argv.index("--trace") + 1])
    seq = sorted(rows[:k], key=lambda r: r[1])
    t0, prev = seq[0][1], None
    for name, s, e in seq:
        nm = dm.get(name.replace(".kd", ""), name); nm = re.sub(r"\(anonymous namespace\)::", "", nm); nm = re.sub(r"^void ", "", nm)
        m = re.match(r"([\w:<>, ]+?)\(", nm); nm = (m.group(1) if m else nm)[:40]
        print("  +%8.1f us  %-40s %7.2f us   gap %6.2f" % ((s - t0) / 1e3, nm, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
        prev = e
