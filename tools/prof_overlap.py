#!/usr/bin/env python3
"""How busy is the GPU in the steady part of a rocprofv3 kernel trace (rocpd sqlite)?  Over the middle half of the trace: the share of
the wall time in which 0 / 1 / 2 / 3+ kernels are in flight, and per kernel the summed duration per frame.
usage: tools/prof_overlap.py <results.db> [frames_per_blend_launch = 2]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
fpl = int(sys.argv[2]) if len(sys.argv) > 2 else 2
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
disp = sorted([t for t in tabs if "kernel_dispatch" in t], key=len)[0]
sym = sorted([t for t in tabs if "kernel_symbol" in t], key=len)[0]
rows = list(db.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (disp, sym)))
t_lo, t_hi = rows[0][1], rows[-1][2]
lo_f, hi_f = (float(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (0.45, 0.85)
a, b = t_lo + (t_hi - t_lo) * lo_f, t_lo + (t_hi - t_lo) * hi_f          # (default: the timed loop sits in the second half of a bench run)
ev = []
per = {}
blends = 0
for name, s, e in rows:
    if e <= a or s >= b:
        continue
    s2, e2 = max(s, a), min(e, b)
    ev.append((s2, 1)); ev.append((e2, -1))
    short = re.sub(r"^.*?(F_\w+(?:<[^>]*>)?|k_(?!twin)\w+(?:<[^>]*>)?).*$", r"\1", name)[:40]
    per[short] = per.get(short, 0.0) + (e2 - s2)
    if "blend" in short and "Li1E" not in short and "1," not in short:
        blends += fpl if short.startswith("F_") else 1            # (a k_twin launch draws fpl frames)
ev.sort()
occ = {}
cur, last = 0, a
for t, d in ev:
    occ[min(cur, 4)] = occ.get(min(cur, 4), 0.0) + (t - last)
    cur += d; last = t
occ[min(cur, 4)] = occ.get(min(cur, 4), 0.0) + (b - last)
wall = b - a
frames = max(1, blends)
print("window %.1f ms, ~%d frames (%.1f us per frame)" % (wall / 1e6, frames, wall / 1e3 / frames))
print("kernels in flight:", "  ".join("%s: %.1f %%" % (("%d" % k if k < 4 else "4+"), 100.0 * v / wall) for k, v in sorted(occ.items())))
print("summed kernel time per frame: %.1f us" % (sum(per.values()) / 1e3 / frames))
for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:16]:
    print("  %-42s %7.2f us per frame" % (k, v / 1e3 / frames))

# ---- per queue: busy share and the gaps between consecutive kernels; the longest intervals with nothing in flight
cols = [r[1] for r in db.execute("pragma table_info(%s)" % disp)]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
if q:
    rows2 = [r for r in db.execute("select d.%s, d.start, d.end, s.kernel_name from %s d join %s s on d.kernel_id = s.id order by d.start" % (q, disp, sym)) if r[2] > a and r[1] < b]
    byq = {}
    for qu, s, e, n in rows2:
        byq.setdefault(qu, []).append((s, e, n))
    for qu, lst in sorted(byq.items()):
        busy = sum(e - s for s, e, _ in lst)
        gaps = [lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1)]
        gaps.sort()
        if len(lst) < 50:
            continue
        print("queue %s: %d kernels, busy %.1f %%, gap median %.1f us, p90 %.1f us, p99 %.1f us, max %.1f us" % (
            qu, len(lst), 100.0 * busy / wall, gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3, gaps[int(len(gaps) * 0.99)] / 1e3, gaps[-1] / 1e3))
    # idle intervals
    ev2 = sorted([(s, 1) for _, s, e, _ in rows2] + [(e, -1) for _, s, e, _ in rows2])
    cur, start, idle = 0, None, []
    for t, d in ev2:
        if cur == 0 and start is not None and d == 1:
            idle.append(t - start)
        cur += d
        if cur == 0:
            start = t
    idle.sort()
    if idle:
        tot = sum(idle)
        print("all-idle intervals: %d, total %.1f %% of the window; median %.1f us, p90 %.1f us, max %.1f us; intervals > 20 us: %d (%.1f %% of the window)" % (
            len(idle), 100.0 * tot / wall, idle[len(idle) // 2] / 1e3, idle[int(len(idle) * 0.9)] / 1e3, idle[-1] / 1e3,
            sum(1 for x in idle if x > 20000), 100.0 * sum(x for x in idle if x > 20000) / wall))

# ---- which kernel boundaries carry a queue's idle time (same queue, consecutive dispatches)
if q:
    agg = {}
    for qu, lst in byq.items():
        for i in range(len(lst) - 1):
            g = lst[i + 1][0] - lst[i][1]
            if g <= 0:
                continue
            sh = lambda n: re.sub(r"^.*?(F_\w+|k_(?!twin)\w+).*$", r"\1", n)[:18]
            k = sh(lst[i][2]) + " -> " + sh(lst[i + 1][2])
            a_ = agg.setdefault(k, [0, 0.0]); a_[0] += 1; a_[1] += g
    print("idle time of the queues by kernel boundary (sum over queues; share of window x queues):")
    for k, (n_, t_) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10]:
        print("  %-40s %6d gaps  avg %6.1f us  %5.1f %%" % (k, n_, t_ / n_ / 1e3, 100.0 * t_ / (wall * max(1, len(byq)))))
