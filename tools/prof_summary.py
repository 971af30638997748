#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share.
usage: tools/prof_summary.py <results.db> [frames]   -> markdown table on stdout"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:<>, ]+?)\(", name)
    return (m.group(1) if m else name)[:48]


def main():
    db = sqlite3.connect(sys.argv[1])
    frames = float(sys.argv[2]) if len(sys.argv) > 2 else None
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print("| kernel | calls | total us | avg us | % |" + (" us/frame |" if frames else ""))
    print("|---|---:|---:|---:|---:|" + ("---:|" if frames else ""))
    tot = 0.0
    for name, calls, total, avg, pct in rows:
        tot += total
        line = "| %s | %d | %.1f | %.2f | %.1f |" % (short(name), calls, total, avg, pct)
        if frames:
            line += " %.1f |" % (total / frames)
        print(line)
    print("\ntotal kernel time: %.1f us" % tot + (" = %.1f us/frame over %g frames" % (tot / frames, frames) if frames else ""))


if __name__ == "__main__":
    main()
