#!/usr/bin/env python
"""Diagnostic (round 6): in a rocprofv3 --kernel-trace database of a bench run, the longest kernels and the longest idle gaps of the GPU (no kernel
running on any stream), with the launches around them.   tools/find_stall.py <results.db>"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
t0 = rows[0][1]
print("kernels", len(rows), "span %.1f ms" % ((rows[-1][2] - t0) / 1e6))
longest = sorted(rows, key=lambda r: r[2] - r[1], reverse=True)[:8]
print("-- longest kernels")
for n, s, e in longest:
    print("  %8.3f ms at %9.3f ms  %s" % ((e - s) / 1e6, (s - t0) / 1e6, n[:90]))
# idle gaps: sweep over the union of busy intervals
gaps = []
busy_end = rows[0][2]
for i, (n, s, e) in enumerate(rows[1:], 1):
    if s > busy_end:
        gaps.append((s - busy_end, busy_end, i))
    busy_end = max(busy_end, e)
gaps.sort(reverse=True)
print("-- longest idle gaps (no kernel on any stream)")
for g, at, i in gaps[:8]:
    print("  %8.3f ms idle from %9.3f ms; before: %s | after: %s" % (g / 1e6, (at - t0) / 1e6, rows[i - 1][0][:50], rows[i][0][:50]))
