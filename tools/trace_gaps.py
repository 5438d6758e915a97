#!/usr/bin/env python
"""PR_TRACE file -> the longest intervals between consecutive marks of each thread, and the marks of all threads around the longest one."""
import sys
rows = [l.split(" ", 2) for l in open(sys.argv[1]).read().splitlines()]
rows = [(int(a), b, c) for a, b, c in rows]
rows.sort()
t0 = rows[0][0]
by = {}
for ns, tid, tag in rows: by.setdefault(tid, []).append((ns, tag))
gaps = []
for tid, v in by.items():
    for (a, ta), (b, tb) in zip(v, v[1:]): gaps.append((b - a, tid, a, ta, tb))
gaps.sort(reverse=True)
for g, tid, a, ta, tb in gaps[:10]: print("%8.3f ms  thread %s  at %9.3f ms: [%s] -> [%s]" % (g / 1e6, tid, (a - t0) / 1e6, ta, tb))
g, tid, a, ta, tb = gaps[int(sys.argv[2]) if len(sys.argv) > 2 else 0]
print("--- all threads around the gap at %.3f ms" % ((a - t0) / 1e6))
for ns, t, tag in rows:
    if a - 3e6 <= ns <= a + g + 2e6: print("%9.3f ms  %s  %s" % ((ns - t0) / 1e6, t, tag))
