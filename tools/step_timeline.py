#!/usr/bin/env python
"""Where a pipelined step's time goes: from a rocprofv3 --kernel-trace database of `bench.py`, over the steady-state middle of
the run: GPU busy fraction (union of kernel intervals), idle gaps, and per kernel the time it runs ALONE vs overlapped.

    rocprofv3 --kernel-trace -d out -o b -- python bench.py --no-cpu-baseline --no-kdtree-extra ; python tools/step_timeline.py out/b_results.db [--dump]

--dump also lists every launch of two consecutive steps from the middle of the run (start relative to the first, duration, queue).
"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = [(n.split("(")[0].replace("void ", "").replace("prk::", ""), s, e) for n, s, e in c.execute("select name, start, end from kernels order by start")]
ras = [i for i, r in enumerate(rows) if r[0].startswith("raster_kernel")]
i0, i1 = ras[len(ras) // 3], ras[2 * len(ras) // 3]
steps = i1 and (2 * len(ras) // 3 - len(ras) // 3)
win = rows[i0:i1]
t0, t1 = win[0][1], rows[i1][1]
ev = []
for n, s, e in win:
    ev.append((s, 1, n)); ev.append((min(e, t1), -1, n))
ev.sort()
busy = 0; alone = {}; total = {}; active = {}; last = t0
for t, d, n in ev:
    dt = t - last
    if active:
        busy += dt
        for k in active: total[k] = total.get(k, 0) + dt
        if len(active) == 1:
            k = next(iter(active)); alone[k] = alone.get(k, 0) + dt
    last = t
    active[n] = active.get(n, 0) + d
    if active[n] <= 0: del active[n]
span = t1 - t0
print(f"steps {steps}, {span / steps / 1e3:.1f} us per step, GPU busy {100 * busy / span:.1f} %, idle {(span - busy) / steps / 1e3:.1f} us per step")
print("| kernel | running (us/step) | running alone (us/step) |\n|---|---:|---:|")
for k in sorted(total, key=lambda k: -total[k]):
    print(f"| `{k[:60]}` | {total[k] / steps / 1e3:.1f} | {alone.get(k, 0) / steps / 1e3:.1f} |")

if "--dump" in sys.argv:
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    mid = len(ras) // 2
    a, b = rows[ras[mid]][1], rows[ras[mid + 2]][1]
    print("\n| start (us) | dur (us) | queue | grid | kernel |\n|---:|---:|---:|---|---|")
    gcol = "grid_size_x || 'x' || grid_size_y" if "grid_size_x" in cols else "''"
    for n, s_, e_, q, gsz in c.execute(f"select name, start, end, {qcol}, {gcol} from kernels where start >= {a} and start < {b} order by start"):
        print(f"| {(s_ - a) / 1e3:.1f} | {(e_ - s_) / 1e3:.1f} | {q} | {gsz} | `{n.split('(')[0].replace('void ', '').replace('prk::', '')[:50]}` |")
