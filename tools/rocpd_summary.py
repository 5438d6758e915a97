#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (*.db) as a per-kernel table (markdown + csv rows).

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.md

Equivalent to `rocprofv3 --kernel-trace --stats` kernel_stats output; also prints PMC counter sums
per kernel when the run collected any (--pmc).
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    print(f"# rocprofv3 kernel summary: {path}\n")
    print("| kernel | calls | total us | avg us | % |")
    print("|---|---:|---:|---:|---:|")
    for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        short = name.split("(")[0].replace("void ", "")
        print(f"| `{short}` | {calls} | {total:.1f} | {avg:.2f} | {pct:.1f} |")
    try:
        rows = list(c.execute(
            "select name, counter_name, count(*), sum(counter_value) from pmc_events group by name, counter_name order by name"))
    except sqlite3.Error as e:
        rows = []
        print(f"\n(no PMC data: {e})")
    if rows:
        print("\n## PMC counters (sum over dispatches)\n")
        print("| kernel | counter | dispatches | sum | per dispatch |")
        print("|---|---|---:|---:|---:|")
        for name, ctr, n, tot in rows:
            short = name.split("(")[0].replace("void ", "")
            print(f"| `{short}` | {ctr} | {n} | {tot:.0f} | {tot / n:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1])
