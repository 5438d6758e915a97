#!/usr/bin/env python
"""List the individual dispatches of kernels matching a substring from a rocprofv3 rocpd database, in start order.

    python tools/rocpd_trace.py x_results.db icp_pass [max_rows]
"""
import sqlite3
import sys


def main(path, needle, limit=200):
    c = sqlite3.connect(path)
    views = [r[0] for r in c.execute("select name from sqlite_master where type in ('view','table')")]
    src = "kernels" if "kernels" in views else None
    if src is None:
        print("views/tables:", views)
        return
    cols = [r[1] for r in c.execute(f"pragma table_info({src})")]
    print("#", cols)
    q = f"select name, start, end, (end-start)/1000.0 from {src} where name like ? order by start limit ?"
    t0 = None
    for name, st, en, us in c.execute(q, (f"%{needle}%", limit)):
        t0 = t0 or st
        print(f"{(st - t0) / 1000.0:12.1f} us  +{us:9.1f} us  {name.split('(')[0][:70]}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 200)
