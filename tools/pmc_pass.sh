#!/bin/bash
# SQ counters of ONE pass of the kd-tree kernels (default: pass 0 of the last 256-hypothesis batch, one pose group), per library variant:
#   tools/pmc_pass.sh <pass> "<counter set 1>" "<counter set 2>" -- lib1.so lib2.so ...     ("-" = the in-tree library)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
pass=$1; shift
sets=(); while [ "$1" != "--" ]; do sets+=("$1"); shift; done; shift
OUT=gpurun_out/pmcp; mkdir -p $OUT
for v in "$@"; do
  if [ "$v" = "-" ]; then unset PR_LIB_PATH; else export PR_LIB_PATH=$GRAFT_REPO_ROOT/$v; fi
  echo "== [$v] pass $pass"
  for c in "${sets[@]}"; do
    rm -rf $OUT/p
    PR_OPTS="pose_groups=1,graph=0" PR_RASTER_MODE=0 timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/p -o p -- python tools/pmc_workload.py 256 nn > $OUT/log.txt 2>&1
    python - $OUT/p $pass <<'PY'
import sqlite3, sys, glob
db = glob.glob(sys.argv[1] + "/**/p_results.db", recursive=True)[0]
c = sqlite3.connect(db)
p = int(sys.argv[2])
cols = [r[1] for r in c.execute("pragma table_info(pmc_events)")]
key = "dispatch_id" if "dispatch_id" in cols else cols[0]
rows = list(c.execute(f"select name, {key}, counter_name, sum(counter_value) from pmc_events group by name, {key}, counter_name order by {key}"))
for k in ("nn_search", "nn_bound", "nn_tree"):
    ids = sorted({r[1] for r in rows if k in r[0]})
    if not ids: continue
    want = ids[-21:][p]
    vals = {r[2]: r[3] for r in rows if r[1] == want}
    print("  %-10s" % k, "  ".join(f"{n}={v:.4g}" for n, v in sorted(vals.items())))
PY
  done
done
rm -rf $OUT/p
