#!/bin/bash
# One counter set over ALL 21 passes of the kd-tree kernels of the last 256-hypothesis batch (one pose group), per library variant:
#   tools/pmc_allpasses.sh "SQ_INSTS_VALU SQ_INSTS_SALU" -- lib1.so lib2.so ...     ("-" = the in-tree library); values in millions
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
sets=(); while [ "$1" != "--" ]; do sets+=("$1"); shift; done; shift
OUT=gpurun_out/pmca; mkdir -p $OUT
for v in "$@"; do
  if [ "$v" = "-" ]; then unset PR_LIB_PATH; else export PR_LIB_PATH=$GRAFT_REPO_ROOT/$v; fi
  echo "== [$v]"
  for c in "${sets[@]}"; do
    rm -rf $OUT/p
    PR_OPTS="pose_groups=1,graph=0" PR_RASTER_MODE=0 timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/p -o p -- python tools/pmc_workload.py 256 nn > $OUT/log.txt 2>&1
    python - $OUT/p <<'PY'
import sqlite3, sys, glob
db = glob.glob(sys.argv[1] + "/**/p_results.db", recursive=True)[0]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(pmc_events)")]
key = "dispatch_id" if "dispatch_id" in cols else cols[0]
rows = list(c.execute(f"select name, {key}, counter_name, sum(counter_value) from pmc_events group by name, {key}, counter_name order by {key}"))
names = sorted({r[2] for r in rows})
for k in ("nn_search", "nn_bound", "nn_tree", "icp_pass"):
    ids = sorted({r[1] for r in rows if k in r[0]})[-21:]
    for n in names:
        vals = [sum(r[3] for r in rows if r[1] == i and r[2] == n) for i in ids]
        print("  %-10s %-22s M:" % (k, n), " ".join("%.1f" % (v / 1e6) for v in vals), " sum %.0f" % (sum(vals) / 1e6))
PY
  done
done
rm -rf $OUT/p
