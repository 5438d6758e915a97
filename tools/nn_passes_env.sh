#!/bin/bash
# like nn_passes.sh, but each argument is "ENV=.. ENV=.." exported for the run (kernel experiments)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/nnp; mkdir -p $OUT
i=0
for envs in "$@"; do
  i=$((i+1))
  env $envs PR_OPTS="pose_groups=1,graph=0" timeout 300 rocprofv3 --kernel-trace -d $OUT/e_$i -o t -- python tools/pmc_workload.py 256 nn > $OUT/log_e_$i.txt 2>&1
  python - $OUT/e_$i/t_results.db "$envs" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, (end-start)/1000.0 from kernels where name like '%nn_search%' or name like '%nn_tree%' or name like '%nn_bound%' or name like '%icp_pass%' or name like '%nn_late%' order by start"))
n = 21
print("==", sys.argv[2])
for key in ("nn_search", "nn_bound", "nn_tree", "icp_pass", "nn_late"):
    v = [r[2] for r in rows if key in r[0]]
    if v: print("%-9s us:" % key, " ".join(f"{x:.0f}" for x in v[-n:]), " sum %.2f ms" % (sum(v[-n:]) / 1e3))
PY
  rm -rf $OUT/e_$i
done
