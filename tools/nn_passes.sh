#!/bin/bash
# Per-launch durations of the kd-tree search / pass kernels over one 256-pose batch (one pose group, so launches do not overlap).
#   tools/nn_passes.sh "nn_run=16,nn_grid=1" [more option sets ...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/nnp; mkdir -p $OUT
for opts in "$@"; do
  tag=$(echo $opts | tr ",=" "__")
  PR_OPTS="pose_groups=1,graph=0,$opts" timeout 300 rocprofv3 --kernel-trace -d $OUT/t_$tag -o t -- python tools/pmc_workload.py 256 nn > $OUT/log_$tag.txt 2>&1
  python - $OUT/t_$tag/t_results.db "$opts" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, (end-start)/1000.0 from kernels where name like '%nn_search%' or name like '%nn_tree%' or name like '%icp_pass%' or name like '%nn_late%' order by start"))
# last batch = last 21 passes
srch = [r[2] for r in rows if 'nn_search' in r[0]]
pss = [r[2] for r in rows if 'icp_pass' in r[0]]
n = 21
print("==", sys.argv[2])
tree = [r[2] for r in rows if 'nn_tree' in r[0]]
if srch: print("search us:", " ".join(f"{v:.0f}" for v in srch[-n:]), " sum %.2f ms" % (sum(srch[-n:]) / 1e3))
if tree: print("tree   us:", " ".join(f"{v:.0f}" for v in tree[-n:]), " sum %.2f ms" % (sum(tree[-n:]) / 1e3))
late = [r[2] for r in rows if 'nn_late' in r[0]]
if late: print("late   us:", " ".join(f"{v:.0f}" for v in late[-n:]), " sum %.2f ms" % (sum(late[-n:]) / 1e3))
print("pass   us:", " ".join(f"{v:.0f}" for v in pss[-n:]), " sum %.2f ms" % (sum(pss[-n:]) / 1e3))
PY
  rm -rf $OUT/t_$tag
done
