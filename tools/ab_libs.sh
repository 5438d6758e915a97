#!/bin/bash
# Same-box A/B of library variants built beforehand (PR_BUILD_OUT=pose_refine_amd/lib/variants/X.so python -m pose_refine_amd.build):
#   tools/ab_libs.sh nn|proj|bench-nn|bench-proj  X.so Y.so ...      (paths relative to the repo; "-" = the in-tree library)
# nn / proj: per-pass kernel times of one 256-hypothesis batch as one pose group (rocprofv3 --kernel-trace of tools/pmc_workload.py);
# bench-*: bench.py throughput, two rounds.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mode=$1; shift
OUT=gpurun_out/ab; mkdir -p $OUT
P="import json,sys; d=json.loads(sys.stdin.read()); print('%.0f poses/s %.3f ms' % (d['value'], d['ms_per_step']))"
for round in 1 2; do for v in "$@"; do
  if [ "$v" = "-" ]; then unset PR_LIB_PATH; else export PR_LIB_PATH=$GRAFT_REPO_ROOT/$v; fi
  echo "== [$v] round $round"
  case $mode in
    nn|proj)
      [ $round = 2 ] && continue
      arg=""; [ $mode = nn ] && arg="nn"
      PR_OPTS="pose_groups=1,graph=0" timeout 300 rocprofv3 --kernel-trace -d $OUT/t -o t -- python tools/pmc_workload.py 256 $arg > $OUT/log.txt 2>&1
      python - $OUT/t/t_results.db <<'PY'
import sqlite3, sys, glob
c = sqlite3.connect(glob.glob(sys.argv[1].replace("t/t_results.db", "t/**/t_results.db"), recursive=True)[0] if not __import__("os").path.exists(sys.argv[1]) else sys.argv[1])
rows = list(c.execute("select name, start, (end-start)/1000.0 from kernels order by start"))
for key in ("nn_search", "nn_bound", "nn_tree", "icp_pass", "nn_late", "raster_kernel"):
    v = [r[2] for r in rows if key in r[0]]
    if v: print("%-13s us:" % key, " ".join(f"{x:.0f}" for x in v[-21:]), " sum %.2f ms" % (sum(v[-21:]) / 1e3))
PY
      rm -rf $OUT/t ;;
    bench-nn) timeout 300 python bench.py --scene nn --steps 30 --warmup 3 --no-cpu-baseline --no-live-pmc 2>/dev/null | tail -1 | python -c "$P" ;;
    bench-proj) timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-live-pmc --no-kdtree-extra 2>/dev/null | tail -1 | python -c "$P" ;;
  esac
done; done
