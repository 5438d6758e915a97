#!/bin/bash
# A/B of compile-time variants INSIDE one GPU session (boxes of the pool differ by several per cent, so variants must share a box):
#   tools/ab_flags.sh "" "-DPR_BRANCHLESS_PROJ=0" ...      each argument = extra hipcc flags of one variant ("" = as committed)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P="import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.0f poses/s %.3f ms frac %.3f launch %.1f us' % (d['value'], d['ms_per_step'], r['frac'], r['avg_launch_us']))"
for round in 1 2; do
for f in "$@"; do
  PR_EXTRA_FLAGS="$f" python -m pose_refine_amd.build --force > /dev/null 2>&1
  echo "== [$f] pipelined / sequential-unfused ${BENCH_ARGS:-}"
  timeout 200 python bench.py --no-cpu-baseline --no-kdtree-extra ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python -c "$P"
  timeout 200 python bench.py --no-cpu-baseline --sequential --fused-solve 0 ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python -c "$P"
done; done
python -m pose_refine_amd.build --force > /dev/null 2>&1
