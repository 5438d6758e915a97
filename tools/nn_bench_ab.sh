#!/bin/bash
# configs[2] throughput (two slots) per compile-flag variant / bench argument set, in ONE GPU session
#   tools/nn_bench_ab.sh flags "" "-DPR_TREE_GX=4" ...     |    tools/nn_bench_ab.sh args "" "--pose-groups 1" ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mode=$1; shift
P="import json,sys; d=json.loads(sys.stdin.read()); print('%.0f poses/s %.3f ms' % (d['value'], d['ms_per_step']))"
for round in 1 2; do for v in "$@"; do
  if [ "$mode" = flags ]; then PR_EXTRA_FLAGS="$v" python -m pose_refine_amd.build --force > /dev/null 2>&1; a=""; else a="$v"; fi
  echo "== [$v]"; timeout 300 python bench.py --scene nn --steps 30 --warmup 3 --no-cpu-baseline $a 2>/dev/null | tail -1 | python -c "$P"
done; done
[ "$mode" = flags ] && python -m pose_refine_amd.build --force > /dev/null 2>&1
