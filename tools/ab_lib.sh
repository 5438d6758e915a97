#!/bin/bash
# same-box A/B of two builds of the library: tools/ab_lib.sh <other.so> [bench args...]  (runs new, other, new, other, new, other)
OTHER=$1; shift
L=pose_refine_amd/lib/libpose_refine_hip.so
cp $L /tmp/new.so
ARGS=${AB_ARGS:-"--poses 256 --steps 100|--poses 512 --steps 60|--poses 1024 --steps 40"}
IFS='|' read -ra SETS <<< "$ARGS"
for v in new other new other new other; do
  if [ $v = new ]; then cp /tmp/new.so $L; else cp $OTHER $L; fi
  for a in "${SETS[@]}"; do
    python bench.py $a --no-cpu-baseline --no-kdtree-extra "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$a', round(d['value']), 'poses/s', round(d['ms_per_step'],4), 'ms frac', round(d['roofline']['frac'],3))"
  done
done
cp /tmp/new.so $L
