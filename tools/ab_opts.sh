#!/bin/bash
# A/B of bench.py argument sets inside one GPU session:  tools/ab_opts.sh "" "--opt points_per_block=3072" ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P="import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.0f poses/s %.3f ms frac %.3f launch %.1f us' % (d['value'], d['ms_per_step'], r['frac'], r['avg_launch_us']))"
for round in 1 2; do for a in "$@"; do echo "== [$a]"; timeout 200 python bench.py --no-cpu-baseline --no-kdtree-extra $a 2>/dev/null | tail -1 | python -c "$P"; done; done
