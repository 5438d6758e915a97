#!/bin/bash
# usage: tools/ab_flags.sh "<flags 1>" "<flags 2>" ... ; rebuilds per variant ON THE GPU BOX and measures the pipelined loop (tools/ab_option.py)
for v in "$@"; do
  echo "=== variant: [$v]"
  PR_EXTRA_FLAGS="$v" python -m pose_refine_amd.build --force > /dev/null 2>&1 || { echo BUILD FAILED; continue; }
  python tools/ab_option.py pose_groups 2 ${AB_POSES:-256} 2>&1 | grep "="
done
python -m pose_refine_amd.build --force > /dev/null 2>&1
