#!/bin/bash
# usage: tools/ab.sh "<flags variant 1>" "<flags variant 2>" ... ; rebuilds the library per variant ON THE GPU BOX and runs tools/tune.py
TUNE_ARGS=${TUNE_ARGS:---poses 256 1024}
for v in "$@"; do
  echo "=== variant: [$v]"
  PR_EXTRA_FLAGS="$v" python -m pose_refine_amd.build --force > /dev/null 2>&1 || { echo BUILD FAILED; continue; }
  python tools/tune.py $TUNE_ARGS 2>&1 | grep 'P='
done
