#!/bin/bash
# kd-tree search A/B of compile-time variants inside ONE GPU session: per variant (extra hipcc flags) the per-pass durations of the
# search / tree / pass kernels over a 256-hypothesis batch and the work counters of passes 0-3.
#   tools/nn_ab_flags.sh "" "-DPR_WIDE_STACK=16" ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for f in "$@"; do
  PR_EXTRA_FLAGS="$f" python -m pose_refine_amd.build --force > /dev/null 2>&1 || { echo "build failed: $f"; continue; }
  echo "#### [$f]"
  bash tools/nn_passes_env.sh "PR_X=0" 2>/dev/null | grep -E "nn_tree|nn_search|nn_bound"
  python tools/nn_counters.py 2>/dev/null | sed -n 3,6p
done
python -m pose_refine_amd.build --force > /dev/null 2>&1
