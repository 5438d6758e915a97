#!/bin/bash
# raster_kernel: hypotheses per workgroup (PR_RASTER_RUN) -- BASELINE configs[4] (1 M triangles) and configs[1] (obj_06)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for r in "$@"; do echo "== PR_RASTER_RUN=$r"
  PR_RASTER_RUN=$r python tools/config5.py 128 2>&1 | tail -1
  PR_RASTER_RUN=$r python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-kdtree-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['phase_ms_per_timed_step'])"
done
for r in "$@"; do PR_RASTER_RUN=$r rocprofv3 --kernel-trace --stats -d gpurun_out/rr$r -o c5 -- python tools/config5.py 128 > /dev/null 2>&1; echo "run $r: $(python tools/rocpd_summary.py gpurun_out/rr$r/c5_results.db | grep -E 'raster_kernel')"; rm -rf gpurun_out/rr$r; done
