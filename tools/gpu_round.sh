#!/bin/bash
# One GPU-box session that produces everything the round's evidence needs (run through gpurun):
#   tests, smoke, bench lines (P=256 proj default, host solve, P=1024, NN), rocprofv3 kernel stats, PMC passes.
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/round; mkdir -p $OUT
python -m pytest tests -m gpu -q 2>&1 | tail -3 > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt
python __graft_entry__.py --smoke 2>&1 | tail -2 > $OUT/smoke.txt; cat $OUT/smoke.txt
python bench.py 2>/dev/null | tail -1 > $OUT/bench_p256_proj.json
python bench.py --steps 40 --warmup 5 --solve host --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_p256_proj_hostsolve.json
python bench.py --sequential --fused-solve 0 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_p256_proj_sequential_unfused.json
python bench.py --steps 60 --poses 1024 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_p1024_proj.json
python bench.py --steps 60 --poses 512 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_p512_proj.json
python bench.py --steps 5 --warmup 2 --scene nn --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_p256_nn.json
rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- python bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2>/dev/null
python tools/rocpd_summary.py $OUT/stats/bench_results.db > $OUT/kernel_stats_bench.md
rocprofv3 --kernel-trace --stats -d $OUT/stats1 -o bench -- python bench.py --sequential --no-cpu-baseline > $OUT/bench_under_rocprof_sequential.json 2>/dev/null
python tools/rocpd_summary.py $OUT/stats1/bench_results.db > $OUT/kernel_stats_bench_sequential.md
python bench.py --sequential --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_p256_proj_sequential.json
for c in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do n=$(echo $c | tr " " "_"); rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc -o pmc_$n -- python tools/pmc_workload.py 256 > /dev/null 2>&1; python tools/rocpd_summary.py $OUT/pmc/pmc_${n}_results.db | sed -n '/PMC counters/,$p' > $OUT/pmc_$n.md; done
for c in "FETCH_SIZE" "WRITE_SIZE"; do rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc -o pmc1024_$c -- python tools/pmc_workload.py 1024 > /dev/null 2>&1; python tools/rocpd_summary.py $OUT/pmc/pmc1024_${c}_results.db | sed -n '/PMC counters/,$p' > $OUT/pmc1024_$c.md; done
rm -rf $OUT/stats $OUT/stats1 $OUT/pmc
lscpu | grep -E 'Model name|^CPU\(s\)|Socket|Core' > $OUT/host_cpu.txt
for f in $OUT/bench_*.json; do echo "== $f"; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print('%.0f poses/s  %.3f ms/step  frac %.3f  launch %.1f us' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us']), d.get('cpu_baseline',''))"; done
