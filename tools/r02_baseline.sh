#!/bin/bash
# Round-2 opening GPU session: the round-1 state re-measured on this round's box + the kd-tree (NN) kernel's PMC passes.
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_base; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_p256_proj.json
timeout 300 python bench.py --steps 10 --warmup 2 --scene nn --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_p256_nn.json
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum" "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  n=$(echo $c | tr " " "_")
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc -o p_$n -- python tools/pmc_workload.py 256 nn > $OUT/log_$n.txt 2>&1
  python tools/rocpd_summary.py $OUT/pmc/p_${n}_results.db | sed -n '/PMC counters/,$p' | grep -E "counter|---|icp_pass|max2zero|fill_i32" > $OUT/pmc_nn_$n.md; cat $OUT/pmc_nn_$n.md
done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o nn -- python tools/pmc_workload.py 256 nn > /dev/null 2>&1
python tools/rocpd_summary.py $OUT/stats/nn_results.db > $OUT/kernel_stats_nn_p256.md
rm -rf $OUT/pmc $OUT/stats
for f in $OUT/bench_*.json; do echo "== $f"; python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1])
print('%.0f poses/s  %.3f ms/step  frac %.3f  launch %.1f us' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us']))"; done
