#!/bin/bash
# SQ / TA / LDS counters of the kernels whose name matches a pattern, over three fused batches (tools/pmc_workload.py):
#   tools/pmc_kernel.sh nn "nn_tree|nn_search" [hypotheses] > out.md          (each counter set in a rocprofv3 run of its own)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SCENE=${1:-nn}; PAT=${2:-nn_tree}; P=${3:-256}
OUT=gpurun_out/pmck; mkdir -p $OUT
for c in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES" "SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS"; do
  n=$(echo $c | tr " " "_")
  PR_OPTS="pose_groups=1,graph=0" timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc -o p_$n -- python tools/pmc_workload.py $P $SCENE > $OUT/log_$n.txt 2>&1
  python tools/rocpd_summary.py $OUT/pmc/p_${n}_results.db | sed -n '/PMC counters/,$p' | grep -E "counter \||$PAT|---"
done
rm -rf $OUT/pmc
