cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/sq; mkdir -p $OUT
for c in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES" "SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do n=$(echo $c | tr " " "_"); PR_RASTER_MODE=0 timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc -o p_$n -- python tools/pmc_workload.py 256 > $OUT/log_$n.txt 2>&1; python tools/rocpd_summary.py $OUT/pmc/p_${n}_results.db | sed -n '/PMC counters/,$p' | grep -E "counter|raster_kernel|icp_pass|---" > $OUT/sq_$n.md; cat $OUT/sq_$n.md; done
rm -rf $OUT/pmc
