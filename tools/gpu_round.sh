#!/bin/bash
# One GPU-box session that produces the round's evidence (run through gpurun): tests, smoke, bench lines, rocprofv3 kernel
# stats, PMC passes, BASELINE configs[4], the 2-rank share-device run, the C++ shard driver.  Output: gpurun_out/round/
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/round; rm -rf $OUT; mkdir -p $OUT                 # (gpurun merges into gpurun_out: start from an empty directory, nothing stale gets copied to profiles/)
B="timeout 600 python bench.py"
summ() { python tools/rocpd_summary.py "$1"; }
# the GPU tests under a kernel trace: test summary + which kernel instantiations of the library they dispatched
rocprofv3 --kernel-trace --stats -d $OUT/cov -o t -- python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt
python tools/kernel_coverage.py $OUT/cov > $OUT/kernel_coverage.md; tail -1 $OUT/kernel_coverage.md; rm -rf $OUT/cov
python __graft_entry__.py --smoke 2>&1 | tail -2 > $OUT/smoke.txt; cat $OUT/smoke.txt
# VALU issue calibration: wave-instructions per second by instruction class -> bench.py's VALU_PEAK
mkdir -p tools/scratch; hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o tools/scratch/valu_peak 2>/dev/null && tools/scratch/valu_peak > $OUT/valu_peak.md 2>&1; tail -2 $OUT/valu_peak.md
$B 2>/dev/null | tail -1 > $OUT/bench_p256_proj.json
$B --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_p256_proj_steps20.json
$B --steps 40 --warmup 5 --solve host --no-cpu-baseline --no-kdtree-extra 2>/dev/null | tail -1 > $OUT/bench_p256_proj_hostsolve.json
$B --sequential --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_p256_proj_sequential.json
$B --sequential --fused-solve 0 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_p256_proj_sequential_unfused.json
$B --steps 60 --poses 1024 --no-cpu-baseline --no-kdtree-extra 2>/dev/null | tail -1 > $OUT/bench_p1024_proj.json
$B --steps 60 --poses 512 --no-cpu-baseline --no-kdtree-extra 2>/dev/null | tail -1 > $OUT/bench_p512_proj.json
$B --steps 20 --warmup 3 --scene nn 2>/dev/null | tail -1 > $OUT/bench_p256_nn.json
# rocprofv3 kernel stats of the same commands
rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- python bench.py --no-cpu-baseline --no-kdtree-extra > $OUT/bench_under_rocprof.json 2>/dev/null
summ $OUT/stats/bench_results.db > $OUT/kernel_stats_bench_p256.md
python tools/step_timeline.py $OUT/stats/bench_results.db > $OUT/step_timeline.md
rocprofv3 --kernel-trace --stats -d $OUT/stats1 -o bench -- python bench.py --sequential --no-cpu-baseline > $OUT/bench_under_rocprof_sequential.json 2>/dev/null
summ $OUT/stats1/bench_results.db > $OUT/kernel_stats_bench_p256_sequential.md
rocprofv3 --kernel-trace --stats -d $OUT/stats2 -o bench -- python bench.py --steps 20 --warmup 3 --scene nn --no-cpu-baseline > $OUT/bench_nn_under_rocprof.json 2>/dev/null
summ $OUT/stats2/bench_results.db > $OUT/kernel_stats_bench_p256_nn.md
# PMC passes (each counter set in a run of its own): HBM bytes of the correspondence kernels, projective and kd-tree
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do n=$(echo $c | tr " " "_")
  PR_RASTER_MODE=0 timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc -o p_$n -- python tools/pmc_workload.py 256 > /dev/null 2>&1
  summ $OUT/pmc/p_${n}_results.db | sed -n '/PMC counters/,$p' | grep -E "counter|---|icp_pass|max2zero|fill_i32|raster_kernel" > $OUT/pmc_proj_$n.md
  PR_RASTER_MODE=0 timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc -o n_$n -- python tools/pmc_workload.py 256 nn > /dev/null 2>&1
  summ $OUT/pmc/n_${n}_results.db | sed -n '/PMC counters/,$p' | grep -E "counter|---|icp_pass|nn_search|nn_bound|nn_tree|max2zero|fill_i32" > $OUT/pmc_nn_$n.md
done
for c in "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum" "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE"; do n=$(echo $c | tr " " "_")
  PR_RASTER_MODE=0 timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc -o s_$n -- python tools/pmc_workload.py 256 nn > /dev/null 2>&1
  summ $OUT/pmc/s_${n}_results.db | sed -n '/PMC counters/,$p' | grep -E "counter|---|nn_search|nn_bound|nn_tree|icp_pass" > $OUT/sq_nn_$n.md
  PR_RASTER_MODE=0 PR_OPTS="pose_groups=1" timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc -o q_$n -- python tools/pmc_workload.py 256 > /dev/null 2>&1
  summ $OUT/pmc/q_${n}_results.db | sed -n '/PMC counters/,$p' | grep -E "counter|---|icp_pass|raster_kernel" > $OUT/sq_proj_$n.md
done
for c in "FETCH_SIZE" "WRITE_SIZE"; do
  PR_OPTS="sub_batch=1024,pose_groups=1" timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc -o big_$c -- python tools/pmc_workload.py 1024 > /dev/null 2>&1
  summ $OUT/pmc/big_${c}_results.db | sed -n '/PMC counters/,$p' | grep -E "counter|---|icp_pass|max2zero|fill_i32|raster_kernel" > $OUT/pmc_proj_p1024_onebatch_$c.md
done
PR_OPTS="sub_batch=1024,pose_groups=1" rocprofv3 --kernel-trace --stats -d $OUT/stats4 -o big -- python tools/pmc_workload.py 1024 > /dev/null 2>&1
summ $OUT/stats4/big_results.db > $OUT/kernel_stats_p1024_onebatch.md
python tools/nn_counters.py > $OUT/nn_work_counters.md 2>/dev/null
bash tools/pmc_allpasses.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS" "TA_BUSY_avr GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" -- - 2>/dev/null | grep -v "^$" > $OUT/sq_nn_allpasses.txt
bash tools/nn_passes_env.sh "PR_OPTS_EXTRA=default" 2>/dev/null | grep -E "==|us:" > $OUT/nn_per_pass_us.txt
bash tools/nn_passes.sh "nn_wide=0" 2>/dev/null | grep -E "==|us:" >> $OUT/nn_per_pass_us.txt
# BASELINE configs[4]: 1M triangles, 1280x720, 128 hypotheses (the per-GPU share of 1024 over 8)
timeout 600 python tools/config5.py 128 --check > $OUT/config5.txt 2>&1; tail -4 $OUT/config5.txt
rocprofv3 --kernel-trace --stats -d $OUT/stats3 -o c5 -- python tools/config5.py 128 > /dev/null 2>&1
summ $OUT/stats3/c5_results.db > $OUT/kernel_stats_config5.md
# two ranks on the one GPU of this box, started by bench.py itself: one process per rank under torch.distributed.run, and one host thread per rank
PR_BENCH_SHARE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 40 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_2ranks_share_device_processes.json
PR_BENCH_SHARE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 40 --warmup 5 --launcher threads 2>/dev/null | tail -1 > $OUT/bench_2ranks_share_device_threads.json
# EIGHT ranks on the one GPU: the driver's SCALE job on this box's 16-CPU cgroup, both launchers; per-rank host CPU over the timed region is in the line
PR_BENCH_SHARE_DEVICE=1 timeout 900 python bench.py --gpus 8 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_8ranks_share_device_processes.json
PR_BENCH_SHARE_DEVICE=1 timeout 900 python bench.py --gpus 8 --steps 20 --warmup 5 --launcher threads 2>/dev/null | tail -1 > $OUT/bench_8ranks_share_device_threads.json
# eight rank THREADS on the one GPU gathering through pr_gather_results' N > 1 branch over the loop-back stand-in for librccl (tests/rccl_loopback)
hipcc -shared -fPIC -O2 tests/rccl_loopback/loopback_rccl.cpp -o tests/rccl_loopback/librccl_loopback.so 2>/dev/null
PR_BENCH_SHARE_DEVICE=1 PR_RCCL_LIBRARY=$PWD/tests/rccl_loopback/librccl_loopback.so timeout 900 python bench.py --gpus 8 --steps 20 --warmup 5 --launcher threads 2>/dev/null | tail -1 > $OUT/bench_8ranks_loopback_gather_threads.json
PR_RCCL_LIBRARY=$PWD/tests/rccl_loopback/librccl_loopback.so timeout 300 python tests/rccl_loopback/run_gather.py 8 2>/dev/null | tail -1 > $OUT/gather_loopback_8ranks.json; cut -c1-300 $OUT/gather_loopback_8ranks.json
# the whole N > 1 machinery with a world of one: torch's RCCL process group and the library's own dlopened RCCL in one process
PR_BENCH_FORCE_COMM=1 timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_force_comm_world1.json
# the host-solve pipeline over time from its first step (round 6: one flag per pose group polled by the helper threads), and the device-solve one
for m in host host device; do timeout 300 python tools/host_warm_probe.py $m 2>/dev/null | tail -1; done > $OUT/pipeline_rate_over_time.txt
for hp in 0 1 0 1 0 1; do timeout 200 python bench.py --steps 40 --warmup 5 --solve host --no-cpu-baseline --no-kdtree-extra --no-live-pmc --opt host_poll=$hp 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('host_poll $hp: %.0f poses/s, largest step %.2f ms, closing fence %.2f ms' % (d['value'], d['step_ms_spread']['max'], d['step_ms_spread']['closing_fence_ms']))"; done > $OUT/host_poll_ab.txt; cat $OUT/host_poll_ab.txt
# a new scene per frame -- what it costs next to a batch, and frames pipelined through the two slots
timeout 300 python tools/scene_frame_time.py 2>/dev/null | grep "valid pixels" > $OUT/scene_frame_time.txt
timeout 300 python tools/frames_pipe.py 2>/dev/null | grep "frame" >> $OUT/scene_frame_time.txt; cat $OUT/scene_frame_time.txt
REPS=4 rocprofv3 --kernel-trace --stats -d $OUT/stats5 -o sf -- python tools/scene_frame_time.py > /dev/null 2>&1
summ $OUT/stats5/sf_results.db | grep -E "^# |^\| kernel|---|kd_|nn_wide|nn_accel|nn_records|nn_frame|nn_grid|nn_gather|scene_proj_prepare|pack_proj|fingerprint" > $OUT/kernel_stats_scene_preparation.md; rm -rf $OUT/stats5
# C++ host: shard driver
g++ -std=c++14 -O2 -pthread -Iinclude tests/cpp/shard_test.cpp -o tests/cpp/shard_test -Lpose_refine_amd/lib -lpose_refine_hip -Wl,-rpath,$PWD/pose_refine_amd/lib && ./tests/cpp/shard_test tests/golden/ 4096 2>&1 | tail -1 > $OUT/shard_test_4096.json; cat $OUT/shard_test_4096.json
python -c "from pose_refine_amd import api; print('visible devices:', api.device_count())" >> $OUT/shard_test_4096.json 2>/dev/null
rm -rf $OUT/stats $OUT/stats1 $OUT/stats2 $OUT/stats3 $OUT/stats4 $OUT/pmc
lscpu | grep -E 'Model name|^CPU\(s\)|Socket|Core' > $OUT/host_cpu.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/host_cpu.txt 2>/dev/null
bash tools/pmc_pass.sh 0 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES" "TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE" -- - 2>&1 | grep -v "^$" > $OUT/sq_nn_pass0.txt
for f in $OUT/bench_*.json; do echo "== $f"; python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1])
r=d['roofline']
print('%.0f poses/s  %.3f ms/step  frac %.3f (algorithmic %.3f, end to end %.3f)  launch %.1f us' % (d['value'], d['ms_per_step'], r['frac'], r['frac_algorithmic'], r['frac_end_to_end'], r['avg_launch_us']), d.get('cpu_baseline',{}).get('value',''), d.get('config2_kdtree',{}).get('value',''), d.get('default_criteria',{}).get('projective',{}).get('value',''), d.get('default_criteria',{}).get('kdtree',{}).get('value',''))"; done
