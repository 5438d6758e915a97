#!/usr/bin/env python
"""bench.py -- refined poses/s of the fused hot path (render -> cloud -> 21-pass ICP) on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic hypotheses per GPU:
BASELINE.json configs[1] -- obj_06.ply, 256-pose batch, 640x480, projective association,
ICPConvergenceCriteria(0,0,20) => exactly 21 correspondence passes / 20 solves per pose
(SURVEY.md 8d).  Model triangles and the scene are resident in HBM before the timed region; the
timed region includes the pose upload (64 B/pose), every kernel, the per-iteration host solve
round trips (PR_SOLVE_HOST) or the device solve, and the result gather.

Multi-GPU: the global batch is cut into contiguous shards (pr_shard_range), rank r refines its shard of
the seeded stream -- no data-path collective -- then ONE RCCL gather of the 72-byte RegistrationResult
records to rank 0 over xGMI: pr_gather_results of the C ABI (grouped ncclSend/ncclRecv on the library's
stream; the communicator's 128-byte id travels over torch.distributed, which also provides the barrier
and the max-over-ranks clock).  PR_BENCH_GATHER=torch uses torch.distributed.gather instead.
  --scaling weak   (default) per-GPU batch fixed: --poses 256 per GPU, or --global-poses G / N
  --scaling strong the global batch is fixed (--global-poses, default 4096 = BASELINE configs[3]) and split over the N ranks
      python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
             bench.py --gpus 8 --global-poses 4096          # configs[3]: 512 hypotheses per GPU

Prints ONE JSON line on rank 0 (contract in the task statement) including `roofline` for the
dominant kernel (the correspondence kernel, HIP-event timed on the library's own stream) and
`cpu_baseline` (the CPU oracle on a bounded sample, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

# The pipelined loop keeps four streams busy side by side (two slots x two pose groups).  The HIP runtime maps every stream of
# the process onto GPU_MAX_HW_QUEUES hardware queues (default 4, shared with torch's and RCCL's streams) and streams that share
# a queue serialise, so ask for 16 (the slots' four, a third pose group for kd-tree scenes, the private contexts of the host-solve threads, torch's
# and RCCL's own) -- before torch initialises the runtime.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12                       # MI355X_MICROARCH.md: 8 TB/s spec
VALU_PEAK = 256 * 4 * 2.4e9 / 4         # wave-instructions/s: 256 CUs x 4 SIMDs, one VALU instruction per 4 cycles at 2.4 GHz = 6.1e11
# SURVEY.md 8d algorithmic bytes of the correspondence kernel: per pass 12 B source read + 24 B
# destination/normal gather, + 12 B write-back on every pass after the first:
# N*(21*36 + 20*12) = 996 N bytes per pose over 21 launches.
BYTES_PER_POINT_PASS0, BYTES_PER_POINT_PASSK = 36.0, 48.0
# Committed rocprofv3 PMC measurements of the correspondence kernel: the fallback of live_pmc_traffic(), which collects the same two counters in
# child passes of this run (counters need rocprofv3 passes
# of their own): fabric-side bytes per point (FETCH_SIZE x2 per the gfx950 correction of MI355X_MICROARCH.md + WRITE_SIZE, both calibrated on
# max2zero_kernel, which moves a known number of bytes) and VALU wave-instructions per point (SQ_INSTS_VALU).  FETCH_SIZE counts what the
# L2s fetch from the fabric, Infinity-Cache hits included (guide, HBM section): the counter figure is FABRIC traffic, an upper bound of DRAM
# traffic -- with <= 512 hypotheses per sub-batch the clouds are Infinity-Cache resident by design.
PMC_TRAFFIC_BYTES_PER_POINT = {"proj": 25.2, "nn": 69.0}       # P = 1024 as ONE sub-batch (clouds spill the Infinity Cache): 23.9 B/point -- the same: it is the cloud read + write-back
PMC_VALU_WAVE_INSTR_PER_POINT = {"proj": 2.085, "nn": None}
PMC_TRAFFIC_SOURCE = {"proj": "profiles/r03/pmc_proj_FETCH_SIZE.md + pmc_proj_WRITE_SIZE.md + sq_proj_SQ_INSTS_VALU*.md (icp_pass_kernel<SceneProjPacked>)",
                      "nn": "profiles/r03/pmc_nn_FETCH_SIZE.md + pmc_nn_WRITE_SIZE.md (search 39.7 + bound 6.3 + task walk 4.9 + winners pass 18.1 B/point)"}


def effective_cpus():
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (cpu.max: a box with 256 online CPUs and a
    quota of 16 runs 256 OpenMP threads on 16 CPUs' worth of time).  Returns (usable, facts)."""
    facts = {"online": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        facts["cgroup_cpu_max"] = f"{q} {per}"
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            facts["cgroup_cfs"] = f"{q} {per}"
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    usable = facts["affinity"]
    if quota is not None:
        usable = max(1, min(usable, int(quota + 0.5)))
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                facts["model"] = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return usable, facts


def flush_c_stdio():
    """RCCL prints its version banner through C stdio when NCCL_DEBUG=VERSION (set in this image); with stdout redirected that text sits in a
    C buffer until the process exits -- i.e. it would land AFTER the JSON line.  Flushing it when it is written keeps the JSON line last."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:                                              # noqa: BLE001
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--poses", type=int, default=256, help="hypotheses per GPU per step (weak scaling)")
    ap.add_argument("--global-poses", type=int, default=0, help="hypotheses per step over ALL GPUs (0: --poses x GPUs); BASELINE configs[3] = 4096 on 8 GPUs")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: per-GPU batch fixed (--poses, or --global-poses / N); strong: the global batch is fixed (--global-poses, default 4096) and split over the ranks")
    ap.add_argument("--scene", choices=["proj", "nn"], default="proj")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--solve", choices=["host", "device"], default=os.environ.get("PR_BENCH_SOLVE", "device"))
    ap.add_argument("--pose-groups", type=int, default=0, help="streams the iteration loop is split over (0 = the library's choice: 2 for projective, 3 for kd-tree scenes)")
    ap.add_argument("--fused-solve", type=int, default=1, help="1: finalize+solve in the tail of the pass kernel (library default)")
    ap.add_argument("--overlap-pass", type=int, default=-1, help="library option overlap_pass (-1: library default)")
    ap.add_argument("--sequential", action="store_true",
                    help="every step through the synchronous single-group path with HIP events around EVERY correspondence launch "
                         "(profile 1): the mode in which rocprofv3's per-launch average and the event average measure the same thing")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="library option (pr_set_option), repeatable -- tuning runs")
    ap.add_argument("--blocking-wait", type=int, default=-1, help="1: pr_refine_wait sleeps instead of spinning (default: 1 when more than one rank shares the host, else 0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--burn-in", type=int, default=0, help="experiments: untimed steps BEFORE the --warmup steps (device clocks up from idle; reported as burn_in_steps).  Default 0: W warm-up steps, K timed steps, nothing else")
    ap.add_argument("--no-live-pmc", action="store_true", help="roofline.traffic from the committed counter passes instead of two rocprofv3 --pmc passes run by this process (rank 0, N = 1)")
    ap.add_argument("--no-kdtree-extra", action="store_true", help="skip the short configs[2] (kd-tree association) measurement appended to the line")
    ap.add_argument("--cpu-poses", type=int, default=0, help="CPU baseline sample size (0 = auto)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    # PR_BENCH_FORCE_COMM=1 (test mode for one-GPU boxes): run the whole N > 1 machinery -- process group on the RCCL backend, C-ABI communicator,
    # pr_gather_results per step -- with a world of ONE rank, so that torch's RCCL and the library's use of it meet in one process
    multi = world > 1 or os.environ.get("PR_BENCH_FORCE_COMM", "0") == "1"
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    # torch first: its bundled HIP runtime must be the one both torch and libpose_refine_hip.so use
    import torch
    import torch.distributed as dist
    import numpy as np
    from pose_refine_amd import api, synth
    from pose_refine_amd import dist as prd

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # PR_BENCH_SHARE_DEVICE=1 is a TEST mode for boxes with one GPU: every rank uses device 0 and the gather runs over
    # gloo on host copies (RCCL refuses two ranks on one device); everything else is the real N>1 code path.
    share_device = os.environ.get("PR_BENCH_SHARE_DEVICE", "0") == "1"
    if share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    api.init(local_rank)
    # the gather of the solved transforms: C ABI (RCCL directly) unless told otherwise or the communicator cannot be formed
    gather_mode = "torch" if (share_device or os.environ.get("PR_BENCH_GATHER", "cabi") == "torch") else "cabi"
    gather_note = None
    if multi and gather_mode == "cabi":
        try:
            ident = [api.comm_id() if rank == 0 else None]
            dist.broadcast_object_list(ident, src=0)               # 128 bytes, once
            api.comm_init_rank(ident[0], rank, world)
        except Exception as e:                                     # noqa: BLE001 -- a failed bootstrap must not cost the measurement
            print(f"[bench] rank {rank}: C-ABI communicator unavailable ({e}); falling back to torch.distributed.gather", file=sys.stderr, flush=True)
            gather_mode = "torch"
            gather_note = f"C-ABI bootstrap failed on rank {rank}: {e}"
        notes = [None] * world
        dist.all_gather_object(notes, gather_note)
        if any(notes):                                             # one rank without a communicator: every rank uses torch's gather, and the line says why
            gather_mode = "torch"
            gather_note = "; ".join(n for n in notes if n)
    if multi:
        dist.barrier()                                              # every rank's communicators exist (and have printed what they print)
    flush_c_stdio()
    api.set_option("blocking_wait", (1 if world > 1 else 0) if args.blocking_wait < 0 else args.blocking_wait)
    api.set_option("solve", api.SOLVE_DEVICE if args.solve == "device" else api.SOLVE_HOST)
    api.set_option("pose_groups", args.pose_groups)
    api.set_option("fused_solve", args.fused_solve)
    if args.overlap_pass >= 0:
        api.set_option("overlap_pass", args.overlap_pass)
    for kv in args.opt:
        name, value = kv.split("=")
        api.set_option(name, int(value))

    W, H, K = synth.WIDTH, synth.HEIGHT, synth.K_TEST
    # this rank's shard: contiguous hypotheses [first, first + P) of the seeded stream (prd.shard_bounds == pr_shard_range of the C ABI)
    if args.scaling == "strong":
        global_poses = args.global_poses or 4096
    else:
        global_poses = args.global_poses or args.poses * world
    first, P = prd.shard_bounds(global_poses, rank, world)
    P_max = prd.shard_bounds(global_poses, 0, world)[1]
    if P == 0:
        raise SystemExit(f"rank {rank}: empty shard ({global_poses} hypotheses over {world} ranks)")
    model = api.Model(os.path.join(ROOT, "tests", "golden", "obj_06.ply"))
    proj = api.compute_proj(K, W, H)
    scene_depth = api.render_host(model, synth.scene_pose()[None], W, H, proj)[0]
    scene = (api.Scene_projective().init_Scene_projective_cuda(scene_depth, K) if args.scene == "proj"
             else api.Scene_nn().init_Scene_nn_cuda(scene_depth, K))
    poses = synth.hypotheses(P, seed=6, first=first)             # this rank's shard of the global batch
    crit = api.ICPConvergenceCriteria(0.0, 0.0, args.iters)

    # P x RegistrationResult (72 B) on the device, double-buffered.  Step k is SUBMITTED on slot k&1 (everything enqueued, no
    # host round trip) and only then is step k-1 waited for and its results handed to the gather -- the GPU always has the
    # next batch queued, and the gather of step k-1 (RCCL's stream) overlaps step k.
    results = [torch.zeros(P * 18, dtype=torch.float32, device="cuda") for _ in range(2)]
    gathered = [torch.zeros(global_poses * 18, dtype=torch.float32, device="cuda") for _ in range(2)] if (multi and rank == 0 and gather_mode == "cabi") else [None, None]
    pending = [None, None]                                       # gather handle per buffer
    inflight = [False, False]                                    # submitted, not yet waited for
    step_no = [0]
    last_sizes = [None]

    def retire(b):
        if not inflight[b]:
            return
        _, sizes = api.refine_wait(b)
        inflight[b] = False
        last_sizes[0] = sizes
        if multi and gather_mode == "cabi":                 # the single RCCL exchange of the job: P x 72 B per rank to rank 0,
            api.gather_results(results[b].data_ptr(), P, global_poses, 0,   # enqueued on the library's stream behind this batch
                               gathered[b].data_ptr() if rank == 0 else None)
        elif multi:
            pending[b] = prd.gather_results(results[b].cpu() if share_device else results[b], world, rank, dst=0, max_count=P_max, async_op=True)

    def step():
        b = step_no[0] & 1
        step_no[0] += 1
        if pending[b] is not None:                              # buffer b was handed to a gather two steps ago
            pending[b].wait()
            torch.cuda.current_stream().synchronize()
            pending[b] = None
        api.refine_submit(b, model, poses, W, H, proj, K, scene, crit, results_dev=results[b].data_ptr())
        inflight[b] = True
        retire(1 - b)

    def fence():
        for b in (0, 1):
            retire(b)
        for b in (0, 1):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None
        if multi and gather_mode == "cabi":
            api.sync()                                           # the library stream carries the gathers
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # Device power state (experiments only, --burn-in N; default none): the first tens of milliseconds of load after process start run at lower
    # clocks on part of the pool's boxes (a 20-step run after 5 warm-up steps: 236-248 k poses/s, after 20 or more: 252-255 k, same box,
    # DESIGN.md section 5).  Such steps would come before the caller's --warmup steps, are never timed and are reported in the line.
    for _ in range(max(0, args.burn_in)):
        step()
    for w in range(args.warmup):
        # the first warm-up step is a timed one: the slot creates its HIP events (46 of them, several microseconds each) on first use, and
        # that belongs to the warm-up like every other first use, not into the timed region's sampled step
        if w == 0 and not args.sequential:
            api.set_option("profile", int(os.environ.get("PR_BENCH_SAMPLE_PROFILE", "3")))
        step()
        if w == 0 and not args.sequential:
            api.set_option("profile", 0)
    cpu0 = time.process_time()
    # Roofline samples: the LAST n_samples steps of the timed region run with profile 3 -- as one pose group, their loop starting when the
    # other slot's batch is complete, HIP events around every correspondence launch -- so that the timed launches have the chip to
    # themselves.  Such a step costs more than a pipelined one (its loop overlaps with nothing) and counts against `value`; at the end of the
    # region the drain it needs is the drain the closing fence needs anyway (sampled in the middle, every sample also cost an empty
    # pipeline afterwards: -3 % at 100 steps, -6 % at the driver's 20).  The batch itself stays asynchronous (its render runs under the previous
    # step's loop, no host round trip): against the synchronous timed call of profile 1 that is +1-5 % at 20 steps, and it keeps the
    # synchronous path's first-use allocations (hundreds of MB, 1-8 ms depending on the box) out of the timed region.  One sampled step
    # (21 launches): the very last one, whose tail has nothing left to overlap with anyway.
    # (one step: a timed loop that has another submitted batch waiting behind it on the other slot's queue runs its launches 2-3 us slower
    # -- 41.1 / 41.0 / 42.0 / 38.7 us over four consecutive timed steps, only the last of which has nothing queued behind it)
    n_samples = min(args.steps, 1)
    api.set_option("profile", 1 if args.sequential else 0)
    api.profile_reset()
    fence()
    t0 = time.perf_counter()
    marks = []
    for i in range(args.steps):
        if i == args.steps - n_samples and not args.sequential:
            api.set_option("profile", int(os.environ.get("PR_BENCH_SAMPLE_PROFILE", "3")))
        step()
        marks.append(time.perf_counter() - t0)
    fence()
    elapsed = time.perf_counter() - t0
    if os.environ.get("PR_BENCH_MARKS") and rank == 0:             # where a run's time went: cumulative ms after every step's submit + previous wait, and the closing fence
        print("[bench] marks ms:", " ".join(f"{1e3 * m:.2f}" for m in marks), "| fence", f"{1e3 * (elapsed - marks[-1]):.2f}", file=sys.stderr, flush=True)
    host_cpu_s = time.process_time() - cpu0                          # CPU time of ALL threads of this process (timed region + the fence before it)
    sizes = last_sizes[0]
    api.set_option("profile", 0)
    prof = api.profile_read()

    gather_ms, gather_n = (api.gather_profile() if (multi and gather_mode == "cabi") else (0.0, 0))
    per_rank_ms = [1e3 * elapsed / args.steps]
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_device else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        mine = 1e3 * elapsed / args.steps
        per_rank_ms = [None] * world
        dist.all_gather_object(per_rank_ms, mine)
        elapsed = float(t.item())

    flush_c_stdio()
    if multi:
        dist.barrier()                                              # nothing of any rank is left to be written before rank 0's line
    if rank == 0:
        total_poses = global_poses * args.steps
        launches = max(1, prof["icp_launches"])
        pts_per_launch = prof["icp_points"] / launches
        bytes_per_launch = prof["icp_bytes"] / launches          # 36 B/point on pass 0, 48 B/point afterwards
        avg_launch_s = prof["icp_kernel_ms"] * 1e-3 / launches
        achieved = bytes_per_launch / avg_launch_s if avg_launch_s > 0 else 0.0
        traffic = PMC_TRAFFIC_BYTES_PER_POINT[args.scene] * pts_per_launch if PMC_TRAFFIC_BYTES_PER_POINT[args.scene] else None
        traffic_source = ("committed rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE passes, "
                          + PMC_TRAFFIC_SOURCE[args.scene] + f": {PMC_TRAFFIC_BYTES_PER_POINT[args.scene]} B/point x points of a launch")
        live = None
        if world == 1 and not args.no_live_pmc and not args.sequential:
            live = live_pmc_traffic(args.scene)                     # two counter passes of the same kernels, run now
        if live:
            traffic = live["bytes_per_point"] * pts_per_launch
            traffic_source = live["source"]
        out = {
            "metric": "refined poses/sec (640x480, 20 ICP iters)",
            "value": total_poses / elapsed,
            "unit": "poses/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "burn_in_steps": max(0, args.burn_in),                   # untimed, before the warm-up steps: device clocks up from idle
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"obj_06.ply, {P_max}-pose batch per GPU, 640x480 synthetic depth, "
                                   f"{'projective' if args.scene == 'proj' else 'kd-tree NN (exact search with the reference tie-breaks: keep-the-winner test and pixel-window scan where the bound allows, order-free task walk over 128-byte wide nodes otherwise, ties repeated by the ordered walk)'} association, "
                                   f"{args.iters} ICP iterations (21 passes), solve on {args.solve}"
                                   + (f", {args.pose_groups or (2 if args.scene == 'proj' else 3)} pose groups" if args.solve == "device" else ""),
                       "poses_per_gpu": P_max, "global_batch": global_poses, "points_per_pose_mean": float(np.mean(sizes)),
                       "parallelism": (f"pose-shard x{world}, no data-path collective, "
                                       + ("no gather (1 rank)" if not multi else
                                          ("1 RCCL gather per step (pr_gather_results)" if gather_mode == "cabi" else
                                           ("1 gloo gather per step on host copies (ranks share one device: test mode)" if share_device else "1 torch.distributed gather per step (RCCL backend)"))))},
            # `bound`: what the counters say limits this kernel.  `frac` is the contract's figure -- SURVEY 8d's ALGORITHMIC bytes over the launch
            # time, against the HBM peak; it exceeds what HBM really carries (the packed 16-byte scene record, Infinity-Cache-resident clouds), so
            # it is a throughput score, not a statement that the kernel sits on the HBM roof: the projective pass is VALU-bound (`valu_frac`).
            "roofline": {"bound": ("valu" if args.scene == "proj" else "l1/lds + valu (cache-resident search: HBM carries only clouds and winners)"),
                         "bound_contract_enum": "hbm",
                         "kernel": ("icp_pass_kernel (correspondence + 29-term reduce" if args.scene == "proj" else
                                                     "one correspondence pass = nn_search_kernel + nn_bound_kernel + nn_tree_wide_kernel + icp_pass_kernel<SceneNNWinners> (search, bound + window, task walk of the queued queries, 29-term reduce over the winners")
                                                    + (" + finalize/solve tail)" if args.fused_solve and args.solve == "device" else ")"),
                         "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK,
                         # (`frac` can exceed 1: SURVEY 8d's algorithmic bytes charge the reference's 24-byte scene gather and a cloud
                         # that streams from HBM; the packed 16-byte scene record and the cache-resident clouds move fewer real bytes)
                         # two readings of the same launches: SURVEY 8d's ALGORITHMIC bytes (what `frac` is), and the HBM bytes the
                         # PMC counters saw for this kernel.  At <= 512 hypotheses per sub-batch the clouds are Infinity-Cache
                         # resident by design and the 16-byte scene records hit L2, so the counter figure is the lower one.
                         "frac_algorithmic": achieved / HBM_PEAK,
                         # committed counter figures x this run's points and launch time: fabric bytes (FETCH_SIZE counts Infinity-Cache hits
                         # too, so this is an UPPER bound of DRAM traffic) and VALU issue slots (SQ_INSTS_VALU wave-instructions / 6.1e11 per s)
                         "frac_fabric_counter": (traffic / avg_launch_s / HBM_PEAK) if (traffic and avg_launch_s > 0) else None,
                         "valu_frac": (PMC_VALU_WAVE_INSTR_PER_POINT[args.scene] * pts_per_launch / avg_launch_s / VALU_PEAK)
                                      if (PMC_VALU_WAVE_INSTR_PER_POINT[args.scene] and avg_launch_s > 0) else None,
                         "valu_peak_wave_instr_per_s": VALU_PEAK,
                         "traffic": traffic,
                         "traffic_source": traffic_source,
                         "traffic_live": live,
                         "residency": "clouds of a sub-batch (<= 512 hypotheses) stay in the 256 MiB Infinity Cache over the 21 passes; scene records are L2-resident",
                         "avg_launch_us": avg_launch_s * 1e6, "launches": int(launches),
                         "algorithmic_bytes_per_launch": bytes_per_launch, "points_per_launch": pts_per_launch,
                         "timing": ("HIP events on the library stream around every launch (--sequential: synchronous single-group steps)" if args.sequential else
                                    f"HIP events on the library stream around every launch of the last {n_samples} steps of the timed region; "
                                    "a timed step stays an asynchronous batch but its loop runs as one pose group and only once the other slot's batch is complete (option profile = 3), so the launch has the chip to itself")},
            "gather": ("none (1 rank)" if not multi else ("pr_gather_results: grouped ncclSend/ncclRecv on the library stream (C ABI over RCCL)" if gather_mode == "cabi" else
                                                            ("torch.distributed.gather (gloo, host copies: ranks share one device)" if share_device else "torch.distributed.gather (RCCL)"))),
            "gather_note": gather_note,
            "gather_event_us": (1e3 * gather_ms / gather_n) if gather_n else None,      # HIP events around the exchange on the library stream, sampled steps (rank 0)
            "gather_events": int(gather_n),
            "per_rank_ms_per_step": per_rank_ms,
            "host_cpu_per_wall": host_cpu_s / elapsed if elapsed > 0 else None,   # rank 0: CPUs kept busy by this process during the timed region (spinning waits count)
            "blocking_wait": bool(api.get_option("blocking_wait")),
            "phase_ms_per_timed_step": {"render": prof["render_ms"] / max(1, launches // (args.iters + 1)),
                                        "cloud": prof["cloud_ms"] / max(1, launches // (args.iters + 1))},
        }
        if world == 1 and args.scene == "proj" and args.solve == "device" and not args.sequential and not args.no_kdtree_extra:
            out["solve_on_host"] = host_solve_extra(args, api, model, poses, W, H, proj, K, scene)
        if world == 1 and args.scene == "proj" and not args.no_kdtree_extra and not args.sequential:
            out["config2_kdtree"] = kdtree_extra(args, api, model, poses, scene_depth, W, H, proj, K)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, args.scene, model.tris, scene_depth, K, W, H)
            if "config2_kdtree" in out:
                out["config2_kdtree"]["cpu_baseline"] = cpu_baseline(args, "nn", model.tris, scene_depth, K, W, H)
        flush_c_stdio()
        print(json.dumps(out), flush=True)

    if multi:
        dist.barrier()
        dist.destroy_process_group()


def live_pmc_traffic(scene_kind):
    """roofline.traffic measured by this run: two `rocprofv3 --kernel-trace --pmc <counter>` passes (FETCH_SIZE, WRITE_SIZE: one counter per
    pass, as MI355X_MICROARCH.md prescribes) of tools/pmc_workload.py -- three 256-hypothesis batches of the same workload as ONE pose group, so
    that a dispatch covers every cloud of the batch -- in child processes, read back from the rocpd databases.  FETCH_SIZE is corrected by
    the factor this very pass shows on max2zero_kernel, which reads and writes a known number of bytes (2.0 on gfx950; outside 1.8-2.2 the
    guide's 2 is used; WRITE_SIZE likewise: 1.0).  Returns None when anything is missing or fails: the line then carries the committed figures."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    kernels = (["icp_pass_kernel<prk::SceneProjPacked"] if scene_kind == "proj" else
               ["nn_search_kernel", "nn_bound_kernel", "nn_tree_wide_kernel", "icp_pass_kernel<prk::SceneNNWinners"])
    try:
        per_dispatch = {}                                          # counter -> KB per dispatch, summed over the kernels of a pass
        calib = {}
        points = None
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            env = dict(os.environ, TMPDIR="/tmp", PR_OPTS="pose_groups=1", PR_RASTER_MODE="0")
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", d, "-o", counter, "--", sys.executable, os.path.join(ROOT, "tools", "pmc_workload.py"), "256"]
                if scene_kind != "proj":
                    cmd.append("nn")
                r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=90)
                for line in r.stdout.splitlines():
                    if line.startswith("points per batch:"):
                        points = int(line.split(":")[1])
                dbs = glob.glob(os.path.join(d, "**", counter + "_results.db"), recursive=True)
                if r.returncode != 0 or not dbs or not points:
                    return None
                con = sqlite3.connect(dbs[0])
                rows = list(con.execute("select name, count(*), sum(counter_value) from pmc_events where counter_name = ? group by name", (counter,)))
                con.close()
                total = 0.0
                for k in kernels:
                    hit = [(n, c, v) for n, c, v in rows if k in n]
                    if not hit:
                        return None
                    total += sum(v for _, _, v in hit) / sum(c for _, c, _ in hit)
                per_dispatch[counter] = total
                # calibration: the workload's two max2zero launches (one scene image, then the 256 images of a public render call) read
                # and write every pixel of 257 images of 640 x 480 int32 -- known bytes against the counter's sum over both
                cal = [v for n, c, v in rows if "max2zero_kernel" in n]
                if cal and cal[0] > 0:
                    calib[counter] = (257 * 640 * 480 * 4 / 1024.0) / cal[0]
        f_fetch = calib.get("FETCH_SIZE", 2.0)
        if not (1.8 <= f_fetch <= 2.2):
            f_fetch = 2.0
        f_write = calib.get("WRITE_SIZE", 1.0)
        if not (0.9 <= f_write <= 1.1):
            f_write = 1.0
        kb = f_fetch * per_dispatch["FETCH_SIZE"] + f_write * per_dispatch["WRITE_SIZE"]
        bpp = kb * 1024.0 / points
        return {"bytes_per_point": bpp, "fetch_kb_per_dispatch": per_dispatch["FETCH_SIZE"], "write_kb_per_dispatch": per_dispatch["WRITE_SIZE"],
                "fetch_correction": f_fetch, "write_correction": f_write, "points_per_dispatch": points,
                "source": (f"this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (one pass each) of tools/pmc_workload.py 256 as one pose group -- "
                           f"({f_fetch:.3f} x {per_dispatch['FETCH_SIZE']:.0f} + {f_write:.3f} x {per_dispatch['WRITE_SIZE']:.0f}) KB per dispatch of {points} points = {bpp:.1f} B/point "
                           f"(fabric side: Infinity-Cache hits included), x points of a launch; both counters corrected by what max2zero_kernel (known bytes) shows in the same pass")}
    except Exception as e:                                         # noqa: BLE001 -- an extra: never at the cost of the line
        print(f"[bench] live PMC pass unavailable ({e}); the line carries the committed counter figures", file=sys.stderr, flush=True)
        return None


def kdtree_extra(args, api, model, poses, scene_depth, W, H, proj, K, steps=20):
    """BASELINE.json configs[2] next to the headline: the same 256-hypothesis batch against the kd-tree scene (Scene_nn), a
    few steps through the two asynchronous slots, plus one instrumented (synchronous) batch whose work counters give the LOGICAL bytes of the
    search (wide nodes x 128 B + leaf points x 16 B + window cells x 16 B + 28 B per query for cloud and winner)."""
    import numpy as np
    crit = api.ICPConvergenceCriteria(0.0, 0.0, args.iters)
    scene = api.Scene_nn().init_Scene_nn_cuda(scene_depth, K)
    for _ in range(2):
        api.refine_batch(model, poses, W, H, proj, K, scene, crit)
    t0 = time.perf_counter()
    for k in range(steps):                                       # the two slots, like the headline loop: step k is enqueued, then step k-1 collected
        api.refine_submit(k & 1, model, poses, W, H, proj, K, scene, crit)
        if k:
            api.refine_wait((k - 1) & 1)
    api.refine_wait((steps - 1) & 1)
    dt = (time.perf_counter() - t0) / steps
    api.set_option("nn_count", 1)
    api.nn_counters(args.iters + 1)
    api.refine_batch(model, poses, W, H, proj, K, scene, crit)
    c = api.nn_counters(args.iters + 1).astype(np.float64)
    api.set_option("nn_count", 0)
    tot = c.sum(0)
    logical = tot[4] * 128 + tot[6] * 16 + tot[7] * 16 + tot[0] * 28          # wide nodes are 128-byte lines, leaf points 16-byte records
    hbm = PMC_TRAFFIC_BYTES_PER_POINT["nn"] * tot[0] if PMC_TRAFFIC_BYTES_PER_POINT["nn"] else None
    return {"workload": f"obj_06.ply, {len(poses)}-pose batch, 640x480, kd-tree nearest-neighbour association (Scene_nn: exact search, "
                        "reference tie-breaks; search kernel = keep-the-winner test and pixel-window scan where the bound allows it, bound kernel = "
                        "descent through the representative points + window, task walk over 128-byte wide nodes for the rest, ties repeated by the ordered walk), "
                        f"{args.iters} ICP iterations, two asynchronous slots",
            "value": len(poses) / dt, "unit": "poses/s", "ms_per_step": dt * 1e3, "steps": steps,
            "queries_per_step": tot[0], "settled_by_pixel_window_frac": tot[1] / max(tot[0], 1.0), "tree_searches_frac": tot[2] / max(tot[0], 1.0),
            "tree_nodes_per_tree_search": tot[4] / max(tot[2], 1.0), "leaf_points_per_tree_search": tot[6] / max(tot[2], 1.0),
            "logical_bytes_per_step": logical, "logical_GBps_over_step": logical / dt / 1e9,
            "hbm_bytes_per_step_counter": hbm, "hbm_GBps_counter_over_step": (hbm / dt / 1e9) if hbm else None,
            "hbm_frac_of_peak": (hbm / dt / HBM_PEAK) if hbm else None,
            "note": "the kd-tree kernels are cache- and latency-bound by design (SURVEY 8d): their working set (0.85 MB tree + 4.9 MB grid) "
                    "lives in L2, so HBM carries only the clouds and winners; `logical` counts every byte the search asks the caches for"}


def host_solve_extra(args, api, model, poses, W, H, proj, K, scene, steps=30):
    """The same batch with the 6x6 solve ON THE HOST, as `north_star` words it ("SVD solve on host"): per iteration and pose group one launch
    whose last workgroup per hypothesis leaves the 29 sums in pinned host memory, the host solve, and the update read back from pinned
    memory by the next launch (PR_SOLVE_HOST).  The headline keeps the iterations on the device.  Two figures: one host thread issuing
    synchronous calls, and two host threads with private contexts (pr_thread_context) taking the batches in turn -- the reference's own
    suggestion for feeding the GPU ("many host threads, each driving its own pose", README.md:15): one thread's render and host work run
    under the other thread's passes."""
    import threading
    crit = api.ICPConvergenceCriteria(0.0, 0.0, args.iters)
    api.set_option("solve", api.SOLVE_HOST)
    try:
        for _ in range(2):
            api.refine_batch(model, poses, W, H, proj, K, scene, crit)
        t0 = time.perf_counter()
        for _ in range(steps):
            api.refine_batch(model, poses, W, H, proj, K, scene, crit)
        dt = (time.perf_counter() - t0) / steps
        n_threads = 2
        barrier = threading.Barrier(n_threads + 1)
        errors = []

        def work():
            try:
                api.thread_context(True)
                for _ in range(3):
                    api.refine_batch(model, poses, W, H, proj, K, scene, crit)
                barrier.wait()
                for _ in range(steps):
                    api.refine_batch(model, poses, W, H, proj, K, scene, crit)
                barrier.wait()
                api.thread_context(False)
            except Exception as e:                                # noqa: BLE001 -- the figure is an extra: a failure must not cost the line
                errors.append(repr(e))
                barrier.abort()
        ts = [threading.Thread(target=work) for _ in range(n_threads)]
        [t.start() for t in ts]
        try:
            barrier.wait(); t1 = time.perf_counter(); barrier.wait(); dt2 = (time.perf_counter() - t1) / (steps * n_threads)
        except threading.BrokenBarrierError:
            dt2 = None
        [t.join() for t in ts]
    finally:
        api.set_option("solve", api.SOLVE_DEVICE)
    out = {"value": len(poses) / dt, "unit": "poses/s", "ms_per_step": dt * 1e3, "steps": steps, "host_threads": 1,
           "note": "PR_SOLVE_HOST, one synchronous call per step: 21 launches per pose group (two groups on two streams, software-pipelined: the host solves one group while the other group's pass runs); the workgroup that delivers a hypothesis' last partial sum stores its 29 totals straight into pinned host memory, the host solves (pivoted LDLT in double, as Eigen) and the next pass reads the update from the pinned array"}
    if dt2:
        out["two_host_threads"] = {"value": len(poses) / dt2, "unit": "poses/s", "ms_per_step": dt2 * 1e3, "steps": steps * n_threads,
                                   "note": "the same synchronous calls from two host threads with private contexts (pr_thread_context), batches taken in turn"}
    elif errors:
        out["two_host_threads"] = {"error": errors[0]}
    return out


def cpu_baseline(args, scene_kind, tris, scene_depth, K, W, H):
    """The CPU oracle (a port of the reference CPU path, oracle/pose_oracle.c at the reference's flags) on a bounded sample of the same
    workload: per-pose render_cpu -> depth2cloud_cpu -> ICP_Point2Plane_cpu with the same fixed 20 iterations, OpenMP across hypotheses
    on the CPUs this process really has (affinity mask capped by the cgroup quota), plus the single-thread rate."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["PR_ORACLE_BUILD"] = "o3"                          # the restatement with the reference's flags (-O3 -fopenmp)
    import oracle_lib as O
    from pose_refine_amd import synth
    usable, facts = effective_cpus()
    threads = int(os.environ.get("PR_BENCH_CPU_THREADS", usable))
    per_pose_s = 0.012 if scene_kind == "proj" else 0.42           # single-thread seconds per hypothesis on the GPU box's host (profiles/r03/cpu_sweep_*.md)
    budget_s = 10.0 if scene_kind == "proj" else 8.0
    n = args.cpu_poses or int(max(2 * threads, min(20000, round(budget_s * threads / per_pose_s))))
    n1 = max(2, int(round((1.5 if scene_kind == "proj" else 1.3) / per_pose_s)))
    oscene = O.ProjScene(scene_depth, K) if scene_kind == "proj" else O.NNScene(scene_depth, K)
    proj = O.compute_proj(K, W, H)
    crit = (0.0, 0.0, args.iters)
    O.set_threads(1)
    t0 = time.perf_counter()
    O.refine_batch(tris, synth.hypotheses(n1), W, H, proj, K, oscene, crit, O.SUM_SEQUENTIAL)
    dt1 = time.perf_counter() - t0
    O.set_threads(threads)
    poses = synth.hypotheses(n)                                     # same seeded stream, longer prefix
    t0 = time.perf_counter()
    _, _, used = O.refine_batch(tris, poses, W, H, proj, K, oscene, crit, O.SUM_SEQUENTIAL)
    dt = time.perf_counter() - t0
    ms_thread, ms_thread1 = 1e3 * dt * used / n, 1e3 * dt1 / n1
    return {"value": n / dt, "unit": "poses/s", "cores": int(used), "kind": "port",
            "sample": f"first {n} hypotheses of the same seeded stream, {dt:.1f} s wall, OpenMP over hypotheses on {used} threads",
            "single_thread": {"value": n1 / dt1, "unit": "poses/s", "sample": f"{n1} hypotheses, {dt1:.1f} s"},
            "ms_per_pose_per_thread": ms_thread, "ms_per_pose_single_thread": ms_thread1, "thread_efficiency": ms_thread1 / ms_thread if ms_thread > 0 else None,
            "host": facts, "threads_rule": "min(affinity mask, cgroup cpu.max quota); PR_BENCH_CPU_THREADS overrides; sweep: profiles/r03/cpu_sweep_*.md",
            "association": "projective" if scene_kind == "proj" else "kd-tree (Scene_nn)",
            "build": "oracle/pose_oracle.c at the reference's flags (-O3 -fopenmp), every thread reusing one depth image and one cloud; per-row check "
                     "against the verbatim reference timings: profiles/r02/cpu_fairness.md"}


if __name__ == "__main__":
    main()
