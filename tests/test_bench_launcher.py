"""`python bench.py --gpus N` starts its N ranks itself.  CPU half: the launcher logic -- the
command line it builds, the device-budget refusal, the fall-back from processes to threads, the thread control plane.
GPU half (-m gpu): both launchers end to end with two ranks sharing the box's one GPU (PR_BENCH_SHARE_DEVICE=1)."""
import json
import os
import subprocess
import sys
import threading
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launch_command_is_the_contracts_line():
    cmd = bench.launch_command(8, 29512, ["--gpus", "8", "--steps", "20", "--warmup", "5"])
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29512"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]


def test_more_ranks_than_devices_is_refused_with_a_clear_message():
    with pytest.raises(SystemExit) as e:
        bench.check_device_budget(8, 1, share_device=False)
    assert "only 1 GPU(s) visible" in str(e.value) and "PR_BENCH_SHARE_DEVICE=1" in str(e.value)
    bench.check_device_budget(8, 1, share_device=True)          # test mode: allowed
    bench.check_device_budget(8, 8, share_device=False)


def test_gpus_n_without_world_size_goes_to_the_launcher(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("PR_BENCH_SHARE_DEVICE", raising=False)
    calls = []
    monkeypatch.setattr(bench, "visible_devices", lambda: 8)
    monkeypatch.setattr(bench, "spawn_processes", lambda args, argv: calls.append(("processes", args.gpus, list(argv))) or 0)
    monkeypatch.setattr(bench, "run_threads", lambda args, note=None: calls.append(("threads", args.gpus, note)) or 0)
    assert bench.main(["--gpus", "8", "--steps", "3"]) == 0
    assert bench.main(["--gpus", "4", "--launcher", "threads"]) == 0
    assert calls == [("processes", 8, ["--gpus", "8", "--steps", "3"]), ("threads", 4, None)]
    monkeypatch.setattr(bench, "visible_devices", lambda: 1)
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2"])
    assert "only 1 GPU(s) visible" in str(e.value)


def test_failed_process_launch_falls_back_to_threads(monkeypatch, capsys):
    notes = []
    monkeypatch.setattr(bench, "run_threads", lambda args, note=None: notes.append(note) or 0)
    monkeypatch.setattr(subprocess, "run", lambda cmd, **kw: types.SimpleNamespace(returncode=1, stdout="rendezvous failed\n"))
    args = bench.parse_args(["--gpus", "2"])
    assert bench.spawn_processes(args, ["--gpus", "2"]) == 0
    assert notes and "fell back to --launcher threads" in notes[0]
    # a launch that printed its line is passed through untouched
    line = json.dumps({"metric": "refined poses/sec (640x480, 20 ICP iters)", "n_gpus": 2})
    monkeypatch.setattr(subprocess, "run", lambda cmd, **kw: types.SimpleNamespace(returncode=0, stdout="banner\n" + line + "\n"))
    notes.clear()
    assert bench.spawn_processes(args, ["--gpus", "2"]) == 0
    assert not notes and capsys.readouterr().out.strip().splitlines()[-1] == line


def test_thread_group_is_a_barrier_and_an_all_gather():
    world = 4
    shared = bench.ThreadGroup.Shared(world)
    got = [None] * world

    def work(r):
        g = bench.ThreadGroup(shared, r)
        a = g.all_gather(("first", r))
        g.barrier()
        b = g.all_gather(r * r)
        got[r] = (a, b)
    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for r in range(world):
        assert got[r] == ([("first", i) for i in range(world)], [i * i for i in range(world)])


def _run_bench(extra, env_extra, gpus=2, share=True):
    env = dict(os.environ, **env_extra)
    if share:
        env["PR_BENCH_SHARE_DEVICE"] = "1"
    else:
        env.pop("PR_BENCH_SHARE_DEVICE", None)
    env.pop("WORLD_SIZE", None)
    # the child is measured, not profiled: when this suite itself runs under rocprofv3 (tools/gpu_round.sh does, for the kernel coverage), the
    # tool's environment must not reach bench.py -- rocprofv3 7.2 aborts in stream_stack.cpp when HIP is driven from Python threads
    for k in list(env):
        if k.startswith(("ROCPROF", "ROCP_", "ROCTX", "HSA_TOOLS", "ROCPROFILER")) or (k == "LD_PRELOAD" and "rocprof" in env[k]):
            env.pop(k)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "6", "--warmup", "2", "--no-config3"] + extra,
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(line) == 1, r.stdout[-2000:]
    assert r.stdout.strip().splitlines()[-1] == line[0]         # the JSON line is the last thing printed
    return json.loads(line[0])


@pytest.mark.gpu
@pytest.mark.parametrize("launcher", ["processes", "threads"])
def test_bench_starts_two_ranks_by_itself(launcher):
    d = _run_bench(["--launcher", launcher], {})
    assert d["n_gpus"] == 2 and len(d["per_rank_ms_per_step"]) == 2
    assert d["launcher"]["kind"] == launcher and d["launcher"]["share_device_test_mode"] is True
    assert d["gather_when"] == "job" and d["gathers_in_timed_region"] == 1
    assert d["config"]["global_batch"] == 512 and d["value"] > 1000
    assert "cpu_baseline" not in d


@pytest.mark.gpu
def test_bench_uneven_shards_gather_per_step():
    d = _run_bench(["--launcher", "threads", "--global-poses", "301", "--gather", "job"], {})
    assert d["gather_when"] == "step" and d["gathers_in_timed_region"] == 6 and "do not divide" in d["gather_when_note"]
    assert d["config"]["poses_per_gpu"] == 151


@pytest.mark.gpu
@pytest.mark.parametrize("launcher", ["processes", "threads"])
def test_bench_starts_eight_ranks_by_itself(launcher):
    """the 8-rank job of the driver's SCALE run, on the one GPU of this box (test mode: every rank on device 0, the gather
    on host copies): eight shards of the seeded stream, eight per-rank clocks, and every rank's host CPU time over its timed region."""
    d = _run_bench(["--launcher", launcher, "--poses", "64"], {}, gpus=8)
    assert d["n_gpus"] == 8 and len(d["per_rank_ms_per_step"]) == 8
    assert d["launcher"]["kind"] == launcher and d["launcher"]["share_device_test_mode"] is True
    assert d["config"]["global_batch"] == 512 and d["config"]["poses_per_gpu"] == 64 and d["value"] > 1000
    assert d["blocking_wait"] is True                              # more than one rank on the host: pr_refine_wait sleeps
    assert len(d["per_rank_host_cpu_ms_per_step"]) == 8 and all(c > 0 for c in d["per_rank_host_cpu_ms_per_step"])
    assert d["gather_when"] == "job" and d["gathers_in_timed_region"] == 1


@pytest.mark.gpu
@pytest.mark.parametrize("extra, when, gathers", [([], "job", 1), (["--global-poses", "509"], "step", 6)])
def test_bench_eight_rank_threads_gather_through_the_c_abi_over_the_loopback_library(extra, when, gathers):
    """`bench.py --gpus 8 --launcher threads` with gather_mode = cabi on a one-GPU box -- the eight rank threads (private contexts
    on device 0) form a communicator of the loop-back stand-in (PR_RCCL_LIBRARY, tests/rccl_loopback) and the job's gather runs through
    pr_gather_results' N > 1 branch: even shards as the job's single exchange, uneven shards (509 over 8) as one exchange per step."""
    from test_gather_loopback_gpu import build_loopback
    d = _run_bench(["--launcher", "threads", "--poses", "64"] + extra, {"PR_RCCL_LIBRARY": build_loopback()}, gpus=8)
    assert d["n_gpus"] == 8 and d["launcher"]["kind"] == "threads"
    assert d["gather"].startswith("pr_gather_results") and "loop-back" in d["gather"], d["gather"]
    assert d["gather_note"] is None and d["gather_when"] == when and d["gathers_in_timed_region"] == gathers
    assert d["value"] > 1000


@pytest.mark.gpu
def test_bench_force_comm_runs_the_rccl_gather_with_a_world_of_one():
    """PR_BENCH_FORCE_COMM=1: the whole N > 1 machinery with ONE rank -- torch's process group on the RCCL backend and the library's own
    dlopened RCCL communicator (pr_comm_init_rank, pr_gather_results) in one process, on every round's GPU box."""
    d = _run_bench([], {"PR_BENCH_FORCE_COMM": "1"}, gpus=1, share=False)
    assert d["n_gpus"] == 1 and d["launcher"]["kind"] == "processes"
    assert d["gather"].startswith("pr_gather_results") and d["gather_when"] == "job" and d["gathers_in_timed_region"] == 1
    assert d["gather_bytes_per_rank"] == 72 * 256 * 6 and d["value"] > 1000
    for key in ("step_ms_spread", "host_cpu_ms_per_step", "roofline"):
        assert key in d
    assert d["roofline"]["frac"] <= 1.0 and d["roofline"]["frac_end_to_end"] > 0


def test_committed_bench_lines_keep_the_contract():
    """The round's committed bench lines (profiles/r06/, written by tools/gpu_round.sh on the GPU box): the contract's keys; `roofline` as the contract
    words it (bound "hbm", achieved = SURVEY 8d's algorithmic GB/s, peak 8000, frac = achieved / peak) with the counters' verdict in
    `roofline.binding` against the CALIBRATED VALU peak; `cpu_baseline` on the one-rank line; the workloads VERDICT r05 missed in the N = 1 line."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r06", "bench_*.json")))
    assert len(files) >= 10
    for f in files:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
            assert key in d, (f, key)
        assert d["unit"] == "poses/s" and d["dtype"] == "f32" and d["vs_baseline"] is None and "workload" in d["config"]
        r = d["roofline"]
        for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_algorithmic", "frac_end_to_end", "binding"):
            assert key in r, (f, key)
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] < 1.2, (f, r["frac"])      # (above 1 at 512+ hypotheses per batch: 8d charges more bytes than the packed scene moves)
        assert r["binding"]["valu_peak_wave_instr_per_s"] == 0.977e12 and "valu_peak.md" in r["binding"]["valu_peak_source"]
        assert abs(d["value"] * d["ms_per_step"] * 1e-3 - d["config"]["global_batch"]) < 1e-3 * d["config"]["global_batch"]      # value = hypotheses of a step / its time
    head = json.loads(open(os.path.join(ROOT, "profiles", "r06", "bench_p256_proj_steps20.json")).read())
    assert head["n_gpus"] == 1 and head["steps"] == 20 and head["warmup"] == 5
    assert head["roofline"]["binding"]["verdict"] == "latency" and 0.2 < head["roofline"]["binding"]["valu_issue_frac"] < 0.45
    assert head["roofline"]["binding"]["valu_constants_stale"] is False
    assert head["cpu_baseline"]["kind"] == "port" and head["cpu_baseline"]["cores"] >= 1 and "sample" in head["cpu_baseline"]
    assert {"projective", "kdtree"} <= set(head["default_criteria"]) and "north_star_solve_on_host" in head["config"]
    for key in ("config3_share_512", "config4_share_128", "config2_kdtree", "solve_on_host"):
        assert len(head[key]["segments_poses_per_s"]) == 3 and head[key]["value"] > 0, key
    assert head["config4_share_128"]["poses_per_gpu"] == 128 and head["config3_share_512"]["poses_per_gpu"] == 512
    loop = json.loads(open(os.path.join(ROOT, "profiles", "r06", "bench_8ranks_loopback_gather_threads.json")).read())
    assert loop["n_gpus"] == 8 and loop["gather"].startswith("pr_gather_results") and "loop-back" in loop["gather"]
    g = json.loads(open(os.path.join(ROOT, "profiles", "r06", "gather_loopback_8ranks.json")).read())
    assert g["world"] == 8 and all(c["ok"] for c in g["cases"]) and g["refine"]["bit_identical_to_unsharded"] and g["errors"] == []
