#!/usr/bin/env python
"""Host-core sweep of bench.py's `cpu_baseline`: what the GPU box's host really offers, and the
thread count at which the CPU restatement of the reference path (oracle/pose_oracle.c, reference flags) is fastest.

    python tools/cpu_sweep.py [proj|nn] > profiles/r03/cpu_sweep_<scene>.md

Prints the scheduler's view of the host (affinity mask, cgroup cpu.max, lscpu) and a table: threads, OMP binding,
poses/s, ms per pose per thread, ratio to the single-thread figure.  Every row runs in a process of its own (the
OpenMP environment is read when libgomp loads).  CPU only -- no GPU call anywhere.
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def host_facts():
    f = {"os.cpu_count": os.cpu_count(), "sched_getaffinity": len(os.sched_getaffinity(0))}
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us",
              "/sys/fs/cgroup/cpuset.cpus.effective", "/sys/fs/cgroup/cpuset/cpuset.cpus"):
        try:
            f[p] = open(p).read().strip()
        except OSError:
            pass
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True).stdout
        for line in out.splitlines():
            k = line.split(":")[0].strip()
            if k in ("Model name", "CPU(s)", "Socket(s)", "Core(s) per socket", "Thread(s) per core", "NUMA node(s)", "CPU max MHz"):
                f["lscpu " + k] = line.split(":", 1)[1].strip()
    except OSError:
        pass
    f["loadavg"] = open("/proc/loadavg").read().strip()
    f["OMP_NUM_THREADS (inherited)"] = os.environ.get("OMP_NUM_THREADS", "(unset)")
    return f


def child(scene, n):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["PR_ORACLE_BUILD"] = "o3"
    import oracle_lib as O
    from pose_refine_amd import synth
    tris = O.ply_load(os.path.join(ROOT, "tests", "golden", "obj_06.ply"))
    K, W, H = synth.K_TEST, synth.WIDTH, synth.HEIGHT
    proj = O.compute_proj(K, W, H)
    scene_depth = O.render(tris, synth.scene_pose()[None], W, H, proj)[0]
    oscene = O.ProjScene(scene_depth, K) if scene == "proj" else O.NNScene(scene_depth, K)
    poses = synth.hypotheses(n)
    t0 = time.perf_counter()
    _, _, threads = O.refine_batch(tris, poses, W, H, proj, K, oscene, (0.0, 0.0, 20), O.SUM_SEQUENTIAL)
    dt = time.perf_counter() - t0
    print(json.dumps({"threads": int(threads), "n": n, "s": dt}))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(sys.argv[2], int(sys.argv[3]))
    scene = sys.argv[1] if len(sys.argv) > 1 else "proj"
    budget_s = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0        # CPU seconds of wall per row, roughly
    facts = host_facts()
    print(f"# cpu_baseline thread sweep ({scene}) -- host of the GPU box\n")
    for k, v in facts.items():
        print(f"* {k}: {v}")
    avail = facts["sched_getaffinity"]
    per_pose = 0.03 if scene == "proj" else 1.3
    counts = sorted({c for c in (1, 2, 4, 8, 16, 32, 64, 96, 128, 192, 256, avail) if c <= avail})
    print("\n| threads | binding | hypotheses | wall s | poses/s | ms per pose per thread | vs 1 thread |")
    print("|---:|---|---:|---:|---:|---:|---:|")
    base = None
    best = None
    for c in counts:
        for bind in ("none", "spread/cores"):
            if c == 1 and bind != "none":
                continue
            env = dict(os.environ, OMP_NUM_THREADS=str(c))
            env.pop("OMP_PROC_BIND", None); env.pop("OMP_PLACES", None)
            if bind != "none":
                env["OMP_PROC_BIND"] = "spread"; env["OMP_PLACES"] = "cores"
            n = max(2 * c, int(round(budget_s * c / per_pose)))
            n = min(n, 40000)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", scene, str(n)], env=env, capture_output=True, text=True)
            try:
                d = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception:                                           # noqa: BLE001
                print(f"| {c} | {bind} | {n} | failed: {r.stderr.strip()[-120:]} |")
                continue
            rate = d["n"] / d["s"]
            ms_thread = 1e3 * d["s"] * d["threads"] / d["n"]
            if base is None:
                base = ms_thread
            if best is None or rate > best[0]:
                best = (rate, d["threads"], bind)
            print(f"| {d['threads']} | {bind} | {d['n']} | {d['s']:.2f} | {rate:.1f} | {ms_thread:.2f} | {ms_thread / base:.2f}x |", flush=True)
    if best:
        print(f"\nbest: {best[0]:.1f} poses/s at {best[1]} threads (binding {best[2]}); single thread {1e3 / base:.1f} poses/s")


if __name__ == "__main__":
    main()
