#!/usr/bin/env python
"""PR_SOLVE_HOST driven the way the reference's README suggests ("many host threads, each driving its own pose", README.md:15): T host
threads with private contexts (pr_thread_context) take the 256-hypothesis batches in turn, every call synchronous -- one thread's render and
host work run under another thread's passes.   python tools/host_solve_threads.py [poses] [batches per thread]"""
import os, sys, time, threading
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pose_refine_amd import api, synth
P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
api.init(0); api.set_option("solve", api.SOLVE_HOST)
model = api.Model(os.path.join(ROOT, "tests/golden/obj_06.ply"))
K = synth.K_TEST; proj = api.compute_proj(K, 640, 480)
sd = api.render_host(model, synth.scene_pose()[None], 640, 480, proj)[0]
scene = api.Scene_projective().init_Scene_projective_cuda(sd, K)
poses = synth.hypotheses(P); crit = api.ICPConvergenceCriteria(0.0, 0.0, 20)
ref = api.refine_batch(model, poses, 640, 480, proj, K, scene, crit)[0].tobytes()
for T in (1, 2, 3, 4, 2):
    bad = []
    barrier = threading.Barrier(T + 1)
    def work():
        api.thread_context(True)
        for _ in range(3): api.refine_batch(model, poses, 640, 480, proj, K, scene, crit)
        barrier.wait()
        for _ in range(N):
            out = api.refine_batch(model, poses, 640, 480, proj, K, scene, crit)
            if out[0].tobytes() != ref: bad.append(1)
        barrier.wait()
        api.thread_context(False)
    ts = [threading.Thread(target=work) for _ in range(T)]
    [t.start() for t in ts]
    barrier.wait(); t0 = time.perf_counter(); barrier.wait(); dt = time.perf_counter() - t0
    [t.join() for t in ts]
    print(f"{T} host thread(s): {1e3 * dt / (N * T):.3f} ms per {P}-hypothesis batch = {P * N * T / dt:.0f} poses/s, mismatches {len(bad)}", flush=True)
