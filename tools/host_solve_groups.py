#!/usr/bin/env python
"""PR_SOLVE_HOST (north_star's "solve on host") as a function of the number of pose groups of its software pipeline:
tools/host_solve_groups.py [poses] [nn]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pose_refine_amd import api, synth
P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kind = sys.argv[2] if len(sys.argv) > 2 else "proj"
api.init(0)
model = api.Model(os.path.join(ROOT, "tests/golden/obj_06.ply"))
K = synth.K_TEST; proj = api.compute_proj(K, 640, 480)
sd = api.render_host(model, synth.scene_pose()[None], 640, 480, proj)[0]
scene = (api.Scene_projective().init_Scene_projective_cuda(sd, K) if kind == "proj" else api.Scene_nn().init_Scene_nn_cuda(sd, K))
poses = synth.hypotheses(P); crit = api.ICPConvergenceCriteria(0.0, 0.0, 20)
ref = None
for rep in range(2):
    for solve, groups in ((0, 1), (0, 2), (0, 3), (0, 4), (1, 2)):
        api.set_option("solve", solve); api.set_option("pose_groups", groups)
        N = 60 if kind == "proj" else 6
        for _ in range(3):
            out = api.refine_batch(model, poses, 640, 480, proj, K, scene, crit)
        t0 = time.perf_counter()
        for _ in range(N):
            out = api.refine_batch(model, poses, 640, 480, proj, K, scene, crit)
        dt = (time.perf_counter() - t0) / N
        blob = out[0].tobytes()
        if ref is None:
            ref = blob
        print(f"solve={'host' if solve == 0 else 'device'} groups={groups}: {dt*1e3:.3f} ms per synchronous batch  {P/dt:.0f} poses/s  identical={blob == ref}", flush=True)
