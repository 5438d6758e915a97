#!/usr/bin/env python
"""Diagnostic (round 6): rate of the host-solve pipeline over time from its first step on, per block of 10 steps.   tools/host_warm_probe.py [device|host]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pose_refine_amd import api, synth
mode = sys.argv[1] if len(sys.argv) > 1 else "host"
W, H, K = synth.WIDTH, synth.HEIGHT, synth.K_TEST
api.init(0)
api.set_option("solve", api.SOLVE_HOST if mode == "host" else api.SOLVE_DEVICE)
model = api.Model(os.path.join(ROOT, "tests", "golden", "obj_06.ply"))
proj = api.compute_proj(K, W, H)
sd = api.render_host(model, synth.scene_pose()[None], W, H, proj)[0]
scene = api.Scene_projective().init_Scene_projective_cuda(sd, K)
crit = api.ICPConvergenceCriteria(0.0, 0.0, 20)
poses = synth.hypotheses(256)
marks = []
t0 = time.perf_counter()
steps = 400
for k in range(steps):
    api.refine_submit(k & 1, model, poses, W, H, proj, K, scene, crit)
    if k: api.refine_wait((k - 1) & 1)
    marks.append(time.perf_counter() - t0)
api.refine_wait((steps - 1) & 1)
m = np.asarray(marks)
out = []
for b in range(0, steps - 10, 10):
    out.append(f"{256 * 10 / (m[b + 10] - m[b]) / 1e3:.0f}")
print(mode, "k poses/s per 10 steps:", " ".join(out))
