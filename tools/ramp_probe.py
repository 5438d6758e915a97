#!/usr/bin/env python
"""Diagnostic (round 6): is the slow start of the pipelined loop (247 -> 269 -> 280 k poses/s over its first thirty steps) a one-time software warm-up or the
device coming up from idle?  The loop runs 60 steps, the process sleeps, the loop runs again.   tools/ramp_probe.py [idle seconds ...]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pose_refine_amd import api, synth
W, H, K = synth.WIDTH, synth.HEIGHT, synth.K_TEST
api.init(0); api.set_option("solve", api.SOLVE_DEVICE)
model = api.Model(os.path.join(ROOT, "tests", "golden", "obj_06.ply"))
proj = api.compute_proj(K, W, H)
sd = api.render_host(model, synth.scene_pose()[None], W, H, proj)[0]
scene = api.Scene_projective().init_Scene_projective_cuda(sd, K)
crit = api.ICPConvergenceCriteria(0.0, 0.0, 20)
poses = synth.hypotheses(256)
def run(steps=60):
    marks = []; t0 = time.perf_counter()
    for k in range(steps):
        api.refine_submit(k & 1, model, poses, W, H, proj, K, scene, crit)
        if k: api.refine_wait((k - 1) & 1)
        marks.append(time.perf_counter() - t0)
    api.refine_wait((steps - 1) & 1)
    m = np.asarray(marks)
    return " ".join(f"{256 * 5 / (m[b + 5] - m[b]) / 1e3:.0f}" for b in range(0, steps - 5, 5))
print("from process start:      ", run())
for idle in [float(a) for a in sys.argv[1:]] or [0.0, 0.005, 0.05, 0.5]:
    time.sleep(idle)
    print(f"after {idle*1e3:6.1f} ms of idle:  ", run())
