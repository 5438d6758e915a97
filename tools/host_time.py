#!/usr/bin/env python
"""Host-side cost of pr_refine_submit / pr_refine_wait in the pipelined loop bench.py runs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import numpy as np
from pose_refine_amd import api, synth
P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
api.init(0); api.set_option("solve", 1)
model = api.Model(os.path.join(ROOT, "tests/golden/obj_06.ply"))
K = synth.K_TEST; proj = api.compute_proj(K, 640, 480)
sd = api.render_host(model, synth.scene_pose()[None], 640, 480, proj)[0]
scene = api.Scene_projective().init_Scene_projective_cuda(sd, K)
poses = synth.hypotheses(P)
crit = api.ICPConvergenceCriteria(0.0, 0.0, 20)
res = [torch.zeros(P * 18, dtype=torch.float32, device="cuda") for _ in range(2)]
ts, tw = [], []
infl = [False, False]
N = 60
t00 = time.perf_counter()
for k in range(N):
    b = k & 1
    t0 = time.perf_counter()
    api.refine_submit(b, model, poses, 640, 480, proj, K, scene, crit, results_dev=res[b].data_ptr())
    t1 = time.perf_counter()
    infl[b] = True
    if infl[1 - b]:
        api.refine_wait(1 - b); infl[1 - b] = False
    t2 = time.perf_counter()
    ts.append(t1 - t0); tw.append(t2 - t1)
for b in (0, 1):
    if infl[b]: api.refine_wait(b)
tot = time.perf_counter() - t00
print(f"P={P}: step {tot/N*1e3:.3f} ms  submit {np.mean(ts[5:])*1e3:.3f} ms  wait {np.mean(tw[5:])*1e3:.3f} ms")
