#!/usr/bin/env python
"""A/B of one library option in the pipelined (two-slot) loop: tools/ab_option.py <option> <v1,v2,...> [poses]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import numpy as np
from pose_refine_amd import api, synth
opt, vals = sys.argv[1], [int(v) for v in sys.argv[2].split(",")]
P = int(sys.argv[3]) if len(sys.argv) > 3 else 256
api.init(0); api.set_option("solve", 1)
model = api.Model(os.path.join(ROOT, "tests/golden/obj_06.ply"))
K = synth.K_TEST; proj = api.compute_proj(K, 640, 480)
sd = api.render_host(model, synth.scene_pose()[None], 640, 480, proj)[0]
scene = api.Scene_projective().init_Scene_projective_cuda(sd, K)
poses = synth.hypotheses(P); crit = api.ICPConvergenceCriteria(0.0, 0.0, 20)
res = [torch.zeros(P * 18, dtype=torch.float32, device="cuda") for _ in range(2)]
infl = [False, False]
def run(N):
    for k in range(N):
        b = k & 1
        api.refine_submit(b, model, poses, 640, 480, proj, K, scene, crit, results_dev=res[b].data_ptr()); infl[b] = True
        if infl[1 - b]: api.refine_wait(1 - b); infl[1 - b] = False
    for b in (0, 1):
        if infl[b]: api.refine_wait(b); infl[b] = False
for rep in range(2):
    for v in vals:
        api.set_option(opt, v)
        run(10); t0 = time.perf_counter(); run(150); dt = (time.perf_counter() - t0) / 150
        print(f"{opt}={v}: {dt*1e3:.4f} ms/step  {P/dt:.0f} poses/s", flush=True)
