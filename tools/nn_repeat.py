#!/usr/bin/env python
"""The kd-tree path's results do not depend on timing: the same 256-hypothesis batch, repeated, must be bit-identical every time
(the task walk's LDS atomics, queue order and second rounds vary from run to run; winners may not)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pose_refine_amd import api, synth
api.init(0); api.set_option("solve", 1)
model = api.Model(os.path.join(ROOT, "tests/golden/obj_06.ply"))
K = synth.K_TEST; W, H = 640, 480; proj = api.compute_proj(K, W, H)
sd = api.render_host(model, synth.scene_pose()[None], W, H, proj)[0]
scene = api.Scene_nn().init_Scene_nn_cuda(sd, K)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
poses = synth.hypotheses(256)
crit = api.ICPConvergenceCriteria(0.0, 0.0, 20)
first = api.refine_batch(model, poses, W, H, proj, K, scene, crit)
bad = 0
for i in range(N):
    api.refine_submit(i & 1, model, poses, W, H, proj, K, scene, crit)
    if i:
        r = api.refine_wait((i - 1) & 1)
        bad += r[0].tobytes() != first[0].tobytes()
r = api.refine_wait((N - 1) & 1); bad += r[0].tobytes() != first[0].tobytes()
print(f"{N} repeats, {bad} differing")
sys.exit(1 if bad else 0)
