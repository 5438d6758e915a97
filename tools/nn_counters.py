#!/usr/bin/env python
"""Work counters of the kd-tree search kernel per ICP pass (option nn_count) for one 256-pose batch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pose_refine_amd import api, synth
P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
api.init(0); api.set_option("solve", 1); api.set_option("nn_count", 1)
for kv in filter(None, os.environ.get("PR_OPTS", "").split(",")):
    k, v = kv.split("="); api.set_option(k, int(v))
model = api.Model(os.path.join(ROOT, "tests/golden/obj_06.ply"))
K = synth.K_TEST; proj = api.compute_proj(K, 640, 480)
sd = api.render_host(model, synth.scene_pose()[None], 640, 480, proj)[0]
scene = api.Scene_nn().init_Scene_nn_cuda(sd, K)
poses = synth.hypotheses(P)
api.refine_batch(model, poses, 640, 480, proj, K, scene, api.ICPConvergenceCriteria(0.0, 0.0, 20))
c = api.nn_counters(21).astype(np.float64)
print("| pass | queries | window % | tree % | pyramid % | nodes / tree query | leaves / tree query | leaf points / tree query | logical MB (32 B node + 12 B point) |")
print("|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
for i, r in enumerate(c):
    q, w, t, p, nd, lf, lp = r[:7]
    t1 = max(t, 1.0)
    print(f"| {i} | {q:.0f} | {100*w/q:.1f} | {100*t/q:.1f} | {100*p/q:.1f} | {nd/t1:.1f} | {lf/t1:.2f} | {lp/t1:.1f} | {(nd*32+lp*12)/1e6:.0f} |")
