#!/usr/bin/env python
"""Latency of ONE ICP_Point2Plane_cuda call (test.cpp:129-172: one cloud of 26 k points, 20 iterations) -- the reference's own usage."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pose_refine_amd import api, synth
api.init(0)
for kv in filter(None, os.environ.get("PR_OPTS", "").split(",")):
    k_, v_ = kv.split("="); api.set_option(k_, int(v_))
model = api.Model(os.path.join(ROOT, "tests/golden/obj_06.ply"))
K = synth.K_TEST; W, H = 640, 480; proj = api.compute_proj(K, W, H)
poses = synth.test_cpp_poses()
depth = api.render_host(model, poses, W, H, proj)
cloud = api.depth2cloud(api.DeviceVector.from_host(depth[0].reshape(-1).astype(np.int32)), W, H, K).to_host()
scenes = {"proj": api.Scene_projective().init_Scene_projective_cuda(depth[1], K), "nn": api.Scene_nn().init_Scene_nn_cuda(depth[1], K)}
crit = api.ICPConvergenceCriteria(0.0, 0.0, 20)
for solve in (api.SOLVE_HOST, api.SOLVE_DEVICE):
    api.set_option("solve", solve)
    for kind, sc in scenes.items():
        ts = []
        for i in range(12):
            dev = api.DeviceVector.from_host(cloud.reshape(-1))
            t0 = time.perf_counter(); r = api.ICP_Point2Plane(dev, sc, crit); ts.append(time.perf_counter() - t0)
        print(f"solve {'host' if solve == api.SOLVE_HOST else 'device'} {kind}: {np.median(ts[2:])*1e3:.3f} ms per call (fitness {r.fitness_:.4f})")
