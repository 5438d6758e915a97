#!/usr/bin/env python
"""pr_icp_batch (clouds in, no render, no camera): P copies of the test.cpp cloud with small offsets, kd-tree and projective scenes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pose_refine_amd import api, synth
api.init(0); api.set_option("solve", 1)
for kv in filter(None, os.environ.get("PR_OPTS", "").split(",")):
    k_, v_ = kv.split("="); api.set_option(k_, int(v_))
model = api.Model(os.path.join(ROOT, "tests/golden/obj_06.ply"))
K = synth.K_TEST; W, H = 640, 480; proj = api.compute_proj(K, W, H)
depth = api.render_host(model, synth.test_cpp_poses(), W, H, proj)
cloud = api.depth2cloud(api.DeviceVector.from_host(depth[0].reshape(-1).astype(np.int32)), W, H, K).to_host().reshape(-1, 3)
scenes = {"proj": api.Scene_projective().init_Scene_projective_cuda(depth[1], K), "nn": api.Scene_nn().init_Scene_nn_cuda(depth[1], K)}
crit = api.ICPConvergenceCriteria(0.0, 0.0, 20)
for kind, sc in scenes.items():
    for P in (1, 4, 16, 64, 256):
        host = np.concatenate([cloud + np.float32(0.00004 * i) for i in range(P)]).astype(np.float32)
        offs = (np.arange(P + 1) * len(cloud)).astype(np.uint32)
        ts = []
        for _ in range(5):
            dev = api.DeviceVector.from_host(host.reshape(-1))
            t0 = time.perf_counter(); r = api.ICP_Point2Plane_batch(dev, offs, sc, crit); ts.append(time.perf_counter() - t0)
        print(f"{kind} P={P}: {np.median(ts[1:])*1e3:.3f} ms per call = {P/np.median(ts[1:]):.0f} clouds/s")
