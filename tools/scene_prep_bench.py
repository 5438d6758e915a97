#!/usr/bin/env python
"""Scene preparation: reference scheme (CPU normals / kd build + upload) vs the device kernels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pose_refine_amd import api, synth
api.init(0)
model = api.Model(os.path.join(ROOT, "tests/golden/obj_06.ply"))
K = synth.K_TEST; proj = api.compute_proj(K, 640, 480)
sd = api.render_host(model, synth.scene_pose()[None], 640, 480, proj)[0]
dev = api.DeviceVector.from_host(sd.reshape(-1))
def timeit(f, n=10):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
print("projective scene: cpu prepare + upload  %.2f ms" % timeit(lambda: api.Scene_projective().init_Scene_projective_cuda(sd, K)))
print("projective scene: device prepare        %.3f ms" % timeit(lambda: api.Scene_projective().init_Scene_projective_device(dev, K)))
print("kd-tree scene:    cpu prepare + upload  %.2f ms" % timeit(lambda: api.Scene_nn().init_Scene_nn_cuda(sd, K)))
print("kd-tree scene:    device prepare+build  %.2f ms" % timeit(lambda: api.Scene_nn().init_Scene_nn_device(dev, K, 640, 480)))
