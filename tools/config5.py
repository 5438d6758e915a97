#!/usr/bin/env python
"""BASELINE.json configs[4]: 1M-triangle synthetic mesh, 1280x720, per-GPU share of 1024 poses (128).
Checks two poses against the CPU oracle (render + cloud + ICP) and times the batch.   tools/config5.py [poses] [--check] [--nn]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from pose_refine_amd import api, synth
P = int(sys.argv[1]) if len(sys.argv) > 1 else 128
check = "--check" in sys.argv
api.init(0); api.set_option("solve", 1)
for kv in filter(None, os.environ.get("PR_OPTS", "").split(",")):      # e.g. PR_OPTS=raster_groups=0
    k, v = kv.split("="); api.set_option(k, int(v))
W, H = 1280, 720
K = synth.intrinsics_720p()
tris = synth.uv_sphere_mesh()
print("triangles", len(tris))
model = api.Model(tris=tris)
proj = api.compute_proj(K, W, H)
poses = synth.hypotheses(P)
sd = api.render_host(model, synth.scene_pose()[None], W, H, proj)[0]
print("scene valid px", int((sd > 0).sum()), "depth range", int(sd[sd > 0].min()), int(sd.max()))
kd = "--nn" in sys.argv                                             # the same workload against the kd-tree scene (165 k scene points)
scene = api.Scene_nn().init_Scene_nn_cuda(sd, K) if kd else api.Scene_projective().init_Scene_projective_cuda(sd, K, W, H)
crit = api.ICPConvergenceCriteria(0.0, 0.0, 20)
res, sizes = api.refine_batch(model, poses, W, H, proj, K, scene, crit)
t0 = time.perf_counter(); n = 3
for _ in range(n): res, sizes = api.refine_batch(model, poses, W, H, proj, K, scene, crit)
dt = (time.perf_counter() - t0) / n
print(f"P={P}: {dt*1e3:.2f} ms/step  {P/dt:.0f} poses/s  mean cloud {sizes.mean():.0f}  mean fitness {res['fitness'].mean():.4f}")
if check:
    import oracle_lib as O
    oproj = O.compute_proj(K, W, H)
    osd = O.render(tris, synth.scene_pose()[None], W, H, oproj)[0]
    assert np.array_equal(osd, sd), "scene render differs from oracle"
    oscene = O.NNScene(sd, K) if kd else O.ProjScene(sd, K)
    ores, osizes, _ = O.refine_batch(tris, poses[:2], W, H, oproj, K, oscene, (0.0, 0.0, 20), O.SUM_CANONICAL, api.get_option("points_per_block"))
    assert np.array_equal(osizes, sizes[:2]), (osizes, sizes[:2])
    assert np.array_equal(ores["fitness"], res["fitness"][:2]), (ores["fitness"], res["fitness"][:2])
    assert np.allclose(ores["T"], res["T"][:2], atol=1e-4)
    print("oracle check ok (2 poses): sizes", osizes.tolist())
