#!/usr/bin/env python
"""Workload for rocprofv3 --pmc passes: 3 fused batches (P poses) + one public render call whose
fill / max2zero kernels move a KNOWN number of bytes (calibration of FETCH_SIZE / WRITE_SIZE)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pose_refine_amd import api, synth
P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
scene_kind = sys.argv[2] if len(sys.argv) > 2 else "proj"
api.init(0); api.set_option("solve", 1); api.set_option("raster_mode", int(os.environ.get("PR_RASTER_MODE", "0")))
for kv in filter(None, os.environ.get("PR_OPTS", "").split(",")):      # e.g. PR_OPTS=nn_run=8,nn_grid=0,pose_groups=1
    k, v = kv.split("="); api.set_option(k, int(v))
model = api.Model(os.path.join(ROOT, "tests/golden/obj_06.ply"))
K = synth.K_TEST; proj = api.compute_proj(K, 640, 480)
sd = api.render_host(model, synth.scene_pose()[None], 640, 480, proj)[0]
scene = api.Scene_projective().init_Scene_projective_cuda(sd, K) if scene_kind == "proj" else api.Scene_nn().init_Scene_nn_cuda(sd, K)
poses = synth.hypotheses(P)
for _ in range(3):
    _, sizes = api.refine_batch(model, poses, 640, 480, proj, K, scene, api.ICPConvergenceCriteria(0.0, 0.0, 20))
print("points per batch:", int(sizes.sum()))
d = api.render(model, poses, 640, 480, proj)      # fill: P*1228800 B written; max2zero: same read + written
print("calibration bytes per fill/max2zero:", P * 640 * 480 * 4)
