#!/usr/bin/env python
"""SURVEY 8f rank 1 ("needed when the scene changes per frame"): what a NEW scene costs on the device -- preparation from a depth image that is
already in HBM (normals + back-projection; for kd-tree scenes the valid-pixel gather and the level-order tree build) and the first refine call
against it (which derives the search structures the library caches per scene), next to the steady step against an unchanged scene.
Two scenes: the bench's (the object alone, about 24 k valid pixels) and a frame-filling one (the object in front of a wall, 307 k valid pixels)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pose_refine_amd import api, synth

W, H = 640, 480
api.init(0); api.set_option("solve", 1)
for kv in filter(None, os.environ.get("PR_OPTS", "").split(",")):
    k_, v_ = kv.split("="); api.set_option(k_, int(v_))
model = api.Model(os.path.join(ROOT, "tests/golden/obj_06.ply"))
K = synth.K_TEST; proj = api.compute_proj(K, W, H)
obj = api.render_host(model, synth.scene_pose()[None], W, H, proj)[0].astype(np.int32)
yy, xx = np.mgrid[0:H, 0:W]
wall = (900 + 0.05 * xx + 0.03 * yy + 3.0 * np.sin(xx / 17.0) * np.cos(yy / 23.0)).astype(np.int32)       # a gently curved wall behind the object
full = np.where(obj > 0, obj, wall).astype(np.int32)
poses = synth.hypotheses(256)
crit = api.ICPConvergenceCriteria(0.0, 0.0, 20)
REPS = int(os.environ.get("REPS", "12"))

def med(v): return float(np.median(v)) * 1e3

def timed(fn):
    api.sync(); t = time.perf_counter(); r = fn(); api.sync(); return time.perf_counter() - t, r

for name, depth in (("object alone", obj), ("object + wall (frame filled)", full)):
    nvalid = int((depth > 0).sum())
    for kind in ("proj", "nn"):
        prep, first, steady = [], [], []
        keep = None
        for rep in range(REPS):
            d = depth.copy(); d[rep % H, rep % W] = 0 if depth[rep % H, rep % W] else 905       # a new frame every time (one pixel differs): nothing cached applies
            dev = api.DeviceVector.from_host(d.reshape(-1))
            # (one scene object re-initialised per frame: it keeps its arrays, so no allocation is timed after the first frames)
            if keep is None: keep = api.Scene_projective() if kind == "proj" else api.Scene_nn()
            scene = keep
            if kind == "proj": t, _ = timed(lambda: scene.init_Scene_projective_device(dev, K, W, H))
            else: t, _ = timed(lambda: scene.init_Scene_nn_device(dev, K, W, H))
            prep.append(t)
            t, _ = timed(lambda: api.refine_batch(model, poses, W, H, proj, K, scene, crit)); first.append(t)
            t, _ = timed(lambda: api.refine_batch(model, poses, W, H, proj, K, scene, crit)); steady.append(t)
            t, _ = timed(lambda: api.refine_batch(model, poses, W, H, proj, K, scene, crit)); steady.append(t)
            del dev
        print(f"{name:30s} {kind:4s} valid pixels {nvalid:6d}: prepare {med(prep[2:]):7.3f} ms   first 256-hypothesis batch {med(first[2:]):7.3f} ms   "
              f"steady batch {med(steady[4:]):7.3f} ms   => a new scene costs {med(prep[2:]) + med(first[2:]) - med(steady[4:]):6.3f} ms", flush=True)
