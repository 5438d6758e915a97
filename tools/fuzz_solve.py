#!/usr/bin/env python
"""Campaign: host solve (pipelined pose groups, fused / separate finalize) against device solve (fused / separate, 1-4 groups, graph on / off,
synchronous and on the slots) on random batches -- records must be identical.   python tools/fuzz_solve.py [seconds] [start seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pose_refine_amd import api, synth
api.init(0)
model = api.Model(os.path.join(ROOT, "tests/golden/obj_06.ply"))
K = synth.K_TEST; W, H = 640, 480; proj = api.compute_proj(K, W, H)
sd = api.render_host(model, synth.scene_pose()[None], W, H, proj)[0]
scenes = {"proj": api.Scene_projective().init_Scene_projective_cuda(sd, K), "nn": api.Scene_nn().init_Scene_nn_cuda(sd, K)}
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = time.time(); n = 0; bad = 0
names = ("solve", "fused_solve", "pose_groups", "graph", "sub_batch")
try:
    while time.time() - t0 < budget:
        rng = np.random.default_rng(seed)
        kind = "nn" if rng.random() < 0.3 else "proj"
        P = int(rng.choice([1, 2, 31, 32, 33, 64, 65, 97, 130, 200] if kind == "proj" else [1, 3, 32, 33, 66]))
        poses = synth.hypotheses(P, seed=seed)
        if P > 2 and rng.random() < 0.4: poses.reshape(-1, 4, 4)[int(rng.integers(P)), 0, 3] += 1e6           # an empty cloud somewhere
        if rng.random() < 0.3: poses.reshape(-1, 4, 4)[:, 2, 3] += float(rng.choice([300.0, -200.0]))
        crit = api.ICPConvergenceCriteria(0.0, 0.0, int(rng.choice([0, 1, 5, 20]))) if rng.random() < 0.6 else api.ICPConvergenceCriteria(float(rng.choice([1e-5, 1e-4])), float(rng.choice([1e-5, 1e-4])), 30)
        ref = None
        combos = [(0, 1, 2, 1, 512), (0, 0, 1, 1, 512), (0, 1, int(rng.integers(1, 5)), 1, 512), (1, 1, 2, 1, 512), (1, 0, int(rng.integers(1, 5)), 0, 512), (1, 1, 1, int(rng.integers(2)), 32)]
        for c in combos:
            for k, v in zip(names, c): api.set_option(k, v)
            out = api.refine_batch(model, poses, W, H, proj, K, scenes[kind], crit)
            blob = out[0].tobytes() + out[1].tobytes()
            if ref is None: ref = blob
            elif blob != ref:
                bad += 1; print("MISMATCH seed", seed, kind, "P", P, "combo", c, flush=True)
        seed += 1; n += 1
finally:
    for k, v in zip(names, (0, 1, 0, 1, 512)): api.set_option(k, v)
print(f"{n} random batches x 6 solve configurations in {time.time()-t0:.0f} s, mismatches: {bad}")
sys.exit(1 if bad else 0)
