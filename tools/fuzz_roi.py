#!/usr/bin/env python
"""Campaign: refinement inside random ROI windows and against random crops of the projective scene, against the oracle (cloud sizes and inlier
counts exact, transforms 1e-4), synchronous and on the asynchronous slots.   python tools/fuzz_roi.py [seconds] [start seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from pose_refine_amd import api, synth
api.init(0); api.set_option("solve", api.SOLVE_DEVICE)
tris = O.ply_load(os.path.join(ROOT, "tests/golden/obj_06.ply"))
model = api.Model(tris=tris)
K = synth.K_TEST; W, H = synth.WIDTH, synth.HEIGHT
proj = O.compute_proj(K, W, H)
sd = O.render(tris, synth.scene_pose()[None], W, H, proj)[0]
gscene = api.Scene_projective().init_Scene_projective_cuda(sd, K)
oscene = O.ProjScene(sd, K)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ys, xs = np.nonzero(sd)
t0 = time.time(); n = 0; bad = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    poses = synth.hypotheses(3, seed=seed)
    crit = (0.0, 0.0, int(rng.choice([1, 4])))
    ppb = api.get_option("points_per_block")
    # a window somewhere around the object (sometimes missing it altogether)
    x0 = int(rng.integers(max(0, xs.min() - 80), xs.max())); y0 = int(rng.integers(max(0, ys.min() - 80), ys.max()))
    w = int(rng.integers(1, W - x0 + 1)); h = int(rng.integers(1, H - y0 + 1))
    if rng.random() < 0.5: w = min(w, int(rng.integers(1, 200))); h = min(h, int(rng.integers(1, 200)))
    roi = (x0, y0, w, h)
    res, sizes = api.refine_batch(model, poses, W, H, proj, K, gscene, api.ICPConvergenceCriteria(*crit), roi=roi)
    ores, osizes, _ = O.refine_batch(tris, poses, W, H, proj, K, oscene, crit, O.SUM_CANONICAL, ppb, roi=roi)
    ok = np.array_equal(sizes, osizes) and np.array_equal(res["fitness"], ores["fitness"]) and np.allclose(res["T"], ores["T"], rtol=0, atol=1e-4)
    api.refine_submit(1, model, poses, W, H, proj, K, gscene, api.ICPConvergenceCriteria(*crit), roi=roi)
    ares, asizes = api.refine_wait(1)
    ok &= np.array_equal(asizes, sizes) and ares.tobytes() == res.tobytes()
    # a crop of the scene (pcd2dep with offsets)
    cx0 = int(rng.integers(0, W - 8)); cy0 = int(rng.integers(0, H - 8)); cw = int(rng.integers(8, W - cx0 + 1)); ch = int(rng.integers(8, H - cy0 + 1))
    win = (cx0, cy0, cw, ch)
    cres, csizes = api.refine_batch(model, poses, W, H, proj, K, gscene.crop(win), api.ICPConvergenceCriteria(*crit))
    cores, cosizes, _ = O.refine_batch(tris, poses, W, H, proj, K, oscene.crop(win), crit, O.SUM_CANONICAL, ppb)
    ok &= np.array_equal(csizes, cosizes) and np.array_equal(cres["fitness"], cores["fitness"]) and np.allclose(cres["T"], cores["T"], rtol=0, atol=1e-4)
    if not ok:
        bad += 1; print("MISMATCH seed", seed, "roi", roi, "crop", win, flush=True)
    seed += 1; n += 1
print(f"{n} random ROI windows + scene crops in {time.time()-t0:.0f} s, mismatches: {bad}")
sys.exit(1 if bad else 0)
