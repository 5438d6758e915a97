#!/usr/bin/env python
"""Stress of the asynchronous slots: random batch sizes / poses alternate over the two slots (growing and shrinking
workspaces, changing grid hints, sub-batches, empty clouds) and every batch is compared bit for bit with the synchronous path.

    [PR_STRESS_TIMED=1] [PR_STRESS_SOLVE=host] python tools/stress_async.py [jobs] [kd-tree fraction]
    (PR_STRESS_TIMED: every third batch is a timed one, profile 3; PR_STRESS_SOLVE=host: the 6x6 solve on the host -- submitted batches run on the slots' helper threads)

With a kd-tree fraction > 0 some jobs run against one of THREE kd-tree scenes: a context holds two sets of derived records (search records, pixel
grid), so three scenes in random order force rebuilds -- also of a set the other slot's batch is reading, which then has to finish first."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pose_refine_amd import api, synth
api.init(0); api.set_option("solve", 0 if os.environ.get("PR_STRESS_SOLVE") == "host" else 1)   # host: the slots' helper threads run the batches
model = api.Model(os.path.join(ROOT, "tests/golden/obj_06.ply"))
K = synth.K_TEST; W, H = 640, 480; proj = api.compute_proj(K, W, H)
sd = api.render_host(model, synth.scene_pose()[None], W, H, proj)[0]
scene = api.Scene_projective().init_Scene_projective_cuda(sd, K)
nn_frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
other = synth.scene_pose().copy(); other.reshape(4, 4)[0, 3] += 15.0; other.reshape(4, 4)[2, 3] += 25.0
sd2 = api.render_host(model, other[None], W, H, proj)[0]
third = synth.scene_pose().copy(); third.reshape(4, 4)[1, 3] -= 12.0; third.reshape(4, 4)[2, 3] -= 20.0
sd3 = api.render_host(model, third[None], W, H, proj)[0]
nn_scenes = [api.Scene_nn().init_Scene_nn_cuda(d, K) for d in (sd, sd2, sd3)] if nn_frac > 0 else []
rng = np.random.default_rng(7)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
jobs = []
for i in range(N):
    use_nn = nn_frac > 0 and rng.random() < nn_frac
    P = int(rng.choice([1, 3, 31, 33, 64, 65]) if use_nn else rng.choice([1, 3, 31, 33, 64, 65, 200, 256, 300, 513, 700]))
    poses = synth.hypotheses(P, seed=100 + i)
    if rng.random() < 0.5:
        poses.reshape(-1, 4, 4)[:, 2, 3] += float(rng.choice([0.0, 400.0, 1500.0, -300.0]))
    if P > 2 and rng.random() < 0.3:
        poses.reshape(-1, 4, 4)[1, 0, 3] += 1e6
    crit = api.ICPConvergenceCriteria(0.0, 0.0, int(rng.choice([0, 3, 20]))) if rng.random() < 0.7 else api.ICPConvergenceCriteria(1e-5, 1e-5, 30)
    jobs.append((poses, crit, nn_scenes[int(rng.integers(3))] if use_nn else scene))
api.set_option("sub_batch", 256)
api.set_option("profile", 1)                                   # timed calls take the synchronous path
refs = [api.refine_batch(model, p, W, H, proj, K, sc, c) for p, c, sc in jobs]
api.set_option("profile", 0)
got = [None] * N
infl = [None, None]
for i, (p, c, sc) in enumerate(jobs):
    b = i & 1
    if infl[b] is not None:
        got[infl[b]] = api.refine_wait(b)
    timed = os.environ.get("PR_STRESS_TIMED") and (i % 3 == 1)            # every third batch a TIMED asynchronous one (profile 3)
    if timed: api.set_option("profile", 3)
    api.refine_submit(b, model, p, W, H, proj, K, sc, c)
    if timed: api.set_option("profile", 0)
    infl[b] = i
for b in (0, 1):
    if infl[b] is not None:
        got[infl[b]] = api.refine_wait(b)
bad = 0
for i in range(N):
    ok = np.array_equal(got[i][1], refs[i][1]) and got[i][0].tobytes() == refs[i][0].tobytes()
    bad += (not ok)
    if not ok: print("MISMATCH job", i, "P", len(jobs[i][0]))
print(f"{N} jobs, {bad} mismatches")
sys.exit(1 if bad else 0)
