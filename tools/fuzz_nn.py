"""Many seeds of the kd-tree variant-agreement tests -- a campaign, not a tracked test."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from pose_refine_amd import api
import test_kdtree_search_gpu as R
Pg = R
api.init(0); api.set_option("solve", api.SOLVE_DEVICE)
t0 = time.time(); budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
bad = []; n = 0; seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    npts = int(rng.choice([1, 2, 7, 40, 300, 2500, 9000, 30000, 70000]))
    ml = int(rng.choice([1, 2, 3, 8, 10, 15, 16, 24]))
    try:
        R.test_task_walk_on_random_scenes_equals_ordered_walks(True, npts, ml, seed)
        if npts >= 300:                                         # (the tracked test asserts fitness > 0.5, which a scene of a few dozen points need not reach)
            api.set_option("solve", api.SOLVE_HOST)
            Pg.test_nn_variants_agree_on_tie_heavy_clouds(True, seed, min(npts, 6000), ml)
            api.set_option("solve", api.SOLVE_DEVICE)
    except AssertionError as e:
        bad.append((seed, npts, ml)); print("MISMATCH", seed, npts, ml, str(e)[:300], flush=True)
        api.set_option("solve", api.SOLVE_DEVICE)
    seed += 1; n += 1
print(f"{n} random kd-tree scenes in {time.time()-t0:.0f} s, mismatches: {len(bad)}", bad[:5])
