"""Many seeds of the two randomised parity tests (render, fused pipeline) -- a campaign, not a tracked test."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from pose_refine_amd import api
import test_render_gpu as FR
import test_refine_batch_gpu as FB
api.init(0)
t0 = time.time()
bad = []
n = 0
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    W, H = int(rng.integers(16, 400)), int(rng.integers(16, 300))
    try:
        FR.test_render_random_scenes(True, seed, W, H)
        FB.test_fused_pipeline_random_scenes(True, seed + 1, W, H)
    except AssertionError as e:
        bad.append((seed, W, H, str(e)[:200]))
        print("MISMATCH", seed, W, H, str(e)[:300], flush=True)
    seed += 2; n += 1
print(f"{n} random scenes in {time.time()-t0:.0f} s, mismatches: {len(bad)}", bad[:5])
