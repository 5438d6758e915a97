#!/usr/bin/env python
"""Campaign: scene preparation on random depth images (random sizes, blobs, holes, noise, 16- and 32-bit, values beyond the 2000 / 65535 gates):
host preparation == device preparation == the oracle's, for projective scenes (points, normals) and kd-tree scenes (points, normals, nodes,
any max_leaf).   python tools/fuzz_scene.py [seconds] [start seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from pose_refine_amd import api
api.init(0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = time.time(); n = 0; bad = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    W, H = (int(rng.integers(300, 1400)), int(rng.integers(200, 800))) if os.environ.get("PR_FUZZ_BIG") else (int(rng.integers(12, 400)), int(rng.integers(12, 300)))   # PR_FUZZ_BIG: up to a million pixels (hundreds of tiles per level of the device build)
    K = np.array([rng.uniform(0.6, 1.5) * W, 0, W / 2 + rng.uniform(-5, 5), 0, rng.uniform(0.6, 1.5) * W, H / 2 + rng.uniform(-5, 5), 0, 0, 1], np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    d = np.zeros((H, W), np.float64)
    for _ in range(int(rng.integers(1, 5))):                      # blobs: tilted planes / bumps
        cx, cy, r = rng.uniform(0, W), rng.uniform(0, H), rng.uniform(5, max(W, H) / 2)
        m = (xx - cx) ** 2 + (yy - cy) ** 2 < r * r
        z = rng.uniform(200, 2500) + rng.uniform(-3, 3) * (xx - cx) + rng.uniform(-3, 3) * (yy - cy) + rng.uniform(0, 30) * np.sin(xx / rng.uniform(2, 20))
        d = np.where(m, z, d)
    d += np.where(d > 0, rng.normal(0, rng.choice([0, 1, 20]), d.shape), 0)
    d[rng.random(d.shape) < rng.choice([0, 0.02, 0.3])] = 0
    if rng.random() < 0.3: d[rng.integers(0, H), :] = rng.choice([70000, 65535, 65536, 1999, 2000, 2001])
    dt = np.int32 if rng.random() < 0.5 else np.uint16
    di = np.clip(d, 0 if dt == np.uint16 else -50, 65535 if dt == np.uint16 else 2 ** 31 - 1).astype(dt)
    ml = int(rng.choice([1, 2, 5, 10, 15, 16, 40]))
    ok = True
    # projective scene
    hp = api.Scene_projective().init_Scene_projective_cuda(di, K, W, H)
    dp = api.Scene_projective().init_Scene_projective_device(api.DeviceVector.from_host(di.reshape(-1)), K, W, H)
    op = O.ProjScene(di, K)
    ok &= np.array_equal(hp.pcd_host, op.pcd) and np.array_equal(hp.normal_host, op.normal)
    ok &= np.array_equal(dp.pcd_buffer.to_host().reshape(-1, 3), op.pcd) and np.array_equal(dp.normal_buffer.to_host().reshape(-1, 3), op.normal)
    # kd-tree scene
    if int((di > 0).sum()) > 0:
        hs = api.Scene_nn().init_Scene_nn_cuda(di, K, max_leaf=ml)
        ds = api.Scene_nn().init_Scene_nn_device(api.DeviceVector.from_host(di.reshape(-1)), K, W, H, max_leaf=ml)
        on = O.NNScene(di, K, max_leaf=ml)
        npt, nn = len(on.pcd), len(on.nodes)
        ok &= len(hs.pcd_host) == npt and np.array_equal(hs.pcd_host, on.pcd) and np.array_equal(hs.normal_host, on.normal) and hs.nodes_host.tobytes() == on.nodes.tobytes()
        ok &= (ds._n_points, ds._n_nodes) == (npt, nn) and np.array_equal(ds.pcd_buffer.to_host()[:3 * npt].reshape(-1, 3), on.pcd) \
            and np.array_equal(ds.normal_buffer.to_host()[:3 * npt].reshape(-1, 3), on.normal) and ds.nodes.to_host()[:nn].tobytes() == on.nodes.tobytes()
    if not ok:
        bad += 1; print("MISMATCH seed", seed, W, H, dt.__name__, "max_leaf", ml, flush=True)
    seed += 1; n += 1
print(f"{n} random depth images in {time.time()-t0:.0f} s, mismatches: {bad}")
sys.exit(1 if bad else 0)
