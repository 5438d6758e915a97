"""CPU: pixel-window half-widths an exact window search would need per pass (bench scene), with an ideal bound (true NN distance)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from pose_refine_amd import synth
from scipy.spatial import cKDTree
K = synth.K_TEST; W, H = 640, 480
proj = O.compute_proj(K, W, H)
tris = O.ply_load(os.path.join(ROOT, "tests/golden/obj_06.ply"))
sd = O.render(tris, synth.scene_pose()[None], W, H, proj)[0]
sc = O.NNScene(sd, K)
kt = cKDTree(sc.pcd.astype(np.float64))
Kf = np.asarray(K, np.float32).reshape(3, 3); fx, fy, cx, cy = float(Kf[0,0]), float(Kf[1,1]), float(Kf[0,2]), float(Kf[1,2])
print("fx", fx, "scene pts", len(sc.pcd), "z range", sc.pcd[:,2].min(), sc.pcd[:,2].max())
poses = synth.hypotheses(16)
for it in (0, 1, 2, 3, 4):
    allw = []; alld = []
    for pi in range(1, 13):
        dep = O.render(tris, poses[pi][None], W, H, proj)[0]
        cloud = O.depth2cloud(dep, K)
        if it:
            _, _, cloud, _ = O.icp(cloud, sc, (0.0, 0.0, it), O.SUM_CANONICAL if hasattr(O, "SUM_CANONICAL") else 0)
        d, _ = kt.query(cloud.astype(np.float64))
        x, y, z = cloud[:, 0], cloud[:, 1], cloud[:, 2]
        r = d * 1.0001
        k = r / (z * (z - r))
        du = fx * k * (z + np.abs(x)); dv = fy * k * (z + np.abs(y))
        w = np.ceil(np.maximum(du, dv) + 1e-3)
        allw.append(w); alld.append(d)
    w = np.concatenate(allw); d = np.concatenate(alld)
    qs = [50, 75, 90, 95, 99, 99.9, 100]
    print(f"pass {it}: d mean {d.mean()*1e3:.2f} mm  half-width quantiles", dict(zip(qs, np.percentile(w, qs))),
          " frac<=2: %.3f <=4: %.3f <=6: %.3f <=8: %.3f <=12 %.3f" % tuple((w <= t).mean() for t in (2, 4, 6, 8, 12)),
          " mean cells (2w+1)^2: %.0f" % ((2 * np.minimum(w, 12) + 1) ** 2).mean())
