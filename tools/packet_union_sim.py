"""packets of queries walked together (pass 0): leaves / points per packet for strips of the queue order and pixel tiles, per-lane-ball union"""
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
nd = sc.nodes
leaf = (nd["child1"] < 0) | (nd["child2"] < 0)
lft = nd["left"][leaf].astype(np.int64); rgt = nd["right"][leaf].astype(np.int64)
L = int(leaf.sum()); lo = np.zeros((L,3),np.float32); hi = np.zeros((L,3),np.float32)
for i in range(L):
    p = sc.pcd[lft[i]:rgt[i]]
    lo[i] = p.min(0); hi[i] = p.max(0)
npts = (rgt - lft)
kt = cKDTree(sc.pcd.astype(np.float64))
poses = synth.hypotheses(8)
def lbdist2(q):
    g = np.maximum(0, np.maximum(lo[None] - q[:,None], q[:,None] - hi[None]))
    return (g*g).sum(-1)
rng = np.random.default_rng(1)
for pi in (1, 3, 5):
    dep = O.render(tris, poses[pi][None], W, H, proj)[0]
    cloud = O.depth2cloud(dep, K)
    ys, xs = np.nonzero(dep.reshape(H, W) > 0)
    dnn, inn = kt.query(cloud.astype(np.float64))
    # bound: true NN for 80 %, 1.05 x for the rest (what the descent delivers)
    slack = np.where(rng.random(len(cloud)) < 0.8, 1.000001, 1.05)
    bound2 = ((dnn * slack) ** 2).astype(np.float32)
    G = lbdist2(cloud)
    single = G <= bound2[:, None]
    print(f"pose {pi}: n={len(cloud)} dnn mean {dnn.mean()*1e3:.1f} mm | single query: leaves {single.sum(1).mean():.1f} pts {(single*npts[None]).sum(1).mean():.0f}")
    for name, key in (("strip8", np.arange(len(cloud)) // 8), ("strip16", np.arange(len(cloud)) // 16), ("strip32", np.arange(len(cloud)) // 32), ("strip64", np.arange(len(cloud)) // 64),
                      ("tile4x4", (ys // 4) * 1000 + xs // 4), ("tile8x2", (ys // 2) * 1000 + xs // 8), ("tile8x8", (ys // 8) * 1000 + xs // 8)):
        order = np.argsort(key, kind="stable"); k = key[order]
        starts = np.flatnonzero(np.r_[True, k[1:] != k[:-1]]); ends = np.r_[starts[1:], len(k)]
        un_l = []; un_p = []; sz = []
        for a, b in zip(starts, ends):
            u = single[order[a:b]].any(0)
            un_l.append(u.sum()); un_p.append((u * npts).sum()); sz.append(b - a)
        un_l = np.array(un_l); un_p = np.array(un_p); sz = np.array(sz)
        print(f"   {name}: packets {len(starts)} fill {sz.mean():.1f} | union leaves/packet {un_l.mean():.1f} pts/packet {un_p.mean():.0f} | point tests per QUERY {(un_p.sum())/len(cloud)*1.0:.0f} -> x{un_p.sum()/len(cloud)/((single*npts[None]).sum(1).mean()):.2f} of single", flush=True)
