"""CPU count behind DESIGN.md section 8 "tile packets": for hypotheses of the bench scene, how many kd-tree leaves / points one query has to look at with a perfect
(slack 1.0) or loose (1.5) bound, against what a PACKET of queries walked together has to look at: row strips of 64 consecutive cloud points, 8 x 8 / 16 x 4 /
4 x 4 pixel tiles; boxes tested against the packet's box + largest radius, or per lane.  Uses the oracle's tree (tests/oracle_lib.py): a measurement helper."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from pose_refine_amd import synth
import ctypes as C
K = synth.K_TEST; W, H = 640, 480
proj = O.compute_proj(K, W, H)
tris = O.ply_load(os.path.join(ROOT, "tests/golden/obj_06.ply"))
sd = O.render(tris, synth.scene_pose()[None], W, H, proj)[0]
sc = O.NNScene(sd, K)
nodes = sc.nodes if hasattr(sc, "nodes") else None
nd = sc.nodes
leaf = (nd["child1"] < 0) | (nd["child2"] < 0)
lb = nd["bbox"][leaf]            # (L,6)
print("scene pts", len(sc.pcd), "nodes", len(nd), "leaves", leaf.sum(), "bbox sample", lb[0], "left/right", nd["left"][leaf][:3], nd["right"][leaf][:3])
cnt = (nd["right"][leaf] - nd["left"][leaf])
print("leaf pts mean", cnt.mean(), cnt.min(), cnt.max())
# which bbox layout? try min = [0:3], max=[3:6] or interleaved
print(sc.pcd.min(0), sc.pcd.max(0), nd["bbox"][0])
L = int(leaf.sum()); lo = np.zeros((L,3),np.float32); hi = np.zeros((L,3),np.float32)
lft = nd["left"][leaf]; rgt = nd["right"][leaf]
for i in range(L):
    p = sc.pcd[lft[i]:rgt[i]+1] if rgt[i] >= lft[i] else sc.pcd[lft[i]:lft[i]+1]
    lo[i] = p.min(0); hi[i] = p.max(0)
npts = np.maximum(rgt - lft + 1, 1)
print("leaf pts (incl)", npts.mean())
from scipy.spatial import cKDTree
kt = cKDTree(sc.pcd)
poses = synth.hypotheses(6)
def lbdist2(qlo, qhi):   # (Q,3) boxes vs leaves -> (Q,L) squared gap
    g = np.maximum(0, np.maximum(lo[None] - qhi[:,None], qlo[:,None] - hi[None]))
    return (g*g).sum(-1)
for pi in range(1, 6):
    dep = O.render(tris, poses[pi][None], W, H, proj)[0]
    cloud = O.depth2cloud(dep, K)
    cloud = cloud[0] if isinstance(cloud, tuple) else cloud
    ys, xs = np.nonzero(dep.reshape(H, W) > 0)
    assert len(ys) == len(cloud), (len(ys), len(cloud))
    dnn, inn = kt.query(cloud)
    for slack in (1.0, 1.5):
        bound = dnn * slack
        G = lbdist2(cloud, cloud)
        single = (G <= (bound**2)[:,None])
        s_leaves = single.sum(1).mean(); s_pts = (single * npts[None]).sum(1).mean()
        out = [f"pose {pi} slack {slack}: n={len(cloud)} dnn mean {dnn.mean()*1e3:.1f} mm | single: leaves {s_leaves:.1f} pts {s_pts:.0f}"]
        for name, key in (("strip64", np.arange(len(cloud)) // 64), ("tile8x8", (ys // 8) * 1000 + xs // 8), ("tile16x4", (ys // 4) * 1000 + xs // 16), ("tile4x4", (ys // 4) * 1000 + xs // 4)):
            order = np.argsort(key, kind="stable"); k = key[order]
            starts = np.flatnonzero(np.r_[True, k[1:] != k[:-1]]); ends = np.r_[starts[1:], len(k)]
            qlo = np.array([cloud[order[a:b]].min(0) for a, b in zip(starts, ends)]); qhi = np.array([cloud[order[a:b]].max(0) for a, b in zip(starts, ends)])
            rmax = np.array([bound[order[a:b]].max() for a, b in zip(starts, ends)])
            Gp = lbdist2(qlo, qhi); m = Gp <= (rmax**2)[:,None]
            sizes = ends - starts
            # tighter: any lane's ball intersects leaf
            anyl = np.array([single[order[a:b]].any(0).sum() for a, b in zip(starts, ends)])
            anyp = np.array([(single[order[a:b]].any(0) * npts).sum() for a, b in zip(starts, ends)])
            out.append(f"   {name}: packets {len(starts)} fill {sizes.mean():.1f} | aabb+rmax leaves/packet {m.sum(1).mean():.0f} pts {(m*npts[None]).sum(1).mean():.0f} | per-lane-ball union leaves {anyl.mean():.0f} pts {anyp.mean():.0f}")
        print("\n".join(out), flush=True)
