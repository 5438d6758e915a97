"""CPU model of the task walk over the wide records: node tasks per query by wide level, leaf tasks, for pass-0 queries of a few hypotheses."""
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
n_nodes = len(nd)
c1 = nd["child1"].astype(np.int64); c2 = nd["child2"].astype(np.int64)
isleaf = (c1 < 0) | (c2 < 0)
left = nd["left"].astype(np.int64); right = nd["right"].astype(np.int64)
# tight boxes per node (bottom-up)
lo = np.zeros((n_nodes, 3), np.float32); hi = np.zeros((n_nodes, 3), np.float32)
for i in range(n_nodes - 1, -1, -1):
    if isleaf[i]:
        p = sc.pcd[left[i]:right[i]]
        lo[i] = p.min(0); hi[i] = p.max(0)
    else:
        lo[i] = np.minimum(lo[c1[i]], lo[c2[i]]); hi[i] = np.maximum(hi[c1[i]], hi[c2[i]])
diag2 = ((hi - lo) ** 2).sum(1)

def build_wide(fan=8):
    wide = []   # list of (level, [child binary ids])
    level_nodes = [0]; lvl = 0
    while level_nodes:
        nxt = []
        for r in level_nodes:
            fr = [r]; sz = [(-1.0 if isleaf[r] else np.inf)]
            while len(fr) < fan:
                cand = [(s, i) for i, s in enumerate(sz) if s >= 0]
                if not cand: break
                pick = max(cand)[1]
                n = fr[pick]
                fr[pick] = c1[n]; sz[pick] = -1.0 if isleaf[c1[n]] else diag2[c1[n]]
                fr.append(c2[n]); sz.append(-1.0 if isleaf[c2[n]] else diag2[c2[n]])
            wide.append((lvl, r, fr))
            nxt += [n for n in fr if not isleaf[n]]
        level_nodes = nxt; lvl += 1
    return wide

def lbdist2(q, blo, bhi):   # (Q,3) vs (B,3) -> (Q,B)
    g = np.maximum(0, np.maximum(blo[None] - q[:, None], q[:, None] - bhi[None]))
    return (g * g).sum(-1)

def simulate(wide, cloud, bound2):
    """returns node tasks per level (mean per query), leaf tasks mean"""
    Q = len(cloud)
    root_of = {r: k for k, (lv, r, fr) in enumerate(wide)}
    visited = {0: np.ones(Q, bool)}          # wide index -> mask of queries with a task on it
    maxl = max(lv for lv, _, _ in wide) + 1
    node_tasks = np.zeros(maxl); leaf_tasks = 0.0; leaf_pts = 0.0; box_tests = 0.0
    for k, (lv, r, fr) in enumerate(wide):
        m = visited.get(k)
        if m is None or not m.any(): continue
        node_tasks[lv] += m.sum()
        idx = np.flatnonzero(m)
        fr = np.array(fr)
        d = lbdist2(cloud[idx], lo[fr], hi[fr]) <= bound2[idx, None]
        for ci, n in enumerate(fr):
            hit = idx[d[:, ci]]
            if isleaf[n]:
                leaf_tasks += len(hit); leaf_pts += len(hit) * (right[n] - left[n])
            else:
                mm = np.zeros(Q, bool); mm[hit] = True
                visited[root_of[n]] = mm
    return node_tasks / Q, leaf_tasks / Q, leaf_pts / Q

if __name__ == "__main__":
    kt = cKDTree(sc.pcd.astype(np.float64))
    poses = synth.hypotheses(8)
    for fan in (8, 16):
        wide = build_wide(fan)
        lv = np.array([w[0] for w in wide])
        print("fan", fan, "wide nodes", len(wide), "per level", np.bincount(lv))
        for pi in (1, 3, 5):
            dep = O.render(tris, poses[pi][None], W, H, proj)[0]
            cloud = O.depth2cloud(dep, K)
            dnn, inn = kt.query(cloud.astype(np.float64))
            for slack in (1.000001, 1.1):
                nt, lt, lp = simulate(wide, cloud, ((dnn * slack) ** 2).astype(np.float32))
                print(f" pose {pi} slack {slack}: dnn mean {dnn.mean()*1e3:.1f} mm node tasks/level {np.round(nt,2)} sum {nt.sum():.1f} leaves {lt:.1f} pts {lp:.0f}", flush=True)
