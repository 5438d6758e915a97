"""CPU model: task walk over the wide records with an extra per-child SLAB bound (oriented: direction n, interval of n.p over the subtree's points)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import importlib.util
spec = importlib.util.spec_from_file_location("wide_sim", os.path.join(ROOT, "tools", "wide_sim.py"))
ws = importlib.util.module_from_spec(spec); spec.loader.exec_module(ws)
O, synth, sc = ws.O, ws.synth, ws.sc
n_nodes, isleaf, left, right, c1, c2, lo, hi = ws.n_nodes, ws.isleaf, ws.left, ws.right, ws.c1, ws.c2, ws.lo, ws.hi
pcd = sc.pcd.astype(np.float64)
nrm = np.asarray(sc.normal, np.float64) if hasattr(sc, "normal") else None
print("normals available:", nrm is not None)
# per-node slab: direction by PCA (smallest eigenvector) or by the normal of the node's middle point
def slabs(mode):
    n = np.zeros((n_nodes, 3)); a = np.zeros(n_nodes); b = np.zeros(n_nodes)
    for i in range(n_nodes):
        p = pcd[left[i]:right[i]]
        if mode == "pca":
            if len(p) >= 3:
                c = p - p.mean(0); w, v = np.linalg.eigh(c.T @ c); d = v[:, 0]
            else: d = np.array([0, 0, 1.0])
        elif mode == "normal":
            d = nrm[(left[i] + right[i]) // 2]
            if not np.isfinite(d).all() or np.linalg.norm(d) < 0.5: d = np.array([0, 0, 1.0])
            d = d / np.linalg.norm(d)
        elif mode == "avgnormal":
            d = nrm[left[i]:right[i]]; d = d[np.isfinite(d).all(1)].sum(0)
            d = d / np.linalg.norm(d) if np.linalg.norm(d) > 1e-6 else np.array([0, 0, 1.0])
        t = p @ d
        n[i] = d; a[i] = t.min(); b[i] = t.max()
    return n, a, b
def simulate(wide, cloud, bound2, slab=None, leaf_only=False):
    Q = len(cloud)
    root_of = {r: k for k, (lv, r, fr) in enumerate(wide)}
    visited = {0: np.ones(Q, bool)}
    node_tasks = 0.0; leaf_tasks = 0.0; leaf_pts = 0.0
    for k, (lv, r, fr) in enumerate(wide):
        m = visited.get(k)
        if m is None or not m.any(): continue
        node_tasks += m.sum()
        idx = np.flatnonzero(m); fr = np.array(fr)
        d = ws.lbdist2(cloud[idx], lo[fr], hi[fr]) <= bound2[idx, None]
        if slab is not None:
            n, a, b = slab
            t = cloud[idx].astype(np.float64) @ n[fr].T          # (Q, B)
            g = np.maximum(0, np.maximum(a[fr][None] - t, t - b[fr][None]))
            ds = (g * g) <= bound2[idx, None]
            if leaf_only: ds = ds | ~isleaf[fr][None]
            d = d & ds
        for ci, nn in enumerate(fr):
            hit = idx[d[:, ci]]
            if isleaf[nn]:
                leaf_tasks += len(hit); leaf_pts += len(hit) * (right[nn] - left[nn])
            else:
                mm = np.zeros(Q, bool); mm[hit] = True
                visited[root_of[nn]] = mm
    return node_tasks / Q, leaf_tasks / Q, leaf_pts / Q
from scipy.spatial import cKDTree
kt = cKDTree(pcd)
poses = synth.hypotheses(8)
wide = ws.build_wide(8)
S = {m: slabs(m) for m in (["pca"] + (["normal", "avgnormal"] if nrm is not None else []))}
for m, (n, a, b) in S.items():
    print(m, "leaf slab thickness mean mm", ((b - a)[isleaf]).mean() * 1e3, " box z-thickness mean mm", ((hi - lo)[isleaf][:, 2]).mean() * 1e3)
for it in (0, 1):
    for pi in (1, 3, 5):
        dep = O.render(ws.tris, poses[pi][None], ws.W, ws.H, ws.proj)[0]
        cloud = O.depth2cloud(dep, ws.K)
        if it: _, _, cloud, _ = O.icp(cloud, sc, (0.0, 0.0, it))
        dnn, _ = kt.query(cloud.astype(np.float64))
        for slack in (1.000001, 1.05):
            b2 = ((dnn * slack) ** 2).astype(np.float32)
            base = simulate(wide, cloud, b2)
            line = f"pass {it} pose {pi} slack {slack}: box only nodes {base[0]:.1f} leaves {base[1]:.1f} pts {base[2]:.0f}"
            for m, sl in S.items():
                r = simulate(wide, cloud, b2, sl); r2 = simulate(wide, cloud, b2, sl, leaf_only=True)
                line += f" | {m}: {r[0]:.1f}/{r[1]:.1f}/{r[2]:.0f} leaf-only {r2[0]:.1f}/{r2[1]:.1f}/{r2[2]:.0f}"
            print(line, flush=True)
