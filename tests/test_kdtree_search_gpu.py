"""GPU tests (-m gpu) of Scene_nn::query (pcd_scene.h:60-136; row a7): every search form -- stackless, per-lane stack, task walk over wide nodes, pixel window -- returns the reference's winner, ties included.
The HIP path is called through the C ABI (pose_refine_amd.api) and held to the CPU oracle on the same inputs: integers, inlier counts and
per-pass sums bit-exact, transforms within 1e-4 (north_star).  @pytest.mark.device_solve = the 6x6 solve runs on the device (the headline
configuration); without it the solve is on the host, as icp.cu:207 does it."""
import ctypes as C
import json
import os
import threading

import numpy as np
import pytest

import oracle_lib as O
from pose_refine_amd import _lib, api, synth
from gpu_common import *  # noqa: F401,F403 -- W, H, TOL_T, inliers, raw_h2d, make_scene, random_mesh ...

pytestmark = pytest.mark.gpu


def test_nn_stack_and_stackless_traversals_agree(gpu, scenario, gscenes):
    """The order-free walk over 128-byte wide nodes / leaf lines (default), the per-lane-stack kd query with compact 32-byte node
    records (child boxes quantised to 16 bits, rounded outwards), the same with exact 64-byte records, and the reference-style
    stackless walk give bit-identical ICP results
    (same winners, same tie-breaks) -- 21 passes on the test.cpp cloud, whose first passes start centimetres off the surface."""
    out = []
    # (1, 1, 1, 1) = default: search kernel + bound kernel + task walk over 128-byte wide nodes; wide 0: binary per-lane walk of the queue;
    # split 0: the search fused into the correspondence pass (what trees without compact records run)
    for stack, compact, wide, split in ((1, 1, 1, 1), (1, 1, 0, 1), (1, 1, 0, 0), (1, 0, 0, 1), (0, 0, 0, 1)):
        api.set_option("nn_stack", stack)
        api.set_option("nn_compact", compact)
        api.set_option("nn_wide", wide)
        api.set_option("nn_split", split)
        dev = api.DeviceVector.from_host(scenario["cloud"].reshape(-1))
        r = api.ICP_Point2Plane(dev, gscenes["nn"], api.ICPConvergenceCriteria(0.0, 0.0, 20))
        out.append((r.transformation_.copy(), r.fitness_, r.inlier_rmse_, dev.to_host()))
    api.set_option("nn_stack", 1)
    api.set_option("nn_compact", 1)
    api.set_option("nn_wide", 1)
    api.set_option("nn_split", 1)
    for o in out[1:]:
        assert np.array_equal(out[0][0], o[0]) and out[0][1] == o[1] and out[0][2] == o[2]
        assert np.array_equal(out[0][3], o[3])


@pytest.mark.parametrize("seed,n,max_leaf", [(1, 4000, 10), (2, 900, 3), (3, 6000, 10), (4, 3000, 24)])    # 24 > 10 points per leaf: no leaf lines, binary walk
def test_nn_variants_agree_on_tie_heavy_clouds(gpu, seed, n, max_leaf):
    """Scene and model points on a coarse lattice with duplicates: many candidate neighbours are at EXACTLY the same distance,
    so the winner is decided by the traversal order alone.  The compact/seeded/near-test stack search, the exact 64-byte
    stack search and the reference-style stackless walk must pick the same neighbours through all passes (bitwise equal
    transforms, scores and transformed clouds), with fixed and with early-exit criteria."""
    import ctypes as C
    from pose_refine_amd import _lib
    rng = np.random.default_rng(seed)
    pts = (np.round(rng.uniform(-0.2, 0.2, size=(n, 3)) * 50) / 50).astype(np.float32)      # 8 mm lattice
    pts = np.concatenate([pts, pts[: n // 10]])                                             # exact duplicates
    nrm = rng.normal(size=pts.shape).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    nodes = np.zeros(2 * len(pts) + 1, _lib.KDNODE); cnt = C.c_uint32()
    _lib.check(_lib.load().pr_kdtree_build(pts.ctypes.data, nrm.ctypes.data, len(pts), max_leaf, nodes.ctypes.data, len(nodes), C.byref(cnt)))
    scene = api.Scene_nn()
    scene.max_dist_diff = 0.1
    scene.pcd_host, scene.normal_host, scene.nodes_host = pts, nrm, np.ascontiguousarray(nodes[:cnt.value])
    scene.pcd_buffer = api.DeviceVector.from_host(pts.reshape(-1))
    scene.normal_buffer = api.DeviceVector.from_host(nrm.reshape(-1))
    scene.nodes = api.DeviceVector.from_host(scene.nodes_host)
    cloud = (np.round(rng.uniform(-0.2, 0.2, size=(3000, 3)) * 100) / 100 + np.float32(0.004)).astype(np.float32)   # 4 mm off the lattice planes
    try:
        for solve in (api.SOLVE_HOST, api.SOLVE_DEVICE):
            api.set_option("solve", solve)
            for crit in ((0.0, 0.0, 12), (1e-5, 1e-5, 30)):
                out = []
                for stack, compact, seeded, wide, split in ((1, 1, 1, 1, 1), (1, 1, 1, 0, 1), (1, 1, 1, 0, 0), (1, 1, 0, 1, 1), (1, 1, 0, 0, 1), (1, 0, 0, 0, 1), (0, 0, 0, 0, 1)):
                    api.set_option("nn_stack", stack); api.set_option("nn_compact", compact); api.set_option("nn_seed", seeded); api.set_option("nn_wide", wide)
                    api.set_option("nn_split", split)
                    dev = api.DeviceVector.from_host(cloud.reshape(-1))
                    r = api.ICP_Point2Plane(dev, scene, api.ICPConvergenceCriteria(*crit))
                    out.append((r.transformation_.copy(), r.fitness_, r.inlier_rmse_, dev.to_host()))
                assert out[-1][1] > 0.5                                                      # the searches do find neighbours
                for o in out[:-1]:
                    assert np.array_equal(out[-1][0], o[0]) and out[-1][1] == o[1] and out[-1][2] == o[2], (solve, crit)
                    assert np.array_equal(out[-1][3], o[3]), (solve, crit)
    finally:
        api.set_option("nn_stack", 1); api.set_option("nn_compact", 1); api.set_option("nn_seed", 1); api.set_option("nn_wide", 1); api.set_option("nn_split", 1)
        api.set_option("solve", api.SOLVE_HOST)


@pytest.mark.device_solve
@pytest.mark.parametrize("n,max_leaf,seed", [(20000, 10, 1), (60000, 10, 2), (3000, 15, 3), (2500, 16, 4), (9, 10, 5), (1, 10, 6)])
def test_task_walk_on_random_scenes_equals_ordered_walks(gpu, n, max_leaf, seed):
    """Random (tie-free) point sets of several sizes -- up to six levels of wide nodes, leaves of up to 15 points (the most a leaf
    reference holds) and of 16 (no wide records: the binary walk runs), a scene that is one leaf, a scene of one point: the task walk,
    the binary per-lane walk and the reference-style stackless walk return bit-identical ICP results, and the first pass' inlier count
    equals a brute-force count in the reference's arithmetic."""
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-0.15, 0.15, size=(n, 3)).astype(np.float32)
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    scene = make_scene(pts, nrm, max_leaf, max_dist=0.02)
    cloud = rng.uniform(-0.16, 0.16, size=(4000, 3)).astype(np.float32)
    m, _, _ = brute_force_first_minimum(cloud[:500], pts, 0.02)
    out = []
    try:
        for stack, wide in ((1, 1), (1, 0), (0, 0)):
            api.set_option("nn_stack", stack); api.set_option("nn_wide", wide)
            dev = api.DeviceVector.from_host(cloud.reshape(-1))
            r = api.ICP_Point2Plane(dev, scene, api.ICPConvergenceCriteria(0.0, 0.0, 4))
            r0 = api.ICP_Point2Plane(api.DeviceVector.from_host(cloud[:500].reshape(-1)), scene, api.ICPConvergenceCriteria(0.0, 0.0, 0))
            out.append((r.transformation_.copy(), r.fitness_, r.inlier_rmse_, dev.to_host(), r0.fitness_))
    finally:
        api.set_option("nn_stack", 1); api.set_option("nn_wide", 1)
    for o in out[1:]:
        assert np.array_equal(out[0][0], o[0]) and out[0][1] == o[1] and out[0][2] == o[2] and np.array_equal(out[0][3], o[3])
    assert round(out[0][4] * 500) == int((m < np.float32(0.02) * np.float32(0.02)).sum())


@pytest.mark.device_solve
def test_graph_cache_keeps_kdtree_batches_of_equal_shape_apart(gpu, scenario, gscenes):
    """two kd-tree batches with the same number of clouds and the same largest cloud but a different LAST cloud lay
    their winner / slack / queue arrays out at different offsets; a captured graph of the first must not be replayed for the second."""
    base = scenario["cloud"]
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 4)                # even max_iteration: the case the advisor describes
    api.set_option("pose_groups", 1)
    try:
        got = {}
        for graph in (1, 0):
            api.set_option("graph", graph)
            res = []
            for last in (2000, 2600, 2000):                       # same P, same max_n (3000), other span
                counts = [1000, 3000, last]
                offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
                cl = np.concatenate([base[:c] for c in counts]).astype(np.float32)
                buf = np.zeros((7000, 3), np.float32); buf[: len(cl)] = cl
                dev = api.DeviceVector.from_host(buf.reshape(-1))
                res.append(api.ICP_Point2Plane_batch(dev, offs, gscenes["nn"], crit).tobytes())
            got[graph] = res
        assert got[1] == got[0]
    finally:
        api.set_option("graph", 1); api.set_option("pose_groups", 0)


@pytest.mark.device_solve
def test_deep_tree_runs_the_24_entry_stacks_in_every_search_form(gpu, scenario, gscenes):
    """A scene of 140 000 points with one point per leaf is 18-19 levels deep: the per-lane stacks take their 24-entry form
    (`icp_pass_kernel<SceneNNDev, true, 24 | 280>`, `nn_tree_kernel<280>`), which no scene of a 640x480 frame reaches.  All searches --
    task walk, binary walk after the search kernel, fused compact and exact stack walks, the stackless walk -- give bit-identical
    results; so does a projective scene whose arrays are used as they are (`icp_pass_kernel<SceneProjAoS>`) against the packed one."""
    rng = np.random.default_rng(77)
    n = 140000
    pts = rng.uniform(-0.2, 0.2, size=(n, 3)).astype(np.float32)
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    scene = make_scene(pts, nrm, 1, max_dist=0.02)
    depth, stack = 0, [(0, 1)]
    nodes = scene.nodes_host
    while stack:                                                  # depth of the tree the build returned
        i, d = stack.pop()
        depth = max(depth, d)
        if nodes["child1"][i] >= 0:
            stack.append((int(nodes["child1"][i]), d + 1)); stack.append((int(nodes["child2"][i]), d + 1))
    assert 16 < depth <= 24
    cloud = rng.uniform(-0.2, 0.2, size=(3000, 3)).astype(np.float32)
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 3)
    names = ("nn_stack", "nn_compact", "nn_wide", "nn_split")
    out = {}
    try:
        for combo in ((1, 1, 1, 1), (1, 1, 0, 1), (1, 1, 0, 0), (1, 0, 0, 0), (0, 0, 0, 0)):
            for k, v in zip(names, combo):
                api.set_option(k, v)
            dev = api.DeviceVector.from_host(cloud.reshape(-1))
            r = api.ICP_Point2Plane(dev, scene, crit)
            out[combo] = (r.transformation_.tobytes(), r.fitness_, r.inlier_rmse_, dev.to_host().tobytes())
        ref = out[(0, 0, 0, 0)]
        assert ref[1] > 0.5
        for combo, o in out.items():
            assert o == ref, combo
        # projective scene: the caller's arrays as they are (no packed records) against the packed scene
        got = []
        for cache in (0, 1):
            api.set_option("scene_cache", cache)
            dev = api.DeviceVector.from_host(scenario["cloud"].reshape(-1))
            r = api.ICP_Point2Plane(dev, gscenes["proj"], api.ICPConvergenceCriteria(0.0, 0.0, 6))
            got.append((r.transformation_.tobytes(), r.fitness_, r.inlier_rmse_, dev.to_host().tobytes()))
        assert got[0] == got[1] and got[0][1] > 0.5
    finally:
        for k, v in zip(names, (1, 1, 1, 1)):
            api.set_option(k, v)
        api.set_option("scene_cache", 1)


@pytest.mark.parametrize("W,H", [(1024, 768), (2048, 1536)])
def test_large_kdtree_scenes(gpu, W, H):
    """kd-tree scenes of 240 k and 900 k points (a frame of 0.8 / 3 M pixels): the tree built on the host, the one built on the device and the
    oracle's are the same; a 200 k-point cloud refined against them equals the oracle's result; the fused path takes the scene as well."""
    rng = np.random.default_rng(W)
    f = 0.9 * W
    K = np.array([f, 0, W / 2, 0, f, H / 2, 0, 0, 1], np.float32)
    tris = random_mesh(rng, 300, 40.0)
    poses = np.stack([random_pose(rng, 160.0)] * 2)
    poses[1] = poses[0]; poses[1][0, 3] += 0.8; poses[1][2, 3] += 1.5
    proj = O.compute_proj(K, W, H)
    ref = O.render(tris, poses, W, H, proj)
    scene = api.Scene_nn().init_Scene_nn_cuda(ref[0], K)
    oscene = O.NNScene(ref[0], K)
    assert len(scene.pcd_host) == len(oscene.pcd) > 200000 and scene.nodes_host.tobytes() == oscene.nodes.tobytes()
    dscene = api.Scene_nn().init_Scene_nn_device(api.DeviceVector.from_host(ref[0].reshape(-1)), K, W, H)
    assert (dscene._n_points, dscene._n_nodes) == (len(oscene.pcd), len(oscene.nodes))
    assert dscene.nodes.to_host()[:len(oscene.nodes)].tobytes() == oscene.nodes.tobytes()
    crit = (0.0, 0.0, 2)
    cl = O.depth2cloud(ref[1], K)[::7][:200000]
    want, _, _, _ = O.icp(cl, oscene, crit, O.SUM_CANONICAL, api.get_option("points_per_block"))
    for sc in (scene, dscene):
        r = api.ICP_Point2Plane(api.DeviceVector.from_host(cl.reshape(-1)), sc, api.ICPConvergenceCriteria(*crit))
        assert r.fitness_ == want["fitness"] and np.allclose(r.transformation_.reshape(-1), want["T"], rtol=0, atol=1e-4)
    model = api.Model(tris=tris)
    res, sizes = api.refine_batch(model, poses, W, H, proj, K, scene, api.ICPConvergenceCriteria(*crit))
    assert [int(s) for s in sizes] == [int((x > 0).sum()) for x in ref] and res["fitness"][0] == 1.0

