"""GPU tests (-m gpu) of init_Scene_projective_* / init_Scene_nn_* / KDTree build on the device (depth_scene.cpp, pcd_scene.cpp:10-184; rows a8-a10, f1) and what the library derives from caller-owned scene arrays.
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


# ---- SURVEY 8f "next" rows: device scene preparation, raw2* conversions ---------------------------------
@pytest.mark.parametrize("dtype", [np.int32, np.uint16])
def test_device_scene_preparation_bit_exact(gpu, scenario, dtype):
    d = scenario["depth"][1].astype(dtype)
    d[100:120, 300:330] = 2500 if dtype == np.uint16 else 70000        # beyond the 2000 mm gate / uint16 saturation
    host = api.Scene_projective().init_Scene_projective_cuda(d, scenario["K"])
    dev = api.Scene_projective().init_Scene_projective_device(api.DeviceVector.from_host(d.reshape(-1)), scenario["K"])
    assert np.array_equal(dev.pcd_buffer.to_host(), host.pcd_host.reshape(-1))
    assert np.array_equal(dev.normal_buffer.to_host(), host.normal_host.reshape(-1))
    ref = O.ProjScene(d, scenario["K"])
    assert np.array_equal(dev.normal_buffer.to_host().reshape(-1, 3), ref.normal)


@pytest.mark.parametrize("dtype", [np.int32, np.uint16])
def test_device_nn_scene_preparation_bit_exact(gpu, scenario, dtype):
    """Device normals + gather + level-order kd-tree build == the CPU preparation, bit for bit (nodes, point order, normals)."""
    d = scenario["depth"][1].astype(dtype)
    host = api.Scene_nn().init_Scene_nn_cuda(d, scenario["K"])
    dev = api.Scene_nn().init_Scene_nn_device(api.DeviceVector.from_host(d.reshape(-1)), scenario["K"], W, H)
    n, m = len(host.pcd_host), len(host.nodes_host)
    assert (dev._n_points, dev._n_nodes) == (n, m)
    assert np.array_equal(dev.pcd_buffer.to_host()[:3 * n], host.pcd_host.reshape(-1))
    assert np.array_equal(dev.normal_buffer.to_host()[:3 * n], host.normal_host.reshape(-1))
    assert dev.nodes.to_host()[:m].tobytes() == host.nodes_host.tobytes()
    # ... and, directly, == the ORACLE's preparation (pcd_scene.cpp:10-184 restated in oracle/pose_oracle.c): points, normals, nodes
    ref = O.NNScene(d, scenario["K"])
    assert (n, m) == (len(ref.pcd), len(ref.nodes))
    assert np.array_equal(dev.pcd_buffer.to_host()[:3 * n], ref.pcd.reshape(-1))
    assert np.array_equal(dev.normal_buffer.to_host()[:3 * n], ref.normal.reshape(-1))
    assert dev.nodes.to_host()[:m].tobytes() == ref.nodes.tobytes()
    # and ICP against the device-built scene gives the same answer
    r0 = api.ICP_Point2Plane(api.DeviceVector.from_host(scenario["cloud"].reshape(-1)), host, api.ICPConvergenceCriteria(0.0, 0.0, 5))
    r1 = api.ICP_Point2Plane(api.DeviceVector.from_host(scenario["cloud"].reshape(-1)), dev, api.ICPConvergenceCriteria(0.0, 0.0, 5))
    assert np.array_equal(r0.transformation_, r1.transformation_) and r0.fitness_ == r1.fitness_


@pytest.mark.parametrize("max_leaf,n", [(1, 700), (3, 700), (10, 5000), (64, 300), (10, 7)])
def test_device_kdtree_build_random_points_with_ties(gpu, max_leaf, n):
    import ctypes as C
    from pose_refine_amd import _lib
    rng = np.random.default_rng(n + max_leaf)
    pts = np.round(rng.normal(size=(n, 3)), 1).astype(np.float32)          # coarse grid -> many equal coordinates (tie rule)
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    hp, hn = pts.copy(), nrm.copy()
    hnodes = np.zeros(2 * n + 1, _lib.KDNODE); cnt = C.c_uint32()
    _lib.check(_lib.load().pr_kdtree_build(hp.ctypes.data, hn.ctypes.data, n, max_leaf, hnodes.ctypes.data, len(hnodes), C.byref(cnt)))
    dp, dn = api.DeviceVector.from_host(pts.reshape(-1)), api.DeviceVector.from_host(nrm.reshape(-1))
    dnodes = api.DeviceVector(2 * n + 1, _lib.KDNODE); dcnt = C.c_uint32()
    _lib.check(_lib.load().pr_kdtree_build_dev(dp.data(), dn.data(), n, max_leaf, dnodes.data(), 2 * n + 1, C.byref(dcnt)))
    assert dcnt.value == cnt.value
    assert dnodes.to_host()[:cnt.value].tobytes() == hnodes[:cnt.value].tobytes()
    assert np.array_equal(dp.to_host(), hp.reshape(-1)) and np.array_equal(dn.to_host(), hn.reshape(-1))
    # the device build against the ORACLE's build_tree (pcd_scene.cpp:45-184), not only against the library's own host build
    op, on = pts.copy(), nrm.copy()
    onodes = np.zeros(2 * n + 1, O.KDNODE)
    ocnt = O.lib().po_kd_build(op.reshape(-1), on.reshape(-1), n, max_leaf, onodes.ctypes.data, len(onodes))
    assert dcnt.value == ocnt and dnodes.to_host()[:ocnt].tobytes() == onodes[:ocnt].tobytes()
    assert np.array_equal(dp.to_host(), op.reshape(-1)) and np.array_equal(dn.to_host(), on.reshape(-1))


@pytest.mark.parametrize("kind,n,max_leaf", [("grid", 40000, 10), ("zeros", 30000, 10), ("fine", 20000, 1), ("fine", 70000, 2), ("planes", 50000, 10)])
def test_device_kdtree_build_whole_levels_at_a_time(gpu, kind, n, max_leaf):
    """The level-order build's passes over whole levels (kd_build.hip): point sets that span dozens of tiles -- a coarse grid (every cut is hit by
    hundreds of points: the alternating tie rule across tile borders), coordinates that are zeros of BOTH signs (the stored extremes keep the sign of the
    first occurrence), leaves of one or two points (more nodes of a level inside a tile than its key table holds: the straight-to-memory path), and
    depth-quantised planes -- against the oracle's build_tree: same nodes, bit for bit, same permutation."""
    import ctypes as C
    from pose_refine_amd import _lib
    rng = np.random.default_rng(n + max_leaf)
    if kind == "grid": pts = np.round(rng.normal(size=(n, 3)) * 3.0, 0).astype(np.float32) / 8
    elif kind == "zeros":
        pts = rng.normal(size=(n, 3)).astype(np.float32)
        z = rng.random((n, 3)) < 0.4
        pts[z] = np.where(rng.random(int(z.sum())) < 0.5, np.float32(0.0), np.float32(-0.0))
    elif kind == "fine": pts = rng.normal(size=(n, 3)).astype(np.float32)
    else:
        u, v = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
        z = np.round((1.0 + 0.1 * u + 0.05 * v) * 1000.0) / 1000.0
        pts = np.stack([u * z, v * z, z], 1).astype(np.float32)
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    op, on = pts.copy(), nrm.copy()
    onodes = np.zeros(2 * n + 1, O.KDNODE)
    ocnt = O.lib().po_kd_build(op.reshape(-1), on.reshape(-1), n, max_leaf, onodes.ctypes.data, len(onodes))
    assert ocnt > 0
    dp, dn = api.DeviceVector.from_host(pts.reshape(-1)), api.DeviceVector.from_host(nrm.reshape(-1))
    dnodes = api.DeviceVector(2 * n + 1, _lib.KDNODE); dcnt = C.c_uint32()
    _lib.check(_lib.load().pr_kdtree_build_dev(dp.data(), dn.data(), n, max_leaf, dnodes.data(), 2 * n + 1, C.byref(dcnt)))
    assert dcnt.value == ocnt
    assert dnodes.to_host()[:ocnt].tobytes() == onodes[:ocnt].tobytes()
    assert dp.to_host().tobytes() == op.tobytes() and dn.to_host().tobytes() == on.tobytes()          # (bytes: -0.0 and 0.0 are different points here)
    # a node array that is too small is an error, as in the reference's restatement (po_kd_build returns 0), not a truncated tree
    small = api.DeviceVector(ocnt - 2, _lib.KDNODE)
    dp2, dn2 = api.DeviceVector.from_host(pts.reshape(-1)), api.DeviceVector.from_host(nrm.reshape(-1))
    assert _lib.load().pr_kdtree_build_dev(dp2.data(), dn2.data(), n, max_leaf, small.data(), ocnt - 2, C.byref(dcnt)) != 0
    assert O.lib().po_kd_build(pts.copy().reshape(-1), nrm.copy().reshape(-1), n, max_leaf, np.zeros(ocnt - 2, O.KDNODE).ctypes.data, ocnt - 2) == 0


@pytest.mark.parametrize("W,H", [(640, 480), (1280, 720)])
def test_device_nn_scene_of_a_frame_filling_image(gpu, W, H):
    """Every pixel valid (a curved wall behind nothing), 640 x 480 and BASELINE configs[4]'s 1280 x 720: gather, normals, the 300 / 900-tile build and the wide
    records derived from it, against the oracle."""
    yy, xx = np.mgrid[0:H, 0:W]
    depth = (900 + 0.05 * xx + 0.03 * yy + 3.0 * np.sin(xx / 17.0) * np.cos(yy / 23.0)).astype(np.int32)
    K = synth.K_TEST if W == 640 else np.array([W * 0.9, 0, W / 2, 0, W * 0.9, H / 2, 0, 0, 1], np.float32)
    dev = api.Scene_nn().init_Scene_nn_device(api.DeviceVector.from_host(depth.reshape(-1)), K, W, H)
    ref = O.NNScene(depth, K)
    npt, nn = len(ref.pcd), len(ref.nodes)
    assert (dev._n_points, dev._n_nodes) == (npt, nn) == (W * H, nn)
    assert np.array_equal(dev.pcd_buffer.to_host()[:3 * npt].reshape(-1, 3), ref.pcd) and np.array_equal(dev.normal_buffer.to_host()[:3 * npt].reshape(-1, 3), ref.normal)
    assert dev.nodes.to_host()[:nn].tobytes() == ref.nodes.tobytes()
    # and a search against it: a cloud of the wall's own points pushed 4 mm towards the camera
    cloud = (ref.pcd[::37 if W == 640 else 151] * np.float32(0.996)).astype(np.float32)
    got = api.ICP_Point2Plane(api.DeviceVector.from_host(cloud.reshape(-1)), dev, api.ICPConvergenceCriteria(0.0, 0.0, 3))
    want, _, _, _ = O.icp(cloud, ref, (0.0, 0.0, 3), O.SUM_CANONICAL, api.get_option("points_per_block"))
    assert got.fitness_ == float(want["fitness"]) and np.allclose(got.transformation_.reshape(-1), want["T"], rtol=0, atol=TOL_T)


# ---- cropped projective scene: pcd2dep / dep2pcd with tl_x, tl_y (common.h:47-73) ------------------------------------------
@pytest.mark.device_solve
def test_cropped_scene_lookup_with_offsets(gpu, model, scenario, gscenes):
    d = scenario["depth"][1]
    ys, xs = np.nonzero(d)
    tight = (int(xs.min()), int(ys.min()), int(xs.max() - xs.min() + 1), int(ys.max() - ys.min() + 1))
    cut = (tight[0] + 40, tight[1] + 30, tight[2] - 70, tight[3] - 50)          # loses part of the scene object
    poses = synth.hypotheses(8)
    crit = (0.0, 0.0, 6)
    ppb = api.get_option("points_per_block")
    full, fsizes = api.refine_batch(model, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], api.ICPConvergenceCriteria(*crit))
    for window, same_as_full in ((tight, True), (cut, False)):
        gs = gscenes["proj"].crop(window)
        res, sizes = api.refine_batch(model, poses, W, H, scenario["proj"], scenario["K"], gs, api.ICPConvergenceCriteria(*crit))
        ores, osizes, _ = O.refine_batch(scenario["tris"], poses, W, H, scenario["proj"], scenario["K"],
                                         scenario["proj_scene"].crop(window), crit, O.SUM_CANONICAL, ppb)
        assert np.array_equal(sizes, osizes) and np.array_equal(res["fitness"], ores["fitness"])
        assert np.allclose(res["T"], ores["T"], rtol=0, atol=TOL_T)
        if same_as_full:
            # every valid scene pixel is inside the window.  Not bit-identical to the full frame: pcd2dep evaluates
            # x/z*fx + cx - tl_x + 0.5f in float, so a projection that lands within an ulp of a pixel boundary may round to the
            # other side once tl_x is subtracted -- which is why parity is taken against the oracle's cropped scene above
            assert np.allclose(res["T"], full["T"], rtol=0, atol=1e-3) and np.allclose(res["fitness"], full["fitness"], atol=2e-3)
        else:
            assert not np.allclose(res["T"], full["T"], rtol=0, atol=1e-5)
        # the un-packed form of the same scene (ICP on clouds goes through the caller's arrays)
        cl = O.depth2cloud(O.render(scenario["tris"], poses[:1], W, H, scenario["proj"])[0], scenario["K"])
        dev = api.DeviceVector.from_host(cl.reshape(-1))
        one = api.ICP_Point2Plane(dev, gs, api.ICPConvergenceCriteria(*crit))
        assert one.fitness_ == res["fitness"][0] and np.array_equal(one.transformation_.reshape(-1), res["T"][0])


@pytest.mark.device_solve
def test_scene_arrays_that_dep2pcd_did_not_produce_are_used_as_they_are(gpu, model, scenario, gscenes):
    """the packed scene rebuilds pcd.x / pcd.y from z.  pcd / normal are caller-owned (public members in the
    reference), so a buffer with other x / y must make the fused path use the arrays themselves -- like pr_icp_batch does."""
    K = scenario["K"]
    s = api.Scene_projective().init_Scene_projective_cuda(scenario["depth"][1], K)
    pcd = s.pcd_host.copy()
    pcd[:, 0] += np.float32(0.004) * (pcd[:, 2] > 0)              # shifted 4 mm in x: no longer dep2pcd's output
    api.check(_lib.load().pr_memcpy_h2d(s.pcd_buffer.data(), pcd.ctypes.data, pcd.nbytes))
    poses = synth.hypotheses(6)
    crit = (0.0, 0.0, 5)
    res, sizes = api.refine_batch(model, poses, W, H, scenario["proj"], K, s, api.ICPConvergenceCriteria(*crit))
    osc = O.ProjScene(scenario["depth"][1], K)
    osc.pcd[:] = pcd
    ores, osizes, _ = O.refine_batch(scenario["tris"], poses, W, H, scenario["proj"], K, osc, crit, O.SUM_CANONICAL,
                                     api.get_option("points_per_block"))
    assert np.array_equal(sizes, osizes) and np.array_equal(res["fitness"], ores["fitness"])
    assert np.allclose(res["T"], ores["T"], rtol=0, atol=TOL_T)
    ref, _ = api.refine_batch(model, poses, W, H, scenario["proj"], K, gscenes["proj"], api.ICPConvergenceCriteria(*crit))
    assert res.tobytes() != ref.tobytes()


# ---- kd-tree input validation --------------------------------------------------------------------------------
@pytest.mark.device_solve
def test_kdtree_that_is_not_a_tree_is_rejected_and_odd_trees_still_match(gpu, scenario):
    s = api.Scene_nn().init_Scene_nn_cuda(scenario["depth"][1], scenario["K"])
    cloud = scenario["cloud"][:4096]
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 2)
    good = api.ICP_Point2Plane(api.DeviceVector.from_host(cloud.reshape(-1)), s, crit)
    nodes = s.nodes_host.copy()
    # (a) a child link that points outside the array
    bad = nodes.copy(); bad["child2"][0] = len(bad) + 5
    s.nodes = api.DeviceVector.from_host(bad)
    with pytest.raises(api.PoseRefineError):
        api.ICP_Point2Plane(api.DeviceVector.from_host(cloud.reshape(-1)), s, crit)
    # (b) parent links that disagree with the child links
    bad = nodes.copy(); bad["parent"][bad["child1"][0]] = 3
    s.nodes = api.DeviceVector.from_host(bad)
    with pytest.raises(api.PoseRefineError):
        api.ICP_Point2Plane(api.DeviceVector.from_host(cloud.reshape(-1)), s, crit)
    # (c) a legal tree whose split values are NOT between the children (the compact 8-byte descent may not be used):
    #     nudging a split value inside the gap keeps left_max <= split <= right_min, nudging it outside must fall back to the
    #     exact records -- results equal the oracle's walk over the same nodes either way
    odd = nodes.copy()
    internal = np.nonzero(odd["child1"] >= 0)[0]
    odd["split_v"][internal[::3]] += np.float32(0.02)             # 2 cm: beyond many right_min values
    s.nodes = api.DeviceVector.from_host(odd)
    got = api.ICP_Point2Plane(api.DeviceVector.from_host(cloud.reshape(-1)), s, crit)
    osc = O.NNScene(scenario["depth"][1], scenario["K"])
    osc.nodes[:] = odd
    ref, _, _, _ = O.icp(cloud, osc, (0.0, 0.0, 2), O.SUM_CANONICAL, api.get_option("points_per_block"))
    assert got.fitness_ == float(ref["fitness"]) and np.allclose(got.transformation_.reshape(-1), ref["T"], rtol=0, atol=TOL_T)
    s.nodes = api.DeviceVector.from_host(nodes)
    again = api.ICP_Point2Plane(api.DeviceVector.from_host(cloud.reshape(-1)), s, crit)
    assert again.fitness_ == good.fitness_ and np.array_equal(again.transformation_, good.transformation_)


@pytest.mark.device_solve
@pytest.mark.parametrize("name", ["identical", "line", "clusters", "one", "two", "eleven"])
@pytest.mark.parametrize("max_leaf", [1, 10])
def test_kdtree_build_on_degenerate_point_sets(gpu, name, max_leaf):
    """A thousand identical points, points on a line, two clusters of exact duplicates, one / two / eleven points: the host build, the
    device build and the oracle's give the same nodes and the same permutation (the alternating tie rule of pcd_scene.cpp decides)."""
    rng = np.random.default_rng(3)
    pts = {"identical": np.tile(np.array([[0.1, 0.2, 0.7]], np.float32), (1000, 1)),
           "line": np.stack([np.linspace(0, 1, 777, dtype=np.float32), np.zeros(777, np.float32), np.full(777, 0.5, np.float32)], 1),
           "clusters": np.concatenate([np.tile(np.array([[0, 0, 1]], np.float32), (300, 1)), np.tile(np.array([[1, 1, 1]], np.float32), (301, 1))]),
           "one": np.array([[0.5, 0.5, 0.5]], np.float32), "two": np.array([[0.5, 0.5, 0.5], [0.1, 0.1, 0.1]], np.float32),
           "eleven": rng.normal(size=(11, 3)).astype(np.float32)}[name]
    n = len(pts)
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    lib = _lib.load()
    hp, hn = pts.copy(), nrm.copy()
    hnodes = np.zeros(2 * n + 1, _lib.KDNODE); cnt = C.c_uint32()
    _lib.check(lib.pr_kdtree_build(hp.ctypes.data, hn.ctypes.data, n, max_leaf, hnodes.ctypes.data, len(hnodes), C.byref(cnt)))
    op, on = pts.copy(), nrm.copy()
    onodes = np.zeros(2 * n + 1, O.KDNODE)
    ocnt = O.lib().po_kd_build(op.reshape(-1), on.reshape(-1), n, max_leaf, onodes.ctypes.data, len(onodes))
    assert ocnt == cnt.value and onodes[:ocnt].tobytes() == hnodes[:ocnt].tobytes() and np.array_equal(op, hp) and np.array_equal(on, hn)
    dp, dn = api.DeviceVector.from_host(pts.reshape(-1)), api.DeviceVector.from_host(nrm.reshape(-1))
    dnodes = api.DeviceVector(2 * n + 1, _lib.KDNODE); dcnt = C.c_uint32()
    _lib.check(lib.pr_kdtree_build_dev(dp.data(), dn.data(), n, max_leaf, dnodes.data(), 2 * n + 1, C.byref(dcnt)))
    assert dcnt.value == cnt.value and dnodes.to_host()[:cnt.value].tobytes() == hnodes[:cnt.value].tobytes()
    assert np.array_equal(dp.to_host().reshape(-1, 3), hp) and np.array_equal(dn.to_host().reshape(-1, 3), hn)
