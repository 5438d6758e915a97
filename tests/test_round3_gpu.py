"""GPU tests (-m gpu) of round 3: the task walk over wide kd-tree records (pcd_scene.h:60-136 semantics kept through ties), the graph
cache key of kd-tree batches, synchronous calls next to a pending slot, pr_free with work in flight."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
from pose_refine_amd import _lib, api, synth

pytestmark = pytest.mark.gpu

W, H = synth.WIDTH, synth.HEIGHT


@pytest.fixture(scope="module")
def gpu():
    api.init(0)
    api.set_option("solve", api.SOLVE_DEVICE)
    yield True
    api.set_option("solve", api.SOLVE_HOST)


@pytest.fixture(scope="module")
def model(gpu, golden_dir):
    return api.Model(os.path.join(golden_dir, "obj_06.ply"))


@pytest.fixture(scope="module")
def gscenes(gpu, scenario):
    d = scenario["depth"][1]
    return dict(proj=api.Scene_projective().init_Scene_projective_cuda(d, scenario["K"]),
                nn=api.Scene_nn().init_Scene_nn_cuda(d, scenario["K"]))


def make_scene(pts, nrm, max_leaf, max_dist=0.1):
    nodes = np.zeros(2 * len(pts) + 1, _lib.KDNODE); cnt = C.c_uint32()
    _lib.check(_lib.load().pr_kdtree_build(pts.ctypes.data, nrm.ctypes.data, len(pts), max_leaf, nodes.ctypes.data, len(nodes), C.byref(cnt)))
    scene = api.Scene_nn()
    scene.max_dist_diff = max_dist
    scene.pcd_host, scene.normal_host, scene.nodes_host = pts, nrm, np.ascontiguousarray(nodes[:cnt.value])
    scene.pcd_buffer = api.DeviceVector.from_host(pts.reshape(-1))
    scene.normal_buffer = api.DeviceVector.from_host(nrm.reshape(-1))
    scene.nodes = api.DeviceVector.from_host(scene.nodes_host)
    return scene


def brute_force_first_minimum(cloud, pts, max_dist):
    """Lowest distance per query in the reference's float arithmetic (pcd_scene.h:88-91) and whether it is attained once."""
    d = cloud[:, None, :].astype(np.float32) - pts[None, :, :].astype(np.float32)
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    m = d2.min(1)
    return m, (d2 == m[:, None]).sum(1), d2.argmin(1)


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


def test_synchronous_call_while_slot_0_is_pending(gpu, model, scenario, gscenes):
    """pr_refine_batch used to be submit(0) + wait(0) and failed with a batch pending on slot 0; it now takes the free slot,
    and reports PR_ERR_INVALID only when both slots are taken."""
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 5)
    a, b = synth.hypotheses(40, first=0), synth.hypotheses(24, first=40)
    want_a = api.refine_batch(model, a, W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit)
    want_b = api.refine_batch(model, b, W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit)
    api.refine_submit(0, model, a, W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit)
    got_b = api.refine_batch(model, b, W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit)       # runs on slot 1
    api.refine_submit(1, model, b, W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit)
    with pytest.raises(api.PoseRefineError):
        api.refine_batch(model, b, W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit)
    got_a = api.refine_wait(0)
    got_b2 = api.refine_wait(1)
    assert got_a[0].tobytes() == want_a[0].tobytes() and got_b[0].tobytes() == want_b[0].tobytes() and got_b2[0].tobytes() == want_b[0].tobytes()


def test_free_with_a_batch_in_flight(gpu, model, scenario, gscenes):
    """pr_free waits for everything the device is running: freeing the device-side results buffer of a batch that was only just
    submitted returns after that batch has finished (the wait that follows finds it done), and the library stays usable."""
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 20)
    poses = synth.hypotheses(64)
    want = api.refine_batch(model, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit)
    dev = api.DeviceVector(64 * 18, np.float32)
    api.refine_submit(0, model, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit, results_dev=dev.data())
    del dev                                                       # pr_free: drains the device first
    api.refine_wait(0)
    again = api.refine_batch(model, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit)
    assert again[0].tobytes() == want[0].tobytes()


def test_raster_repeated_64_hypotheses_is_deterministic(gpu, model, scenario):
    """The depth resolve is an integer atomicMin: twenty renders of the same 64 hypotheses give the same bits (and the oracle's)."""
    poses = synth.hypotheses(64)
    ref = O.render(scenario["tris"], poses[:3], W, H, scenario["proj"])
    imgs = [np.asarray(api.render_host(model, poses, W, H, scenario["proj"])) for _ in range(20)]
    for im in imgs[1:]:
        assert np.array_equal(im, imgs[0])
    assert np.array_equal(imgs[0][:3].reshape(3, -1), ref.reshape(3, -1))


def raw_d2d(dst_dev: int, src_dev: int, nbytes: int):
    """A write the library cannot see: the HIP runtime called directly."""
    hip = C.CDLL("libamdhip64.so.7")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(dst_dev, src_dev, nbytes, 3) == 0
    assert hip.hipDeviceSynchronize() == 0


def test_scene_rewritten_behind_the_librarys_back_is_noticed(gpu, model, scenario):
    """the packed projective scene (like the kd-tree search records) is cached by the address of the caller's arrays.  A frame
    that is replaced as a whole through a raw hipMemcpy (no pr_invalidate) changes the sampled fingerprint every asynchronous batch takes
    of those arrays: the batch is repeated with fresh caches and returns what a new scene object returns.  (A kd-tree scene of another
    frame has other point and node counts, which are part of its cache key.)"""
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 6)
    poses = synth.hypotheses(48)
    K = scenario["K"]
    a = api.Scene_projective().init_Scene_projective_cuda(scenario["depth"][1], K)
    b = api.Scene_projective().init_Scene_projective_cuda(scenario["depth"][0], K)       # another frame of the same size
    want_b = api.refine_batch(model, poses, W, H, scenario["proj"], K, b, crit)
    first = api.refine_batch(model, poses, W, H, scenario["proj"], K, a, crit)      # the caches of `a` are built here
    assert first[0].tobytes() != want_b[0].tobytes()
    raw_d2d(a.normal_buffer.data(), b.normal_buffer.data(), a.normal_buffer.size() * 4)
    raw_d2d(a.pcd_buffer.data(), b.pcd_buffer.data(), a.pcd_buffer.size() * 4)
    got = api.refine_batch(model, poses, W, H, scenario["proj"], K, a, crit)
    assert got[0].tobytes() == want_b[0].tobytes() and np.array_equal(got[1], want_b[1])


def test_kdtree_scene_rewritten_behind_the_librarys_back_is_noticed_by_a_bare_icp_call(gpu, scenario):
    """The same for the kd-tree search records and a synchronous call (pr_icp_nn checks a cache hit on the spot): points and nodes of a
    scene are replaced by those of a shifted copy with the same counts, through raw copies."""
    rng = np.random.default_rng(11)
    pts = rng.uniform(-0.1, 0.1, size=(5000, 3)).astype(np.float32)
    nrm = rng.normal(size=(5000, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    a = make_scene(pts.copy(), nrm.copy(), 10)
    b = make_scene((pts + np.float32(0.003)).astype(np.float32), nrm.copy(), 10)
    if len(a.nodes_host) != len(b.nodes_host):
        pytest.skip("the shifted copy built a tree of another size")
    cloud = rng.uniform(-0.1, 0.1, size=(3000, 3)).astype(np.float32)
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 5)
    run = lambda sc: api.ICP_Point2Plane(api.DeviceVector.from_host(cloud.reshape(-1)), sc, crit)
    want_b, first = run(b), run(a)                                # the search records of `b`, then of `a`, are built here
    assert not np.array_equal(first.transformation_, want_b.transformation_)
    for dst, src in ((a.pcd_buffer, b.pcd_buffer), (a.normal_buffer, b.normal_buffer)):
        raw_d2d(dst.data(), src.data(), dst.size() * 4)
    raw_d2d(a.nodes.data(), b.nodes.data(), len(a.nodes_host) * 52)
    got = run(a)
    assert np.array_equal(got.transformation_, want_b.transformation_) and got.fitness_ == want_b.fitness_


def test_deep_tree_runs_the_24_entry_stacks_and_every_dataflow_form(gpu, scenario, gscenes):
    """A scene of 140 000 points with one point per leaf is 18-19 levels deep: the per-lane stacks take their 24-entry form
    (`icp_pass_kernel<SceneNNDev, true, 24 | 280>`, `nn_tree_kernel<280>`, `icp_flow_kernel<SceneNNDev, true, 24>`), which no scene of
    a 640x480 frame reaches.  All searches -- task walk, binary walk after the search kernel, fused compact and exact stack walks, the
    stackless walk, each as launch-per-pass loop and as the persistent dataflow kernel -- give bit-identical results; so does the
    dataflow kernel on a projective scene whose arrays are used as they are (`icp_flow_kernel<SceneProjAoS>`)."""
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
    names = ("nn_stack", "nn_compact", "nn_wide", "nn_split", "icp_flow")
    out = {}
    try:
        for combo in ((1, 1, 1, 1, 0), (1, 1, 0, 1, 0), (1, 1, 0, 0, 0), (1, 0, 0, 0, 0), (0, 0, 0, 0, 0), (1, 1, 0, 0, 1), (1, 0, 0, 0, 1), (0, 0, 0, 0, 1)):
            for k, v in zip(names, combo):
                api.set_option(k, v)
            dev = api.DeviceVector.from_host(cloud.reshape(-1))
            r = api.ICP_Point2Plane(dev, scene, crit)
            out[combo] = (r.transformation_.tobytes(), r.fitness_, r.inlier_rmse_, dev.to_host().tobytes())
        ref = out[(0, 0, 0, 0, 0)]
        assert ref[1] > 0.5
        for combo, o in out.items():
            assert o == ref, combo
        # projective scene without the packed records, dataflow against launch-per-pass
        api.set_option("scene_cache", 0)
        got = []
        for flow in (0, 1):
            api.set_option("icp_flow", flow)
            dev = api.DeviceVector.from_host(scenario["cloud"].reshape(-1))
            r = api.ICP_Point2Plane(dev, gscenes["proj"], api.ICPConvergenceCriteria(0.0, 0.0, 6))
            got.append((r.transformation_.tobytes(), r.fitness_, r.inlier_rmse_, dev.to_host().tobytes()))
        assert got[0] == got[1] and got[0][1] > 0.5
    finally:
        for k, v in zip(names, (1, 1, 1, 1, 0)):
            api.set_option(k, v)
        api.set_option("scene_cache", 1)


def _pathological(base):
    p = base.copy().reshape(-1, 4, 4)
    p[1, 0, 0] = np.nan
    p[3, 2, 3] = np.inf
    p[5] = 0.0
    p[7, :3, 3] = [1e30, -1e30, 1e30]
    p[9, :3, :3] *= -1.0                      # mirrored
    p[11, 2, 3] = 1e-6                        # the camera inside the object
    p[13, 2, 3] = -700.0                      # the object behind the camera
    p[15, :3, :3] *= 1e6                      # a giant
    p[17, :3, :3] *= 1e-9                     # a speck
    p[19, 0, 3] = -np.inf
    p[21, :3, :3] = np.nan
    p[23, 3, :] = [1, 2, 3, 4]                # a last row that is not 0 0 0 1 (the renderer never reads it)
    return p.reshape(base.shape), [1, 3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23]


def test_pathological_hypotheses_do_no_harm(gpu, model, scenario, gscenes):
    """Hypotheses with NaN / infinite / zero / mirrored / astronomically scaled matrices in a batch: nothing faults, the other hypotheses
    of the batch are refined bit for bit as without them (synchronous path, both solves, both scenes, and on the asynchronous slots), and
    the render of the finite oddities equals the oracle's (a mirrored object, one behind the camera, a speck, a foreign last row)."""
    good = synth.hypotheses(40, seed=3)
    bad, idx_bad = _pathological(good)
    idx_ok = [i for i in range(40) if i not in idx_bad]
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 6)
    try:
        for solve in (api.SOLVE_DEVICE, api.SOLVE_HOST):
            api.set_option("solve", solve)
            for kind in ("proj", "nn"):
                ref = api.refine_batch(model, good, W, H, scenario["proj"], scenario["K"], gscenes[kind], crit)
                out = api.refine_batch(model, bad, W, H, scenario["proj"], scenario["K"], gscenes[kind], crit)
                assert all(ref[0][i].tobytes() == out[0][i].tobytes() and ref[1][i] == out[1][i] for i in idx_ok), (solve, kind)
                if solve == api.SOLVE_DEVICE:
                    for b in (0, 1):
                        api.refine_submit(b, model, bad, W, H, scenario["proj"], scenario["K"], gscenes[kind], crit)
                    for b in (0, 1):
                        o2 = api.refine_wait(b)
                        assert all(o2[0][i].tobytes() == ref[0][i].tobytes() for i in idx_ok), (kind, b)
    finally:
        api.set_option("solve", api.SOLVE_DEVICE)
    finite = [9, 13, 17, 23]
    got = api.render_host(model, bad[finite], W, H, scenario["proj"])
    want = O.render(scenario["tris"], bad[finite], W, H, scenario["proj"])
    assert np.array_equal(got, want)
    assert (got[0] > 0).sum() > 20000 and (got[3] > 0).sum() > 20000      # the mirrored object and the foreign last row do render


@pytest.mark.parametrize("kind", ["proj", "nn"])
def test_cloud_with_non_finite_and_absurd_points_matches_the_oracle(gpu, scenario, gscenes, kind):
    """400 of the cloud's points replaced by NaN, infinities, 1e30, zero depth, negative depth, the origin and denormal-small values: the
    reference's rules decide each of them (a projection that is NaN or out of range is rejected, common.h:63-73 / depth_scene.h:38-45; a
    kd-tree query that finds nothing within max_dist is no correspondence) -- same inlier counts and transforms as the CPU oracle, per
    cloud and inside a ragged batch, nothing faults."""
    cloud = scenario["cloud"]
    bad = cloud.copy()
    ix = np.random.default_rng(0).choice(len(bad), 400, replace=False)
    bad[ix[:50]] = np.nan
    bad[ix[50:100], 2] = np.inf
    bad[ix[100:150]] = [1e30, -1e30, 1e30]
    bad[ix[150:200], 2] = 0.0
    bad[ix[200:250], 2] = -0.5
    bad[ix[250:300]] = 0.0
    bad[ix[300:350], 0] = -np.inf
    bad[ix[350:400]] *= 1e-30
    crit = (0.0, 0.0, 1)                                          # two passes: the non-finite points go through one rigid update as well
    ppb = api.get_option("points_per_block")
    want, _, _, _ = O.icp(bad, scenario["proj_scene" if kind == "proj" else "nn_scene"], crit, O.SUM_CANONICAL, ppb)
    dev = api.DeviceVector.from_host(bad.reshape(-1))
    r = api.ICP_Point2Plane(dev, gscenes[kind], api.ICPConvergenceCriteria(*crit))
    assert r.fitness_ == want["fitness"] and 0.5 < r.fitness_ < 1.0
    assert np.allclose(r.transformation_.reshape(-1), want["T"], rtol=0, atol=1e-4)
    offs = np.array([0, len(bad), len(bad) + len(cloud)], np.uint32)
    both = api.DeviceVector.from_host(np.concatenate([bad, cloud]).reshape(-1))
    res = api.ICP_Point2Plane_batch(both, offs, gscenes[kind], api.ICPConvergenceCriteria(*crit))
    clean, _, _, _ = O.icp(cloud, scenario["proj_scene" if kind == "proj" else "nn_scene"], crit, O.SUM_CANONICAL, ppb)
    assert res[0]["fitness"] == want["fitness"] and res[1]["fitness"] == clean["fitness"]


@pytest.mark.parametrize("kind", ["proj", "nn"])
def test_more_clouds_than_a_launch_has_rows(gpu, scenario, gscenes, kind):
    """70 000 clouds in one ICP_Point2Plane_batch call (the hypothesis index is the y dimension of the launches: the list runs in
    pieces): every spot-checked cloud equals the same cloud refined on its own, host and device solve give the same records."""
    cloud = scenario["cloud"]
    n_clouds, per = 70000, 8
    starts = np.random.default_rng(1).integers(0, len(cloud) - per, n_clouds)
    cl = np.concatenate([cloud[s:s + per] for s in starts])
    offs = (np.arange(n_clouds + 1) * per).astype(np.uint32)
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 3)
    got = {}
    try:
        for solve in (api.SOLVE_DEVICE, api.SOLVE_HOST):
            api.set_option("solve", solve)
            res = api.ICP_Point2Plane_batch(api.DeviceVector.from_host(cl.reshape(-1)), offs, gscenes[kind], crit)
            got[solve] = res.tobytes()
            for i in (0, 1, 32767, 32768, 40001, 65535, 65536, n_clouds - 1):
                one = api.ICP_Point2Plane(api.DeviceVector.from_host(cl[i * per:(i + 1) * per].reshape(-1)), gscenes[kind], crit)
                assert one.fitness_ == res[i]["fitness"] and np.array_equal(one.transformation_.reshape(-1), res[i]["T"]), (solve, i)
    finally:
        api.set_option("solve", api.SOLVE_DEVICE)
    assert got[api.SOLVE_DEVICE] == got[api.SOLVE_HOST]


@pytest.mark.parametrize("kind", ["proj", "nn"])
def test_six_hundred_iterations(gpu, scenario, gscenes, kind):
    """max_iteration = 600 with zero thresholds (601 passes, a long captured loop): host and device solve agree bit for bit and with the oracle."""
    cl = scenario["cloud"][:6000]
    crit = (0.0, 0.0, 600)
    out = []
    try:
        for solve in (api.SOLVE_DEVICE, api.SOLVE_HOST):
            api.set_option("solve", solve)
            r = api.ICP_Point2Plane(api.DeviceVector.from_host(cl.reshape(-1)), gscenes[kind], api.ICPConvergenceCriteria(*crit))
            out.append((r.transformation_.tobytes(), r.fitness_, r.inlier_rmse_))
    finally:
        api.set_option("solve", api.SOLVE_DEVICE)
    assert out[0] == out[1]
    want, _, _, _ = O.icp(cl, scenario["proj_scene" if kind == "proj" else "nn_scene"], crit, O.SUM_CANONICAL, api.get_option("points_per_block"))
    assert out[0][1] == want["fitness"] and np.allclose(np.frombuffer(out[0][0], np.float32), want["T"], rtol=0, atol=1e-4)


def test_reduction_tree_and_grouping_options_at_their_extremes(gpu, model, scenario, gscenes):
    """points_per_block 1024 / 65 536 (one point step per workgroup / one workgroup per cloud), 1 and 4 pose groups, sub-batches of 32 on a
    batch of 33: every combination equals the oracle in the tree that points_per_block selects."""
    poses = synth.hypotheses(33, seed=2)
    crit = (0.0, 0.0, 5)
    cl = O.depth2cloud(O.render(scenario["tris"], poses[32:33], W, H, scenario["proj"])[0], scenario["K"])
    try:
        for ppb in (1024, 65536):
            api.set_option("points_per_block", ppb)
            want, _, _, _ = O.icp(cl, scenario["proj_scene"], crit, O.SUM_CANONICAL, ppb)
            for groups, sub in ((1, 32), (4, 32), (3, 512)):
                api.set_option("pose_groups", groups); api.set_option("sub_batch", sub)
                res, sizes = api.refine_batch(model, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], api.ICPConvergenceCriteria(*crit))
                assert sizes[32] == len(cl) and res[32]["fitness"] == want["fitness"] and np.allclose(res[32]["T"], want["T"], rtol=0, atol=1e-4), (ppb, groups, sub)
    finally:
        api.set_option("points_per_block", 3072); api.set_option("pose_groups", 0); api.set_option("sub_batch", 512)


def test_four_host_threads_on_the_shared_context(gpu, model, scenario, gscenes):
    """Calls of different kinds from four threads at once on the process' shared context: every result equals the one computed alone."""
    import threading
    poses = synth.hypotheses(24, seed=8)
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 5)
    cloud = scenario["cloud"]

    def run(what):
        if what == "refine":
            return api.refine_batch(model, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit)[0].tobytes()
        if what == "nn":
            return api.refine_batch(model, poses[:8], W, H, scenario["proj"], scenario["K"], gscenes["nn"], crit)[0].tobytes()
        if what == "render":
            return api.render_host(model, poses[:4], W, H, scenario["proj"]).tobytes()
        return api.ICP_Point2Plane(api.DeviceVector.from_host(cloud.reshape(-1)), gscenes["proj"], crit).transformation_.tobytes()
    kinds = ("refine", "nn", "render", "icp")
    alone = {k: run(k) for k in kinds}
    bad = []

    def worker(tid):
        for k in range(10):
            what = kinds[(tid + k) % 4]
            if run(what) != alone[what]:
                bad.append((tid, k, what))
    ts = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not bad


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


@pytest.mark.parametrize("W,H,stride,tlx,tly,dt", [(97, 61, 3, 0, 0, np.int32), (101, 57, 7, 5, 9, np.uint16), (64, 48, 64, 0, 0, np.int32), (33, 2, 5, 1, 1, np.int32), (1, 1, 1, 0, 0, np.uint16)])
def test_depth2cloud_strides_that_do_not_divide_the_frame(gpu, W, H, stride, tlx, tly, dt):
    rng = np.random.default_rng(W * H + stride)
    d = (rng.integers(0, 3, size=(H, W)) * rng.integers(200, 900, size=(H, W))).astype(dt)
    K = np.array([80, 0, W / 2, 0, 82, H / 2, 0, 0, 1], np.float32)
    got = api.depth2cloud(api.DeviceVector.from_host(d.reshape(-1)), W, H, K, stride, tlx, tly, dtype=dt).to_host().reshape(-1, 3)
    want = O.depth2cloud(d, K, stride, tlx, tly)
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.parametrize("roi", [(0, 0, 1, 1), (639, 479, 1, 1), (320, 240, 1, 1), (0, 0, 640, 1), (0, 0, 1, 480), (300, 200, 37, 53)])
def test_render_roi_extremes(gpu, model, scenario, roi):
    poses = synth.hypotheses(3, seed=1)
    assert np.array_equal(api.render_host(model, poses, W, H, scenario["proj"], roi), O.render(scenario["tris"], poses, W, H, scenario["proj"], roi))


@pytest.mark.parametrize("kind,P", [("proj", 96), ("nn", 40)])
def test_timed_asynchronous_batches(gpu, model, scenario, gscenes, kind, P):
    """Option profile = 3: batches submitted on the slots carry HIP events around their launches and stay asynchronous.  Results equal the
    untimed ones bit for bit; the accounts read after pr_refine_wait hold one entry per pass and sub-batch, the points of every cloud
    per pass, and 36 / 48 algorithmic bytes per point on edge / inner passes -- as the synchronous timed path (profile 1) reports them."""
    poses_a, poses_b = synth.hypotheses(P, seed=21), synth.hypotheses(P, seed=22)
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 6)
    api.set_option("sub_batch", 64)                               # P = 96 runs as two sub-batches
    try:
        api.set_option("profile", 0)
        ref_a = api.refine_batch(model, poses_a, W, H, scenario["proj"], scenario["K"], gscenes[kind], crit)
        ref_b = api.refine_batch(model, poses_b, W, H, scenario["proj"], scenario["K"], gscenes[kind], crit)
        api.set_option("profile", 1)
        api.profile_reset()
        api.refine_batch(model, poses_a, W, H, scenario["proj"], scenario["K"], gscenes[kind], crit)
        sync = api.profile_read()
        api.set_option("profile", 3)
        api.profile_reset()
        api.refine_submit(0, model, poses_a, W, H, scenario["proj"], scenario["K"], gscenes[kind], crit)
        api.refine_submit(1, model, poses_b, W, H, scenario["proj"], scenario["K"], gscenes[kind], crit)     # timed as well: runs after slot 0's batch
        got_a = api.refine_wait(0)
        one = api.profile_read()
        got_b = api.refine_wait(1)
        both = api.profile_read()
    finally:
        api.set_option("profile", 0); api.set_option("sub_batch", 512)
    assert got_a[0].tobytes() == ref_a[0].tobytes() and np.array_equal(got_a[1], ref_a[1])
    assert got_b[0].tobytes() == ref_b[0].tobytes() and np.array_equal(got_b[1], ref_b[1])
    n_sub = (P + 63) // 64
    assert one["icp_launches"] == 7 * n_sub and sync["icp_launches"] == 7        # (the synchronous path does not split 96 hypotheses)
    assert one["icp_points"] == 7 * int(ref_a[1].sum()) == sync["icp_points"] and one["icp_bytes"] == sync["icp_bytes"] == int(ref_a[1].sum()) * (2 * 36 + 5 * 48)
    assert both["icp_launches"] == 14 * n_sub and both["icp_points"] == 7 * int(ref_a[1].sum() + ref_b[1].sum())
    assert 0 < one["icp_kernel_ms"] < 50 and one["render_ms"] > 0 and one["cloud_ms"] > 0


@pytest.mark.parametrize("kind", ["proj", "nn"])
def test_host_solve_batches_from_two_threads_with_private_contexts(gpu, model, scenario, gscenes, kind):
    """The reference's way of feeding the GPU (README.md:15): host threads, each with its own context, issue whole batches with the solve on
    the host.  Every batch equals the one computed alone."""
    import threading
    poses = synth.hypotheses(70, seed=31)
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 6)
    api.set_option("solve", api.SOLVE_HOST)
    try:
        alone = api.refine_batch(model, poses, W, H, scenario["proj"], scenario["K"], gscenes[kind], crit)
        bad, errs = [], []

        def work():
            try:
                api.thread_context(True)
                for _ in range(6):
                    out = api.refine_batch(model, poses, W, H, scenario["proj"], scenario["K"], gscenes[kind], crit)
                    if out[0].tobytes() != alone[0].tobytes() or not np.array_equal(out[1], alone[1]):
                        bad.append(1)
                api.thread_context(False)
            except Exception as e:                               # noqa: BLE001
                errs.append(e)
        ts = [threading.Thread(target=work) for _ in range(2)]
        [t.start() for t in ts]
        [t.join() for t in ts]
    finally:
        api.set_option("solve", api.SOLVE_DEVICE)
    assert not errs and not bad, (errs, len(bad))


def test_misused_entry_points_return_codes_and_leave_no_trace(gpu):
    """Null and foreign pointers, a double free, unknown options and slots: error codes (a null destination of pr_fill_i32 used to reach the
    kernel and fault the GPU), and the call after a refused one works (the runtime's sticky copy of an error is not found by the next launch)."""
    lib = _lib.load()
    assert lib.pr_fill_i32(None, 10, 5) == -3
    assert lib.pr_malloc(None, 16) == -3
    host = np.zeros(16, np.float32)
    assert lib.pr_memcpy_h2d(None, host.ctypes.data, 64) != 0          # refused by the runtime ...
    d = api.DeviceVector(64, np.int32)
    assert lib.pr_fill_i32(d.data(), 64, 7) == 0                          # ... and the next launch does not inherit that error
    assert np.array_equal(d.to_host(), np.full(64, 7, np.int32))
    q = C.c_void_p()
    assert lib.pr_malloc(C.byref(q), 1024) == 0 and lib.pr_free(q) == 0 and lib.pr_free(q) != 0 and lib.pr_free(None) == 0
    assert lib.pr_fill_i32(d.data(), 64, 9) == 0 and int(d.to_host()[0]) == 9
    assert lib.pr_set_option(None, 1) == -3 and lib.pr_set_option(b"nonsense", 1) == -3 and lib.pr_refine_wait(7) == -3 and lib.pr_refine_wait(-1) == -3


def test_empty_inputs_come_without_arrays(gpu, model, scenario, gscenes):
    """What an empty device_vector hands over is a null pointer and a zero count: a render of no hypotheses, a cloud of no points (alone, in a
    list of such, a list of none), an empty render stack for raw2depth -- the reference's answers (nothing / the identity with fitness 0,
    icp.cu:183), not 'bad arguments'."""
    e0 = np.zeros((0, 16), np.float32)
    assert api.render(model, e0, W, H, scenario["proj"]).size() == 0
    assert api.render_host(model, e0, W, H, scenario["proj"]).shape == (0, H, W)
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 3)
    try:
        for solve in (api.SOLVE_HOST, api.SOLVE_DEVICE):
            api.set_option("solve", solve)
            for sc in (gscenes["proj"], gscenes["nn"]):
                r = api.ICP_Point2Plane(api.DeviceVector(0, np.float32), sc, crit)
                assert r.fitness_ == 0.0 and r.inlier_rmse_ == 0.0 and np.array_equal(r.transformation_, np.eye(4, dtype=np.float32))
                res = api.ICP_Point2Plane_batch(api.DeviceVector(0, np.float32), np.zeros(4, np.uint32), sc, crit)
                assert len(res) == 3 and not res["fitness"].any() and all(np.array_equal(t.reshape(4, 4), np.eye(4, dtype=np.float32)) for t in res["T"])
                assert len(api.ICP_Point2Plane_batch(api.DeviceVector(0, np.float32), np.zeros(1, np.uint32), sc, crit)) == 0
                out = api.refine_batch(model, e0, W, H, scenario["proj"], scenario["K"], sc, crit)
                assert len(out[0]) == 0 and len(out[1]) == 0
    finally:
        api.set_option("solve", api.SOLVE_DEVICE)
    d16, m8 = api.raw2depth_mask(api.DeviceVector(0, np.int32))
    assert d16.size == 0 and m8.size == 0
