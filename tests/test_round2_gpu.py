"""GPU tests (-m gpu) of what round 2 added to the boundary: ROI refinement and cropped projective scenes (SURVEY 8f
rank 3: renderer.h:199, common.h:47-73 tl_x / tl_y), the safety of address-keyed caches, kd-tree input
validation, per-thread contexts (the reference's threading contract, README.md:15) and the C-ABI gather."""
import ctypes as C
import json
import os
import threading

import numpy as np
import pytest

import oracle_lib as O
from pose_refine_amd import _lib, api, synth

pytestmark = pytest.mark.gpu

W, H = synth.WIDTH, synth.HEIGHT
TOL_T = 1e-4


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


def raw_hip():
    """The HIP runtime the library itself uses, for writes the library cannot see."""
    return C.CDLL("libamdhip64.so.7")


def raw_h2d(dst_dev: int, arr: np.ndarray):
    hip = raw_hip()
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(dst_dev, arr.ctypes.data, arr.nbytes, 1) == 0
    assert hip.hipDeviceSynchronize() == 0


# ---- ROI refinement: renderer.h:199 + icp.h:57-60 (cuda_renderer/test.cpp:116-157 is the reference's ROI test) ---------
@pytest.mark.parametrize("solve", [api.SOLVE_HOST, api.SOLVE_DEVICE])
def test_refine_roi_that_contains_every_silhouette_equals_full_frame(gpu, model, scenario, gscenes, solve):
    api.set_option("solve", solve)
    try:
        poses = synth.hypotheses(16)
        crit = api.ICPConvergenceCriteria(0.0, 0.0, 6)
        full, fsizes = api.refine_batch(model, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit)
        depth = O.render(scenario["tris"], poses, W, H, scenario["proj"])
        ys, xs = np.nonzero(depth.max(axis=0))
        roi = (int(xs.min()), int(ys.min()), int(xs.max() - xs.min() + 1), int(ys.max() - ys.min() + 1))
        crop, csizes = api.refine_batch(model, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit, roi=roi)
        assert np.array_equal(fsizes, csizes) and full.tobytes() == crop.tobytes()
    finally:
        api.set_option("solve", api.SOLVE_DEVICE)


@pytest.mark.parametrize("roi", [(160, 80, 320, 240), (300, 200, 150, 120), (0, 0, 64, 48)])
def test_refine_roi_against_oracle(gpu, model, scenario, gscenes, roi):
    """An ROI that cuts through the object: the cloud is the rendered pixels inside the window with full-frame
    coordinates (render with roi, depth2cloud with tl = roi.xy) -- compared with the oracle doing exactly that."""
    poses = synth.hypotheses(6)
    crit = (0.0, 0.0, 5)
    res, sizes = api.refine_batch(model, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"],
                                  api.ICPConvergenceCriteria(*crit), roi=roi)
    ores, osizes, _ = O.refine_batch(scenario["tris"], poses, W, H, scenario["proj"], scenario["K"], scenario["proj_scene"], crit,
                                     O.SUM_CANONICAL, api.get_option("points_per_block"), roi=roi)
    assert np.array_equal(sizes, osizes)
    assert np.array_equal(res["fitness"], ores["fitness"])
    assert np.allclose(res["T"], ores["T"], rtol=0, atol=TOL_T)
    # asynchronous slots take the same window
    api.refine_submit(1, model, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], api.ICPConvergenceCriteria(*crit), roi=roi)
    ares, asizes = api.refine_wait(1)
    assert np.array_equal(asizes, sizes) and ares.tobytes() == res.tobytes()
    with pytest.raises(api.PoseRefineError):
        api.refine_batch(model, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], roi=(600, 0, 100, 100))


# ---- cropped projective scene: pcd2dep / dep2pcd with tl_x, tl_y (common.h:47-73) ------------------------------------------
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


# ---- caches keyed by the caller's addresses -----------------------------------------------------------------------------
def test_rewritten_triangle_buffer_is_never_rendered_with_a_stale_box(gpu, scenario, gscenes):
    """the asynchronous path sizes a batch from a host copy of the model box keyed by (pointer, size).
    The buffer is rewritten here behind the library's back (raw hipMemcpy, same address, same triangle count) with a mesh
    twice the size: the batch must come out as if the box had been recomputed."""
    tris = scenario["tris"][:20000].copy()
    big = (tris * np.float32(1.6)).astype(np.float32)
    poses = synth.hypotheses(40)
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 4)
    m = api.Model(tris=tris)
    r0, s0 = api.refine_batch(m, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit)      # caches the small box
    ref_big, ref_sizes = api.refine_batch(api.Model(tris=big), poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit)
    raw_h2d(m.device_tris().data(), big)
    r1, s1 = api.refine_batch(m, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit)
    assert np.array_equal(s1, ref_sizes) and r1.tobytes() == ref_big.tobytes()
    assert not np.array_equal(s0, s1)
    # and through the two asynchronous slots, with the stale box detected at wait time
    raw_h2d(m.device_tris().data(), tris)
    api.refine_submit(0, m, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit)
    api.refine_submit(1, m, poses[::-1].copy(), W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit)
    a, sa = api.refine_wait(0)
    b, sb = api.refine_wait(1)
    assert np.array_equal(sa, s0) and a.tobytes() == r0.tobytes()
    assert np.array_equal(sb, s0[::-1]) and b.tobytes() == r0[::-1].tobytes()


def test_scene_cache_follows_writes(gpu, model, scenario):
    """The packed projective scene is cached by the address of the caller's arrays: writes through the library drop it,
    writes the library cannot see need pr_invalidate (documented contract), scene_cache=0 never caches."""
    K = scenario["K"]
    d1, d0 = scenario["depth"][1], scenario["depth"][0]
    poses = synth.hypotheses(8)
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 5)
    sa = api.Scene_projective().init_Scene_projective_cuda(d1, K)
    sb = api.Scene_projective().init_Scene_projective_cuda(d0, K)
    ra, _ = api.refine_batch(model, poses, W, H, scenario["proj"], K, sa, crit)
    rb, _ = api.refine_batch(model, poses, W, H, scenario["proj"], K, sb, crit)
    assert ra.tobytes() != rb.tobytes()
    lib = _lib.load()
    # (1) overwrite scene A's arrays with scene B's content through the library: the cache must notice
    n = sa.pcd_buffer.size() * 4
    api.check(lib.pr_memcpy_d2d(sa.pcd_buffer.data(), sb.pcd_buffer.data(), n))
    api.check(lib.pr_memcpy_d2d(sa.normal_buffer.data(), sb.normal_buffer.data(), n))
    r1, _ = api.refine_batch(model, poses, W, H, scenario["proj"], K, sa, crit)
    assert r1.tobytes() == rb.tobytes()
    # (2) back to scene A behind the library's back + pr_invalidate
    sa2 = api.Scene_projective().init_Scene_projective_cuda(d1, K)
    raw_h2d(sa.pcd_buffer.data(), sa2.pcd_host)
    raw_h2d(sa.normal_buffer.data(), sa2.normal_host)
    api.invalidate(sa.pcd_buffer.data(), n)
    api.invalidate(sa.normal_buffer.data())                      # 0 bytes = the whole allocation
    r2, _ = api.refine_batch(model, poses, W, H, scenario["proj"], K, sa, crit)
    assert r2.tobytes() == ra.tobytes()
    # (3) caches off: raw writes are picked up without any announcement
    api.set_option("scene_cache", 0)
    try:
        raw_h2d(sa.pcd_buffer.data(), sb.pcd_host)
        raw_h2d(sa.normal_buffer.data(), sb.normal_host)
        r3, _ = api.refine_batch(model, poses, W, H, scenario["proj"], K, sa, crit)
        assert r3.tobytes() == rb.tobytes()
    finally:
        api.set_option("scene_cache", 1)


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


# ---- the reference's threading contract: many host threads, each refining its own hypothesis (README.md:15) --------------
def test_host_threads_with_private_contexts(gpu, scenario, gscenes):
    cloud = scenario["cloud"]
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 8)
    shifts = [np.array([0.001 * k, -0.0005 * k, 0.0008 * k], np.float32) for k in range(6)]
    want = []
    for sh in shifts:
        want.append(api.ICP_Point2Plane(api.DeviceVector.from_host((cloud + sh).reshape(-1)), gscenes["proj"], crit))
    got = [None] * len(shifts)
    errs = []

    def work(k):
        try:
            api.thread_context(True)                             # own stream + workspaces, like cudaStreamPerThread
            for _ in range(3):
                dev = api.DeviceVector.from_host((cloud + shifts[k]).reshape(-1))
                got[k] = api.ICP_Point2Plane(dev, gscenes["proj"], crit)
                dev.free()
            api.thread_context(False)
        except Exception as e:                                   # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(k,)) for k in range(len(shifts))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    for g, w in zip(got, want):
        assert g.fitness_ == w.fitness_ and g.inlier_rmse_ == w.inlier_rmse_ and np.array_equal(g.transformation_, w.transformation_)


def test_async_slots_in_private_contexts_and_across_shutdown(gpu, model, scenario, gscenes):
    """Every host thread with a private context has two asynchronous slots of its own (and the Python mirror keeps each thread's
    in-flight output arrays alive separately); pr_shutdown releases a context, the next call builds a new one."""
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 5)
    poses = synth.hypotheses(70)
    api.set_option("nn_count", 1)                                # instrumented kd-tree runs are synchronous
    api.set_option("profile", 1)                                 # timed calls are synchronous
    try:
        want, want_sizes = api.refine_batch(model, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit)
        want_nn, _ = api.refine_batch(model, poses[:20], W, H, scenario["proj"], scenario["K"], gscenes["nn"], crit)
    finally:
        api.set_option("profile", 0)
        api.set_option("nn_count", 0)

    def both_slots(tag):
        for _ in range(3):
            api.refine_submit(0, model, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit)
            api.refine_submit(1, model, poses[:20], W, H, scenario["proj"], scenario["K"], gscenes["nn"], crit)
            a, sa = api.refine_wait(0)
            b, _ = api.refine_wait(1)
            assert a.tobytes() == want.tobytes() and np.array_equal(sa, want_sizes), tag
            assert b.tobytes() == want_nn.tobytes(), tag

    both_slots("shared context")
    api.shutdown()
    both_slots("after pr_shutdown")
    errs = []

    def work(k):
        try:
            api.thread_context(True)
            both_slots(f"private context of thread {k}")
            api.thread_context(False)
        except Exception as e:                                   # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    both_slots("shared context again")


# ---- kd-tree scenes on the asynchronous two-slot path (VERDICT r01 missing #4) -------------------------------------------------
def test_kdtree_batches_on_both_slots_equal_the_synchronous_path(gpu, model, scenario, gscenes):
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 6)
    batches = [synth.hypotheses(70, first=0), synth.hypotheses(96, first=70), synth.hypotheses(33, first=166)]
    api.set_option("nn_count", 1)                                # an instrumented run is synchronous (refine_impl + icp_drive)
    try:
        want = [api.refine_batch(model, b, W, H, scenario["proj"], scenario["K"], gscenes["nn"], crit) for b in batches]
    finally:
        api.set_option("nn_count", 0)
    want_proj = api.refine_batch(model, batches[1], W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit)
    for rep in range(3):
        # slot 0 and slot 1 in flight together, kd-tree next to kd-tree and next to a projective batch
        api.refine_submit(0, model, batches[0], W, H, scenario["proj"], scenario["K"], gscenes["nn"], crit)
        api.refine_submit(1, model, batches[1], W, H, scenario["proj"], scenario["K"], gscenes["nn"], crit)
        r0 = api.refine_wait(0)
        api.refine_submit(0, model, batches[1], W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit)
        r1 = api.refine_wait(1)
        api.refine_submit(1, model, batches[2], W, H, scenario["proj"], scenario["K"], gscenes["nn"], crit)
        rp = api.refine_wait(0)
        r2 = api.refine_wait(1)
        for got, exp in ((r0, want[0]), (r1, want[1]), (r2, want[2]), (rp, want_proj)):
            assert np.array_equal(got[1], exp[1])
            assert got[0].tobytes() == exp[0].tobytes()
    # and against the oracle (kd-tree association is slow on the CPU: the first four hypotheses)
    ores, osizes, _ = O.refine_batch(scenario["tris"], batches[0][:4], W, H, scenario["proj"], scenario["K"], scenario["nn_scene"],
                                     (0.0, 0.0, 6), O.SUM_CANONICAL, api.get_option("points_per_block"))
    assert np.array_equal(osizes, want[0][1][:4]) and np.array_equal(ores["fitness"], want[0][0]["fitness"][:4])
    assert np.allclose(ores["T"], want[0][0]["T"][:4], rtol=0, atol=TOL_T)


# ---- C-ABI gather (RCCL); one GPU here: world 1, the communicator and the collective still run ------------------------------
def test_cabi_gather_world1(gpu, model, scenario, gscenes):
    P = 40
    poses = synth.hypotheses(P)
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 3)
    res, _ = api.refine_batch(model, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit)
    send = api.DeviceVector.from_host(res.view(np.float32).reshape(-1))
    recv = api.DeviceVector(P * 18, np.float32)
    api.comm_init_rank(api.comm_id(), 0, 1)                      # ncclCommInitRank with one rank
    try:
        assert api.comm_rank() == (0, 1)
        api.gather_results(send.data(), P, P, 0, recv.data())
        api.sync()
        assert recv.to_host().tobytes() == res.tobytes()
        with pytest.raises(api.PoseRefineError):
            api.gather_results(send.data(), P - 1, P, 0, recv.data())      # not this rank's shard size
    finally:
        api.comm_destroy()
    api.comm_init_all(1)                                          # the single-process form (ncclCommInitAll)
    try:
        recv2 = api.DeviceVector(P * 18, np.float32)
        api.gather_results(send.data(), P, P, 0, recv2.data())
        api.sync()
        assert recv2.to_host().tobytes() == res.tobytes()
    finally:
        api.comm_destroy()


# ---- the reference's own known answers, against the HIP result directly (VERDICT r01 missing #9) ---------------------------
def test_hip_results_against_reference_known_answers(gpu, model, scenario, gscenes, golden_dir):
    """tests/golden/survey_8c.json holds what the verbatim reference CPU path returned for test.cpp's scenario (sequential
    sums).  The HIP path sums in its own fixed tree, so: transforms within 1e-4, inlier counts within the +-3 the
    reference's own OpenMP reduction wobbles by, cloud size and render checksums exact."""
    with open(os.path.join(golden_dir, "survey_8c.json")) as f:
        gold = json.load(f)
    depth = api.render_host(model, scenario["poses"], W, H, scenario["proj"])
    for i in range(2):
        g = gold["render"][i]
        v = depth[i][depth[i] > 0]
        assert (int(v.size), int(v.sum()), int(v.min()), int(v.max())) == (g["valid"], g["sum"], g["min"], g["max"])
    dev_depth = api.render(model, scenario["poses"][:1], W, H, scenario["proj"])
    n = gold["cloud_points"]
    for key, kind, crit in (("proj_default", "proj", (1e-5, 1e-5, 30)), ("proj_fixed20", "proj", (0.0, 0.0, 20)),
                            ("nn_default", "nn", (1e-5, 1e-5, 30)), ("nn_fixed20", "nn", (0.0, 0.0, 20))):
        if key not in gold["icp"]:
            continue
        for solve in (api.SOLVE_HOST, api.SOLVE_DEVICE):
            api.set_option("solve", solve)
            cloud = api.depth2cloud(dev_depth, W, H, scenario["K"])
            assert cloud.size() // 3 == n
            r = api.ICP_Point2Plane(cloud, gscenes[kind], api.ICPConvergenceCriteria(*crit))
            g = gold["icp"][key]
            assert abs(int(round(r.fitness_ * n)) - g["inliers"]) <= 3, (key, r.fitness_ * n, g["inliers"])
            assert r.inlier_rmse_ == pytest.approx(g["rmse"], rel=2e-4)
            for row, want in enumerate(g["T_rows"]):
                assert np.allclose(r.transformation_[row], np.array(want, np.float32), rtol=0, atol=TOL_T), (key, row)
    api.set_option("solve", api.SOLVE_DEVICE)
