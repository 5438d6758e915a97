"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs.  Integer outputs (depth images, cloud sizes, inlier counts) must be bit-exact;
per-pass 29-float sums are bit-exact against the oracle's canonical tree; transforms within 1e-4
(BASELINE.json north_star tolerance) -- in practice they are bit-identical too.
"""
import os

import numpy as np
import pytest

import oracle_lib as O
from pose_refine_amd import api, synth

pytestmark = pytest.mark.gpu

W, H = synth.WIDTH, synth.HEIGHT
TOL_T = 1e-4          # north_star: "transforms within 1e-4"


@pytest.fixture(scope="module")
def gpu():
    api.init(0)
    return True


@pytest.fixture(scope="module")
def model(gpu, golden_dir):
    return api.Model(os.path.join(golden_dir, "obj_06.ply"))


@pytest.fixture(scope="module")
def gscenes(gpu, scenario):
    d = scenario["depth"][1]
    return dict(proj=api.Scene_projective().init_Scene_projective_cuda(d, scenario["K"]),
                nn=api.Scene_nn().init_Scene_nn_cuda(d, scenario["K"]))


def inliers(fitness, n):
    return np.rint(np.asarray(fitness, np.float64) * np.asarray(n, np.float64)).astype(np.int64)


# ---- renderer: cuda_renderer/test.cpp:79-106 (full frame) and :122-149 (ROI) ------------------------
def test_render_matches_cpu_bit_exact(model, scenario):
    poses = np.concatenate([scenario["poses"], synth.hypotheses(6)[1:]])
    ref = O.render(scenario["tris"], poses, W, H, scenario["proj"])
    got_host = api.render_host(model, poses, W, H, scenario["proj"])
    assert np.array_equal(got_host, ref)
    keep = api.render(model, poses, W, H, scenario["proj"])           # render_cuda_keep_in_gpu
    assert np.array_equal(keep.to_host().reshape(ref.shape), ref)


def test_render_100_identical_poses_like_reference_test(model, scenario):
    poses = np.repeat(scenario["poses"][:1], 100, axis=0)              # cuda_renderer/test.cpp:63
    got = api.render_host(model, poses, W, H, scenario["proj"])
    assert np.abs(got.astype(np.int64) - scenario["depth"][0][None].astype(np.int64)).sum() == 0


def test_render_roi(model, scenario):
    roi = (160, 80, 320, 240)                                          # cuda_renderer/test.cpp:122
    poses = scenario["poses"]
    ref = O.render(scenario["tris"], poses, W, H, scenario["proj"], roi)
    got = api.render_host(model, poses, W, H, scenario["proj"], roi)
    assert got.shape == ref.shape and np.array_equal(got, ref)
    with pytest.raises(api.PoseRefineError):
        api.render_host(model, poses, W, H, scenario["proj"], (600, 0, 100, 100))   # roi out of image


def test_render_offscreen_and_empty(model, scenario):
    far = scenario["poses"][:1].copy()
    far[0, 0, 3] = 5000.0                                              # pushed out of the frustum sideways
    assert api.render_host(model, far, W, H, scenario["proj"]).sum() == 0
    assert api.render_host(model, np.zeros((0, 16), np.float32), W, H, scenario["proj"]).size == 0


# ---- depth -> cloud -------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.int32, np.uint16])
def test_depth2cloud_bit_exact(gpu, scenario, dtype):
    d = scenario["depth"][0].astype(dtype)
    dev = api.DeviceVector.from_host(d.reshape(-1))
    got = api.depth2cloud(dev, W, H, scenario["K"], dtype=dtype).to_host().reshape(-1, 3)
    ref = O.depth2cloud(d, scenario["K"])
    assert got.shape == ref.shape and np.array_equal(got, ref)


def test_depth2cloud_crop_offsets_and_empty(gpu, scenario):
    d = np.ascontiguousarray(scenario["depth"][0][80:320, 160:480])    # cropped render, tl = (160, 80)
    dev = api.DeviceVector.from_host(d.reshape(-1))
    got = api.depth2cloud(dev, 320, 240, scenario["K"], 1, 160, 80).to_host().reshape(-1, 3)
    assert np.array_equal(got, O.depth2cloud(d, scenario["K"], 1, 160, 80))
    z = api.DeviceVector.from_host(np.zeros(64 * 48, np.int32))
    assert api.depth2cloud(z, 64, 48, scenario["K"]).size() == 0


def test_depth2cloud_stride2(gpu, scenario):
    d = scenario["depth"][0]
    dev = api.DeviceVector.from_host(d.reshape(-1))
    got = api.depth2cloud(dev, W, H, scenario["K"], 2).to_host().reshape(-1, 3)
    assert np.array_equal(got, O.depth2cloud(d, scenario["K"], 2))


# ---- single-cloud ICP: test.cpp:153-172 -----------------------------------------------------------
@pytest.mark.parametrize("kind", ["proj", "nn"])
@pytest.mark.parametrize("crit", [(1e-5, 1e-5, 30), (0.0, 0.0, 20)])
@pytest.mark.parametrize("solve", [api.SOLVE_HOST, api.SOLVE_DEVICE])
def test_icp_single_cloud(gpu, scenario, gscenes, kind, crit, solve):
    api.set_option("solve", solve)
    try:
        cloud = scenario["cloud"]
        dev = api.DeviceVector.from_host(cloud.reshape(-1))
        res = api.ICP_Point2Plane(dev, gscenes[kind], api.ICPConvergenceCriteria(*crit))
        oscene = scenario["proj_scene" if kind == "proj" else "nn_scene"]
        ores, passes, ocloud, _ = O.icp(cloud, oscene, crit, O.SUM_CANONICAL, api.get_option("points_per_block"))
        n = len(cloud)
        assert inliers(res.fitness_, n) == inliers(ores["fitness"], n)          # bit-exact inlier count
        assert res.fitness_ == float(ores["fitness"])
        assert res.inlier_rmse_ == pytest.approx(float(ores["inlier_rmse"]), rel=1e-6)
        assert np.allclose(res.transformation_, ores["T"].reshape(4, 4), rtol=0, atol=TOL_T)
        # the reference mutates the caller's cloud (test.cpp:129 comment)
        assert np.array_equal(dev.to_host().reshape(-1, 3), ocloud)              # bit for bit: same updates, same order
        if solve == api.SOLVE_HOST:
            assert np.array_equal(res.transformation_, ores["T"].reshape(4, 4)) or \
                np.allclose(res.transformation_, ores["T"].reshape(4, 4), rtol=0, atol=1e-6)
    finally:
        api.set_option("solve", api.SOLVE_HOST)


@pytest.mark.parametrize("kind", ["proj", "nn"])
def test_first_pass_sums_bit_exact(gpu, scenario, gscenes, kind):
    """max_iteration=0 -> exactly one correspondence pass: fitness and rmse are functions of the
    canonical-tree sums [28] and [27], so equality here pins the reduction order bit for bit."""
    cloud = scenario["cloud"]
    dev = api.DeviceVector.from_host(cloud.reshape(-1))
    res = api.ICP_Point2Plane(dev, gscenes[kind], api.ICPConvergenceCriteria(0.0, 0.0, 0))
    oscene = scenario["proj_scene" if kind == "proj" else "nn_scene"]
    s = O.sum29(cloud, oscene, O.SUM_CANONICAL, api.get_option("points_per_block"))
    assert res.fitness_ == np.float32(s[28] / np.float32(len(cloud)))
    assert res.inlier_rmse_ == np.float32(np.sqrt(np.float32(s[27] / s[28])))
    assert np.array_equal(res.transformation_, np.eye(4, dtype=np.float32))


def test_host_and_device_solve_agree_bitwise(gpu, scenario, gscenes):
    """Host solve (sums finalized in the tail of the pass and stored into pinned host memory, or by the separate finalize launch + copy:
    fused_solve 1 / 0) and device solve (fused tail / separate finalize+solve launch): same sums in the same order, same solver source."""
    out = []
    try:
        for solve, fused in ((api.SOLVE_HOST, 1), (api.SOLVE_HOST, 0), (api.SOLVE_DEVICE, 1), (api.SOLVE_DEVICE, 0)):
            api.set_option("solve", solve); api.set_option("fused_solve", fused)
            dev = api.DeviceVector.from_host(scenario["cloud"].reshape(-1))
            r = api.ICP_Point2Plane(dev, gscenes["proj"], api.ICPConvergenceCriteria(0.0, 0.0, 20))
            out.append((r.transformation_.copy(), r.fitness_, r.inlier_rmse_, dev.to_host()))
    finally:
        api.set_option("solve", api.SOLVE_HOST); api.set_option("fused_solve", 1)
    for o in out[1:]:
        assert np.array_equal(out[0][0], o[0]) and out[0][1] == o[1] and out[0][2] == o[2]
        assert np.array_equal(out[0][3], o[3])


@pytest.mark.parametrize("kind,P,groups", [("proj", 70, 2), ("proj", 130, 4), ("nn", 66, 2)])
def test_host_solve_pose_groups_agree_bitwise(gpu, model, scenario, gscenes, kind, P, groups):
    """The host-solve loop runs the batch as software-pipelined pose groups; records must not depend on the grouping, on where the
    sums are finalized, or (early-exit criteria) on groups finishing at different iterations."""
    poses = synth.hypotheses(P, seed=5)
    poses[3] = poses[3].copy(); poses[3].reshape(4, 4)[0, 3] += 1.0e6          # an empty cloud in the first group
    try:
        for crit in ((0.0, 0.0, 20), (1e-5, 1e-5, 30)):
            out = []
            for g, fused in ((1, 0), (groups, 1), (groups, 0)):
                api.set_option("pose_groups", g); api.set_option("fused_solve", fused)
                out.append(api.refine_batch(model, poses, W, H, scenario["proj"], scenario["K"], gscenes[kind], api.ICPConvergenceCriteria(*crit)))
            for o in out[1:]:
                assert np.array_equal(out[0][1], o[1]) and out[0][0].tobytes() == o[0].tobytes(), crit
    finally:
        api.set_option("pose_groups", 0); api.set_option("fused_solve", 1)


# ---- edge cases -----------------------------------------------------------------------------------
def test_icp_degenerate_clouds(gpu, scenario, gscenes):
    # no correspondences at all: count==0 -> identity, fitness 0 (icp.cu:183)
    far = scenario["cloud"] + np.array([0, 0, 5.0], np.float32)
    dev = api.DeviceVector.from_host(far.reshape(-1))
    for kind in ("proj", "nn"):
        r = api.ICP_Point2Plane(dev, gscenes[kind])
        assert r.fitness_ == 0.0 and r.inlier_rmse_ == 0.0 and np.array_equal(r.transformation_, np.eye(4, dtype=np.float32))
    # ragged batch incl. an empty cloud and sizes that are not multiples of 4 (scalar tail path,
    # unaligned cloud starts)
    cl = scenario["cloud"]
    parts = [cl[:1], cl[:0], cl[:4099], cl[5:2054], cl]
    offs = np.cumsum([0] + [len(p) for p in parts]).astype(np.uint32)
    dev = api.DeviceVector.from_host(np.concatenate(parts).reshape(-1))
    crit = (0.0, 0.0, 5)
    res = api.ICP_Point2Plane_batch(dev, offs, gscenes["proj"], api.ICPConvergenceCriteria(*crit))
    ppb = api.get_option("points_per_block")
    for i, p in enumerate(parts):
        o, _, _, _ = O.icp(p, scenario["proj_scene"], crit, O.SUM_CANONICAL, ppb)
        assert res[i]["fitness"] == o["fitness"], i
        assert np.allclose(res[i]["T"], o["T"], rtol=0, atol=TOL_T), i


# ---- fused batch: BASELINE.json configs[1]/[2] at parity-test size -----------------------------------
@pytest.mark.parametrize("kind,P", [("proj", 24), ("nn", 6), ("nn", 24)])
@pytest.mark.parametrize("solve", [api.SOLVE_HOST, api.SOLVE_DEVICE])
def test_refine_batch_against_oracle(gpu, model, scenario, gscenes, kind, P, solve):
    api.set_option("solve", solve)
    try:
        poses = synth.hypotheses(P)
        crit = (0.0, 0.0, 20)
        res, sizes = api.refine_batch(model, poses, W, H, scenario["proj"], scenario["K"], gscenes[kind],
                                      api.ICPConvergenceCriteria(*crit))
        oscene = scenario["proj_scene" if kind == "proj" else "nn_scene"]
        ores, osizes, _ = O.refine_batch(scenario["tris"], poses, W, H, scenario["proj"], scenario["K"], oscene, crit,
                                         O.SUM_CANONICAL, api.get_option("points_per_block"))
        assert np.array_equal(sizes, osizes)
        assert np.array_equal(inliers(res["fitness"], sizes), inliers(ores["fitness"], osizes))
        assert np.array_equal(res["fitness"], ores["fitness"])
        assert np.allclose(res["inlier_rmse"], ores["inlier_rmse"], rtol=1e-6, atol=0)
        assert np.allclose(res["T"], ores["T"], rtol=0, atol=TOL_T)
    finally:
        api.set_option("solve", api.SOLVE_HOST)


def test_refine_batch_default_criteria_early_exit(gpu, model, scenario, gscenes):
    poses = synth.hypotheses(8)
    crit = (1e-5, 1e-5, 30)
    res, sizes = api.refine_batch(model, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"],
                                  api.ICPConvergenceCriteria(*crit))
    ores, osizes, _ = O.refine_batch(scenario["tris"], poses, W, H, scenario["proj"], scenario["K"], scenario["proj_scene"],
                                     crit, O.SUM_CANONICAL, api.get_option("points_per_block"))
    assert np.array_equal(sizes, osizes) and np.array_equal(res["fitness"], ores["fitness"])
    assert np.allclose(res["T"], ores["T"], rtol=0, atol=TOL_T)


# ---- full-size properties (configs[1]: 256 hypotheses) --------------------------------------------
def test_full_batch_properties(gpu, model, scenario, gscenes):
    """At BASELINE size the oracle is too slow for every pose; check size-independent properties:
    batch results are independent of batch composition (each pose equals its single-pose run),
    the batch is permutation-equivariant, and pose 0 equals the oracle run of pose 0."""
    P = 256
    poses = synth.hypotheses(P)
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 20)
    res, sizes = api.refine_batch(model, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit)
    perm = np.random.default_rng(3).permutation(P)
    res_p, sizes_p = api.refine_batch(model, poses[perm], W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit)
    assert np.array_equal(sizes[perm], sizes_p)
    assert np.array_equal(res["T"][perm], res_p["T"]) and np.array_equal(res["fitness"][perm], res_p["fitness"])
    for i in (0, 17, 255):
        one, s1 = api.refine_batch(model, poses[i:i + 1], W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit)
        assert s1[0] == sizes[i] and np.array_equal(one["T"][0], res["T"][i]) and one["fitness"][0] == res["fitness"][i]
    ores, osizes, _ = O.refine_batch(scenario["tris"], poses[:1], W, H, scenario["proj"], scenario["K"], scenario["proj_scene"],
                                     (0.0, 0.0, 20), O.SUM_CANONICAL, api.get_option("points_per_block"))
    assert sizes[0] == osizes[0] and res["fitness"][0] == ores["fitness"][0]
    assert np.allclose(res["T"][0], ores["T"][0], rtol=0, atol=TOL_T)
    # every hypothesis is a rigid transform: R^T R = I, det = +1
    R = res["T"].reshape(P, 4, 4)[:, :3, :3].astype(np.float64)
    assert np.allclose(np.einsum("pij,pik->pjk", R, R), np.eye(3), atol=1e-5)
    assert np.allclose(np.linalg.det(R), 1.0, atol=1e-5)


def test_no_device_option_errors(gpu):
    with pytest.raises(api.PoseRefineError):
        api.set_option("no_such_option", 1)
    with pytest.raises(api.PoseRefineError):
        api.set_option("points_per_block", 1000)


def test_fused_raster_modes_agree(gpu, model, scenario, gscenes):
    """LDS-band raster (raster_mode=1) vs the reference-style global atomicMin raster (default) inside the fused path:
    identical cloud sizes and bit-identical results; includes a close-up pose whose pixel box needs
    several LDS bands and a pose partly outside the image."""
    poses = synth.hypotheses(12)
    close = scenario["poses"][0].copy(); close[2, 3] = 110.0            # object fills most of the frame -> many bands
    off = scenario["poses"][0].copy(); off[0, 3] = 120.0                # partly outside the image
    behind = scenario["poses"][0].copy(); behind[2, 3] = 20.0           # camera inside the object's box -> full-frame fallback
    poses = np.concatenate([poses, close[None], off[None], behind[None]])
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 4)
    out = []
    for mode in (1, 0):
        api.set_option("raster_mode", mode)
        out.append(api.refine_batch(model, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit))
    api.set_option("raster_mode", 0)                             # library default
    assert np.array_equal(out[0][1], out[1][1])
    assert out[0][0].tobytes() == out[1][0].tobytes()
    ref = O.render(scenario["tris"], poses[-3:], W, H, scenario["proj"])
    assert np.array_equal(out[0][1][-3:], (ref > 0).reshape(3, -1).sum(1))


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


@pytest.mark.parametrize("kind,P", [("proj", 70), ("nn", 5)])
def test_dataflow_and_multilaunch_icp_agree_bitwise(gpu, model, scenario, gscenes, kind, P):
    """The persistent dataflow kernel (all iterations in one launch, option icp_flow=1) and the launch-per-pass loop run the same
    canonical tree and the same solver: results must be bit-identical, with fixed and with early-exit criteria."""
    poses = synth.hypotheses(P)
    api.set_option("solve", api.SOLVE_DEVICE)
    try:
        for crit in ((0.0, 0.0, 20), (1e-5, 1e-5, 30)):
            out = []
            for flow in (1, 0):
                api.set_option("icp_flow", flow)
                out.append(api.refine_batch(model, poses, W, H, scenario["proj"], scenario["K"], gscenes[kind],
                                            api.ICPConvergenceCriteria(*crit)))
            assert np.array_equal(out[0][1], out[1][1])
            assert out[0][0].tobytes() == out[1][0].tobytes(), crit
    finally:
        api.set_option("icp_flow", 0)
        api.set_option("solve", api.SOLVE_HOST)


@pytest.mark.parametrize("P,raster_mode", [(5, 0), (96, 0), (5, 1)])
def test_async_slots_match_synchronous_path_bitwise(gpu, model, scenario, gscenes, P, raster_mode):
    """pr_refine_submit / pr_refine_wait (host-computed pixel boxes, device-side start state, no mid-step read-back, two batches
    in flight) against the synchronous path (profile=1 forces it): records and cloud sizes must be bit-identical, with fixed
    and with early-exit criteria, for host and device result buffers, with an empty-cloud hypothesis in the batch."""
    poses_a = synth.hypotheses(P, seed=11)
    poses_b = synth.hypotheses(P, seed=12)
    poses_b[1] = poses_b[1].copy()
    poses_b[1].reshape(4, 4)[0, 3] += 1.0e6                      # a kilometre to the side: off-screen -> empty cloud -> identity result
    api.set_option("solve", api.SOLVE_DEVICE)
    api.set_option("raster_mode", raster_mode)                   # 1: not covered by the asynchronous path -> submit runs synchronously (replayed graphs)
    api.set_option("sub_batch", 40)                              # P=96 runs as three sub-batches of 32 that reuse the same workspace
    try:
        for crit in ((0.0, 0.0, 20), (1e-5, 1e-5, 30)):
            c = api.ICPConvergenceCriteria(*crit)
            api.set_option("profile", 1)
            ref_a = api.refine_batch(model, poses_a, W, H, scenario["proj"], scenario["K"], gscenes["proj"], c)
            ref_b = api.refine_batch(model, poses_b, W, H, scenario["proj"], scenario["K"], gscenes["proj"], c)
            api.set_option("profile", 0)
            # the grid of an asynchronous batch is sized from the clouds of the batch before it: make that a batch of tiny
            # clouds (3 m farther away) so that every workgroup of the next one has to walk several 2048-point blocks
            far = poses_a.copy()
            far.reshape(-1, 4, 4)[:, 2, 3] += 3000.0
            _, far_sizes = api.refine_batch(model, far, W, H, scenario["proj"], scenario["K"], gscenes["proj"], c)
            assert 0 < far_sizes.max() < 2048 < ref_a[1].max()
            api.refine_submit(0, model, poses_a, W, H, scenario["proj"], scenario["K"], gscenes["proj"], c)
            dev_b = api.DeviceVector(P * 18, np.float32)
            api.refine_submit(1, model, poses_b, W, H, scenario["proj"], scenario["K"], gscenes["proj"], c, results_dev=dev_b.data())
            got_a = api.refine_wait(0)
            _, sizes_b = api.refine_wait(1)
            got_b = np.frombuffer(dev_b.to_host().tobytes(), dtype=ref_b[0].dtype)
            assert np.array_equal(got_a[1], ref_a[1]) and np.array_equal(sizes_b, ref_b[1])
            assert got_a[0].tobytes() == ref_a[0].tobytes(), crit
            assert got_b.tobytes() == ref_b[0].tobytes(), crit
            assert ref_b[1][1] == 0 and np.array_equal(ref_b[0]["T"][1].reshape(4, 4), np.eye(4, dtype=np.float32))
        with pytest.raises(api.PoseRefineError):
            api.refine_wait(0)                                   # nothing pending
    finally:
        api.set_option("profile", 0)
        api.set_option("raster_mode", 0)
        api.set_option("sub_batch", 512)
        api.set_option("solve", api.SOLVE_HOST)


def test_submit_wait_contract_errors(gpu, model, scenario, gscenes):
    """A slot holds one batch at a time; waiting on an idle slot, an out-of-range slot and frames beyond the raster's
    coordinate packing (8192 per side, 2^24 pixels) are refused with PR_ERR_INVALID, and the slot stays usable."""
    poses = synth.hypotheses(3)
    c = api.ICPConvergenceCriteria(0.0, 0.0, 2)
    api.set_option("solve", api.SOLVE_DEVICE)
    try:
        api.refine_submit(0, model, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], c)
        with pytest.raises(api.PoseRefineError):
            api.refine_submit(0, model, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], c)   # still pending
        res, sizes = api.refine_wait(0)
        assert (sizes > 0).all() and np.isfinite(res["T"]).all()
        with pytest.raises(api.PoseRefineError):
            api.refine_wait(0)
        with pytest.raises(api.PoseRefineError):
            api.refine_submit(2, model, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], c)
        with pytest.raises(api.PoseRefineError):
            api.refine_batch(model, poses, 8200, 16, scenario["proj"], scenario["K"], gscenes["proj"], c)
        again, _ = api.refine_batch(model, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], c)
        assert again.tobytes() == res.tobytes()
    finally:
        api.set_option("solve", api.SOLVE_HOST)


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


def test_raw2depth_mask(gpu, model, scenario):
    poses = scenario["poses"]
    raw = api.render(model, poses, W, H, scenario["proj"])
    d16, m8 = api.raw2depth_mask(raw)
    ref = scenario["depth"].reshape(-1)
    assert np.array_equal(d16, ref.astype(np.uint16)) and np.array_equal(m8, np.where(ref > 0, 255, 0).astype(np.uint8))
    d_only, none = api.raw2depth_mask(raw, want_mask=False)
    assert none is None and np.array_equal(d_only, d16)
    big = api.DeviceVector.from_host(np.array([70000, -5, 0, 65535, 1, 2, 3], np.int32))      # truncation, not saturation
    d, m = api.raw2depth_mask(big)
    assert d.tolist() == [70000 % 65536, 65531, 0, 65535, 1, 2, 3] and m.tolist() == [255, 0, 0, 255, 255, 255, 255]


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
