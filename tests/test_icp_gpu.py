"""GPU tests (-m gpu) of cuda_icp::ICP_Point2Plane_cuda<Scene> (icp.cu:156-223; rows a6, a11-a14): per-pass sums in the canonical tree, host and device solve, criteria, degenerate and absurd clouds.
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


@pytest.mark.device_solve
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


@pytest.mark.device_solve
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


@pytest.mark.device_solve
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


@pytest.mark.device_solve
def test_reduction_tree_and_grouping_options_at_their_extremes(gpu, model, scenario, gscenes):
    """points_per_block 1024 / 65 536 (one point step per workgroup / one workgroup per cloud), 1 and 4 pose groups, sub-batches of 32 on a
    batch of 33: every combination equals the oracle in the tree that points_per_block selects."""
    poses = synth.hypotheses(33, seed=2)
    crit = (0.0, 0.0, 5)
    cl = O.depth2cloud(O.render(scenario["tris"], poses[32:33], W, H, scenario["proj"])[0], scenario["K"])
    ppb_default = api.get_option("points_per_block")
    try:
        for ppb in (1024, 65536):
            api.set_option("points_per_block", ppb)
            want, _, _, _ = O.icp(cl, scenario["proj_scene"], crit, O.SUM_CANONICAL, ppb)
            for groups, sub in ((1, 32), (4, 32), (3, 512)):
                api.set_option("pose_groups", groups); api.set_option("sub_batch", sub)
                res, sizes = api.refine_batch(model, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], api.ICPConvergenceCriteria(*crit))
                assert sizes[32] == len(cl) and res[32]["fitness"] == want["fitness"] and np.allclose(res[32]["T"], want["T"], rtol=0, atol=1e-4), (ppb, groups, sub)
    finally:
        api.set_option("points_per_block", ppb_default); api.set_option("pose_groups", 0); api.set_option("sub_batch", 512)


@pytest.mark.device_solve
def test_host_solve_sampled_call_times_one_pass(gpu, model, scenario, gscenes):
    """with the solve on the host and option profile = 2, one call in sample_period times one of its passes -- also when the
    batch is large enough for the host loop to split it into pose groups (the sampled call runs as one group; the others keep their pipeline).
    Records do not depend on it."""
    poses = synth.hypotheses(128, seed=5)
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 5)
    api.set_option("solve", api.SOLVE_HOST)
    try:
        ref = api.refine_batch(model, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit)
        api.set_option("profile", 2); api.set_option("sample_period", 2)
        api.profile_reset()
        got = [api.refine_batch(model, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit) for _ in range(4)]
        prof = api.profile_read()
    finally:
        api.set_option("profile", 0); api.set_option("sample_period", 32); api.set_option("solve", api.SOLVE_DEVICE)
    assert all(g[0].tobytes() == ref[0].tobytes() for g in got)
    assert prof["icp_launches"] == 2 and prof["icp_points"] == 2 * int(ref[1].sum()) and prof["icp_kernel_ms"] > 0


@pytest.mark.parametrize("kind", ["proj", "nn"])
def test_host_solve_flag_polling_equals_the_stream_wait(gpu, model, scenario, gscenes, kind):
    """Round 6: with the solve on the host the helper polls ONE flag per pose group in pinned memory (stored by the workgroup that completes the group,
    behind every hypothesis' sums) instead of waiting for the stream.  Records are identical with the polling on and off (option host_poll), through the
    synchronous call and through the slots' helper threads, fixed and early-exit criteria; and no flag ever reached the host before one of its rows
    (each row carries the iteration's tag behind its sums; the library would fall back to the stream wait and count it)."""
    K, proj = scenario["K"], scenario["proj"]
    poses = synth.hypotheses(150, seed=9)
    poses[5] = poses[5].copy(); poses[5].reshape(4, 4)[0, 3] += 1.0e6          # an empty cloud: a hypothesis that never delivers
    got = {}
    try:
        api.set_option("solve", api.SOLVE_HOST)
        for poll in (1, 0):
            api.set_option("host_poll", poll)
            for crit in ((0.0, 0.0, 12), (1e-5, 1e-5, 30)):
                c = api.ICPConvergenceCriteria(*crit)
                sync = api.refine_batch(model, poses, W, H, proj, K, gscenes[kind], c)
                for k in range(3):
                    api.refine_submit(k & 1, model, poses, W, H, proj, K, gscenes[kind], c)
                    if k:
                        slot = api.refine_wait((k - 1) & 1)
                        assert slot[0].tobytes() == sync[0].tobytes() and np.array_equal(slot[1], sync[1])
                api.refine_wait(0)
                got[(poll, crit)] = sync
        for crit in ((0.0, 0.0, 12), (1e-5, 1e-5, 30)):
            assert got[(1, crit)][0].tobytes() == got[(0, crit)][0].tobytes()
        assert api.get_option("stat_flag_overtook") == 0
    finally:
        api.set_option("host_poll", 1); api.set_option("solve", api.SOLVE_HOST)
