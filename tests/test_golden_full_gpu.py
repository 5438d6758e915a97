"""GPU tests (-m gpu): EVERY hypothesis of BASELINE.json configs[1..4] against full-size oracle fixtures (tests/golden/config{1,2,3,4}.npz,
made in the build container by tools/make_golden.py from oracle/pose_oracle.c -- /root/reference/cuda_icp/icp.cpp:125-188 behind render_cpu
and depth2cloud_cpu -- in its canonical summation mode).  Cloud sizes and fitness (= inlier count / cloud size, icp.cpp:186) array_equal,
inlier rmse 1e-6 relative, transforms within 1e-4 (north_star).  The HIP path is called through the C ABI (pose_refine_amd.api):
the synchronous call and the asynchronous slots bench.py drives, fixed-20 criteria and the reference's defaults (icp.h:42-45: hypotheses
leave the loop at different passes)."""
import os

import numpy as np
import pytest

from pose_refine_amd import api, dist, synth

pytestmark = pytest.mark.gpu
TOL_T = 1e-4          # north_star: "transforms within 1e-4"
FIXED = (0.0, 0.0, 20)
DEFAULT = (1e-5, 1e-5, 30)


@pytest.fixture(scope="module")
def dev():
    api.init(0)
    api.set_option("solve", api.SOLVE_DEVICE)
    yield True
    api.set_option("solve", api.SOLVE_HOST)


def load(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name))
    assert int(g["ppb"]) == api.get_option("points_per_block"), "fixtures follow the library's default points_per_block: re-run tools/make_golden.py"
    return g


def hold(res, sizes, g, tag):
    """every hypothesis of a batch against the fixture"""
    assert np.array_equal(sizes, g[tag + "_sizes"])
    assert np.array_equal(res["fitness"], g[tag + "_fitness"])                       # inlier counts bit-exact (count / N in float, icp.cpp:186)
    inl = np.rint(res["fitness"].astype(np.float64) * sizes).astype(np.int64)
    assert np.array_equal(inl, np.rint(g[tag + "_fitness"].astype(np.float64) * g[tag + "_sizes"]).astype(np.int64))
    assert np.allclose(res["inlier_rmse"], g[tag + "_rmse"], rtol=1e-6, atol=0)
    err = np.abs(res["T"].reshape(len(sizes), 16) - g[tag + "_T"]).max()
    assert err <= TOL_T, err
    return err


def obj06(golden_dir):
    model = api.Model(os.path.join(golden_dir, "obj_06.ply"))
    K = synth.K_TEST
    W, H = synth.WIDTH, synth.HEIGHT
    proj = api.compute_proj(K, W, H)
    scene_depth = api.render_host(model, synth.scene_pose()[None], W, H, proj)[0]
    return model, K, W, H, proj, scene_depth


@pytest.mark.parametrize("crit,tag", [(FIXED, "fixed20"), (DEFAULT, "default")])
def test_config1_all_256_hypotheses_projective(dev, golden_dir, crit, tag):
    g = load(golden_dir, "config1.npz")
    model, K, W, H, proj, sd = obj06(golden_dir)
    scene = api.Scene_projective().init_Scene_projective_cuda(sd, K)
    poses = synth.hypotheses(256)
    res, sizes = api.refine_batch(model, poses, W, H, proj, K, scene, api.ICPConvergenceCriteria(*crit))
    hold(res, sizes, g, tag)
    api.refine_submit(0, model, poses, W, H, proj, K, scene, api.ICPConvergenceCriteria(*crit))      # what bench.py times
    ares, asizes = api.refine_wait(0)
    hold(ares, asizes, g, tag)
    if tag == "default":                                          # the early exit is exercised: not every hypothesis runs the same number of passes
        assert len(np.unique(g["default_T"], axis=0)) == 256 and not np.array_equal(g["default_T"], g["fixed20_T"])


@pytest.mark.parametrize("solve", [api.SOLVE_DEVICE, api.SOLVE_HOST])
def test_config1_all_256_both_solve_modes(dev, golden_dir, solve):
    g = load(golden_dir, "config1.npz")
    model, K, W, H, proj, sd = obj06(golden_dir)
    scene = api.Scene_projective().init_Scene_projective_cuda(sd, K)
    api.set_option("solve", solve)
    try:
        res, sizes = api.refine_batch(model, synth.hypotheses(256), W, H, proj, K, scene, api.ICPConvergenceCriteria(*FIXED))
    finally:
        api.set_option("solve", api.SOLVE_DEVICE)
    hold(res, sizes, g, "fixed20")


@pytest.mark.parametrize("crit,tag", [(FIXED, "fixed20"), (DEFAULT, "default")])
def test_config2_all_256_hypotheses_kdtree(dev, golden_dir, crit, tag):
    g = load(golden_dir, "config2.npz")
    model, K, W, H, proj, sd = obj06(golden_dir)
    scene = api.Scene_nn().init_Scene_nn_cuda(sd, K)
    poses = synth.hypotheses(256)
    res, sizes = api.refine_batch(model, poses, W, H, proj, K, scene, api.ICPConvergenceCriteria(*crit))
    hold(res, sizes, g, tag)
    api.refine_submit(1, model, poses, W, H, proj, K, scene, api.ICPConvergenceCriteria(*crit))
    ares, asizes = api.refine_wait(1)
    hold(ares, asizes, g, tag)


def test_config2_all_256_device_built_scene(dev, golden_dir):
    """the same 256 hypotheses against a scene whose normals, points and kd-tree were made ON THE DEVICE"""
    g = load(golden_dir, "config2.npz")
    model, K, W, H, proj, sd = obj06(golden_dir)
    scene = api.Scene_nn().init_Scene_nn_device(api.DeviceVector.from_host(sd.reshape(-1)), K, W, H)
    res, sizes = api.refine_batch(model, synth.hypotheses(256), W, H, proj, K, scene, api.ICPConvergenceCriteria(*FIXED))
    hold(res, sizes, g, "fixed20")


def test_config3_all_512_hypotheses_of_rank_5(dev, golden_dir):
    g = load(golden_dir, "config3.npz")
    model, K, W, H, proj, sd = obj06(golden_dir)
    scene = api.Scene_projective().init_Scene_projective_cuda(sd, K)
    first, count = dist.shard_bounds(4096, 5, 8)
    assert first == int(g["first"]) and count == 512
    res, sizes = api.refine_batch(model, synth.hypotheses(count, first=first), W, H, proj, K, scene, api.ICPConvergenceCriteria(*FIXED))
    hold(res, sizes, g, "fixed20")


def test_config4_all_128_hypotheses_1m_triangles(dev, golden_dir):
    g = load(golden_dir, "config4.npz")
    W, H = 1280, 720
    K = synth.intrinsics_720p()
    model = api.Model(tris=synth.uv_sphere_mesh())
    proj = api.compute_proj(K, W, H)
    sd = api.render_host(model, synth.scene_pose()[None], W, H, proj)[0]
    scene = api.Scene_projective().init_Scene_projective_cuda(sd, K, W, H)
    res, sizes = api.refine_batch(model, synth.hypotheses(128), W, H, proj, K, scene, api.ICPConvergenceCriteria(*FIXED))
    hold(res, sizes, g, "fixed20")
