"""GPU tests (-m gpu) of cuda_renderer::render / render_host / raw2* (renderer.cu:83-187,189-439; rows a2-a4, f2 of SURVEY 8): the HIP raster against the oracle, bit for bit.
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


@pytest.mark.device_solve
def test_raster_repeated_64_hypotheses_is_deterministic(gpu, model, scenario):
    """The depth resolve is an integer atomicMin: twenty renders of the same 64 hypotheses give the same bits (and the oracle's)."""
    poses = synth.hypotheses(64)
    ref = O.render(scenario["tris"], poses[:3], W, H, scenario["proj"])
    imgs = [np.asarray(api.render_host(model, poses, W, H, scenario["proj"])) for _ in range(20)]
    for im in imgs[1:]:
        assert np.array_equal(im, imgs[0])
    assert np.array_equal(imgs[0][:3].reshape(3, -1), ref.reshape(3, -1))


@pytest.mark.device_solve
@pytest.mark.parametrize("roi", [(0, 0, 1, 1), (639, 479, 1, 1), (320, 240, 1, 1), (0, 0, 640, 1), (0, 0, 1, 480), (300, 200, 37, 53)])
def test_render_roi_extremes(gpu, model, scenario, roi):
    poses = synth.hypotheses(3, seed=1)
    assert np.array_equal(api.render_host(model, poses, W, H, scenario["proj"], roi), O.render(scenario["tris"], poses, W, H, scenario["proj"], roi))


@pytest.mark.parametrize("seed,W,H", [(1, 97, 61), (2, 64, 48), (3, 333, 200), (4, 640, 480), (5, 130, 517), (6, 65, 65)])
def test_render_random_scenes(gpu, seed, W, H):
    rng = np.random.default_rng(seed)
    K = np.array([rng.uniform(0.7, 1.6) * W, 0, W / 2 + rng.uniform(-9, 9), 0, rng.uniform(0.7, 1.6) * W, H / 2 + rng.uniform(-9, 9), 0, 0, 1], np.float32)
    tris = random_mesh(rng, 400, 40.0)
    poses = np.stack([random_pose(rng, d) for d in (300.0, 150.0, 90.0, 600.0, 45.0)])
    poses[4, 2, 3] = 10.0                            # camera inside the soup: vertices behind the camera, huge boxes
    proj = O.compute_proj(K, W, H)
    assert np.array_equal(api.compute_proj(K, W, H), proj)
    ref = O.render(tris, poses, W, H, proj)
    model = api.Model(tris=tris)
    assert np.array_equal(api.render_host(model, poses, W, H, proj), ref)
    # ROI: random crop inside the image
    x0, y0 = int(rng.integers(0, W // 2)), int(rng.integers(0, H // 2))
    roi = (x0, y0, int(rng.integers(1, W - x0 + 1)), int(rng.integers(1, H - y0 + 1)))
    assert np.array_equal(api.render_host(model, poses, W, H, proj, roi), O.render(tris, poses, W, H, proj, roi))


@pytest.mark.parametrize("seed,W,H", [(21, 97, 61), (22, 160, 120)])
def test_render_mesh_with_non_finite_and_absurd_vertices(gpu, seed, W, H):
    """Triangles with NaN, infinite, 1e30 and denormal-small vertices among ordinary ones: the image equals the oracle's (the loop
    bounds of renderer.cu:100-125 decide what such a triangle touches), with and without an ROI, and through the fused path."""
    rng = np.random.default_rng(seed)
    K = np.array([1.1 * W, 0, W / 2, 0, 1.1 * W, H / 2, 0, 0, 1], np.float32)
    tris = random_mesh(rng, 300, 40.0)
    tris[10, 0, 0] = np.nan
    tris[11, 1] = np.nan
    tris[12, 2, 2] = np.inf
    tris[13, 0] = [np.inf, -np.inf, np.inf]
    tris[14] *= 1e30
    tris[15, 1] = [1e30, 1e30, 1e30]
    tris[16] *= 1e-38
    tris[17, 2, 1] = -np.inf
    tris[18] = 0.0
    poses = np.stack([random_pose(rng, d) for d in (300.0, 120.0, 60.0)])
    proj = O.compute_proj(K, W, H)
    ref = O.render(tris, poses, W, H, proj)
    assert (ref > 0).sum() > 100
    model = api.Model(tris=tris)
    assert np.array_equal(api.render_host(model, poses, W, H, proj), ref)
    roi = (W // 5, H // 4, W // 2, H // 2)
    assert np.array_equal(api.render_host(model, poses, W, H, proj, roi), O.render(tris, poses, W, H, proj, roi))
    # fused path (per-pose pixel boxes from the mesh's box -- which is not finite here): cloud sizes as the oracle renders them
    scene = api.Scene_projective().init_Scene_projective_cuda(ref[0], K, W, H)
    _, sizes = api.refine_batch(model, poses, W, H, proj, K, scene, api.ICPConvergenceCriteria(0.0, 0.0, 2))
    assert [int(s) for s in sizes] == [int((r > 0).sum()) for r in ref]


@pytest.mark.parametrize("W,H", [(8192, 2048), (2048, 8192), (4096, 4096)])
def test_largest_frames(gpu, W, H):
    """The largest frames the pixel packing admits (8192 on a side, 2^24 pixels): render against the oracle bit for bit, and the fused
    path's clouds and first pass against it (projective scene made from the render)."""
    rng = np.random.default_rng(W + H)
    f = 0.9 * max(W, H)
    K = np.array([f, 0, W / 2, 0, f, H / 2, 0, 0, 1], np.float32)
    tris = random_mesh(rng, 300, 40.0)
    poses = np.stack([random_pose(rng, d) for d in (160.0, 110.0)])
    proj = O.compute_proj(K, W, H)
    ref = O.render(tris, poses, W, H, proj)
    assert (ref[0] > 0).sum() > 100000
    model = api.Model(tris=tris)
    got = api.render_host(model, poses, W, H, proj)
    assert np.array_equal(got, ref)
    scene = api.Scene_projective().init_Scene_projective_cuda(ref[0], K, W, H)
    crit = (0.0, 0.0, 1)
    res, sizes = api.refine_batch(model, poses, W, H, proj, K, scene, api.ICPConvergenceCriteria(*crit))
    assert [int(s) for s in sizes] == [int((r > 0).sum()) for r in ref]
    oscene = O.ProjScene(ref[0], K)
    ppb = api.get_option("points_per_block")
    for i in range(2):
        want, _, _, _ = O.icp(O.depth2cloud(ref[i], K), oscene, crit, O.SUM_CANONICAL, ppb)
        assert res[i]["fitness"] == want["fitness"] and np.allclose(res[i]["T"], want["T"], rtol=0, atol=1e-4)
    # the asynchronous path (device solve) sizes its sub-batches so that the workspace of one stays within a few GiB: 150 hypotheses
    # of a 16 M-pixel frame run as three sub-batches of 50..64; the first and the last equal the synchronous records
    if W * H == 1 << 24:
        many = np.concatenate([poses] * 75)
        api.set_option("solve", api.SOLVE_DEVICE)
        try:
            res2, sizes2 = api.refine_batch(model, many, W, H, proj, K, scene, api.ICPConvergenceCriteria(*crit))
        finally:
            api.set_option("solve", api.SOLVE_HOST)
        assert res2[:2].tobytes() == res.tobytes() and res2[-2:].tobytes() == res.tobytes() and np.array_equal(sizes2[-2:], sizes)


@pytest.mark.parametrize("name,Kd", [("zero", [0] * 9), ("nan", [np.nan] * 9), ("mirrored", [-50, 0, 24, 0, -50, 16, 0, 0, 1])])
def test_degenerate_intrinsics(gpu, name, Kd):
    """Intrinsics that are all zero, NaN or negative: the render equals the oracle's (compute_proj and the viewport arithmetic decide),
    scenes can be made from them and refinement runs to the end -- nothing faults."""
    rng = np.random.default_rng(6)
    W, H = 48, 32
    K = np.array([1.1 * W, 0, W / 2, 0, 1.1 * W, H / 2, 0, 0, 1], np.float32)
    tris = random_mesh(rng, 200, 40.0)
    model = api.Model(tris=tris)
    poses = np.stack([random_pose(rng, d) for d in (300.0, 150.0, 220.0, 90.0)])
    scene_depth = O.render(tris, poses[:1], W, H, O.compute_proj(K, W, H))[0]
    Kd = np.array(Kd, np.float32)
    pj = api.compute_proj(Kd, W, H)
    assert np.array_equal(pj, O.compute_proj(Kd, W, H), equal_nan=True)
    assert np.array_equal(api.render_host(model, poses, W, H, pj), O.render(tris, poses, W, H, O.compute_proj(Kd, W, H)))
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 3)
    for scene in (api.Scene_projective().init_Scene_projective_cuda(scene_depth, Kd, W, H), api.Scene_nn().init_Scene_nn_cuda(scene_depth, Kd)):
        res, sizes = api.refine_batch(model, poses, W, H, pj, Kd, scene, crit)
        assert len(res) == 4
