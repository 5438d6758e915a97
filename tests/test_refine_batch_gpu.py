"""GPU tests (-m gpu) of the fused path render -> cloud -> ICP for a batch of hypotheses (test.cpp:143-172 per hypothesis), full frame and ROI (row f3), against the oracle.
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


# ---- ROI refinement: renderer.h:199 + icp.h:57-60 (cuda_renderer/test.cpp:116-157 is the reference's ROI test) ---------
@pytest.mark.device_solve
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


@pytest.mark.device_solve
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


@pytest.mark.device_solve
def test_pathological_hypotheses_do_no_harm(gpu, model, scenario, gscenes):
    """Hypotheses with NaN / infinite / zero / mirrored / astronomically scaled matrices in a batch: nothing faults, the other hypotheses
    of the batch are refined bit for bit as without them (synchronous path, both solves, both scenes, and on the asynchronous slots), and
    the render of the finite oddities equals the oracle's (a mirrored object, one behind the camera, a speck, a foreign last row)."""
    good = synth.hypotheses(40, seed=3)
    bad, idx_bad = pathological_hypotheses(good)
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


@pytest.mark.device_solve
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


@pytest.mark.parametrize("seed,W,H", [(11, 97, 61), (12, 333, 200), (13, 640, 480), (14, 70, 130)])
def test_fused_pipeline_random_scenes(gpu, seed, W, H):
    """render -> cloud -> ICP on random geometry: cloud sizes, inlier counts (fitness) and transforms vs the oracle, for both
    associations, both solvers, both raster modes."""
    rng = np.random.default_rng(seed)
    f = rng.uniform(0.9, 1.3) * W
    K = np.array([f, 0, W / 2 + rng.uniform(-5, 5), 0, f * rng.uniform(0.95, 1.05), H / 2 + rng.uniform(-5, 5), 0, 0, 1], np.float32)
    tris = random_mesh(rng, 600, 45.0)
    base = random_pose(rng, 320.0)
    proj = O.compute_proj(K, W, H)
    scene_depth = O.render(tris, base[None], W, H, proj)[0]
    if seed % 2 == 0:
        scene_depth = scene_depth.astype(np.uint16)                # CV_16U scenes
    poses = []
    for _ in range(7):
        p = base.copy()
        p[:3, 3] += rng.normal(size=3).astype(np.float32) * 6.0
        poses.append(p)
    off = base.copy(); off[0, 3] += 4000.0                           # renders nothing: empty cloud
    poses = np.stack(poses + [off])
    model = api.Model(tris=tris)
    crit = (0.0, 0.0, 6)
    for kind in ("proj", "nn"):
        if kind == "proj":
            gs = api.Scene_projective().init_Scene_projective_cuda(scene_depth, K, W, H)
            osc = O.ProjScene(scene_depth, K)
        else:
            if int((scene_depth > 0).sum()) == 0:
                continue
            gs = api.Scene_nn().init_Scene_nn_cuda(scene_depth, K)
            osc = O.NNScene(scene_depth, K)
        ores, osizes, _ = O.refine_batch(tris, poses, W, H, proj, K, osc, crit, O.SUM_CANONICAL, api.get_option("points_per_block"))
        for solve in (api.SOLVE_HOST, api.SOLVE_DEVICE):
            for raster_mode in (0, 1):
                api.set_option("solve", solve); api.set_option("raster_mode", raster_mode)
                try:
                    res, sizes = api.refine_batch(model, poses, W, H, proj, K, gs, api.ICPConvergenceCriteria(*crit))
                finally:
                    api.set_option("solve", api.SOLVE_HOST); api.set_option("raster_mode", 0)
                assert np.array_equal(sizes, osizes), (kind, solve, raster_mode)
                assert sizes[-1] == 0 and res["fitness"][-1] == 0
                assert np.array_equal(res["fitness"], ores["fitness"]), (kind, solve, raster_mode)
                assert np.allclose(res["T"], ores["T"], rtol=0, atol=1e-4), (kind, solve, raster_mode)


def test_more_hypotheses_than_a_launch_has_rows_and_empty_models(gpu):
    """On a 48 x 32 frame: 70 000 hypotheses in one render call and 40 000 / 6 000 in one refinement call (projective / kd-tree; the
    hypothesis index is the y dimension of the launches, so such calls run in pieces) -- spot checks against the oracle's render and
    against the same hypotheses refined in a call of their own; a model of no triangles (no array at all) and of one triangle."""
    rng = np.random.default_rng(5)
    W, H = 48, 32
    K = np.array([1.1 * W, 0, W / 2, 0, 1.1 * W, H / 2, 0, 0, 1], np.float32)
    proj = O.compute_proj(K, W, H)
    tris = random_mesh(rng, 200, 40.0)
    model = api.Model(tris=tris)
    base = [random_pose(rng, d) for d in (300.0, 150.0, 220.0)]
    P = 70000
    poses = np.stack([base[i % 3] for i in range(P)]).copy()
    poses[:, 0, 3] += (np.arange(P) % 97).astype(np.float32) * 0.5
    d = api.render_host(model, poses, W, H, proj)
    for i in (0, 1, 32767, 32768, 65535, 65536, P - 1):
        assert np.array_equal(d[i], O.render(tris, poses[i:i + 1], W, H, proj)[0]), i
    scene_depth = O.render(tris, np.stack(base[:1]), W, H, proj)[0]
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 3)
    try:
        for kind, Pn in (("proj", 40000), ("nn", 6000)):
            scene = api.Scene_projective().init_Scene_projective_cuda(scene_depth, K, W, H) if kind == "proj" else api.Scene_nn().init_Scene_nn_cuda(scene_depth, K)
            for solve in (api.SOLVE_DEVICE, api.SOLVE_HOST):
                api.set_option("solve", solve)
                out = api.refine_batch(model, poses[:Pn], W, H, proj, K, scene, crit)
                for i0 in (0, 511, 512, 20000 % Pn, Pn - 3):
                    ref = api.refine_batch(model, poses[i0:i0 + 3], W, H, proj, K, scene, crit)
                    assert out[0][i0:i0 + 3].tobytes() == ref[0].tobytes() and np.array_equal(out[1][i0:i0 + 3], ref[1]), (kind, solve, i0)
    finally:
        api.set_option("solve", api.SOLVE_HOST)
    scene = api.Scene_projective().init_Scene_projective_cuda(scene_depth, K, W, H)
    for n in (0, 1):
        m = api.Model(tris=tris[:n])
        assert np.array_equal(api.render_host(m, poses[:5], W, H, proj), O.render(tris[:n], poses[:5], W, H, proj) if n else np.zeros((5, H, W), np.int32))
        res, sizes = api.refine_batch(m, poses[:5], W, H, proj, K, scene, crit)
        assert [int(s) for s in sizes] == [int((r > 0).sum()) for r in O.render(tris[:n], poses[:5], W, H, proj)] if n else not sizes.any()


@pytest.mark.device_solve
@pytest.mark.parametrize("W,H", [(96, 64), (97, 61)])
def test_packed_clouds_and_boxes_at_their_capacity_bound(gpu, W, H):
    """The fused paths pack the clouds and the pixel boxes of a batch one behind the other; the capacity they reserve is the largest
    box per hypothesis.  A wall that fills the frame makes every box the whole frame and every cloud W x H points -- the sum of the
    packed sizes then EQUALS the reservation -- next to a hypothesis that renders nothing and one that fills part of the frame:
    sizes, inlier counts and transforms of the synchronous call and of the asynchronous slots against the oracle."""
    K = np.array([1.2 * W, 0, W / 2, 0, 1.2 * W, H / 2, 0, 0, 1], np.float32)
    proj = O.compute_proj(K, W, H)
    s = 4000.0                                                        # a wall far larger than the view, 300 mm in front of the camera
    wall = np.array([[[-s, -s, 0], [s, -s, 0], [s, s, 0]], [[-s, -s, 0], [s, s, 0], [-s, s, 0]]], np.float32)
    small = wall * np.float32(0.01)                                   # ... and a 80 mm patch of it, 20 mm nearer
    small[:, :, 2] -= 20.0
    tris = np.concatenate([wall, small]).astype(np.float32)
    base = np.eye(4, dtype=np.float32); base[2, 3] = 300.0
    scene_depth = O.render(tris, base[None], W, H, proj)[0]
    poses = []
    for k in range(11):
        p = base.copy(); p[0, 3] += 0.7 * k; p[2, 3] += 1.5 * k
        poses.append(p)
    gone = base.copy(); gone[2, 3] = -500.0                           # behind the camera: renders nothing
    poses = np.stack(poses[:5] + [gone] + poses[5:])
    crit = (0.0, 0.0, 4)
    osc = O.ProjScene(scene_depth, K)
    ores, osizes, _ = O.refine_batch(tris, poses, W, H, proj, K, osc, crit, O.SUM_CANONICAL, api.get_option("points_per_block"))
    assert osizes.max() == W * H and osizes[5] == 0
    model = api.Model(tris=tris)
    gs = api.Scene_projective().init_Scene_projective_cuda(scene_depth, K, W, H)
    res, sizes = api.refine_batch(model, poses, W, H, proj, K, gs, api.ICPConvergenceCriteria(*crit))
    assert np.array_equal(sizes, osizes) and np.array_equal(res["fitness"], ores["fitness"])
    assert np.allclose(res["T"], ores["T"], rtol=0, atol=1e-4)
    for b in (0, 1, 0):
        api.refine_submit(b, model, poses, W, H, proj, K, gs, api.ICPConvergenceCriteria(*crit))
        ares, asizes = api.refine_wait(b)
        assert np.array_equal(asizes, osizes) and np.array_equal(ares["fitness"], ores["fitness"])
        assert np.array_equal(ares["T"], res["T"])
