"""BASELINE.json configs[2], configs[3] and configs[4] at their full / per-GPU size (-m gpu).

configs[2]: obj_06.ply, 256 hypotheses, kd-tree nearest-neighbour association.

configs[3]: obj_06.ply, 4096 hypotheses over 8 GPUs -> 512 per GPU, projective association.
configs[4]: 1M-triangle synthetic mesh (UV sphere 1000 x 500 quads, bumpy radius), 1280x720,
            1024 hypotheses over 8 GPUs -> 128 per GPU.  This is the case that stresses the raster differently:
            a 36 MB triangle stream per hypothesis and many sub-pixel triangles fighting over the same pixels
            (cuda_renderer/renderer.cu:124-149 atomicMin).

The oracle runs on a few hypotheses (it needs seconds per 1M-triangle render); everything else is checked through
size-independent properties: batch results do not depend on batch composition or order, every shard of the seeded
stream equals the same hypotheses refined alone, and the result is a rigid transform.
"""
import os

import numpy as np
import pytest

import oracle_lib as O
from pose_refine_amd import api, dist, synth

pytestmark = pytest.mark.gpu
TOL_T = 1e-4


@pytest.fixture(scope="module")
def gpu():
    api.init(0)
    api.set_option("solve", api.SOLVE_DEVICE)
    yield True
    api.set_option("solve", api.SOLVE_HOST)


def rigid(T):
    R = T.reshape(-1, 4, 4)[:, :3, :3].astype(np.float64)
    return np.allclose(np.einsum("pij,pik->pjk", R, R), np.eye(3), atol=1e-5) and np.allclose(np.linalg.det(R), 1.0, atol=1e-5)


def test_config3_share_512_hypotheses_projective(gpu, scenario, golden_dir):
    W, H, K = synth.WIDTH, synth.HEIGHT, scenario["K"]
    world, n_total = 8, 4096
    model = api.Model(os.path.join(golden_dir, "obj_06.ply"))
    scene = api.Scene_projective().init_Scene_projective_cuda(scenario["depth"][1], K)
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 20)
    rank = 5                                                      # any rank's shard of the global seeded stream
    first, count = dist.shard_bounds(n_total, rank, world)
    assert (first, count) == api.shard_range(n_total, rank, world) == (2560, 512)
    poses = synth.hypotheses(count, first=first)
    res, sizes = api.refine_batch(model, poses, W, H, scenario["proj"], K, scene, crit)
    assert rigid(res["T"]) and (sizes > 15000).all()
    # the asynchronous two-slot form (what bench.py runs) is bit-identical
    api.refine_submit(1, model, poses, W, H, scenario["proj"], K, scene, crit)
    ares, asizes = api.refine_wait(1)
    assert np.array_equal(asizes, sizes) and ares.tobytes() == res.tobytes()
    # composition / order independence
    perm = np.random.default_rng(11).permutation(count)
    pres, psizes = api.refine_batch(model, poses[perm], W, H, scenario["proj"], K, scene, crit)
    assert np.array_equal(psizes, sizes[perm]) and pres.tobytes() == res[perm].tobytes()
    # hypotheses 0 / 255 / 511 of the shard against the oracle (same reduction tree: inlier counts bit-exact)
    pick = np.array([0, 255, 511])
    ores, osizes, _ = O.refine_batch(scenario["tris"], poses[pick], W, H, scenario["proj"], K, scenario["proj_scene"],
                                     (0.0, 0.0, 20), O.SUM_CANONICAL, api.get_option("points_per_block"))
    assert np.array_equal(sizes[pick], osizes)
    assert np.array_equal(res["fitness"][pick], ores["fitness"])
    assert np.allclose(res["T"][pick], ores["T"], rtol=0, atol=TOL_T)
    # shard boundaries: the last hypothesis of rank 4 and the first of rank 6 are the neighbours in the global stream
    edge = synth.hypotheses(514, first=first - 1)
    eres, _ = api.refine_batch(model, edge, W, H, scenario["proj"], K, scene, crit)
    assert eres[1:513].tobytes() == res.tobytes()


@pytest.fixture(scope="module")
def config5(gpu):
    W, H = 1280, 720
    K = synth.intrinsics_720p()
    tris = synth.uv_sphere_mesh()
    assert len(tris) == 1_000_000
    model = api.Model(tris=tris)
    proj = api.compute_proj(K, W, H)
    scene_depth = api.render_host(model, synth.scene_pose()[None], W, H, proj)[0]
    return dict(W=W, H=H, K=K, tris=tris, model=model, proj=proj, scene_depth=scene_depth,
                scene=api.Scene_projective().init_Scene_projective_cuda(scene_depth, K, W, H))


def test_config2_256_hypotheses_kdtree(gpu, scenario, golden_dir):
    """configs[2] at full size.  The oracle's kd-tree ICP needs ~1.5 s per hypothesis on the CPU: three hypotheses go to it, the
    rest is held by properties -- order and batch composition do not matter, the slots and the synchronous path agree."""
    model = api.Model(os.path.join(golden_dir, "obj_06.ply"))
    W, H, K = synth.WIDTH, synth.HEIGHT, scenario["K"]
    scene = api.Scene_nn().init_Scene_nn_cuda(scenario["depth"][1], K)
    P = 256
    poses = synth.hypotheses(P)
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 20)
    res, sizes = api.refine_batch(model, poses, W, H, scenario["proj"], K, scene, crit)
    assert rigid(res["T"]) and np.all(res["fitness"] > 0.5)
    perm = np.random.default_rng(5).permutation(P)
    res_p, sizes_p = api.refine_batch(model, poses[perm], W, H, scenario["proj"], K, scene, crit)
    assert np.array_equal(sizes[perm], sizes_p) and res[perm].tobytes() == res_p.tobytes()
    for i in (0, 101, 255):
        one, s1 = api.refine_batch(model, poses[i:i + 1], W, H, scenario["proj"], K, scene, crit)
        assert s1[0] == sizes[i] and one.tobytes() == res[i:i + 1].tobytes()
    api.set_option("nn_count", 1)                                # instrumented = synchronous path
    try:
        res_s, sizes_s = api.refine_batch(model, poses, W, H, scenario["proj"], K, scene, crit)
    finally:
        api.set_option("nn_count", 0)
    assert np.array_equal(sizes, sizes_s) and res.tobytes() == res_s.tobytes()
    pick = [0, 101, 255]
    ores, osizes, _ = O.refine_batch(scenario["tris"], poses[pick], W, H, scenario["proj"], K, scenario["nn_scene"], (0.0, 0.0, 20),
                                     O.SUM_CANONICAL, api.get_option("points_per_block"))
    assert np.array_equal(sizes[pick], osizes) and np.array_equal(res["fitness"][pick], ores["fitness"])
    assert np.allclose(res["T"][pick], ores["T"], rtol=0, atol=TOL_T)


def test_render_is_the_pixelwise_minimum_over_any_split_of_the_mesh(gpu, scenario, golden_dir):
    """Size-independent property of the raster (renderer.cu:124-149: atomicMin per pixel): rendering the whole mesh equals the
    pixelwise minimum (over covered pixels) of rendering any two parts of it -- 64 hypotheses, full frame, three different splits."""
    tris = scenario["tris"].reshape(-1, 9)
    W, H = synth.WIDTH, synth.HEIGHT
    poses = synth.hypotheses(64)
    full = api.render_host(tris, poses, W, H, scenario["proj"])
    big = np.iinfo(np.int32).max
    for cut in (1, len(tris) // 3, len(tris) - 7):
        a = api.render_host(tris[:cut], poses, W, H, scenario["proj"]).astype(np.int64)
        b = api.render_host(tris[cut:], poses, W, H, scenario["proj"]).astype(np.int64)
        a[a == 0] = big; b[b == 0] = big
        m = np.minimum(a, b); m[m == big] = 0
        assert np.array_equal(m.astype(np.int32), full)


def test_config5_scene_render_bit_exact(config5):
    c = config5
    oproj = O.compute_proj(c["K"], c["W"], c["H"])
    assert np.array_equal(oproj, c["proj"])
    ref = O.render(c["tris"], synth.scene_pose()[None], c["W"], c["H"], oproj)[0]
    assert np.array_equal(ref, c["scene_depth"])
    assert (ref > 0).sum() > 100_000 and ref.max() < 2000           # depth stays below get_normal's 2000 mm gate (common.cpp:33)
    # the fused path's raster (pixel boxes, no full-frame clear) sees the same image: cloud size == valid pixels
    _, sizes = api.refine_batch(c["model"], synth.scene_pose()[None], c["W"], c["H"], c["proj"], c["K"], c["scene"],
                                api.ICPConvergenceCriteria(0.0, 0.0, 0))
    assert sizes[0] == (ref > 0).sum()


def test_config5_share_128_hypotheses(config5):
    c = config5
    W, H, K = c["W"], c["H"], c["K"]
    P = 128                                                          # 1024 hypotheses over 8 GPUs
    poses = synth.hypotheses(P)
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 20)
    res, sizes = api.refine_batch(c["model"], poses, W, H, c["proj"], K, c["scene"], crit)
    assert rigid(res["T"]) and (sizes > 100_000).all()
    # two hypotheses end to end against the oracle: render + cloud + 21-pass ICP
    oscene = O.ProjScene(c["scene_depth"], K)
    pick = np.array([0, 77])
    ores, osizes, _ = O.refine_batch(c["tris"], poses[pick], W, H, c["proj"], K, oscene, (0.0, 0.0, 20), O.SUM_CANONICAL,
                                     api.get_option("points_per_block"))
    assert np.array_equal(sizes[pick], osizes)
    assert np.array_equal(res["fitness"][pick], ores["fitness"])
    assert np.allclose(res["inlier_rmse"][pick], ores["inlier_rmse"], rtol=1e-6, atol=0)
    assert np.allclose(res["T"][pick], ores["T"], rtol=0, atol=TOL_T)
    # permutation / composition over all 128
    perm = np.random.default_rng(5).permutation(P)
    pres, psizes = api.refine_batch(c["model"], poses[perm], W, H, c["proj"], K, c["scene"], crit)
    assert np.array_equal(psizes, sizes[perm]) and pres.tobytes() == res[perm].tobytes()
    one, s1 = api.refine_batch(c["model"], poses[100:101], W, H, c["proj"], K, c["scene"], crit)
    assert s1[0] == sizes[100] and one.tobytes() == res[100:101].tobytes()
    # the host-solve loop (the reference's structure) gives the same bits at this size too
    api.set_option("solve", api.SOLVE_HOST)
    try:
        hres, hsizes = api.refine_batch(c["model"], poses[:16], W, H, c["proj"], K, c["scene"], crit)
    finally:
        api.set_option("solve", api.SOLVE_DEVICE)
    assert np.array_equal(hsizes, sizes[:16]) and hres.tobytes() == res[:16].tobytes()
