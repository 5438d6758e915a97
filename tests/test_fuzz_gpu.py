"""Randomised differential tests (-m gpu): odd image sizes, random meshes (degenerate, behind-camera, huge and sub-pixel
triangles), random intrinsics, ROI / stride / offsets, uint16 scenes -- HIP path vs the CPU oracle, bit for bit."""
import numpy as np
import pytest

import oracle_lib as O
from pose_refine_amd import api

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    api.init(0)
    return True


def random_mesh(rng, n, scale):
    """Triangle soup around the origin: mixed sizes, a few exactly degenerate triangles, duplicated triangles."""
    centers = rng.normal(size=(n, 1, 3)) * scale
    size = np.exp(rng.uniform(np.log(0.002), np.log(0.5), size=(n, 1, 1))) * scale
    tris = (centers + rng.normal(size=(n, 3, 3)) * size).astype(np.float32)
    tris[0, 1] = tris[0, 0]                         # two equal vertices -> zero area
    tris[1, 2] = tris[1, 1] = tris[1, 0]            # a point
    tris[2] = tris[3]                               # duplicate
    return np.ascontiguousarray(tris)


def random_pose(rng, dist):
    a = rng.normal(size=3); a /= np.linalg.norm(a)
    ang = rng.uniform(0, np.pi)
    Kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = R.astype(np.float32)
    T[:3, 3] = (rng.normal(size=3) * dist * 0.08 + np.array([0, 0, dist])).astype(np.float32)
    return T


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


@pytest.mark.parametrize("seed", [21, 22, 23])
def test_depth2cloud_random_images(gpu, seed):
    rng = np.random.default_rng(seed)
    W, H = int(rng.integers(3, 300)), int(rng.integers(3, 300))
    K = np.array([500.0, 0, W / 2, 0, 510.0, H / 2, 0, 0, 1], np.float32)
    d = np.where(rng.random((H, W)) < 0.3, rng.integers(1, 3000, (H, W)), 0).astype(np.int32)
    d[rng.integers(0, H), :] = -7                                  # negative depths are "not > 0"
    for dtype in (np.int32, np.uint16):
        dd = d.astype(dtype) if dtype == np.int32 else np.clip(d, 0, 65535).astype(np.uint16)
        dev = api.DeviceVector.from_host(dd.reshape(-1))
        for stride, tlx, tly in ((1, 0, 0), (1, 17, 5), (2, 0, 0), (3, 1, 2)):
            got = api.depth2cloud(dev, W, H, K, stride, tlx, tly, dtype=dtype).to_host().reshape(-1, 3)
            assert np.array_equal(got, O.depth2cloud(dd, K, stride, tlx, tly)), (dtype, stride)


def test_async_slots_random_job_stream(gpu):
    """Random batch sizes / poses / criteria alternate over the two asynchronous slots (workspaces grow and shrink, the grid
    hint of a batch comes from whatever ran before it, sub-batches of 256, hypotheses with empty clouds) and every batch
    is compared bit for bit with the synchronous path."""
    import os
    from pose_refine_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    model = api.Model(os.path.join(root, "tests", "golden", "obj_06.ply"))
    K = synth.K_TEST; W, H = 640, 480
    proj = api.compute_proj(K, W, H)
    sd = api.render_host(model, synth.scene_pose()[None], W, H, proj)[0]
    scene = api.Scene_projective().init_Scene_projective_cuda(sd, K)
    rng = np.random.default_rng(7)
    jobs = []
    for i in range(14):
        P = int(rng.choice([1, 3, 31, 33, 64, 65, 200, 300, 513]))
        poses = synth.hypotheses(P, seed=100 + i)
        if rng.random() < 0.5:
            poses.reshape(-1, 4, 4)[:, 2, 3] += float(rng.choice([0.0, 400.0, 1500.0, -300.0]))
        if P > 2 and rng.random() < 0.3:
            poses.reshape(-1, 4, 4)[1, 0, 3] += 1e6                # off-screen hypothesis -> empty cloud
        crit = (api.ICPConvergenceCriteria(0.0, 0.0, int(rng.choice([0, 3, 20]))) if rng.random() < 0.7
                else api.ICPConvergenceCriteria(1e-5, 1e-5, 30))
        jobs.append((poses, crit))
    api.set_option("solve", api.SOLVE_DEVICE)
    api.set_option("sub_batch", 256)
    try:
        api.set_option("profile", 1)                               # forces the synchronous path
        refs = [api.refine_batch(model, p, W, H, proj, K, scene, c) for p, c in jobs]
        api.set_option("profile", 0)
        got, inflight = [None] * len(jobs), [None, None]
        for i, (p, c) in enumerate(jobs):
            b = i & 1
            if inflight[b] is not None:
                got[inflight[b]] = api.refine_wait(b)
            api.refine_submit(b, model, p, W, H, proj, K, scene, c)
            inflight[b] = i
        for b in (0, 1):
            if inflight[b] is not None:
                got[inflight[b]] = api.refine_wait(b)
        for i, (g, r) in enumerate(zip(got, refs)):
            assert np.array_equal(g[1], r[1]), i
            assert g[0].tobytes() == r[0].tobytes(), (i, len(jobs[i][0]))
    finally:
        api.set_option("profile", 0)
        api.set_option("sub_batch", 512)
        api.set_option("solve", api.SOLVE_HOST)


def test_async_slots_with_two_alternating_kdtree_scenes(gpu):
    """Kd-tree batches on the two slots against TWO scenes: the traversal records and the pixel grid are shared by the slots and
    hold one scene at a time, so a batch for the other scene rebuilds them while the previous batch may still be in flight
    (the rebuild drains the slots first).  Mixed with projective batches; every result bit for bit equal to the synchronous path."""
    import os
    from pose_refine_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    model = api.Model(os.path.join(root, "tests", "golden", "obj_06.ply"))
    K = synth.K_TEST; W, H = 640, 480
    proj = api.compute_proj(K, W, H)
    sd = api.render_host(model, synth.scene_pose()[None], W, H, proj)[0]
    other = synth.scene_pose().copy()
    other.reshape(4, 4)[0, 3] += 15.0; other.reshape(4, 4)[2, 3] += 25.0
    sd2 = api.render_host(model, other[None], W, H, proj)[0]
    scenes = [api.Scene_projective().init_Scene_projective_cuda(sd, K), api.Scene_nn().init_Scene_nn_cuda(sd, K),
              api.Scene_nn().init_Scene_nn_cuda(sd2, K)]
    rng = np.random.default_rng(11)
    jobs = []
    for i in range(18):
        which = int(rng.integers(3))
        P = int(rng.choice([1, 3, 31, 33, 64, 65])) if which else int(rng.choice([33, 200, 300]))
        poses = synth.hypotheses(P, seed=300 + i)
        if P > 2 and rng.random() < 0.3:
            poses.reshape(-1, 4, 4)[1, 0, 3] += 1e6                # off-screen hypothesis -> empty cloud
        jobs.append((poses, api.ICPConvergenceCriteria(0.0, 0.0, int(rng.choice([0, 3, 8]))), scenes[which]))
    api.set_option("solve", api.SOLVE_DEVICE)
    try:
        api.set_option("profile", 1)                               # forces the synchronous path
        refs = [api.refine_batch(model, p, W, H, proj, K, sc, c) for p, c, sc in jobs]
        api.set_option("profile", 0)
        got, inflight = [None] * len(jobs), [None, None]
        for i, (p, c, sc) in enumerate(jobs):
            b = i & 1
            if inflight[b] is not None:
                got[inflight[b]] = api.refine_wait(b)
            api.refine_submit(b, model, p, W, H, proj, K, sc, c)
            inflight[b] = i
        for b in (0, 1):
            if inflight[b] is not None:
                got[inflight[b]] = api.refine_wait(b)
        for i, (g, r) in enumerate(zip(got, refs)):
            assert np.array_equal(g[1], r[1]), i
            assert g[0].tobytes() == r[0].tobytes(), (i, len(jobs[i][0]))
    finally:
        api.set_option("profile", 0)
        api.set_option("solve", api.SOLVE_HOST)


def test_arrival_protocol_stress_tiny_clouds(gpu):
    """The fused finalize + solve tail hands a hypothesis' partial sums from the workgroups that produce them to the one that draws
    the last ticket of the arrival counter (system-scope stores -> s_waitcnt -> agent-scope atomic -> system-scope loads).  That
    rests on how the hardware orders those accesses, so it is hammered here where it is most exposed: hundreds of hypotheses whose
    clouds have ONE (sometimes two or three) 2048-point blocks (a far-away object), i.e. almost every workgroup is a last arrival and hundreds of
    tails run in the same few microseconds, over four pose groups on four streams and both asynchronous slots, 30 times.  Every
    repetition must reproduce the synchronous, un-fused result bit for bit."""
    import os
    from pose_refine_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    model = api.Model(os.path.join(root, "tests", "golden", "obj_06.ply"))
    K = synth.K_TEST; W, H = 640, 480
    proj = api.compute_proj(K, W, H)
    far = synth.scene_pose().copy(); far[2, 3] = 1400.0
    sd = api.render_host(model, far[None], W, H, proj)[0]              # the scene: the object at 1.4 m (~1.1 k pixels)
    scene = api.Scene_projective().init_Scene_projective_cuda(sd, K)
    P = 768
    poses = synth.hypotheses(P, seed=11)
    poses.reshape(-1, 4, 4)[:, 2, 3] += 1080.0                         # hypotheses at ~1.4 m: clouds of one 2048-point block
    poses.reshape(-1, 4, 4)[::7, 2, 3] -= 500.0                        # every seventh at ~0.9 m: two or three blocks
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 12)
    api.set_option("solve", api.SOLVE_DEVICE)
    try:
        api.set_option("fused_solve", 0); api.set_option("pose_groups", 1); api.set_option("profile", 1)
        ref, ref_sizes = api.refine_batch(model, poses, W, H, proj, K, scene, crit)
        api.set_option("profile", 0); api.set_option("fused_solve", 1); api.set_option("pose_groups", 4)
        assert 0 < ref_sizes.max() <= 3 * 2048 and (ref_sizes <= 2048).mean() > 0.5
        for rep in range(15):
            api.refine_submit(0, model, poses, W, H, proj, K, scene, crit)
            api.refine_submit(1, model, poses[::-1].copy(), W, H, proj, K, scene, crit)
            a, sa = api.refine_wait(0)
            b, sb = api.refine_wait(1)
            assert np.array_equal(sa, ref_sizes) and a.tobytes() == ref.tobytes(), rep
            assert np.array_equal(sb, ref_sizes[::-1]) and b.tobytes() == ref[::-1].tobytes(), rep
    finally:
        api.set_option("profile", 0); api.set_option("fused_solve", 1); api.set_option("pose_groups", 0)
        api.set_option("solve", api.SOLVE_HOST)


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
