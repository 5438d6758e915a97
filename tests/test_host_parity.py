"""CPU tests of the product's HOST-side code (pr_host.cpp, reached through the C ABI) against the
oracle: these functions run on the CPU in the reference too (model import, compute_proj, scene
preparation, kd-tree build, 6x6 solver).  No GPU needed."""
import os

import numpy as np
import pytest

import oracle_lib as O
from pose_refine_amd import api, synth, _lib


def test_ply_loader_matches(golden_dir, obj06_tris):
    m = api.Model(os.path.join(golden_dir, "obj_06.ply"))
    assert m.tris.shape == (31468, 3, 3)
    assert np.array_equal(m.tris, obj06_tris)
    with pytest.raises(api.PoseRefineError):
        api.Model(os.path.join(golden_dir, "does_not_exist.ply"))


def test_compute_proj_matches():
    for K, w, h in [(synth.K_TEST, 640, 480), (synth.intrinsics_720p(), 1280, 720)]:
        assert np.array_equal(api.compute_proj(K, w, h), O.compute_proj(K, w, h))
    # renderer.cpp:161-185 closed form
    p = api.compute_proj(synth.K_TEST, 640, 480).reshape(4, 4)
    assert p[3, 2] == 1 and p[0, 0] == np.float32(2 * synth.K_TEST[0] / 640) and p[1, 1] < 0


def test_projective_scene_preparation_matches(scenario):
    d = scenario["depth"][1]
    for depth in (d, d.astype(np.uint16)):
        s = api.Scene_projective.__new__(api.Scene_projective)
        pcd = np.zeros((640 * 480, 3), np.float32); nrm = np.zeros_like(pcd)
        k = np.ascontiguousarray(scenario["K"], np.float32)
        dd = np.ascontiguousarray(depth)
        _lib.check(_lib.load().pr_scene_proj_prepare(dd.ctypes.data, int(dd.dtype == np.int32), k.ctypes.data, 640, 480,
                                                    pcd.ctypes.data, nrm.ctypes.data))
        ref = O.ProjScene(depth, scenario["K"])
        assert np.array_equal(pcd, ref.pcd) and np.array_equal(nrm, ref.normal)
    assert (np.linalg.norm(ref.normal, axis=1) > 0).sum() > 15000


def test_normals_gates(scenario):
    """get_normal leaves zeros for depth >= 2000 mm and in the 5 px border (common.cpp:33,52-60)."""
    d16 = np.full((60, 80), 1000, np.uint16)
    d16[:, 40:] = 2500
    out = np.zeros((60 * 80, 3), np.float32)
    k = np.ascontiguousarray(scenario["K"], np.float32)
    _lib.check(_lib.load().pr_get_normal(d16.ctypes.data, 80, 60, k.ctypes.data, out.ctypes.data))
    assert np.array_equal(out, O.get_normal(d16, scenario["K"]))
    n = out.reshape(60, 80, 3)
    assert not n[:5].any() and not n[:, :5].any() and not n[-6:].any() and not n[:, 40:].any()
    assert np.allclose(n[30, 20], [0, 0, -1])


def test_kdtree_build_matches(scenario):
    d = scenario["depth"][1]
    # build without touching the device: call the C ABI directly
    h, w = d.shape
    pcd = np.zeros((w * h, 3), np.float32); nrm = np.zeros_like(pcd)
    nodes = np.zeros(2 * w * h + 1, _lib.KDNODE)
    import ctypes as C
    npts, nn = C.c_uint32(), C.c_uint32()
    k = np.ascontiguousarray(scenario["K"], np.float32)
    dd = np.ascontiguousarray(d)
    _lib.check(_lib.load().pr_scene_nn_prepare(dd.ctypes.data, 1, k.ctypes.data, w, h, 10, pcd.ctypes.data, nrm.ctypes.data,
                                              nodes.ctypes.data, len(nodes), C.byref(npts), C.byref(nn)))
    ref = scenario["nn_scene"]
    assert npts.value == len(ref.pcd) and nn.value == len(ref.nodes)
    assert np.array_equal(pcd[:npts.value], ref.pcd) and np.array_equal(nrm[:npts.value], ref.normal)
    assert nodes[:nn.value].tobytes() == ref.nodes.tobytes()


@pytest.mark.parametrize("max_leaf", [1, 3, 10, 64])
def test_kdtree_build_random_points_with_ties(max_leaf):
    rng = np.random.default_rng(7)
    pts = np.round(rng.normal(size=(700, 3)), 1).astype(np.float32)      # coarse grid -> many equal coordinates
    nrm = rng.normal(size=(700, 3)).astype(np.float32)
    a_p, a_n = pts.copy(), nrm.copy(); b_p, b_n = pts.copy(), nrm.copy()
    na = np.zeros(2 * 700 + 1, _lib.KDNODE); nb = np.zeros(2 * 700 + 1, O.KDNODE)
    import ctypes as C
    cnt = C.c_uint32()
    _lib.check(_lib.load().pr_kdtree_build(a_p.ctypes.data, a_n.ctypes.data, 700, max_leaf, na.ctypes.data, len(na), C.byref(cnt)))
    cb = O.lib().po_kd_build(b_p.reshape(-1), b_n.reshape(-1), 700, max_leaf, nb.ctypes.data, len(nb))
    assert cnt.value == cb and na[:cb].tobytes() == nb[:cb].tobytes()
    assert np.array_equal(a_p, b_p) and np.array_equal(a_n, b_n)


def test_solver_matches_oracle():
    """Same algorithm (pivoted LDLT, quaternion composition); sin/cos differ (fixed polynomial vs libm)
    by < 1 ulp in double, invisible after the float cast except in rare last-bit cases."""
    rng = np.random.default_rng(11)
    worst = 0.0
    for i in range(200):
        J = rng.normal(size=(40, 6)).astype(np.float32) * (10.0 ** rng.uniform(-2, 1))
        A = (J.T @ J).astype(np.float32)
        b = (J.T @ rng.normal(size=40) * 0.05).astype(np.float32)
        T = api.eigen_slover_666(A, b)
        ref = O.solve666(A, b)
        worst = max(worst, float(np.abs(T - ref).max()))
        assert T[3].tolist() == [0, 0, 0, 1]
    assert worst <= 2.4e-7
    # large angles exercise the range reduction of the polynomial sin/cos
    A = np.zeros((6, 6), np.float32)
    for ang in (0.5, 1.0, 2.5, -3.0, 7.0):
        b = np.array([ang * 0.01, -ang * 0.01, ang * 0.01, 0, 0, 0], np.float32)   # x = b / 0.01
        assert np.allclose(api.eigen_slover_666(A, b), O.solve666(A, b), atol=3e-7)


def test_shard_range_covers_everything():
    for n, w in [(4096, 8), (1024, 8), (7, 3), (2, 4), (0, 2)]:
        seen = []
        for r in range(w):
            first, cnt = api.shard_range(n, r, w)
            seen += list(range(first, first + cnt))
        assert seen == list(range(n))
