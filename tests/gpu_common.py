"""Helpers shared by the GPU test files (-m gpu): frame size, tolerances, raw HIP writes the library cannot see, hand-made kd-tree scenes,
random meshes and poses."""
import ctypes as C
import json
import os
import threading

import numpy as np
import pytest

import oracle_lib as O
from pose_refine_amd import _lib, api, synth


W, H = synth.WIDTH, synth.HEIGHT


TOL_T = 1e-4          # north_star: "transforms within 1e-4"


def inliers(fitness, n):
    return np.rint(np.asarray(fitness, np.float64) * np.asarray(n, np.float64)).astype(np.int64)


def raw_hip():
    """The HIP runtime the library itself uses, for writes the library cannot see."""
    return C.CDLL("libamdhip64.so.7")


def raw_h2d(dst_dev: int, arr: np.ndarray):
    hip = raw_hip()
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(dst_dev, arr.ctypes.data, arr.nbytes, 1) == 0
    assert hip.hipDeviceSynchronize() == 0


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


def raw_d2d(dst_dev: int, src_dev: int, nbytes: int):
    """A write the library cannot see: the HIP runtime called directly."""
    hip = C.CDLL("libamdhip64.so.7")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(dst_dev, src_dev, nbytes, 3) == 0
    assert hip.hipDeviceSynchronize() == 0


def pathological_hypotheses(base):
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
