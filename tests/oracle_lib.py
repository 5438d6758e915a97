"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE (the parity checker).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product package pose_refine_amd never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
# PR_ORACLE_BUILD=o3 selects the build with the reference's optimisation flags (oracle/Makefile): bench.py's cpu_baseline uses
# it; every parity test uses the default build (-O2 -ffp-contract=off: bit-reproducible sums)
_SO = os.path.join(_ORACLE_DIR, "liboracle_o3.so" if os.environ.get("PR_ORACLE_BUILD") == "o3" else "liboracle.so")

SCENE_PROJ, SCENE_NN = 0, 1
SUM_SEQUENTIAL, SUM_CANONICAL = 0, 1

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
u16p = np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")

KDNODE = np.dtype([("parent", "<i4"), ("child1", "<i4"), ("child2", "<i4"), ("split_v", "<f4"),
                   ("bbox", "<f4", (6,)), ("split_dim", "<i4"), ("left", "<i4"), ("right", "<i4")])
assert KDNODE.itemsize == 52
RESULT = np.dtype([("T", "<f4", (16,)), ("inlier_rmse", "<f4"), ("fitness", "<f4")])
assert RESULT.itemsize == 72


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Roi(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("width", C.c_int), ("height", C.c_int)]


class Criteria(C.Structure):
    _fields_ = [("relative_fitness", C.c_float), ("relative_rmse", C.c_float), ("max_iteration", C.c_int)]


class SceneProj(C.Structure):
    _fields_ = [("width", C.c_size_t), ("height", C.c_size_t), ("max_dist_diff", C.c_float),
                ("K", C.c_float * 9), ("pcd", C.c_void_p), ("normal", C.c_void_p), ("tl_x", C.c_size_t), ("tl_y", C.c_size_t)]


class SceneNN(C.Structure):
    _fields_ = [("max_dist_diff", C.c_float), ("pcd", C.c_void_p), ("normal", C.c_void_p), ("nodes", C.c_void_p)]


def build(force: bool = False) -> str:
    src = [os.path.join(_ORACLE_DIR, f) for f in ("pose_oracle.c", "pose_oracle.h", "Makefile")]
    stale = force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    if stale:
        subprocess.run(["make", "-C", _ORACLE_DIR, "-s"], check=True)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.po_compute_proj.argtypes = [f32p, C.c_int, C.c_int, C.c_float, C.c_float, f32p]
        L.po_render.argtypes = [f32p, C.c_size_t, f32p, C.c_size_t, C.c_size_t, C.c_size_t, f32p, Roi, i32p]
        L.po_depth2cloud_i32.restype = C.c_size_t
        L.po_depth2cloud_i32.argtypes = [i32p, C.c_uint32, C.c_uint32, f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.po_depth2cloud_u16.restype = C.c_size_t
        L.po_depth2cloud_u16.argtypes = [u16p, C.c_uint32, C.c_uint32, f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.po_get_normal.argtypes = [u16p, C.c_int, C.c_int, f32p, f32p]
        L.po_depth_i32_to_u16.argtypes = [i32p, u16p, C.c_size_t]
        L.po_scene_proj_init.argtypes = [C.c_void_p, C.c_int, f32p, C.c_size_t, C.c_size_t, f32p, f32p]
        L.po_scene_nn_gather.restype = C.c_size_t
        L.po_scene_nn_gather.argtypes = [C.c_void_p, C.c_int, f32p, C.c_int, C.c_int, f32p, f32p]
        L.po_kd_build.restype = C.c_size_t
        L.po_kd_build.argtypes = [f32p, f32p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t]
        L.po_solve666.argtypes = [f32p, f32p, f32p]
        L.po_mat4_mul.argtypes = [f32p, f32p, f32p]
        L.po_icp.restype = C.c_int
        L.po_icp.argtypes = [f32p, C.c_size_t, C.c_int, C.c_void_p, Criteria, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
        L.po_sum29.argtypes = [f32p, C.c_size_t, C.c_int, C.c_void_p, C.c_int, C.c_uint32, f32p]
        L.po_set_threads.restype = C.c_int
        L.po_set_threads.argtypes = [C.c_int]
        L.po_refine_batch.restype = C.c_int
        L.po_refine_batch.argtypes = [f32p, C.c_size_t, f32p, C.c_size_t, C.c_size_t, C.c_size_t, f32p, f32p,
                                      C.c_int, C.c_void_p, Criteria, C.c_int, C.c_uint32, Roi, C.c_void_p, u32p]
        L.po_ply_count.restype = C.c_size_t
        L.po_ply_count.argtypes = [C.c_char_p, C.c_void_p]
        L.po_ply_load.restype = C.c_int
        L.po_ply_load.argtypes = [C.c_char_p, f32p, C.c_size_t]
        L.po_query_nn.restype = C.c_int
        L.po_query_nn.argtypes = [C.c_void_p, Vec3, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def compute_proj(K, width, height, near=10.0, far=10000.0):
    out = np.zeros(16, np.float32)
    lib().po_compute_proj(_f32(K).reshape(-1), width, height, near, far, out)
    return out


def ply_load(path):
    n = lib().po_ply_count(path.encode(), None)
    tris = np.zeros((n, 3, 3), np.float32)
    got = lib().po_ply_load(path.encode(), tris.reshape(-1), n)
    assert got == n, (got, n)
    return tris


def render(tris, poses, width, height, proj, roi=(0, 0, 0, 0)):
    tris = _f32(tris).reshape(-1, 9)
    poses = _f32(poses).reshape(-1, 16)
    rw, rh = (roi[2], roi[3]) if roi[2] > 0 and roi[3] > 0 else (width, height)
    out = np.empty(len(poses) * rw * rh, np.int32)
    lib().po_render(tris.reshape(-1), len(tris), poses.reshape(-1), len(poses), width, height,
                    _f32(proj).reshape(-1), Roi(*roi), out)
    return out.reshape(len(poses), rh, rw)


def depth2cloud(depth, K, stride=1, tl_x=0, tl_y=0):
    h, w = depth.shape
    cap = np.empty((w * h, 3), np.float32)
    if depth.dtype == np.uint16:
        n = lib().po_depth2cloud_u16(np.ascontiguousarray(depth), w, h, _f32(K).reshape(-1), stride, tl_x, tl_y, cap.ctypes.data)
    else:
        n = lib().po_depth2cloud_i32(np.ascontiguousarray(depth, np.int32), w, h, _f32(K).reshape(-1), stride, tl_x, tl_y, cap.ctypes.data)
    return np.ascontiguousarray(cap[:n])


def get_normal(depth16, K):
    h, w = depth16.shape
    out = np.zeros((h * w, 3), np.float32)
    lib().po_get_normal(np.ascontiguousarray(depth16, np.uint16), w, h, _f32(K).reshape(-1), out.reshape(-1))
    return out


class ProjScene:
    """Host buffers + descriptor of Scene_projective (depth_scene.h:7-48)."""

    def __init__(self, depth, K, max_dist_diff=0.1):
        h, w = depth.shape
        self.width, self.height = w, h
        self.K = _f32(K).reshape(-1)
        self.pcd = np.zeros((w * h, 3), np.float32)
        self.normal = np.zeros((w * h, 3), np.float32)
        is32 = depth.dtype != np.uint16
        d = np.ascontiguousarray(depth, np.int32 if is32 else np.uint16)
        lib().po_scene_proj_init(d.ctypes.data, int(is32), self.K, w, h, self.pcd.reshape(-1), self.normal.reshape(-1))
        self.max_dist_diff = max_dist_diff
        self.desc = SceneProj(w, h, max_dist_diff, (C.c_float * 9)(*self.K), self.pcd.ctypes.data, self.normal.ctypes.data, 0, 0)
        self.kind = SCENE_PROJ

    def crop(self, window):
        """The scene restricted to window = (x, y, width, height): arrays of the window, pcd2dep offsets tl = (x, y)."""
        x, y, w, h = (int(v) for v in window)
        out = object.__new__(ProjScene)
        out.width, out.height, out.K, out.max_dist_diff, out.kind = w, h, self.K, self.max_dist_diff, SCENE_PROJ
        out.pcd = np.ascontiguousarray(self.pcd.reshape(self.height, self.width, 3)[y:y + h, x:x + w]).reshape(-1, 3)
        out.normal = np.ascontiguousarray(self.normal.reshape(self.height, self.width, 3)[y:y + h, x:x + w]).reshape(-1, 3)
        out.desc = SceneProj(w, h, self.max_dist_diff, (C.c_float * 9)(*self.K), out.pcd.ctypes.data, out.normal.ctypes.data, x, y)
        return out

    def ptr(self):
        return C.addressof(self.desc)


class NNScene:
    """Host buffers + descriptor of Scene_nn / KDTree_cpu (pcd_scene.h:27-137)."""

    def __init__(self, depth, K, max_dist_diff=0.1, max_leaf=10):
        h, w = depth.shape
        self.K = _f32(K).reshape(-1)
        pcd = np.zeros((w * h, 3), np.float32)
        nrm = np.zeros((w * h, 3), np.float32)
        is32 = depth.dtype != np.uint16
        d = np.ascontiguousarray(depth, np.int32 if is32 else np.uint16)
        n = lib().po_scene_nn_gather(d.ctypes.data, int(is32), self.K, w, h, pcd.reshape(-1), nrm.reshape(-1))
        self.pcd = np.ascontiguousarray(pcd[:n])
        self.normal = np.ascontiguousarray(nrm[:n])
        nodes = np.zeros(2 * n + 1, KDNODE)
        cnt = lib().po_kd_build(self.pcd.reshape(-1), self.normal.reshape(-1), n, max_leaf, nodes.ctypes.data, len(nodes))
        self.nodes = np.ascontiguousarray(nodes[:cnt])
        self.max_dist_diff = max_dist_diff
        self.desc = SceneNN(max_dist_diff, self.pcd.ctypes.data, self.normal.ctypes.data, self.nodes.ctypes.data)
        self.kind = SCENE_NN

    def ptr(self):
        return C.addressof(self.desc)

    def query(self, src):
        dst = Vec3(); nrm = Vec3()
        win = C.c_int(); d2 = C.c_float(); vis = C.c_uint32()
        ok = lib().po_query_nn(self.ptr(), Vec3(*[float(v) for v in src]), C.addressof(dst), C.addressof(nrm),
                               C.addressof(win), C.addressof(d2), C.addressof(vis))
        return ok, win.value, d2.value, vis.value


def icp(cloud, scene, criteria=(1e-5, 1e-5, 30), sum_mode=SUM_SEQUENTIAL, ppb=2048, trace=False):
    """Runs po_icp on a COPY of cloud; returns (result record, passes, final cloud, trace or None)."""
    cl = np.array(cloud, dtype=np.float32, order="C", copy=True)
    res = np.zeros(1, RESULT)
    tr = np.zeros((criteria[2] + 1, 29), np.float32) if trace else None
    passes = lib().po_icp(cl.reshape(-1), len(cl), scene.kind, scene.ptr(), Criteria(*criteria), sum_mode, ppb,
                          res.ctypes.data, tr.ctypes.data if trace else None)
    return res[0], passes, cl, (tr[:passes] if trace else None)


def sum29(cloud, scene, sum_mode=SUM_SEQUENTIAL, ppb=2048):
    out = np.zeros(29, np.float32)
    cl = _f32(cloud)
    lib().po_sum29(cl.reshape(-1), len(cl), scene.kind, scene.ptr(), sum_mode, ppb, out)
    return out


def solve666(A, b):
    T = np.zeros(16, np.float32)
    lib().po_solve666(_f32(A).reshape(-1), _f32(b).reshape(-1), T)
    return T.reshape(4, 4)


def set_threads(n=0):
    """OpenMP threads of refine_batch (0: query); returns the setting in force."""
    return int(lib().po_set_threads(int(n)))


def refine_batch(tris, poses, width, height, proj, K, scene, criteria=(0.0, 0.0, 20),
                 sum_mode=SUM_SEQUENTIAL, ppb=2048, roi=(0, 0, 0, 0)):
    tris = _f32(tris).reshape(-1, 9)
    poses = _f32(poses).reshape(-1, 16)
    res = np.zeros(len(poses), RESULT)
    sizes = np.zeros(len(poses), np.uint32)
    threads = lib().po_refine_batch(tris.reshape(-1), len(tris), poses.reshape(-1), len(poses), width, height,
                                    _f32(proj).reshape(-1), _f32(K).reshape(-1), scene.kind, scene.ptr(),
                                    Criteria(*criteria), sum_mode, ppb, Roi(*roi), res.ctypes.data, sizes)
    return res, sizes, threads
