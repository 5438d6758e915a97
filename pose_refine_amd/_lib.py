"""ctypes binding of the C ABI (include/pose_refine.h) -> pose_refine_amd/lib/libpose_refine_hip.so.

The shared library is built in-tree by ``pose_refine_amd.build`` (hipcc --offload-arch=gfx950).
There is no fallback of any kind: a missing library raises ImportError-like RuntimeError, a missing
GPU makes every device entry point fail with PR_ERR_NO_DEVICE (raised as PoseRefineError).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (PR_LIB_PATH: same-box A/B runs of kernel variants, tools/ab_libs.sh -- the product is the in-tree library)
LIB_PATH = os.environ.get("PR_LIB_PATH") or os.path.join(_HERE, "lib", "libpose_refine_hip.so")

PR_OK = 0
PR_ERR_NO_DEVICE, PR_ERR_HIP, PR_ERR_INVALID, PR_ERR_IO, PR_ERR_NOMEM, PR_ERR_COMM = -1, -2, -3, -4, -5, -6
SCENE_PROJ, SCENE_NN, SCENE_PROJ_CROP = 0, 1, 2
COMM_ID_BYTES = 128
SOLVE_HOST, SOLVE_DEVICE = 0, 1

KDNODE = np.dtype([("parent", "<i4"), ("child1", "<i4"), ("child2", "<i4"), ("split_v", "<f4"),
                   ("bbox", "<f4", (6,)), ("split_dim", "<i4"), ("left", "<i4"), ("right", "<i4")])
RESULT = np.dtype([("T", "<f4", (16,)), ("inlier_rmse", "<f4"), ("fitness", "<f4")])
assert KDNODE.itemsize == 52 and RESULT.itemsize == 72


class PoseRefineError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"pose_refine error {code}: {msg}")
        self.code = code


class Roi(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("width", C.c_int), ("height", C.c_int)]


class Criteria(C.Structure):
    _fields_ = [("relative_fitness", C.c_float), ("relative_rmse", C.c_float), ("max_iteration", C.c_int)]


class SceneProjDesc(C.Structure):
    _fields_ = [("width", C.c_uint64), ("height", C.c_uint64), ("max_dist_diff", C.c_float), ("K", C.c_float * 9),
                ("pcd", C.c_void_p), ("normal", C.c_void_p)]


class SceneProjCropDesc(C.Structure):
    _fields_ = [("view", SceneProjDesc), ("tl_x", C.c_uint32), ("tl_y", C.c_uint32)]


SCENE_NN_CAM_MAGIC = 0x4d414350          # PR_SCENE_NN_CAM_MAGIC: the camera hint of a pr_scene_nn is read only with it


class SceneNNDesc(C.Structure):
    _fields_ = [("max_dist_diff", C.c_float), ("pcd", C.c_void_p), ("normal", C.c_void_p), ("nodes", C.c_void_p),
                ("n_points", C.c_uint32), ("n_nodes", C.c_uint32),
                ("cam_fx", C.c_float), ("cam_fy", C.c_float), ("cam_cx", C.c_float), ("cam_cy", C.c_float), ("cam_w", C.c_uint32), ("cam_h", C.c_uint32), ("cam_magic", C.c_uint32)]


# name -> (restype, argtypes); this table is also what tests/test_cabi_symbols.py checks against the header
_vp, _sz, _u32, _i32 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int
SIGNATURES = {
    "pr_last_error": (C.c_char_p, []),
    "pr_version": (C.c_char_p, []),
    "pr_abi_version": (_i32, []),
    "pr_device_count": (_i32, []),
    "pr_init": (_i32, [_i32]),
    "pr_set_device": (_i32, [_i32]),
    "pr_thread_context": (_i32, [_i32]),
    "pr_shutdown": (_i32, []),
    "pr_sync": (_i32, []),
    "pr_malloc": (_i32, [C.POINTER(_vp), _sz]),
    "pr_free": (_i32, [_vp]),
    "pr_memcpy_h2d": (_i32, [_vp, _vp, _sz]),
    "pr_memcpy_d2h": (_i32, [_vp, _vp, _sz]),
    "pr_memcpy_d2d": (_i32, [_vp, _vp, _sz]),
    "pr_fill_i32": (_i32, [_vp, _sz, C.c_int32]),
    "pr_invalidate": (_i32, [_vp, _sz]),
    "pr_scene_proj_crop_dev": (_i32, [_vp, _vp, _sz, _sz, Roi, _vp, _vp]),
    "pr_mesh_count": (_i32, [C.c_char_p, C.POINTER(_sz), C.POINTER(_sz)]),
    "pr_mesh_load": (_i32, [C.c_char_p, _vp, _sz, C.POINTER(_sz), _vp, _sz, C.POINTER(_sz), _vp, _vp, _vp]),
    "pr_ply_count": (_i32, [C.c_char_p, C.POINTER(_sz), C.POINTER(_sz)]),
    "pr_ply_load": (_i32, [C.c_char_p, _vp, _sz, C.POINTER(_sz)]),
    "pr_compute_proj": (None, [_vp, _i32, _i32, C.c_float, C.c_float, _vp]),
    "pr_get_normal": (_i32, [_vp, _i32, _i32, _vp, _vp]),
    "pr_scene_proj_prepare": (_i32, [_vp, _i32, _vp, _sz, _sz, _vp, _vp]),
    "pr_scene_proj_prepare_dev": (_i32, [_vp, _i32, _vp, _sz, _sz, _vp, _vp]),
    "pr_raw2depth_mask": (_i32, [_vp, _sz, _vp, _vp]),
    "pr_scene_nn_prepare": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _sz, C.POINTER(_u32), C.POINTER(_u32)]),
    "pr_kdtree_build_dev": (_i32, [_vp, _vp, _sz, _i32, _vp, _sz, C.POINTER(_u32)]),
    "pr_scene_nn_prepare_dev": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _sz, C.POINTER(_u32), C.POINTER(_u32)]),
    "pr_kdtree_build": (_i32, [_vp, _vp, _sz, _i32, _vp, _sz, C.POINTER(_u32)]),
    "pr_solve_666": (None, [_vp, _vp, _vp]),
    "pr_mat4_mul": (None, [_vp, _vp, _vp]),
    "pr_render": (_i32, [_vp, _sz, _vp, _sz, _sz, _sz, _vp, Roi, _vp]),
    "pr_render_to_host": (_i32, [_vp, _sz, _vp, _sz, _sz, _sz, _vp, Roi, _vp]),
    "pr_depth2cloud_i32": (_i32, [_vp, _u32, _u32, _vp, _u32, _u32, _u32, C.POINTER(_vp), C.POINTER(_u32)]),
    "pr_depth2cloud_u16": (_i32, [_vp, _u32, _u32, _vp, _u32, _u32, _u32, C.POINTER(_vp), C.POINTER(_u32)]),
    "pr_icp_proj": (_i32, [_vp, _u32, _vp, Criteria, _vp]),
    "pr_icp_nn": (_i32, [_vp, _u32, _vp, Criteria, _vp]),
    "pr_icp_batch": (_i32, [_vp, _vp, _u32, _i32, _vp, Criteria, _vp]),
    "pr_refine_batch": (_i32, [_vp, _sz, _vp, _u32, _u32, _u32, _vp, _vp, _i32, _vp, Criteria, _vp, _vp]),
    "pr_refine_batch_dev": (_i32, [_vp, _sz, _vp, _u32, _u32, _u32, _vp, _vp, _i32, _vp, Criteria, _vp, _vp]),
    "pr_refine_submit": (_i32, [_i32, _vp, _sz, _vp, _u32, _u32, _u32, _vp, _vp, _i32, _vp, Criteria, _vp, _vp, _vp]),
    "pr_refine_batch_roi": (_i32, [_vp, _sz, _vp, _u32, _u32, _u32, _vp, _vp, _i32, _vp, Criteria, Roi, _vp, _vp]),
    "pr_refine_submit_roi": (_i32, [_i32, _vp, _sz, _vp, _u32, _u32, _u32, _vp, _vp, _i32, _vp, Criteria, Roi, _vp, _vp, _vp]),
    "pr_refine_wait": (_i32, [_i32]),
    "pr_comm_id": (_i32, [_vp]),
    "pr_comm_init_rank": (_i32, [_vp, _i32, _i32]),
    "pr_comm_init_all": (_i32, [_i32]),
    "pr_comm_rank": (_i32, [C.POINTER(_i32), C.POINTER(_i32)]),
    "pr_comm_destroy": (_i32, []),
    "pr_gather_results": (_i32, [_vp, _u32, _u32, _i32, _vp]),
    "pr_shard_range": (None, [_u32, _u32, _u32, C.POINTER(_u32), C.POINTER(_u32)]),
    "pr_set_option": (_i32, [C.c_char_p, _i32]),
    "pr_get_option": (_i32, [C.c_char_p, C.POINTER(_i32)]),
    "pr_nn_counters": (_i32, [_vp, _u32]),
    "pr_profile_reset": (_i32, []),
    "pr_profile_read": (_i32, [C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "pr_gather_profile": (_i32, [C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "pr_profile_launches": (_i32, [_vp, _u32, C.POINTER(_u32)]),
    "pr_profile_nn": (_i32, [_vp, C.POINTER(C.c_uint64)]),
    "pr_debug_contrib29": (_i32, [_vp, _u32, _i32, _vp, _vp, _i32, _vp]),
    "pr_stats": (_i32, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
}

_lib = None


def load():
    """Load (once) the in-tree shared library and bind every entry point of the header."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -m pose_refine_amd.build` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int):
    if rc != PR_OK:
        raise PoseRefineError(rc, load().pr_last_error().decode(errors="replace"))


def ptr(a: np.ndarray) -> int:
    return a.ctypes.data
