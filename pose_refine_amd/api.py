"""Host-side mirror of the reference's public API for the hot path, on top of the C ABI.

Names, argument meaning and error behaviour follow the reference headers so that tests read like
the reference's own drivers (``test.cpp``, ``cuda_renderer/test.cpp``):

  cuda_renderer::Model / compute_proj / render / render_host        cuda_renderer/renderer.h:27-248
  cuda_icp::depth2cloud / ICP_Point2Plane / RegistrationResult /
            ICPConvergenceCriteria                                  cuda_icp/icp.h:26-120
  ::Scene_projective / ::Scene_nn / KDTree_cuda                     cuda_icp/scene/**.h
  ::device_vector_holder<T>                                         cuda_icp/scene/common.h:16-44

The C++ adapters in include/cuda_renderer and include/cuda_icp are the compiled drop-in; this module
is the same surface for python callers (tests, bench.py).  Everything that touches the device goes
through libpose_refine_hip.so -- nothing here computes on the CPU except what the reference also
computes on the CPU (model import, scene preparation).
"""
from __future__ import annotations

import ctypes as C
import threading
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import (COMM_ID_BYTES, Criteria, KDNODE, RESULT, Roi, SCENE_NN, SCENE_PROJ, SCENE_PROJ_CROP, SOLVE_DEVICE, SOLVE_HOST,
                   PoseRefineError, SceneNNDesc, SceneProjCropDesc, SceneProjDesc, check, ptr)


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a if shape is None else a.reshape(shape)


# ------------------------------------------------------------------------------------------------
# library / options
# ------------------------------------------------------------------------------------------------
def init(device: int = 0):
    check(_lib.load().pr_init(device))


def set_device(device: int):
    """Bind the calling thread to the shared context of ``device`` (one host thread per GPU)."""
    check(_lib.load().pr_set_device(device))


def shutdown():
    """``pr_shutdown``: release the calling thread's context (workspaces, streams, communicator).  Device buffers handed out by
    this module (DeviceVector, Model, scenes) stay valid; the next call re-creates the context."""
    _slots().clear()
    check(_lib.load().pr_shutdown())


def thread_context(enable: bool = True):
    """Private context (own stream / workspaces) for the calling thread: the reference's "many host threads,
    each driving its own pose" usage (README.md:15) without queueing on the device's shared context."""
    check(_lib.load().pr_thread_context(1 if enable else 0))


def invalidate(dev_ptr: int, nbytes: int = 0):
    """Announce a write to device memory the library could not see (drops derived scene data cached by address)."""
    check(_lib.load().pr_invalidate(int(dev_ptr), int(nbytes)))


def device_count() -> int:
    return _lib.load().pr_device_count()


# ---- the job's one collective (C ABI over RCCL) -------------------------------------------------------------------------
def comm_id() -> bytes:
    buf = (C.c_ubyte * COMM_ID_BYTES)()
    check(_lib.load().pr_comm_id(buf))
    return bytes(buf)


def comm_init_rank(comm_id_bytes: bytes, rank: int, world: int):
    buf = (C.c_ubyte * COMM_ID_BYTES).from_buffer_copy(comm_id_bytes)
    check(_lib.load().pr_comm_init_rank(buf, rank, world))


def comm_init_all(n_devices: int):
    check(_lib.load().pr_comm_init_all(n_devices))


def comm_destroy():
    check(_lib.load().pr_comm_destroy())


def comm_rank():
    r, w = C.c_int(), C.c_int()
    check(_lib.load().pr_comm_rank(C.byref(r), C.byref(w)))
    return r.value, w.value


def gather_results(send_dev: int, n_local: int, n_total: int, root: int = 0, recv_dev: Optional[int] = None):
    """``pr_gather_results``: this rank's shard (device memory) -> ``recv_dev`` on root, global hypothesis order.
    Enqueued on the context's stream."""
    check(_lib.load().pr_gather_results(int(send_dev) if send_dev else None, n_local, n_total, root, int(recv_dev) if recv_dev else None))


def sync():
    check(_lib.load().pr_sync())


def set_option(name: str, value: int):
    check(_lib.load().pr_set_option(name.encode(), int(value)))


def get_option(name: str) -> int:
    v = C.c_int()
    check(_lib.load().pr_get_option(name.encode(), C.byref(v)))
    return v.value


def profile_reset():
    check(_lib.load().pr_profile_reset())


def profile_read():
    ms, rms, cms = C.c_double(), C.c_double(), C.c_double()
    n, pts, nbytes = C.c_uint64(), C.c_uint64(), C.c_uint64()
    check(_lib.load().pr_profile_read(C.byref(ms), C.byref(n), C.byref(pts), C.byref(nbytes), C.byref(rms), C.byref(cms)))
    return dict(icp_kernel_ms=ms.value, icp_launches=n.value, icp_points=pts.value, icp_bytes=nbytes.value,
                render_ms=rms.value, cloud_ms=cms.value)


def profile_launches() -> np.ndarray:
    """``pr_profile_launches``: the timed correspondence launches since the last reset, one by one, in microseconds."""
    n = C.c_uint32()
    check(_lib.load().pr_profile_launches(None, 0, C.byref(n)))
    out = np.zeros(n.value, np.float32)
    if n.value:
        check(_lib.load().pr_profile_launches(ptr(out), n.value, C.byref(n)))
    return out


def profile_nn():
    """``pr_profile_nn``: accumulated time of the four kernels of the timed kd-tree passes -- (ms[4]: search, bound, task walk, winners pass; passes)."""
    ms = np.zeros(4, np.float64)
    n = C.c_uint64()
    check(_lib.load().pr_profile_nn(ptr(ms), C.byref(n)))
    return ms, n.value


def stats():
    """``pr_stats``: (asynchronous batches repeated by the stale-cache safety net, timings dropped after a failed event call)."""
    a, b = C.c_uint64(), C.c_uint64()
    check(_lib.load().pr_stats(C.byref(a), C.byref(b)))
    return a.value, b.value


def gather_profile():
    """HIP-event time of the gathers issued while option ``profile`` was on: (total ms, count).  Waits for the library stream."""
    ms, n = C.c_double(), C.c_uint64()
    check(_lib.load().pr_gather_profile(C.byref(ms), C.byref(n)))
    return ms.value, n.value


def nn_counters(passes: int = 21) -> np.ndarray:
    """Per-pass work counters of the kd-tree search kernel (option ``nn_count``): (passes, 8) uint64 --
    queries, settled by window, tree searches, pyramid descents, tree nodes, leaves, leaf points, window cells."""
    out = np.zeros((passes, 8), np.uint64)
    check(_lib.load().pr_nn_counters(ptr(out), passes))
    return out


def shard_range(n_items: int, rank: int, world: int):
    first, count = C.c_uint32(), C.c_uint32()
    _lib.load().pr_shard_range(n_items, rank, world, C.byref(first), C.byref(count))
    return first.value, count.value


# ------------------------------------------------------------------------------------------------
# device_vector_holder<T>
# ------------------------------------------------------------------------------------------------
class DeviceVector:
    """RAII device buffer (``device_vector_holder<T>``: common.h:16-44 / renderer.h:161-187)."""

    def __init__(self, count: int = 0, dtype=np.float32, _adopt: Optional[int] = None):
        self.dtype = np.dtype(dtype)
        self.count = int(count)
        self._ptr = None
        if _adopt is not None:
            self._ptr = int(_adopt)
        elif count > 0:
            p = C.c_void_p()
            check(_lib.load().pr_malloc(C.byref(p), self.count * self.dtype.itemsize))
            self._ptr = p.value

    @classmethod
    def from_host(cls, arr: np.ndarray, dtype=None) -> "DeviceVector":
        arr = np.ascontiguousarray(arr, dtype=dtype)
        dv = cls(arr.size if arr.dtype.fields is None else arr.shape[0], arr.dtype)
        if arr.nbytes:
            check(_lib.load().pr_memcpy_h2d(dv._ptr, ptr(arr), arr.nbytes))
        return dv

    def data(self) -> int:
        return self._ptr or 0

    def size(self) -> int:
        return self.count

    def to_host(self) -> np.ndarray:
        out = np.empty(self.count, self.dtype)
        if self.count:
            check(_lib.load().pr_memcpy_d2h(ptr(out), self._ptr, out.nbytes))
        return out

    def free(self):
        if self._ptr:
            _lib.load().pr_free(self._ptr)
            self._ptr = None
            self.count = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------
# cuda_renderer
# ------------------------------------------------------------------------------------------------
class Model:
    """``cuda_renderer::Model(fileName)`` (renderer.cpp:11-104): ``tris`` feeds the path; ``vertices``, ``faces``,
    ``bbox_min`` / ``bbox_max`` are the reference's other public members (renderer.h:143-147)."""

    def __init__(self, file_name: Optional[str] = None, tris: Optional[np.ndarray] = None):
        if tris is not None:
            self.tris = _f32(tris, (-1, 3, 3))
            self.vertices = self.tris.reshape(-1, 3)
            self.faces = np.arange(len(self.vertices), dtype=np.int32).reshape(-1, 3)
            self.bbox_min, self.bbox_max = (self.vertices.min(0), self.vertices.max(0)) if len(self.vertices) else (np.zeros(3, np.float32),) * 2
        else:
            lib = _lib.load()
            nt, nv = C.c_size_t(), C.c_size_t()
            check(lib.pr_mesh_count(file_name.encode(), C.byref(nt), C.byref(nv)))
            buf = np.zeros((nt.value, 3, 3), np.float32)
            self.vertices = np.zeros((nv.value, 3), np.float32)
            self.faces = np.zeros((nt.value, 3), np.int32)
            self.bbox_min, self.bbox_max = np.zeros(3, np.float32), np.zeros(3, np.float32)
            check(lib.pr_mesh_load(file_name.encode(), ptr(buf), nt.value, C.byref(nt), ptr(self.vertices), nv.value, C.byref(nv),
                                   ptr(self.faces), ptr(self.bbox_min), ptr(self.bbox_max)))
            self.tris = buf
        self._dev: Optional[DeviceVector] = None

    def device_tris(self) -> DeviceVector:
        """Triangles resident on the device (the ``device_vector_holder<Triangle>`` overloads)."""
        if self._dev is None:
            self._dev = DeviceVector.from_host(self.tris.reshape(-1), np.float32)
        return self._dev


def compute_proj(K, width: int, height: int, near: float = 10.0, far: float = 10000.0) -> np.ndarray:
    out = np.zeros(16, np.float32)
    k = _f32(K, -1)
    _lib.load().pr_compute_proj(ptr(k), width, height, near, far, ptr(out))
    return out


def _tris_dev(tris) -> DeviceVector:
    if isinstance(tris, Model):
        return tris.device_tris()
    if isinstance(tris, DeviceVector):
        return tris
    return DeviceVector.from_host(_f32(tris, -1))


def render(tris, poses, width: int, height: int, proj, roi: Sequence[int] = (0, 0, 0, 0)) -> DeviceVector:
    """``cuda_renderer::render`` -> ``render_cuda_keep_in_gpu`` (renderer.cu:269-336): int32 depth in mm
    for every pose, kept on the device, 0 where nothing was drawn."""
    td = _tris_dev(tris)
    poses = _f32(poses, (-1, 16))
    rw, rh = (roi[2], roi[3]) if roi[2] > 0 and roi[3] > 0 else (width, height)
    out = DeviceVector(len(poses) * rw * rh, np.int32)
    pj = _f32(proj, -1)
    check(_lib.load().pr_render(td.data(), td.size() // 9, ptr(poses), len(poses), width, height, ptr(pj), Roi(*roi), out.data()))
    return out


def raw2depth_mask(raw: DeviceVector, want_depth: bool = True, want_mask: bool = True):
    """``raw2depth_uint16_cuda`` / ``raw2mask_uint8_cuda`` / ``raw2depth_mask_cuda`` (renderer.cu:338-439) for a whole
    render stack: returns (uint16 depth or None, uint8 mask or None) as flat host arrays."""
    n = raw.size()
    d = np.empty(n, np.uint16) if want_depth else None
    m = np.empty(n, np.uint8) if want_mask else None
    check(_lib.load().pr_raw2depth_mask(raw.data(), n, ptr(d) if want_depth else None, ptr(m) if want_mask else None))
    return d, m


def render_host(tris, poses, width: int, height: int, proj, roi: Sequence[int] = (0, 0, 0, 0)) -> np.ndarray:
    """``cuda_renderer::render_host`` -> ``render_cuda`` (renderer.cu:189-267): result on the host."""
    td = _tris_dev(tris)
    poses = _f32(poses, (-1, 16))
    rw, rh = (roi[2], roi[3]) if roi[2] > 0 and roi[3] > 0 else (width, height)
    out = np.empty((len(poses), rh, rw), np.int32)
    pj = _f32(proj, -1)
    check(_lib.load().pr_render_to_host(td.data(), td.size() // 9, ptr(poses), len(poses), width, height, ptr(pj), Roi(*roi), ptr(out)))
    return out


# ------------------------------------------------------------------------------------------------
# cuda_icp
# ------------------------------------------------------------------------------------------------
@dataclass
class ICPConvergenceCriteria:           # icp.h:38-50
    relative_fitness: float = 1e-5
    relative_rmse: float = 1e-5
    max_iteration: int = 30

    def c(self) -> Criteria:
        return Criteria(self.relative_fitness, self.relative_rmse, self.max_iteration)


@dataclass
class RegistrationResult:               # icp.h:26-36
    transformation_: np.ndarray
    inlier_rmse_: float
    fitness_: float

    @classmethod
    def from_record(cls, r) -> "RegistrationResult":
        return cls(np.array(r["T"], np.float32).reshape(4, 4), float(r["inlier_rmse"]), float(r["fitness"]))


def depth2cloud(depth_dev, width: int, height: int, K, stride: int = 1, tl_x: int = 0, tl_y: int = 0,
                dtype=np.int32, offset_elems: int = 0) -> DeviceVector:
    """``cuda_icp::depth2cloud_cuda<T>`` (icp.cu:256-291): DEVICE depth pointer in, device cloud out.
    ``depth_dev`` is a DeviceVector (or raw device address); offset_elems selects an image of a stack."""
    base = depth_dev.data() if isinstance(depth_dev, DeviceVector) else int(depth_dev)
    dt = np.dtype(dtype)
    if isinstance(depth_dev, DeviceVector) and (offset_elems + width * height) * dt.itemsize > depth_dev.size() * depth_dev.dtype.itemsize:
        raise ValueError(f"depth buffer of {depth_dev.size()} {depth_dev.dtype} values holds no {width} x {height} {dt} image at element {offset_elems}")
    k = _f32(K, -1)
    out, n = C.c_void_p(), C.c_uint32()
    fn = _lib.load().pr_depth2cloud_u16 if dt == np.uint16 else _lib.load().pr_depth2cloud_i32
    check(fn(base + offset_elems * dt.itemsize, width, height, ptr(k), stride, tl_x, tl_y, C.byref(out), C.byref(n)))
    return DeviceVector(n.value * 3, np.float32, _adopt=out.value)


class Scene_projective:
    """``::Scene_projective`` (depth_scene.h:7-48) initialised like ``init_Scene_projective_cuda``
    (depth_scene.cu:3-20): CPU preparation (back-projection + normals), then two uploads."""

    def __init__(self):
        self.width, self.height, self.max_dist_diff = 640, 480, 0.1
        self.K = np.zeros(9, np.float32)
        self.pcd_buffer: Optional[DeviceVector] = None
        self.normal_buffer: Optional[DeviceVector] = None
        self.pcd_host = self.normal_host = None

    def init_Scene_projective_cuda(self, scene_depth: np.ndarray, scene_K, width: int = 640, height: int = 480,
                                   max_dist_diff: float = 0.1):
        if scene_depth.dtype not in (np.uint16, np.int32):          # depth_scene.cpp:11-12 assert
            raise ValueError("scene depth must be CV_16U or CV_32S")
        # the reference reads scene_depth.at(r, c) for r < height, c < width (depth_scene.cpp:17-30) and trusts the caller; an image
        # of another size than the (width, height) given -- the defaults are 640 x 480 as in depth_scene.h -- is an error here
        if scene_depth.ndim != 2 or scene_depth.shape != (height, width):
            raise ValueError(f"scene depth is {scene_depth.shape}, expected ({height}, {width}): pass width / height of the image")
        self.width, self.height, self.max_dist_diff = width, height, max_dist_diff
        self.K = _f32(scene_K, -1)
        d = np.ascontiguousarray(scene_depth)
        pcd = np.zeros((width * height, 3), np.float32)
        nrm = np.zeros((width * height, 3), np.float32)
        check(_lib.load().pr_scene_proj_prepare(ptr(d), int(d.dtype == np.int32), ptr(self.K), width, height, ptr(pcd), ptr(nrm)))
        self.pcd_host, self.normal_host = pcd, nrm
        self.pcd_buffer = DeviceVector.from_host(pcd.reshape(-1))
        self.normal_buffer = DeviceVector.from_host(nrm.reshape(-1))
        return self

    def init_Scene_projective_device(self, scene_depth_dev: "DeviceVector", scene_K, width: int = 640, height: int = 480,
                                     max_dist_diff: float = 0.1):
        """the same initialisation with the depth image already on the device and the
        preparation (back-projection + normals) done by a kernel -- no CPU work, no PCIe traffic."""
        if scene_depth_dev.dtype not in (np.uint16, np.int32):
            raise ValueError("scene depth must be CV_16U or CV_32S")
        if scene_depth_dev.size() != width * height:
            raise ValueError(f"scene depth holds {scene_depth_dev.size()} values, expected {width} x {height}: pass width / height of the image")
        self.width, self.height, self.max_dist_diff = width, height, max_dist_diff
        self.K = _f32(scene_K, -1)
        if self.pcd_buffer is None or self.pcd_buffer.size() != width * height * 3:        # a scene object that is re-initialised frame after frame keeps its arrays
            self.pcd_buffer = DeviceVector(width * height * 3, np.float32)
            self.normal_buffer = DeviceVector(width * height * 3, np.float32)
        check(_lib.load().pr_scene_proj_prepare_dev(scene_depth_dev.data(), int(scene_depth_dev.dtype == np.int32), ptr(self.K),
                                                    width, height, self.pcd_buffer.data(), self.normal_buffer.data()))
        return self

    kind = SCENE_PROJ
    tl_x = tl_y = 0

    def desc(self):
        view = SceneProjDesc(self.width, self.height, self.max_dist_diff, (C.c_float * 9)(*self.K),
                             self.pcd_buffer.data(), self.normal_buffer.data())
        if self.kind == SCENE_PROJ_CROP:
            return SceneProjCropDesc(view, self.tl_x, self.tl_y)
        return view

    def crop(self, window: Sequence[int]) -> "Scene_projective":
        """The same scene restricted to ``window`` = (x, y, width, height): arrays of the window's size and the
        ``tl_x / tl_y`` offsets of pcd2dep / dep2pcd (common.h:47-73) -- SURVEY 8f rank 3."""
        x, y, w, h = (int(v) for v in window)
        out = Scene_projective()
        out.kind = SCENE_PROJ_CROP
        out.width, out.height, out.max_dist_diff, out.K = w, h, self.max_dist_diff, self.K
        out.tl_x, out.tl_y = x, y
        out.pcd_buffer = DeviceVector(w * h * 3, np.float32)
        out.normal_buffer = DeviceVector(w * h * 3, np.float32)
        check(_lib.load().pr_scene_proj_crop_dev(self.pcd_buffer.data(), self.normal_buffer.data(), self.width, self.height,
                                                 Roi(x, y, w, h), out.pcd_buffer.data(), out.normal_buffer.data()))
        return out


class Scene_nn:
    """``::Scene_nn`` + ``KDTree_cuda`` (pcd_scene.h:27-137) initialised like ``init_Scene_nn_cuda``
    (pcd_scene.cu:3-20): CPU normals + kd-tree build, then three uploads."""

    kind = SCENE_NN

    def __init__(self):
        self.max_dist_diff = 0.1
        self.pcd_buffer = self.normal_buffer = self.nodes = None
        self.pcd_host = self.normal_host = self.nodes_host = None
        self.camera = None                                          # (fx, fy, cx, cy, w, h) of the depth image the scene was made from, if known

    def init_Scene_nn_cuda(self, scene_depth: np.ndarray, scene_K, max_leaf: int = 10, max_dist_diff: float = 0.1):
        if scene_depth.dtype not in (np.uint16, np.int32):          # pcd_scene.cpp:6-7 assert
            raise ValueError("scene depth must be CV_16U or CV_32S")
        h, w = scene_depth.shape
        k = _f32(scene_K, -1)
        d = np.ascontiguousarray(scene_depth)
        pcd = np.zeros((w * h, 3), np.float32)
        nrm = np.zeros((w * h, 3), np.float32)
        nodes = np.zeros(2 * w * h + 1, KDNODE)
        npts, nnodes = C.c_uint32(), C.c_uint32()
        check(_lib.load().pr_scene_nn_prepare(ptr(d), int(d.dtype == np.int32), ptr(k), w, h, max_leaf, ptr(pcd), ptr(nrm),
                                              ptr(nodes), len(nodes), C.byref(npts), C.byref(nnodes)))
        self.max_dist_diff = max_dist_diff
        self.camera = (float(k[0]), float(k[4]), float(k[2]), float(k[5]), int(w), int(h))
        self.pcd_host = np.ascontiguousarray(pcd[:npts.value])
        self.normal_host = np.ascontiguousarray(nrm[:npts.value])
        self.nodes_host = np.ascontiguousarray(nodes[:nnodes.value])
        self.pcd_buffer = DeviceVector.from_host(self.pcd_host.reshape(-1))
        self.normal_buffer = DeviceVector.from_host(self.normal_host.reshape(-1))
        self.nodes = DeviceVector.from_host(self.nodes_host)
        return self

    def init_Scene_nn_device(self, scene_depth_dev: "DeviceVector", scene_K, width: int, height: int, max_leaf: int = 10,
                             max_dist_diff: float = 0.1):
        """normals, valid-pixel gather and the level-order kd-tree build all on the device."""
        k = _f32(scene_K, -1)
        px = width * height
        if scene_depth_dev.size() != px:
            raise ValueError(f"scene depth holds {scene_depth_dev.size()} values, expected {width} x {height}")
        if self.pcd_buffer is None or self.pcd_buffer.size() != px * 3 or self.nodes is None or self.nodes.size() != 2 * px + 1:    # (kept across re-initialisations)
            self.pcd_buffer = DeviceVector(px * 3, np.float32)
            self.normal_buffer = DeviceVector(px * 3, np.float32)
            self.nodes = DeviceVector(2 * px + 1, KDNODE)
        npts, nnodes = C.c_uint32(), C.c_uint32()
        check(_lib.load().pr_scene_nn_prepare_dev(scene_depth_dev.data(), int(scene_depth_dev.dtype == np.int32), ptr(k), width, height, max_leaf,
                                                  self.pcd_buffer.data(), self.normal_buffer.data(), self.nodes.data(), 2 * px + 1,
                                                  C.byref(npts), C.byref(nnodes)))
        self.max_dist_diff = max_dist_diff
        self.camera = (float(k[0]), float(k[4]), float(k[2]), float(k[5]), int(width), int(height))
        self._n_points, self._n_nodes = npts.value, nnodes.value
        self.pcd_host = self.normal_host = self.nodes_host = None
        return self

    def desc(self) -> SceneNNDesc:
        n_pts = len(self.pcd_host) if self.pcd_host is not None else self._n_points
        n_nodes = len(self.nodes_host) if self.nodes_host is not None else self._n_nodes
        cam = self.camera if getattr(self, "camera", None) else (0.0, 0.0, 0.0, 0.0, 0, 0)
        return SceneNNDesc(self.max_dist_diff, self.pcd_buffer.data(), self.normal_buffer.data(), self.nodes.data(), n_pts, n_nodes, *cam,
                           _lib.SCENE_NN_CAM_MAGIC if cam[4] else 0)


def ICP_Point2Plane(model_pcd: DeviceVector, scene, criteria: ICPConvergenceCriteria = ICPConvergenceCriteria()) -> RegistrationResult:
    """``cuda_icp::ICP_Point2Plane_cuda<Scene>`` (icp.cu:156-223).  Mutates ``model_pcd`` in place."""
    res = np.zeros(1, RESULT)
    d = scene.desc()
    if scene.kind == SCENE_PROJ_CROP:                            # no single-cloud entry point of its own: a batch of one
        off = np.array([0, model_pcd.size() // 3], np.uint32)
        check(_lib.load().pr_icp_batch(model_pcd.data(), ptr(off), 1, scene.kind, C.addressof(d), criteria.c(), ptr(res)))
        return RegistrationResult.from_record(res[0])
    fn = _lib.load().pr_icp_nn if scene.kind == SCENE_NN else _lib.load().pr_icp_proj
    check(fn(model_pcd.data(), model_pcd.size() // 3, C.addressof(d), criteria.c(), ptr(res)))
    return RegistrationResult.from_record(res[0])


def ICP_Point2Plane_batch(clouds: DeviceVector, offsets, scene, criteria: ICPConvergenceCriteria = ICPConvergenceCriteria()) -> np.ndarray:
    """Many clouds against one scene (one launch per iteration).  offsets: P+1 point offsets."""
    offsets = np.ascontiguousarray(offsets, np.uint32)
    if len(offsets) == 0:
        raise ValueError("offsets: P + 1 point offsets (at least one)")
    if int(offsets[-1]) * 3 > clouds.size():
        raise ValueError(f"offsets end at point {int(offsets[-1])}, the cloud buffer holds {clouds.size() // 3}")
    res = np.zeros(len(offsets) - 1, RESULT)
    d = scene.desc()
    check(_lib.load().pr_icp_batch(clouds.data(), ptr(offsets), len(offsets) - 1, scene.kind, C.addressof(d), criteria.c(), ptr(res)))
    return res


def debug_contrib29(model_pcd: DeviceVector, scene, update=None, packed: bool = False) -> np.ndarray:
    """``pr_debug_contrib29``: the (n, 29) per-point terms of one correspondence pass; ``update`` (4x4 or None) is applied to the cloud first."""
    n = model_pcd.size() // 3
    out = np.zeros((n, 29), np.float32)
    d = scene.desc()
    u = _f32(update, -1) if update is not None else None
    check(_lib.load().pr_debug_contrib29(model_pcd.data(), n, scene.kind, C.addressof(d), ptr(u) if u is not None else None, int(packed), ptr(out)))
    return out


def refine_batch(tris, poses, width: int, height: int, proj, K, scene,
                 criteria: ICPConvergenceCriteria = ICPConvergenceCriteria(), results_dev: Optional[int] = None,
                 roi: Optional[Sequence[int]] = None):
    """Fused hypothesis refinement (test.cpp:143-172 for a batch): render -> cloud -> ICP on the device.
    Returns (records[P] of RESULT dtype or None when results_dev is given, cloud sizes[P]).
    ``roi`` = (x, y, width, height): render only inside the window (renderer.h:199), clouds with tl = (x, y)."""
    td = _tris_dev(tris)
    poses = _f32(poses, (-1, 16))
    pj, k = _f32(proj, -1), _f32(K, -1)
    sizes = np.zeros(len(poses), np.uint32)
    d = scene.desc()
    if roi is not None:
        if results_dev is not None:
            raise ValueError("roi and results_dev together: use refine_submit")
        res = np.zeros(len(poses), RESULT)
        check(_lib.load().pr_refine_batch_roi(td.data(), td.size() // 9, ptr(poses), len(poses), width, height, ptr(pj), ptr(k),
                                              scene.kind, C.addressof(d), criteria.c(), Roi(*roi), ptr(res), ptr(sizes)))
        return res, sizes
    if results_dev is None:
        res = np.zeros(len(poses), RESULT)
        check(_lib.load().pr_refine_batch(td.data(), td.size() // 9, ptr(poses), len(poses), width, height, ptr(pj), ptr(k),
                                          scene.kind, C.addressof(d), criteria.c(), ptr(res), ptr(sizes)))
        return res, sizes
    check(_lib.load().pr_refine_batch_dev(td.data(), td.size() // 9, ptr(poses), len(poses), width, height, ptr(pj), ptr(k),
                                          scene.kind, C.addressof(d), criteria.c(), int(results_dev), ptr(sizes)))
    return None, sizes


_tls = threading.local()


def _slots() -> dict:
    """Outputs (and input owners) of the batches in flight, per host thread: a thread with a private context
    (``thread_context``) has slots of its own, and the arrays the library writes into at ``refine_wait`` must stay alive until then."""
    d = getattr(_tls, "inflight", None)
    if d is None:
        d = _tls.inflight = {}
    return d


def refine_submit(slot: int, tris, poses, width: int, height: int, proj, K, scene,
                  criteria: ICPConvergenceCriteria = ICPConvergenceCriteria(), results_dev: Optional[int] = None,
                  roi: Sequence[int] = (0, 0, 0, 0), also_host: bool = False):
    """Asynchronous ``refine_batch``: enqueue the batch on ``slot`` (0 or 1) and return at once; ``refine_wait(slot)``
    delivers what ``refine_batch`` returns.  With two slots, batch k+1 is enqueued while batch k runs.  ``results_dev``: the records go to
    that device block (a sharded job's gather reads them there); ``also_host``: and to the host as well, like the reference's by-value return."""
    td = _tris_dev(tris)
    poses = _f32(poses, (-1, 16))
    pj, k = _f32(proj, -1), _f32(K, -1)
    sizes = np.zeros(len(poses), np.uint32)
    res = np.zeros(len(poses), RESULT) if (results_dev is None or also_host) else None
    d = scene.desc()
    check(_lib.load().pr_refine_submit_roi(int(slot), td.data(), td.size() // 9, ptr(poses), len(poses), width, height, ptr(pj), ptr(k),
                                           scene.kind, C.addressof(d), criteria.c(), Roi(*roi), ptr(res) if res is not None else None,
                                           int(results_dev) if results_dev is not None else None, ptr(sizes)))
    _slots()[int(slot)] = (res, sizes, td, scene)              # keep the output arrays (and the inputs' owners) alive


def refine_wait(slot: int):
    """Block until the batch submitted on ``slot`` is finished; returns (records or None, cloud sizes)."""
    check(_lib.load().pr_refine_wait(int(slot)))
    res, sizes, _, _ = _slots().pop(int(slot))
    return res, sizes


def eigen_slover_666(A, b) -> np.ndarray:
    """``cuda_icp::eigen_slover_666`` (icp.cpp:29-45, public in icp.h:54)."""
    T = np.zeros(16, np.float32)
    a, bb = _f32(A, -1), _f32(b, -1)
    _lib.load().pr_solve_666(ptr(a), ptr(bb), ptr(T))
    return T.reshape(4, 4)
