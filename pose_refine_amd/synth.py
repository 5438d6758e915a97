"""Synthetic inputs of the benchmark / parity scenarios (pure numpy, float32 arithmetic).

Mirrors the input construction of the reference's only end-to-end driver, ``test.cpp:22-46``
(intrinsics K, R_ren, the two poses, the ``10/180*3.14f`` angle) and the hypothesis sampling of
SURVEY.md section 8d (std::mt19937(6), +-10 degrees per axis, +-20 mm per axis, pose 0 = the exact
``test.cpp`` model pose).  Nothing here touches the GPU or the oracle.
"""
from __future__ import annotations

import numpy as np

WIDTH, HEIGHT = 640, 480

# test.cpp:26
K_TEST = np.array([572.4114, 0.0, 325.2611,
                   0.0, 573.57043, 242.04899,
                   0.0, 0.0, 1.0], dtype=np.float32)

# test.cpp:29-30
R_REN = np.array([0.34768538, 0.93761126, 0.00000000,
                  0.70540612, -0.26157897, -0.65877056,
                  -0.61767070, 0.22904489, -0.75234390], dtype=np.float32).reshape(3, 3)
T_MODEL = np.array([0.0, 0.0, 300.0], dtype=np.float32)     # test.cpp:31
T_SCENE = np.array([20.0, 20.0, 320.0], dtype=np.float32)   # test.cpp:32


def _mm32(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """float32 matrix product with double accumulation (what cv::Mat CV_32F '*' does)."""
    return (a.astype(np.float64) @ b.astype(np.float64)).astype(np.float32)


def euler_zyx(theta) -> np.ndarray:
    """helper.h:187-209 eulerAnglesToRotationMatrix: R = Rz * Ry * Rx, float32 entries."""
    tx, ty, tz = (np.float32(t) for t in theta)
    c, s = np.cos, np.sin
    rx = np.array([[1, 0, 0], [0, c(tx), -s(tx)], [0, s(tx), c(tx)]], dtype=np.float32)
    ry = np.array([[c(ty), 0, s(ty)], [0, 1, 0], [-s(ty), 0, c(ty)]], dtype=np.float32)
    rz = np.array([[c(tz), -s(tz), 0], [s(tz), c(tz), 0], [0, 0, 1]], dtype=np.float32)
    return _mm32(_mm32(rz, ry), rx)


def pose_matrix(R: np.ndarray, t: np.ndarray) -> np.ndarray:
    """Row-major 4x4 (Model::mat4x4::init_from_cv(R, t), renderer.h:125-140)."""
    m = np.eye(4, dtype=np.float32)
    m[:3, :3] = R.astype(np.float32)
    m[:3, 3] = t.astype(np.float32)
    return m


def test_cpp_angle() -> np.float32:
    """test.cpp:34 -- 10 degrees with pi ~ 3.14f, evaluated in float."""
    return np.float32(np.float32(np.float32(10.0) / np.float32(180.0)) * np.float32(3.14))


def test_cpp_poses() -> np.ndarray:
    """The two poses of test.cpp:29-46: [0] model pose, [1] scene pose. Shape (2,4,4) float32."""
    a = test_cpp_angle()
    r2 = _mm32(euler_zyx((a, a, a)), R_REN)
    return np.stack([pose_matrix(R_REN, T_MODEL), pose_matrix(r2, T_SCENE)])


def scene_pose() -> np.ndarray:
    return test_cpp_poses()[1]


class _StdUniformFloat:
    """std::uniform_real_distribution<float> over std::mt19937 as libstdc++ evaluates it:
    generate_canonical<float,24> takes ONE 32-bit draw, converts it to float, divides by 2^32
    (clamping a result of 1.0 to the float just below), then a + (b-a)*u in float."""

    def __init__(self, seed: int):
        self._bits = np.random.MT19937()
        # std::mt19937(seed) uses the classic init_genrand seeding == numpy's _legacy_seeding
        self._bits._legacy_seeding(seed)

    def draw(self, lo: float, hi: float) -> np.float32:
        raw = np.uint32(self._bits.random_raw())
        u = np.float32(np.float32(raw) / np.float32(4294967296.0))
        if u >= np.float32(1.0):
            u = np.nextafter(np.float32(1.0), np.float32(0.0))
        lo32, hi32 = np.float32(lo), np.float32(hi)
        return np.float32(lo32 + np.float32(np.float32(hi32 - lo32) * u))


def hypotheses(n: int, seed: int = 6, first: int = 0) -> np.ndarray:
    """SURVEY.md section 8d hypothesis set: poses[first:first+n] of the seeded stream.

    pose 0 is the exact test.cpp model pose; pose i>0 is
    R_i = Rzyx(alpha_i) * R_S, t_i = t_S + delta_i around the scene pose S with
    alpha ~ U(-10deg, 10deg)^3 (radians, true pi) and delta ~ U(-20, 20)^3 mm, drawn in
    (ax, ay, az, dx, dy, dz) order per pose.  Every pose index consumes its six draws, so a
    shard [first, first+n) of the stream is identical no matter how the batch is split over GPUs.
    """
    rng = _StdUniformFloat(seed)
    s = scene_pose()
    rs, ts = s[:3, :3], s[:3, 3]
    lim = np.float32(np.float32(10.0) * np.float32(np.pi) / np.float32(180.0))
    out = np.empty((n, 4, 4), dtype=np.float32)
    for i in range(first + n):
        d = [rng.draw(-lim, lim) for _ in range(3)] + [rng.draw(-20.0, 20.0) for _ in range(3)]
        if i < first:
            continue
        if i == 0:
            out[0] = test_cpp_poses()[0]
            continue
        r = _mm32(euler_zyx(d[:3]), rs)
        t = (ts + np.array(d[3:], dtype=np.float32)).astype(np.float32)
        out[i - first] = pose_matrix(r, t)
    return out


def uv_sphere_mesh(n_lon: int = 1000, n_lat: int = 500) -> np.ndarray:
    """closed UV-sphere grid, n_lon x n_lat quads split into two
    triangles each (default exactly 1 000 000 triangles), radius 60*(1+0.1 sin(5 theta) sin(7 phi)) mm.
    Returns (T,3,3) float32 triangle vertices."""
    theta = (np.arange(n_lon + 1, dtype=np.float64) % n_lon) * (2.0 * np.pi / n_lon)   # longitude, wraps
    phi = np.arange(n_lat + 1, dtype=np.float64) * (np.pi / n_lat)                    # latitude 0..pi
    th, ph = np.meshgrid(theta, phi, indexing="xy")          # (n_lat+1, n_lon+1)
    r = 60.0 * (1.0 + 0.1 * np.sin(5.0 * th) * np.sin(7.0 * ph))
    v = np.stack([r * np.sin(ph) * np.cos(th), r * np.sin(ph) * np.sin(th), r * np.cos(ph)], axis=-1)
    v = v.astype(np.float32)
    a = v[:-1, :-1]; b = v[:-1, 1:]; c = v[1:, :-1]; d = v[1:, 1:]
    t1 = np.stack([a, c, b], axis=-2).reshape(-1, 3, 3)
    t2 = np.stack([b, c, d], axis=-2).reshape(-1, 3, 3)
    return np.ascontiguousarray(np.concatenate([t1, t2], axis=0))


def intrinsics_720p() -> np.ndarray:
    """SURVEY.md section 8d config 5 intrinsics (focal x2, principal-point offset from centre x2)."""
    return np.array([1144.8228, 0.0, 650.5222, 0.0, 1147.14086, 364.09798, 0.0, 0.0, 1.0], dtype=np.float32)


def load_ply_triangles(path: str) -> np.ndarray:
    """ASCII PLY -> (T,3,3) float32 triangles in face order (stands in for the assimp import of
    cuda_renderer/renderer.cpp:16-104 on the python side; the C-ABI has pr_ply_load too)."""
    with open(path, "r") as f:
        nv = nf = 0
        while True:
            line = f.readline()
            if not line:
                raise ValueError("bad ply header")
            if line.startswith("element vertex"):
                nv = int(line.split()[2])
            elif line.startswith("element face"):
                nf = int(line.split()[2])
            elif line.startswith("end_header"):
                break
        verts = np.loadtxt([f.readline() for _ in range(nv)], dtype=np.float64)[:, :3].astype(np.float32)
        faces = np.loadtxt([f.readline() for _ in range(nf)], dtype=np.int64)
    faces = faces[faces[:, 0] == 3][:, 1:4]
    return np.ascontiguousarray(verts[faces])
