"""cuda_icp/geometry.h is the one piece of the reference this image can compile as it lies (std-only without CUDA_ON):
oracle/Makefile `ref` builds it verbatim under oracle/ref_driver/geometry_ref.cpp, `fixtures` writes the bit patterns it
computes to tests/golden/geometry_h.json.  Everything that restates that header is compared with it bit for bit:
the C++ adapter (include/cuda_icp/geometry.h, through the very same driver), the oracle's po_mat4_mul and the library's
pr_mat4_mul (the source the device-side ICP loop accumulates its transform with).  CPU only -- row a15 of SURVEY.md 8a."""
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from pose_refine_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gold(golden_dir):
    with open(os.path.join(golden_dir, "geometry_h.json")) as f:
        return json.load(f)


def f32(bits):
    return np.array(bits, np.uint32).view(np.float32)


def test_adapter_header_reproduces_the_reference_header(gold, tmp_path):
    exe = str(tmp_path / "geometry_twin")
    subprocess.run(["g++", "-std=c++14", "-O2", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "include", "cuda_icp"),
                    os.path.join(ROOT, "oracle", "ref_driver", "geometry_ref.cpp"), "-o", exe], check=True)
    got = json.loads(subprocess.run([exe], check=True, capture_output=True, text=True).stdout)
    assert got["cases"] == gold["cases"] and got["identity4"] == gold["identity4"]


def test_mat4_products_bitwise(gold):
    lib = _lib.load()
    for c in gold["cases"]:
        A, B, want = f32(c["A"]), f32(c["B"]), np.array(c["A_mul_B"], np.uint32)
        out = np.zeros(16, np.float32)
        O.lib().po_mat4_mul(A, B, out)
        assert np.array_equal(out.view(np.uint32), want)
        out2 = np.zeros(16, np.float32)
        lib.pr_mat4_mul(A.ctypes.data, B.ctypes.data, out2.ctypes.data)
        assert np.array_equal(out2.view(np.uint32), want)
        acc = B.copy()                                             # in place, as `T = E * T` does
        lib.pr_mat4_mul(A.ctypes.data, acc.ctypes.data, acc.ctypes.data)
        assert np.array_equal(acc.view(np.uint32), want)


def test_fixture_is_self_consistent(gold):
    """numpy re-derivation of the order-independent entries (guards against a stale or hand-edited fixture)."""
    assert len(gold["cases"]) == 24
    assert np.array_equal(f32(gold["identity4"]).reshape(4, 4), np.eye(4, dtype=np.float32))
    for c in gold["cases"]:
        A, p, q = f32(c["A"]).reshape(4, 4), f32(c["p"]), f32(c["q"])
        assert np.array_equal(f32(c["A_transpose"]).reshape(4, 4), A.T)
        assert np.array_equal(f32(c["p_plus_q"]), p + q) and np.array_equal(f32(c["p_minus_q"]), p - q)
        assert np.allclose(f32(c["cross_pq"]), np.cross(p.astype(np.float64), q.astype(np.float64)), atol=1e-6)
        assert np.allclose(f32(c["A_mul_B"]).reshape(4, 4), A.astype(np.float64) @ f32(c["B"]).reshape(4, 4).astype(np.float64), rtol=1e-5, atol=1e-3)
        assert c["vec3i_of_1000p"] == [int(np.float32(np.float32(v) * np.float32(1000.0)) + np.float32(0.5)) for v in p]
