"""The C-ABI library loads on a machine without a GPU and exports every symbol that
include/pose_refine.h declares; device entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from pose_refine_amd import _lib, api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "pose_refine.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pr_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    names = declared_functions()
    assert len(names) >= 30
    assert sorted(_lib.SIGNATURES) == names


def test_every_declared_symbol_is_exported():
    lib = C.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in pose_refine.h but not exported"
    _lib.load()
    assert b"gfx950" in _lib.load().pr_version()


def test_struct_layouts():
    assert C.sizeof(_lib.SceneProjDesc) == 72          # Scene_projective is 72 B in the reference
    assert _lib.KDNODE.itemsize == 52 and _lib.RESULT.itemsize == 72
    assert C.sizeof(_lib.Roi) == 16 and C.sizeof(_lib.Criteria) == 12


def test_device_entry_points_fail_loudly_without_gpu():
    if api.device_count() > 0:
        pytest.skip("a GPU is visible; the no-device path is covered on the CPU-only box")
    with pytest.raises(api.PoseRefineError) as e:
        api.init(0)
    assert e.value.code == _lib.PR_ERR_NO_DEVICE
    with pytest.raises(api.PoseRefineError):
        api.DeviceVector(16)
    with pytest.raises(api.PoseRefineError):
        api.render_host(np.zeros((1, 3, 3), np.float32), np.eye(4, dtype=np.float32)[None], 64, 48, np.eye(4, dtype=np.float32))
