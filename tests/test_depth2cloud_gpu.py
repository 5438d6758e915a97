"""GPU tests (-m gpu) of cuda_icp::depth2cloud_cuda<T> (icp.cu:228-291; row a5): counts, offsets, strides, both depth types.
The HIP path is called through the C ABI (pose_refine_amd.api) and held to the CPU oracle on the same inputs: integers, inlier counts and
per-pass sums bit-exact, transforms within 1e-4 (north_star).  @pytest.mark.device_solve = the 6x6 solve runs on the device (the headline
configuration); without it the solve is on the host, as icp.cu:207 does it."""
import ctypes as C
import json
import os
import threading

import numpy as np
import pytest

import oracle_lib as O
from pose_refine_amd import _lib, api, synth
from gpu_common import *  # noqa: F401,F403 -- W, H, TOL_T, inliers, raw_h2d, make_scene, random_mesh ...

pytestmark = pytest.mark.gpu


# ---- depth -> cloud -------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.int32, np.uint16])
def test_depth2cloud_bit_exact(gpu, scenario, dtype):
    d = scenario["depth"][0].astype(dtype)
    dev = api.DeviceVector.from_host(d.reshape(-1))
    got = api.depth2cloud(dev, W, H, scenario["K"], dtype=dtype).to_host().reshape(-1, 3)
    ref = O.depth2cloud(d, scenario["K"])
    assert got.shape == ref.shape and np.array_equal(got, ref)


def test_depth2cloud_crop_offsets_and_empty(gpu, scenario):
    d = np.ascontiguousarray(scenario["depth"][0][80:320, 160:480])    # cropped render, tl = (160, 80)
    dev = api.DeviceVector.from_host(d.reshape(-1))
    got = api.depth2cloud(dev, 320, 240, scenario["K"], 1, 160, 80).to_host().reshape(-1, 3)
    assert np.array_equal(got, O.depth2cloud(d, scenario["K"], 1, 160, 80))
    z = api.DeviceVector.from_host(np.zeros(64 * 48, np.int32))
    assert api.depth2cloud(z, 64, 48, scenario["K"]).size() == 0


def test_depth2cloud_stride2(gpu, scenario):
    d = scenario["depth"][0]
    dev = api.DeviceVector.from_host(d.reshape(-1))
    got = api.depth2cloud(dev, W, H, scenario["K"], 2).to_host().reshape(-1, 3)
    assert np.array_equal(got, O.depth2cloud(d, scenario["K"], 2))


@pytest.mark.device_solve
@pytest.mark.parametrize("W,H,stride,tlx,tly,dt", [(97, 61, 3, 0, 0, np.int32), (101, 57, 7, 5, 9, np.uint16), (64, 48, 64, 0, 0, np.int32), (33, 2, 5, 1, 1, np.int32), (1, 1, 1, 0, 0, np.uint16)])
def test_depth2cloud_strides_that_do_not_divide_the_frame(gpu, W, H, stride, tlx, tly, dt):
    rng = np.random.default_rng(W * H + stride)
    d = (rng.integers(0, 3, size=(H, W)) * rng.integers(200, 900, size=(H, W))).astype(dt)
    K = np.array([80, 0, W / 2, 0, 82, H / 2, 0, 0, 1], np.float32)
    got = api.depth2cloud(api.DeviceVector.from_host(d.reshape(-1)), W, H, K, stride, tlx, tly, dtype=dt).to_host().reshape(-1, 3)
    want = O.depth2cloud(d, K, stride, tlx, tly)
    assert got.shape == want.shape and np.array_equal(got, want)


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
