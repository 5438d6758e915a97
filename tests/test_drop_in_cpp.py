"""The C++ adapter headers (include/cuda_renderer, include/cuda_icp) keep the reference's API
source-compatible on top of the C ABI.  tests/cpp/drop_in_test.cpp is the GPU half of the
reference's own end-to-end driver (test.cpp:22-46,143-172) written with the reference's names."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "drop_in_test")


def compile_driver():
    lib_dir = os.path.join(ROOT, "pose_refine_amd", "lib")
    cmd = ["g++", "-std=c++14", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "drop_in_test.cpp"), "-o", EXE,
           "-L" + lib_dir, "-lpose_refine_hip", "-Wl,-rpath," + lib_dir]
    subprocess.run(cmd, check=True)
    return EXE


def test_adapters_compile_as_plain_cxx14():
    """Host code is plain C++14 (the reference's standard, CMakeLists.txt:2) -- no hipcc needed."""
    from pose_refine_amd import build
    build.build()
    assert os.path.exists(compile_driver())


@pytest.mark.gpu
def test_drop_in_driver_matches_oracle(scenario):
    import oracle_lib as O
    exe = compile_driver()
    out = subprocess.run([exe, os.path.join(ROOT, "tests", "golden") + "/"], check=True, capture_output=True, text=True).stdout
    got = json.loads(out[out.index("{"):])
    assert got["n_triangles"] == 31468 and got["cloud_points"] == len(scenario["cloud"])
    assert got["depth_sum"] == [int(scenario["depth"][0].sum()), int(scenario["depth"][1].sum())]
    assert got["pose_renderer"]["depth_sum"] == got["depth_sum"]                 # PoseRenderer + raw2depth_mask
    assert got["pose_renderer"]["mask_px"] == [int((scenario["depth"][i] > 0).sum()) for i in range(2)]
    from pose_refine_amd import api
    ppb = api.get_option("points_per_block")                      # the C++ driver runs with the library's defaults: the oracle sums in the same tree
    for key, sk, crit in [("proj_default", "proj_scene", (1e-5, 1e-5, 30)), ("nn_fixed20", "nn_scene", (0.0, 0.0, 20))]:
        ref, _, _, _ = O.icp(scenario["cloud"], scenario[sk], crit, O.SUM_CANONICAL, ppb)
        assert np.float32(got[key]["fitness"]) == ref["fitness"]
        assert np.allclose(np.array(got[key]["T"], np.float32), ref["T"], rtol=0, atol=1e-4)
        assert got[key]["rmse"] == pytest.approx(float(ref["inlier_rmse"]), rel=1e-6)


def test_cpu_twins_match_survey_known_answers(golden_dir):
    """render_cpu / depth2cloud_cpu / init_Scene_*_cpu / ICP_Point2Plane_cpu of the adapter headers (the CPU half of
    test.cpp:48-129) reproduce the reference's known answers -- no GPU involved."""
    from pose_refine_amd import build
    build.build()
    lib_dir = os.path.join(ROOT, "pose_refine_amd", "lib")
    exe = os.path.join(ROOT, "tests", "cpp", "cpu_twins_test")
    subprocess.run(["g++", "-std=c++14", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "cpu_twins_test.cpp"), "-o", exe,
                    "-L" + lib_dir, "-lpose_refine_hip", "-Wl,-rpath," + lib_dir], check=True)
    out = subprocess.run([exe, golden_dir + "/"], check=True, capture_output=True, text=True).stdout
    got = json.loads(out[out.index("{"):])
    with open(os.path.join(golden_dir, "survey_8c.json")) as f:
        gold = json.load(f)
    assert got["depth_sum"] == [gold["render"][0]["sum"], gold["render"][1]["sum"]]
    assert got["cloud_points"] == gold["cloud_points"] and got["kd_nodes"] == gold["kdtree"]["nodes"]
    for key in ("proj_default", "nn_default"):
        g = gold["icp"][key]
        assert int(round(got[key]["fitness"] * gold["cloud_points"])) == g["inliers"]
        assert got[key]["rmse"] == pytest.approx(g["rmse"], rel=1e-6)
        T = np.array(got[key]["T"], np.float32).reshape(4, 4)
        for r, row in enumerate(g["T_rows"]):
            assert np.allclose(T[r], np.array(row, np.float32), rtol=0, atol=1e-6)


def compile_thrust_driver():
    lib_dir = os.path.join(ROOT, "pose_refine_amd", "lib")
    exe = os.path.join(ROOT, "tests", "cpp", "thrust_holder_test")
    hipcc = "/opt/rocm/bin/hipcc"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "thrust_holder_test.cpp"), "-o", exe,
                    "-L" + lib_dir, "-lpose_refine_hip", "-Wl,-rpath," + lib_dir], check=True)
    return exe


def test_holders_expose_thrust_pointers_under_hipcc():
    """device_vector_holder::begin_thr / end_thr (renderer.h:175-177, common.h:30-33) exist when the adapters are compiled by
    hipcc with rocThrust: the reference's thrust::copy lines build unchanged.  (g++ builds use upload / download.)"""
    from pose_refine_amd import build
    build.build()
    assert os.path.exists(compile_thrust_driver())


@pytest.mark.gpu
def test_thrust_copy_through_holders_runs():
    exe = compile_thrust_driver()
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert out.strip().endswith("OK")


def compile_shard_driver():
    lib_dir = os.path.join(ROOT, "pose_refine_amd", "lib")
    exe = os.path.join(ROOT, "tests", "cpp", "shard_test")
    subprocess.run(["g++", "-std=c++14", "-O2", "-Wall", "-pthread", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "shard_test.cpp"), "-o", exe,
                    "-L" + lib_dir, "-lpose_refine_hip", "-Wl,-rpath," + lib_dir], check=True)
    return exe


def test_cpp_shard_driver_compiles():
    from pose_refine_amd import build
    build.build()
    assert os.path.exists(compile_shard_driver())


@pytest.mark.gpu
def test_cpp_host_shards_over_all_visible_gpus(golden_dir):
    """tests/cpp/shard_test.cpp: one host thread per visible GPU, pr_comm_init_all + pr_gather_results (RCCL), virtual ranks and
    private-context threads -- all bit-identical to the unsharded batch (no Python on that side)."""
    exe = compile_shard_driver()
    r = subprocess.run([exe, golden_dir + "/"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    got = json.loads(r.stdout[r.stdout.rindex("{"):])
    assert got["failures"] == 0 and got["devices"] >= 1 and got["mean_fitness"] > 0.3


@pytest.mark.gpu
def test_adapters_at_their_edges(golden_dir):
    """No hypotheses, an image in which nothing is visible, clouds without points, holders that never held memory -- through the C++ adapters:
    empty vectors and identity results (icp.cu:183), no error exit."""
    from pose_refine_amd import build
    build.build()
    lib_dir = os.path.join(ROOT, "pose_refine_amd", "lib")
    exe = os.path.join(ROOT, "tests", "cpp", "edge_cases_test")
    subprocess.run(["g++", "-std=c++14", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "edge_cases_test.cpp"),
                    "-o", exe, "-L" + lib_dir, "-lpose_refine_hip", "-Wl,-rpath," + lib_dir], check=True)
    out = subprocess.run([exe, golden_dir + "/"], check=True, capture_output=True, text=True).stdout
    got = json.loads(out[out.index("{\"render_none\""):])
    assert got["render_none"] == 0 and got["keep_none"] == 0 and got["empty_cloud"] == 0 and got["pose_renderer_none"] == 0
    assert got["full_cloud"] > 20000 and got["identity"] == [1, 1, 1, 1] and got["full_fitness"] == 1.0
