import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "device_solve: the test starts with the 6x6 solve on the device (PR_SOLVE_DEVICE); default: on the host")


# ---- GPU fixtures shared by the -m gpu files ---------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def gpu():
    from pose_refine_amd import api
    api.init(0)
    return True


@pytest.fixture(autouse=True)
def _solve_default(request):
    """@pytest.mark.device_solve: the headline configuration (iterations stay on the device); restored to the library default afterwards."""
    if request.node.get_closest_marker("device_solve") is None:
        yield
        return
    from pose_refine_amd import api
    api.init(0)
    api.set_option("solve", api.SOLVE_DEVICE)
    yield
    api.set_option("solve", api.SOLVE_HOST)


@pytest.fixture(scope="module")
def model(gpu, golden_dir):
    from pose_refine_amd import api
    return api.Model(os.path.join(golden_dir, "obj_06.ply"))


@pytest.fixture(scope="module")
def gscenes(gpu, scenario):
    from pose_refine_amd import api
    d = scenario["depth"][1]
    return dict(proj=api.Scene_projective().init_Scene_projective_cuda(d, scenario["K"]),
                nn=api.Scene_nn().init_Scene_nn_cuda(d, scenario["K"]))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def obj06_tris():
    import oracle_lib as O
    return O.ply_load(os.path.join(GOLDEN, "obj_06.ply"))


@pytest.fixture(scope="session")
def scenario(obj06_tris):
    """test.cpp:22-86 scenario evaluated by the ORACLE: depth of both poses, model cloud, both scenes."""
    import numpy as np
    import oracle_lib as O
    from pose_refine_amd import synth
    K = synth.K_TEST
    proj = O.compute_proj(K, synth.WIDTH, synth.HEIGHT)
    poses = synth.test_cpp_poses()
    depth = O.render(obj06_tris, poses, synth.WIDTH, synth.HEIGHT, proj)
    cloud = O.depth2cloud(depth[0], K)
    return dict(K=K, proj=proj, poses=poses, depth=depth, cloud=cloud, tris=obj06_tris,
                proj_scene=O.ProjScene(depth[1], K), nn_scene=O.NNScene(depth[1], K))
