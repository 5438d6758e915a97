"""Full-size oracle fixtures for BASELINE.json configs[1..4]  ->  tests/golden/config{1,2,3,4}.npz  (VERDICT r04 item 2).

    python tools/make_golden.py [1 2 3 4]            # in the BUILD container (CPU only), a few minutes on 8 cores

Runs oracle/pose_oracle.c (the CPU restatement of /root/reference/cuda_icp/icp.cpp:125-188 with render_cpu and depth2cloud_cpu in
front of it, summation mode PO_SUM_CANONICAL at the library's points_per_block) over EVERY hypothesis of a configuration and stores,
per hypothesis: cloud size, the 16 + 2 floats of the RegistrationResult (icp.h:33-35) for fixed-20 criteria and -- for configs[1] and
[2] -- for the reference's default criteria (1e-5, 1e-5, 30; icp.h:42-45: hypotheses drop out at different passes).  The GPU tests
(tests/test_golden_full_gpu.py) then hold every hypothesis of the HIP path to these numbers instead of the one to three the oracle
has time for on the GPU box.

Provenance: fixtures are OUTPUT DATA of this repository's own oracle on seeded synthetic inputs (pose_refine_amd/synth.py,
SURVEY.md 8d); no reference source or binary is involved.  The oracle itself is pinned as DESIGN.md section 2 says.  Each file
records the git revision of oracle/pose_oracle.c it was made with and the points_per_block the sums follow.
"""
import hashlib
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import oracle_lib as O  # noqa: E402
from pose_refine_amd import dist, synth  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
PPB = 3072          # the library's default points_per_block (pr_runtime.h); the GPU test asserts the option still has this value
FIXED = (0.0, 0.0, 20)
DEFAULT = (1e-5, 1e-5, 30)


def oracle_rev():
    src = os.path.join(ROOT, "oracle", "pose_oracle.c")
    sha = hashlib.sha256(open(src, "rb").read()).hexdigest()[:16]
    try:
        rev = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%h", "--", "oracle/pose_oracle.c"], capture_output=True, text=True).stdout.strip()
    except OSError:
        rev = "?"
    return "oracle/pose_oracle.c sha256:%s git:%s" % (sha, rev)


def run(tris, poses, W, H, proj, K, scene, crit):
    t = time.time()
    res, sizes, threads = O.refine_batch(tris, poses, W, H, proj, K, scene, crit, O.SUM_CANONICAL, PPB)
    print("   %d hypotheses, criteria %s: %.1f s on %d threads" % (len(poses), crit, time.time() - t, threads), flush=True)
    return res, sizes


def pack(out, tag, res, sizes):
    out[tag + "_sizes"] = sizes.astype(np.uint32)
    out[tag + "_T"] = np.ascontiguousarray(res["T"], np.float32).reshape(len(res), 16)
    out[tag + "_fitness"] = np.ascontiguousarray(res["fitness"], np.float32)
    out[tag + "_rmse"] = np.ascontiguousarray(res["inlier_rmse"], np.float32)


def save(name, out, note):
    out["ppb"] = np.uint32(PPB)
    out["provenance"] = np.array("tools/make_golden.py; %s; %s" % (oracle_rev(), note))
    path = os.path.join(GOLDEN, name)
    np.savez_compressed(path, **out)
    print("  ->", path, os.path.getsize(path), "bytes", flush=True)


def obj06():
    tris = O.ply_load(os.path.join(GOLDEN, "obj_06.ply"))
    K = synth.K_TEST
    W, H = synth.WIDTH, synth.HEIGHT
    proj = O.compute_proj(K, W, H)
    scene_depth = O.render(tris, synth.scene_pose()[None], W, H, proj)[0]
    return tris, K, W, H, proj, scene_depth


def config1():
    print("configs[1]: obj_06, 256 hypotheses, projective", flush=True)
    tris, K, W, H, proj, sd = obj06()
    scene = O.ProjScene(sd, K)
    poses = synth.hypotheses(256)
    out = {}
    pack(out, "fixed20", *run(tris, poses, W, H, proj, K, scene, FIXED))
    pack(out, "default", *run(tris, poses, W, H, proj, K, scene, DEFAULT))
    save("config1.npz", out, "synth.hypotheses(256), Scene_projective of synth.scene_pose(), 640x480")


def config2():
    print("configs[2]: obj_06, 256 hypotheses, kd-tree", flush=True)
    tris, K, W, H, proj, sd = obj06()
    scene = O.NNScene(sd, K)
    poses = synth.hypotheses(256)
    out = {}
    pack(out, "fixed20", *run(tris, poses, W, H, proj, K, scene, FIXED))
    pack(out, "default", *run(tris, poses, W, H, proj, K, scene, DEFAULT))
    save("config2.npz", out, "synth.hypotheses(256), Scene_nn of synth.scene_pose(), 640x480")


def config3():
    print("configs[3]: obj_06, rank 5's 512 of 4096 hypotheses, projective", flush=True)
    tris, K, W, H, proj, sd = obj06()
    scene = O.ProjScene(sd, K)
    first, count = dist.shard_bounds(4096, 5, 8)
    poses = synth.hypotheses(count, first=first)
    out = {"first": np.uint32(first)}
    pack(out, "fixed20", *run(tris, poses, W, H, proj, K, scene, FIXED))
    save("config3.npz", out, "synth.hypotheses(512, first=2560): rank 5 of 8 over 4096")


def config4():
    print("configs[4]: 1M-triangle mesh, 1280x720, 128 hypotheses (one GPU's share of 1024)", flush=True)
    W, H = 1280, 720
    K = synth.intrinsics_720p()
    tris = synth.uv_sphere_mesh()
    proj = O.compute_proj(K, W, H)
    sd = O.render(tris, synth.scene_pose()[None], W, H, proj)[0]
    scene = O.ProjScene(sd, K)
    poses = synth.hypotheses(128)
    out = {}
    pack(out, "fixed20", *run(tris, poses, W, H, proj, K, scene, FIXED))
    save("config4.npz", out, "synth.uv_sphere_mesh(), synth.intrinsics_720p(), synth.hypotheses(128), 1280x720")


if __name__ == "__main__":
    O.build()
    which = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]
    for c in which:
        {1: config1, 2: config2, 3: config3, 4: config4}[c]()
