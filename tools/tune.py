#!/usr/bin/env python
"""Quick on-GPU A/B harness: runs the fused batch with option sweeps and prints per-phase timings."""
import argparse, os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pose_refine_amd import api, synth

PROFILE = 1
def run(model, poses, proj, K, scene, crit, steps=8, warm=2):
    for _ in range(warm): api.refine_batch(model, poses, 640, 480, proj, K, scene, crit)
    api.set_option("profile", PROFILE); api.profile_reset()
    t0 = time.perf_counter()
    for _ in range(steps): res, sizes = api.refine_batch(model, poses, 640, 480, proj, K, scene, crit)
    dt = (time.perf_counter() - t0) / steps
    api.set_option("profile", 0)
    p = api.profile_read()
    n = max(1, p["icp_launches"])
    return dict(ms_step=dt * 1e3, poses_s=len(poses) / dt, icp_us=p["icp_kernel_ms"] * 1e3 / n, render_ms=p["render_ms"] / steps,
                cloud_ms=p["cloud_ms"] / steps, gbs=p["icp_bytes"] / (p["icp_kernel_ms"] * 1e-3) / 1e9 if p["icp_kernel_ms"] else 0,
                chk=float(np.sum(res["fitness"])))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--poses", type=int, nargs="+", default=[256])
    ap.add_argument("--scene", default="proj")
    ap.add_argument("--opt", action="append", default=[], help="name=v1,v2,... sweep")
    ap.add_argument("--solve", default="device")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--profile", type=int, default=1)
    a = ap.parse_args()
    global PROFILE
    PROFILE = 0 if a.no_profile else a.profile
    api.init(0)
    api.set_option("solve", 1 if a.solve == "device" else 0)
    model = api.Model(os.path.join(ROOT, "tests/golden/obj_06.ply"))
    K = synth.K_TEST; proj = api.compute_proj(K, 640, 480)
    sd = api.render_host(model, synth.scene_pose()[None], 640, 480, proj)[0]
    scene = api.Scene_projective().init_Scene_projective_cuda(sd, K) if a.scene == "proj" else api.Scene_nn().init_Scene_nn_cuda(sd, K)
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 20)
    sweeps = [("none", [0])]
    if a.opt:
        sweeps = [(o.split("=")[0], [int(v) for v in o.split("=")[1].split(",")]) for o in a.opt]
    for P in a.poses:
        poses = synth.hypotheses(P)
        for name, vals in sweeps:
            for v in vals:
                if name != "none": api.set_option(name, v)
                r = run(model, poses, proj, K, scene, crit, a.steps)
                print(f"P={P} {name}={v}: " + " ".join(f"{k}={val:.4g}" for k, val in r.items()), flush=True)
main()
