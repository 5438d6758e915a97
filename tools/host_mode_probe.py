#!/usr/bin/env python
"""Diagnostic (round 6): the host-solve pipeline (PR_SOLVE_HOST through the two slots' helper threads) runs in one of two modes from process to
process (~245 k or ~210 k poses/s).  Is the mode a property of the PROCESS or of the helper threads / their streams?  The pipeline is run, the
context is shut down (helper threads, streams and all) and started again, several times in one process.   tools/host_mode_probe.py [rounds]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
if os.environ.get("PROBE_TORCH"):
    import torch
    torch.cuda.set_device(0); torch.cuda.init()
    if os.environ["PROBE_TORCH"] == "2": torch.zeros(4, device="cuda").sum().item()
from pose_refine_amd import api, synth, _lib

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
W, H, K = synth.WIDTH, synth.HEIGHT, synth.K_TEST
for rnd in range(rounds):
    api.init(0)
    api.set_option("solve", api.SOLVE_HOST)
    model = api.Model(os.path.join(ROOT, "tests", "golden", "obj_06.ply"))
    proj = api.compute_proj(K, W, H)
    sd = api.render_host(model, synth.scene_pose()[None], W, H, proj)[0]
    scene = api.Scene_projective().init_Scene_projective_cuda(sd, K)
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 20)
    poses = synth.hypotheses(256)
    rdev = api.DeviceVector(256 * 18, np.float32) if os.environ.get("PROBE_RESULTS_DEV") else None
    kw = dict(results_dev=rdev.data(), also_host=True) if rdev is not None else {}
    rates = []
    for part in range(3):
        for k in range(4):
            api.refine_submit(k & 1, model, poses, W, H, proj, K, scene, crit, **kw)
            if k: api.refine_wait((k - 1) & 1)
        api.refine_wait(1)
        t0 = time.perf_counter(); steps = 40
        for k in range(steps):
            api.refine_submit(k & 1, model, poses, W, H, proj, K, scene, crit, **kw)
            if k: api.refine_wait((k - 1) & 1)
        api.refine_wait((steps - 1) & 1)
        rates.append(256 * steps / (time.perf_counter() - t0))
    print("round", rnd, " ".join(f"{r/1e3:.1f}k" for r in rates), flush=True)
    del scene, model, rdev
    _lib.check(_lib.load().pr_shutdown())
