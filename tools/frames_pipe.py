#!/usr/bin/env python
"""A NEW scene every frame, frames pipelined through the two asynchronous slots: the scene prepared right before its submit (two scene
objects), one frame ahead (three), or ahead by one to three helper threads with private contexts; host time per frame by call.  The figures of
profiles/r05/README.md "A new scene per frame"."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import faulthandler, signal
faulthandler.register(signal.SIGUSR1, all_threads=True)          # kill -USR1: where every thread is (a stuck run)
from pose_refine_amd import api, synth
W, H = 640, 480
api.init(0); api.set_option("solve", 1)
for kv in filter(None, os.environ.get("PR_OPTS", "").split(",")):
    k_, v_ = kv.split("="); api.set_option(k_, int(v_))
model = api.Model(os.path.join(ROOT, "tests/golden/obj_06.ply"))
K = synth.K_TEST; proj = api.compute_proj(K, W, H)
obj = api.render_host(model, synth.scene_pose()[None], W, H, proj)[0].astype(np.int32)
poses = synth.hypotheses(256)
crit = api.ICPConvergenceCriteria(0.0, 0.0, 20)
N = 40
for kind in ("proj", "nn"):
    scenes = [api.Scene_projective() if kind == "proj" else api.Scene_nn() for _ in range(2)]
    devs = []
    for f in range(N + 4):
        d = obj.copy(); d[f % H, f % W] = 0 if d[f % H, f % W] else 905
        devs.append(api.DeviceVector.from_host(d.reshape(-1)))
    def prep(k):
        s = scenes[k & 1]
        if kind == "proj": s.init_Scene_projective_device(devs[k], K, W, H)
        else: s.init_Scene_nn_device(devs[k], K, W, H)
        return s
    # synchronous frames
    api.sync(); t0 = time.perf_counter()
    for k in range(N // 2):
        api.refine_batch(model, poses, W, H, proj, K, prep(k), crit)
    api.sync(); sync_ms = 1e3 * (time.perf_counter() - t0) / (N // 2)
    # pipelined frames: prepare frame k while frame k-1 runs in the other slot
    for k in range(4):
        api.refine_submit(k & 1, model, poses, W, H, proj, K, prep(k), crit)
        if k: api.refine_wait((k - 1) & 1)
    api.refine_wait(3 & 1)
    api.sync(); t0 = time.perf_counter()
    for k in range(4, N + 4):
        api.refine_submit(k & 1, model, poses, W, H, proj, K, prep(k), crit)
        if k > 4: api.refine_wait((k - 1) & 1)
    api.refine_wait((N + 3) & 1)
    api.sync(); pipe_ms = 1e3 * (time.perf_counter() - t0) / N
    # three scene objects: the next frame's scene is prepared while BOTH slots hold a batch
    scenes = [api.Scene_projective() if kind == "proj" else api.Scene_nn() for _ in range(3)]
    def prep3(k):
        s = scenes[k % 3]
        if kind == "proj": s.init_Scene_projective_device(devs[k % len(devs)], K, W, H)
        else: s.init_Scene_nn_device(devs[k % len(devs)], K, W, H)
        return s
    nxt = prep3(0)
    for k in range(4):
        api.refine_submit(k & 1, model, poses, W, H, proj, K, nxt, crit)
        nxt = prep3(k + 1)
        if k: api.refine_wait((k - 1) & 1)
    api.refine_wait(3 & 1)
    api.sync(); t0 = time.perf_counter()
    ts = [0.0, 0.0, 0.0]
    for k in range(4, N + 4):
        a = time.perf_counter()
        api.refine_submit(k & 1, model, poses, W, H, proj, K, nxt, crit)
        b = time.perf_counter()
        nxt = prep3(k + 1)
        c = time.perf_counter()
        if k > 4: api.refine_wait((k - 1) & 1)
        d = time.perf_counter()
        ts[0] += b - a; ts[1] += c - b; ts[2] += d - c
    api.refine_wait((N + 3) & 1)
    api.sync(); pipe3_ms = 1e3 * (time.perf_counter() - t0) / N
    print(f"{kind}:   host time per frame: submit {1e3*ts[0]/N:.3f} ms, prepare {1e3*ts[1]/N:.3f} ms, wait {1e3*ts[2]/N:.3f} ms", flush=True)
    print(f"{kind}: prepared one frame ahead (three scene objects): {pipe3_ms:.3f} ms/frame", flush=True)
    # helper threads (private contexts: own stream and workspaces) prepare scenes ahead; the main thread only submits and waits
    import threading, queue
    for n_prod in (1, 2, 3):
        n_obj = 3 * n_prod
        scenes4 = [api.Scene_projective() if kind == "proj" else api.Scene_nn() for _ in range(n_obj)]
        ready = [queue.Queue() for _ in range(n_prod)]
        # every producer has scene objects of its own (round 6: with ONE pool two quick producers could take every free object for frames that are not due
        # yet and starve the one whose frame is -- the main thread then waits for ever; it showed once the preparation had lost its copy-command stalls)
        free = [queue.Queue() for _ in range(n_prod)]
        for i in range(n_obj): free[i % n_prod].put(i)
        def producer(pi):
            api.thread_context(True)
            for k in range(pi, N + 8, n_prod):
                i = free[pi].get()
                s = scenes4[i]
                if kind == "proj": s.init_Scene_projective_device(devs[k % len(devs)], K, W, H)
                else: s.init_Scene_nn_device(devs[k % len(devs)], K, W, H)
                ready[pi].put(i)
            api.thread_context(False)
        ths = [threading.Thread(target=producer, args=(pi,)) for pi in range(n_prod)]
        for th in ths: th.start()
        held = {}
        t0 = None
        for k in range(N + 8):
            if k == 8: api.sync(); t0 = time.perf_counter()
            i = ready[k % n_prod].get()
            api.refine_submit(k & 1, model, poses, W, H, proj, K, scenes4[i], crit)
            held[k] = i
            if k:
                api.refine_wait((k - 1) & 1)
                j = held.pop(k - 1); free[j % n_prod].put(j)
        api.refine_wait((N + 7) & 1)
        api.sync(); thr_ms = 1e3 * (time.perf_counter() - t0) / N
        for th in ths: th.join()
        print(f"{kind}: scenes prepared ahead by {n_prod} helper thread(s): {thr_ms:.3f} ms/frame", flush=True)
    print(f"{kind}: new scene every frame: synchronous {sync_ms:.3f} ms/frame, pipelined through the two slots {pipe_ms:.3f} ms/frame", flush=True)
