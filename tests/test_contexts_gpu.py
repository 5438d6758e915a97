"""GPU tests (-m gpu) of contexts and host threads (README.md:15: many host threads, each refining its own hypothesis), the C-ABI gather, misuse of entry points.
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


def test_no_device_option_errors(gpu):
    with pytest.raises(api.PoseRefineError):
        api.set_option("no_such_option", 1)
    with pytest.raises(api.PoseRefineError):
        api.set_option("points_per_block", 1000)


# ---- the reference's threading contract: many host threads, each refining its own hypothesis (README.md:15) --------------
@pytest.mark.device_solve
def test_host_threads_with_private_contexts(gpu, scenario, gscenes):
    cloud = scenario["cloud"]
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 8)
    shifts = [np.array([0.001 * k, -0.0005 * k, 0.0008 * k], np.float32) for k in range(6)]
    want = []
    for sh in shifts:
        want.append(api.ICP_Point2Plane(api.DeviceVector.from_host((cloud + sh).reshape(-1)), gscenes["proj"], crit))
    got = [None] * len(shifts)
    errs = []

    def work(k):
        try:
            api.thread_context(True)                             # own stream + workspaces, like cudaStreamPerThread
            for _ in range(3):
                dev = api.DeviceVector.from_host((cloud + shifts[k]).reshape(-1))
                got[k] = api.ICP_Point2Plane(dev, gscenes["proj"], crit)
                dev.free()
            api.thread_context(False)
        except Exception as e:                                   # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(k,)) for k in range(len(shifts))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    for g, w in zip(got, want):
        assert g.fitness_ == w.fitness_ and g.inlier_rmse_ == w.inlier_rmse_ and np.array_equal(g.transformation_, w.transformation_)


@pytest.mark.device_solve
def test_async_slots_in_private_contexts_and_across_shutdown(gpu, model, scenario, gscenes):
    """Every host thread with a private context has two asynchronous slots of its own (and the Python mirror keeps each thread's
    in-flight output arrays alive separately); pr_shutdown releases a context, the next call builds a new one."""
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 5)
    poses = synth.hypotheses(70)
    api.set_option("nn_count", 1)                                # instrumented kd-tree runs are synchronous
    api.set_option("profile", 1)                                 # timed calls are synchronous
    try:
        want, want_sizes = api.refine_batch(model, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit)
        want_nn, _ = api.refine_batch(model, poses[:20], W, H, scenario["proj"], scenario["K"], gscenes["nn"], crit)
    finally:
        api.set_option("profile", 0)
        api.set_option("nn_count", 0)

    def both_slots(tag):
        for _ in range(3):
            api.refine_submit(0, model, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit)
            api.refine_submit(1, model, poses[:20], W, H, scenario["proj"], scenario["K"], gscenes["nn"], crit)
            a, sa = api.refine_wait(0)
            b, _ = api.refine_wait(1)
            assert a.tobytes() == want.tobytes() and np.array_equal(sa, want_sizes), tag
            assert b.tobytes() == want_nn.tobytes(), tag

    both_slots("shared context")
    api.shutdown()
    both_slots("after pr_shutdown")
    errs = []

    def work(k):
        try:
            api.thread_context(True)
            both_slots(f"private context of thread {k}")
            api.thread_context(False)
        except Exception as e:                                   # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    both_slots("shared context again")


# ---- C-ABI gather (RCCL); one GPU here: world 1, the communicator and the collective still run ------------------------------
@pytest.mark.device_solve
def test_cabi_gather_world1(gpu, model, scenario, gscenes):
    P = 40
    poses = synth.hypotheses(P)
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 3)
    res, _ = api.refine_batch(model, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit)
    send = api.DeviceVector.from_host(res.view(np.float32).reshape(-1))
    recv = api.DeviceVector(P * 18, np.float32)
    api.comm_init_rank(api.comm_id(), 0, 1)                      # ncclCommInitRank with one rank
    try:
        assert api.comm_rank() == (0, 1)
        api.gather_results(send.data(), P, P, 0, recv.data())
        api.sync()
        assert recv.to_host().tobytes() == res.tobytes()
        with pytest.raises(api.PoseRefineError):
            api.gather_results(send.data(), P - 1, P, 0, recv.data())      # not this rank's shard size
    finally:
        api.comm_destroy()
    api.comm_init_all(1)                                          # the single-process form (ncclCommInitAll)
    try:
        recv2 = api.DeviceVector(P * 18, np.float32)
        api.gather_results(send.data(), P, P, 0, recv2.data())
        api.sync()
        assert recv2.to_host().tobytes() == res.tobytes()
    finally:
        api.comm_destroy()


@pytest.mark.device_solve
def test_four_host_threads_on_the_shared_context(gpu, model, scenario, gscenes):
    """Calls of different kinds from four threads at once on the process' shared context: every result equals the one computed alone."""
    import threading
    poses = synth.hypotheses(24, seed=8)
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 5)
    cloud = scenario["cloud"]

    def run(what):
        if what == "refine":
            return api.refine_batch(model, poses, W, H, scenario["proj"], scenario["K"], gscenes["proj"], crit)[0].tobytes()
        if what == "nn":
            return api.refine_batch(model, poses[:8], W, H, scenario["proj"], scenario["K"], gscenes["nn"], crit)[0].tobytes()
        if what == "render":
            return api.render_host(model, poses[:4], W, H, scenario["proj"]).tobytes()
        return api.ICP_Point2Plane(api.DeviceVector.from_host(cloud.reshape(-1)), gscenes["proj"], crit).transformation_.tobytes()
    kinds = ("refine", "nn", "render", "icp")
    alone = {k: run(k) for k in kinds}
    bad = []

    def worker(tid):
        for k in range(10):
            what = kinds[(tid + k) % 4]
            if run(what) != alone[what]:
                bad.append((tid, k, what))
    ts = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not bad


@pytest.mark.device_solve
@pytest.mark.parametrize("kind", ["proj", "nn"])
def test_host_solve_batches_from_two_threads_with_private_contexts(gpu, model, scenario, gscenes, kind):
    """The reference's way of feeding the GPU (README.md:15): host threads, each with its own context, issue whole batches with the solve on
    the host.  Every batch equals the one computed alone."""
    import threading
    poses = synth.hypotheses(70, seed=31)
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 6)
    api.set_option("solve", api.SOLVE_HOST)
    try:
        alone = api.refine_batch(model, poses, W, H, scenario["proj"], scenario["K"], gscenes[kind], crit)
        bad, errs = [], []

        def work():
            try:
                api.thread_context(True)
                for _ in range(6):
                    out = api.refine_batch(model, poses, W, H, scenario["proj"], scenario["K"], gscenes[kind], crit)
                    if out[0].tobytes() != alone[0].tobytes() or not np.array_equal(out[1], alone[1]):
                        bad.append(1)
                api.thread_context(False)
            except Exception as e:                               # noqa: BLE001
                errs.append(e)
        ts = [threading.Thread(target=work) for _ in range(2)]
        [t.start() for t in ts]
        [t.join() for t in ts]
    finally:
        api.set_option("solve", api.SOLVE_DEVICE)
    assert not errs and not bad, (errs, len(bad))


@pytest.mark.device_solve
def test_misused_entry_points_return_codes_and_leave_no_trace(gpu):
    """Null and foreign pointers, a double free, unknown options and slots: error codes (a null destination of pr_fill_i32 used to reach the
    kernel and fault the GPU), and the call after a refused one works (the runtime's sticky copy of an error is not found by the next launch)."""
    lib = _lib.load()
    assert lib.pr_fill_i32(None, 10, 5) == -3
    assert lib.pr_malloc(None, 16) == -3
    host = np.zeros(16, np.float32)
    assert lib.pr_memcpy_h2d(None, host.ctypes.data, 64) != 0          # refused by the runtime ...
    d = api.DeviceVector(64, np.int32)
    assert lib.pr_fill_i32(d.data(), 64, 7) == 0                          # ... and the next launch does not inherit that error
    assert np.array_equal(d.to_host(), np.full(64, 7, np.int32))
    q = C.c_void_p()
    assert lib.pr_malloc(C.byref(q), 1024) == 0 and lib.pr_free(q) == 0 and lib.pr_free(q) != 0 and lib.pr_free(None) == 0
    assert lib.pr_fill_i32(d.data(), 64, 9) == 0 and int(d.to_host()[0]) == 9
    assert lib.pr_set_option(None, 1) == -3 and lib.pr_set_option(b"nonsense", 1) == -3 and lib.pr_refine_wait(7) == -3 and lib.pr_refine_wait(-1) == -3


@pytest.mark.device_solve
@pytest.mark.parametrize("kind", ["proj", "nn"])
def test_scenes_prepared_by_a_helper_thread_while_batches_run(gpu, model, scenario, kind):
    """The per-frame pattern of tools/frames_pipe.py: a helper thread with a private context prepares the NEXT frames' scenes on the device (its own stream and
    build workspace) while the caller's two slots refine against the scenes it handed over before.  What the helper writes is seen by the caller's context
    (the write log is the library's, not a context's: the caller's cached records of a re-used scene object are stale the moment it is re-initialised)."""
    import queue
    K, proj = scenario["K"], scenario["proj"]
    poses = synth.hypotheses(40)
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 5)
    base = scenario["depth"][1].astype(np.int32)
    depths = []
    for i in range(3):
        d = base.copy(); d[d > 0] += 2 * i; d[(30 * i) % H::9, ::4] = 0
        depths.append(d)
    def make(obj, dev):
        return obj.init_Scene_projective_device(dev, K, W, H) if kind == "proj" else obj.init_Scene_nn_device(dev, K, W, H)
    devs = [api.DeviceVector.from_host(d.reshape(-1)) for d in depths]
    new = (lambda: api.Scene_projective()) if kind == "proj" else (lambda: api.Scene_nn())
    want = [api.refine_batch(model, poses, W, H, proj, K, make(new(), dv), crit) for dv in devs]
    objs = [new() for _ in range(4)]
    ready, free, errors = queue.Queue(), queue.Queue(), []
    for i in range(4): free.put(i)
    N = 14
    def producer():
        try:
            api.thread_context(True)
            for k in range(N):
                i = free.get()
                make(objs[i], devs[k % 3])
                ready.put(i)
            api.thread_context(False)
        except Exception as e:                                       # noqa: BLE001 -- reported by the main thread
            errors.append(e); ready.put(-1)
    th = threading.Thread(target=producer); th.start()
    held = {}
    for k in range(N):
        i = ready.get()
        assert i >= 0, errors
        api.refine_submit(k & 1, model, poses, W, H, proj, K, objs[i], crit)
        held[k] = i
        if k:
            res, sizes = api.refine_wait((k - 1) & 1)
            assert res.tobytes() == want[(k - 1) % 3][0].tobytes() and np.array_equal(sizes, want[(k - 1) % 3][1])
            free.put(held.pop(k - 1))
    res, _ = api.refine_wait((N - 1) & 1)
    assert res.tobytes() == want[(N - 1) % 3][0].tobytes()
    th.join()
    assert not errors
