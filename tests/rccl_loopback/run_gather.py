"""Eight ranks as host threads of one process on ONE GPU, each with a private context and a communicator of the loop-back stand-in
(PR_RCCL_LIBRARY, tests/rccl_loopback/loopback_rccl.cpp): the N > 1 branch of pr_gather_results -- grouped ncclSend / ncclRecv with
per-rank offsets and counts, csrc/pr_comm.cpp -- executed with uneven shards, empty shards and a root other than rank 0, then with real
refinement results (every rank refines its shard; the gathered block must equal the unsharded batch bit for bit).
Run by tests/test_gather_loopback_gpu.py in a process of its own (the library binds its RCCL entry points once per process).  Prints one JSON line."""
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pose_refine_amd import api, synth                             # noqa: E402

WORLD = int(sys.argv[1]) if len(sys.argv) > 1 else 8
assert os.environ.get("PR_RCCL_LIBRARY"), "set PR_RCCL_LIBRARY to the loop-back library"


def main():
    api.init(0)
    ident = api.comm_id()
    W, H, K = synth.WIDTH, synth.HEIGHT, synth.K_TEST
    model = api.Model(os.path.join(ROOT, "tests", "golden", "obj_06.ply"))
    proj = api.compute_proj(K, W, H)
    scene_depth = api.render_host(model, synth.scene_pose()[None], W, H, proj)[0]
    scene = api.Scene_projective().init_Scene_projective_cuda(scene_depth, K)
    crit = api.ICPConvergenceCriteria(0.0, 0.0, 3)
    api.set_option("solve", api.SOLVE_DEVICE)
    n_refine = 67                                                   # 67 hypotheses over 8 ranks: shards of 9 and 8
    poses = synth.hypotheses(n_refine)
    whole, _ = api.refine_batch(model, poses, W, H, proj, K, scene, crit)
    cases = [(4096, 0), (4099, 3 % WORLD), (5, 6 % WORLD), (WORLD, WORLD - 1), (1, 0)]          # (hypotheses, root): even, uneven, empty shards, one each, one in all
    bar = threading.Barrier(WORLD)
    errors, report = [], {"world": WORLD, "cases": [], "refine": None}

    def rank_main(rank):
        # Device buffers are kept until every gather is over: pr_free waits for every other context of the device (it locks them one by one), and
        # the stand-in's ncclGroupEnd blocks on the host until the peers have issued theirs (RCCL itself only enqueues) -- a rank that frees a buffer
        # while another is inside the gather, waiting for a root that in turn waits for the freeing rank, would never return.
        keep = []
        try:
            api.init(0)
            api.thread_context(True)
            api.comm_init_rank(ident, rank, WORLD)                  # blocks until all WORLD ranks have joined, like ncclCommInitRank
            assert api.comm_rank() == (rank, WORLD)
            for n_total, root in cases:
                first, count = api.shard_range(n_total, rank, WORLD)
                rec = np.zeros((count, 18), np.float32)
                rec[:] = (np.arange(first, first + count, dtype=np.float32)[:, None] * 32.0 + np.arange(18, dtype=np.float32)[None, :])   # record g, word k = 32 g + k (exact in f32)
                send = api.DeviceVector.from_host(rec.reshape(-1)) if count else api.DeviceVector(18, np.float32)
                recv = api.DeviceVector(max(1, n_total) * 18, np.float32) if rank == root else None
                keep += [send, recv]
                bar.wait()
                api.gather_results(send.data() if count else None, count, n_total, root, recv.data() if recv else None)
                api.sync()
                if rank == root:
                    got = recv.to_host()[: n_total * 18].reshape(n_total, 18)
                    want = np.arange(n_total, dtype=np.float32)[:, None] * 32.0 + np.arange(18, dtype=np.float32)[None, :]
                    ok = bool(np.array_equal(got, want))
                    report["cases"].append({"n_total": n_total, "root": root, "ok": ok})
                    if not ok:
                        errors.append(f"gather of {n_total} to root {root}: wrong order or content")
                bar.wait()
            # real results: every rank refines its shard into device memory, one gather to rank 2
            first, count = api.shard_range(n_refine, rank, WORLD)
            res = api.DeviceVector(max(1, count) * 18, np.float32)
            api.refine_submit(0, model, poses[first:first + count], W, H, proj, K, scene, crit, results_dev=res.data())
            api.refine_wait(0)
            recv2 = api.DeviceVector(n_refine * 18, np.float32) if rank == 2 % WORLD else None
            keep += [res, recv2]
            bar.wait()
            api.gather_results(res.data(), count, n_refine, 2 % WORLD, recv2.data() if recv2 else None)
            api.sync()
            if rank == 2 % WORLD:
                got = recv2.to_host().view(np.uint8)
                same = got.tobytes() == whole.tobytes()
                report["refine"] = {"hypotheses": n_refine, "root": 2 % WORLD, "bit_identical_to_unsharded": bool(same)}
                if not same:
                    errors.append("gathered refinement results differ from the unsharded batch")
            bar.wait()
            api.comm_destroy()
            del keep[:]                                              # (every gather is over)
            api.thread_context(False)
        except BaseException as e:                                  # noqa: BLE001
            import traceback
            errors.append("".join(traceback.format_exception(type(e), e, e.__traceback__)))
            bar.abort()

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(WORLD)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    report["errors"] = errors
    print(json.dumps(report), flush=True)
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main())
