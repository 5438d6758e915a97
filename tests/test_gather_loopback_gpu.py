"""The N > 1 branch of pr_gather_results (csrc/pr_comm.cpp: grouped ncclSend / ncclRecv, per-rank offsets and counts) EXECUTED on a box
with one GPU: eight ranks as host threads with private contexts over a loop-back stand-in for librccl (tests/rccl_loopback: test
infrastructure, selected with PR_RCCL_LIBRARY; real RCCL refuses two ranks on one device).  SURVEY 8e's contract: contiguous shards,
one gather of 72-byte records, global hypothesis order on the root.  The reference has nothing here (test.cpp:14)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOOP_DIR = os.path.join(ROOT, "tests", "rccl_loopback")
LOOP_SO = os.path.join(LOOP_DIR, "librccl_loopback.so")


def build_loopback():
    src = os.path.join(LOOP_DIR, "loopback_rccl.cpp")
    if not os.path.exists(LOOP_SO) or os.path.getmtime(LOOP_SO) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O2", src, "-o", LOOP_SO], check=True)
    return LOOP_SO


def test_loopback_library_builds_and_exports_what_the_library_binds():
    import ctypes
    so = build_loopback()
    lib = ctypes.CDLL(so)
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommInitAll", "ncclCommDestroy", "ncclGroupStart", "ncclGroupEnd", "ncclSend", "ncclRecv", "ncclGetErrorString"):
        assert hasattr(lib, name), name


@pytest.mark.gpu
@pytest.mark.parametrize("world", [8, 3])
def test_gather_with_more_than_one_rank_runs_and_keeps_global_order(world):
    so = build_loopback()
    env = dict(os.environ, PR_RCCL_LIBRARY=so, GPU_MAX_HW_QUEUES="16")
    r = subprocess.run([sys.executable, os.path.join(LOOP_DIR, "run_gather.py"), str(world)], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert got["errors"] == [] and got["world"] == world
    assert len(got["cases"]) == 5 and all(c["ok"] for c in got["cases"])
    assert got["refine"]["bit_identical_to_unsharded"]


@pytest.mark.gpu
def test_a_forced_rccl_library_that_cannot_be_opened_is_an_error_not_a_fallback():
    code = ("import sys; sys.path.insert(0, %r)\nfrom pose_refine_amd import api\napi.init(0)\n"
            "try:\n    api.comm_id()\n    print('NO ERROR')\nexcept Exception as e:\n    print('ERR', e)\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PR_RCCL_LIBRARY="/nonexistent/librccl.so"), capture_output=True, text=True, timeout=300)
    assert "ERR" in r.stdout and "PR_RCCL_LIBRARY" in r.stdout, r.stdout + r.stderr
