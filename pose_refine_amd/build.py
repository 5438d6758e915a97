"""In-tree build of the HIP shared library (hipcc, gfx950 only).

    python -m pose_refine_amd.build [--force]

The .so is git-ignored but travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.environ.get("PR_BUILD_OUT") or os.path.join(HERE, "lib", "libpose_refine_hip.so")      # PR_BUILD_OUT: a variant library for an A/B run
# one translation unit per stage of the path (kernels + their launchers), the C ABI and the host-side code; headers = shared device code
SOURCES = ["raster.hip", "d2c.hip", "icp_pass.hip", "icp_debug.hip", "nn_search.hip", "nn_build.hip", "kd_build.hip", "scene_prep.hip",
           "pr_context.cpp", "pr_scene.cpp", "pr_icp.cpp", "pr_refine.cpp", "pr_comm.cpp", "pr_host.cpp"]
HEADERS = ["pr_internal.h", "pr_runtime.h", "pr_solver.inl", "pr_tuning.h", "pr_device.h", "pr_launch.h", "proj_query.h", "nn_query.h", "icp_accumulate.h",
           "icp_solve_device.h"]
DEPS = SOURCES + HEADERS + [os.path.join(ROOT, "include", "pose_refine.h")]
OBJ_DIR = os.path.join(HERE, "lib", "obj") if not os.environ.get("PR_BUILD_OUT") else os.environ["PR_BUILD_OUT"] + ".obj"
# -ffp-contract=off: no FMA contraction anywhere (bit-parity with the CPU restatement);
# division and sqrt stay IEEE-correct (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
# -fno-slp-vectorize: the SLP vectoriser packs the 29-term accumulation into v_pk_mul_f32 / v_pk_add_f32 plus ~230 v_mov to
# arrange the pairs; on gfx950 packed f32 is no faster per element, so the unpacked code is ~8 % quicker (measured A/B).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-Wno-unused-value"]
LINK_FLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-rpath,/opt/rocm/lib"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found")
    return exe


def is_stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d if os.path.isabs(d) else os.path.join(CSRC, d)) > t for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the translation units side by side (a unit is recompiled when it, a header or the flags changed), then link."""
    if not force and not is_stale():
        return OUT
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ_DIR, exist_ok=True)
    extra = os.environ.get("PR_EXTRA_FLAGS", "").split()
    base = [hipcc()] + FLAGS + extra + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    stamp = os.path.join(OBJ_DIR, "flags.txt")
    flags_now = " ".join(base)
    flags_same = os.path.exists(stamp) and open(stamp).read() == flags_now
    newest_header = max(os.path.getmtime(d if os.path.isabs(d) else os.path.join(CSRC, d)) for d in DEPS if d not in SOURCES)

    def compile_one(name: str) -> str:
        obj = os.path.join(OBJ_DIR, name.rsplit(".", 1)[0] + ".o")
        src = os.path.join(CSRC, name)
        if force or not flags_same or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_header):
            cmd = base + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    open(stamp, "w").write(flags_now)
    cmd = [hipcc()] + LINK_FLAGS + objs + ["-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
