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
OUT = os.path.join(HERE, "lib", "libpose_refine_hip.so")
SOURCES = ["pr_kernels.hip", "pr_api.cpp", "pr_host.cpp"]
DEPS = SOURCES + ["pr_internal.h", "pr_solver.inl", os.path.join(ROOT, "include", "pose_refine.h")]
# -ffp-contract=off: no FMA contraction anywhere (bit-parity with the CPU restatement);
# division and sqrt stay IEEE-correct (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
# -fno-slp-vectorize: the SLP vectoriser packs the 29-term accumulation into v_pk_mul_f32 / v_pk_add_f32 plus ~230 v_mov to
# arrange the pairs; on gfx950 packed f32 is no faster per element, so the unpacked code is ~8 % quicker (measured A/B).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared",
         "-Wno-unused-value", "-Wl,-rpath,/opt/rocm/lib"]


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
    if not force and not is_stale():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    extra = os.environ.get("PR_EXTRA_FLAGS", "").split()
    cmd = [hipcc()] + FLAGS + extra + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + \
          [os.path.join(CSRC, s) for s in SOURCES] + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
