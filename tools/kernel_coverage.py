#!/usr/bin/env python
"""Which kernels of libpose_refine_hip.so did a profiled run dispatch?  ("every kernel of the library is
reached by a tracked -m gpu test".)

    rocprofv3 --kernel-trace -d gpurun_out/cov -o t -- python -m pytest tests -m gpu -q
    python tools/kernel_coverage.py gpurun_out/cov > profiles/r03/kernel_coverage.md

Left column: every kernel symbol of the library's gfx950 code object (one per template instantiation; read from the .so with
llvm-readelf on the embedded code object, demangled); right: dispatch count over all rocpd databases under the directory (the test
run's own process and the child processes it starts).  Exit code 1 when a kernel was never dispatched.
"""
import glob
import os
import re
import sqlite3
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def library_kernels():
    so = os.path.join(ROOT, "pose_refine_amd", "lib", "libpose_refine_hip.so")
    names = set()
    with tempfile.TemporaryDirectory() as d:
        # the fat binary sits in section .hip_fatbin; clang-offload-bundler lists and unbundles the gfx950 code object
        fat = os.path.join(d, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat], check=True)
        blob = open(fat, "rb").read()
        # bundles: magic "__CLANG_OFFLOAD_BUNDLE__", then ELF images; take every embedded AMDGPU ELF
        at = 0
        k = 0
        while True:
            at = blob.find(b"\x7fELF", at)
            if at < 0:
                break
            p = os.path.join(d, f"co{k}.elf")
            open(p, "wb").write(blob[at:])
            r = subprocess.run([f"{LLVM}/llvm-readelf", "-s", "-W", p], capture_output=True, text=True)
            mangled = [ln.split()[-1] for ln in r.stdout.splitlines() if " FUNC " in ln and "_kernel" in ln and " UND " not in ln]
            if mangled:
                dem = subprocess.run(["c++filt"], input="\n".join(mangled), capture_output=True, text=True).stdout.splitlines()
                names.update(dem)
            at += 4
            k += 1
    return names


def norm(n):
    n = n.replace("void ", "")
    n = re.sub(r"\s*\[clone .*\]$", "", n)
    n = re.sub(r"\(.*\)$", "", n)          # argument list
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"(\d+)u\b", r"\1", n)       # 272u -> 272
    return n.strip()


def main(d):
    seen = {}
    for db in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        c = sqlite3.connect(db)
        try:
            for name, calls in c.execute("select name,total_calls from top_kernels"):
                seen[norm(name)] = seen.get(norm(name), 0) + calls
        except sqlite3.Error:
            tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
            kd = [t for t in tabs if "kernel_dispatch" in t]
            sy = [t for t in tabs if "kernel_symbol" in t]
            if kd and sy:
                for name, calls in c.execute(f"select s.kernel_name, count(*) from {kd[0]} k join {sy[0]} s on k.kernel_id = s.id group by s.kernel_name"):
                    seen[norm(name)] = seen.get(norm(name), 0) + calls
    lib = sorted({norm(n) for n in library_kernels()})
    print(f"# kernel coverage of the GPU test run ({len(lib)} kernel instantiations in libpose_refine_hip.so)\n")
    print("| kernel | dispatches |")
    print("|---|---:|")
    missing = 0
    for n in lib:
        calls = seen.get(n, 0)
        missing += calls == 0
        print(f"| `{n}` | {calls if calls else '**0**'} |")
    other = sorted(k for k in seen if k not in lib)
    print(f"\n{len(lib) - missing} of {len(lib)} dispatched; not from this library (PyTorch, rocThrust test, runtime fills): {len(other)} kernels")
    return 1 if missing else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
