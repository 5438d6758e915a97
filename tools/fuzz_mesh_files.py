#!/usr/bin/env python
"""Mutation fuzz of the model importers (CPU only): valid PLY (ASCII / binary), OBJ, glTF (+ .bin) and GLB files with random bytes flipped,
ranges overwritten, truncated or duplicated, loaded through pr_mesh_load in a child process -- an importer may refuse a file, it may not
crash, hang or return triangles that index outside their vertices.   python tools/fuzz_mesh_files.py [mutations per format] [seed]

With PR_MESH_HARNESS=<binary> the files go through a sanitizer build of the importers instead (AddressSanitizer + UBSan):
    g++ -std=c++17 -O1 -g -fsanitize=address,undefined -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -Ipose_refine_amd/csrc \
        tools/mesh_sanitize.cpp pose_refine_amd/csrc/pr_host.cpp -o mesh_asan
(harness.cpp = tools/mesh_sanitize.cpp: pr_mesh_count / pr_mesh_load on every argv file, prints "ok N refused M", prh::set_error a stub)."""
import os, sys, json, struct, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def child(paths):
    from pose_refine_amd import api
    ok = refused = 0
    for p in paths:
        try:
            m = api.Model(p)
            assert m.tris.shape[0] == m.faces.shape[0]
            if len(m.faces):
                assert m.faces.min() >= 0                      # (per-mesh indices for glTF: only the sign is checked here)
            ok += 1
        except api.PoseRefineError:
            refused += 1
    print(json.dumps({"ok": ok, "refused": refused}))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(sys.argv[2:])
    n_mut = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    import test_mesh_import as T
    from pose_refine_amd import api
    ref = api.Model(os.path.join(ROOT, "tests", "golden", "obj_06.ply"))
    verts, faces = ref.vertices[:400], ref.faces[(ref.faces < 400).all(1)][:500]
    tot = {"ok": 0, "refused": 0, "crashed": 0}
    with tempfile.TemporaryDirectory() as d:
        from pathlib import Path
        dp = Path(d)
        bases = {}
        for fmt in ("ascii", "binary_little_endian", "binary_big_endian"):
            p = str(dp / f"base_{fmt}.ply"); T.write_ply(p, verts, faces, fmt, extra=(fmt != "ascii")); bases[p] = None
        with open(dp / "base.obj", "w") as f:
            for v in verts: f.write("v %r %r %r\n" % tuple(float(c) for c in v))
            for fc in faces: f.write("f %d/1/1 %d//2 %d\n" % tuple(int(i) + 1 for i in fc))
        bases[str(dp / "base.obj")] = None
        class R: pass
        r = R(); r.vertices = ref.vertices; r.faces = ref.faces; r.tris = ref.tris
        for container in ("bin", "glb", "base64"):
            sub = dp / container; sub.mkdir()
            p = T._gltf_scene(r, sub, container)[0]
            bases[p] = str(sub / "scene.bin") if container == "bin" else None
        for base, side in bases.items():
            data = open(base, "rb").read()
            paths = []
            for k in range(n_mut):
                b = bytearray(data)
                kind = rng.integers(6)
                # mutate the head of the file more often than the bulk: that is where the structure is
                lim = len(b) if rng.random() < 0.4 else min(len(b), 2500)
                if kind == 0:
                    for _ in range(int(rng.integers(1, 8))): b[int(rng.integers(lim))] = int(rng.integers(256))
                elif kind == 1:
                    b = b[: int(rng.integers(0, len(b)))]
                elif kind == 2:
                    at = int(rng.integers(lim)); ln = int(rng.integers(1, 64)); b[at:at + ln] = bytes(rng.integers(0, 256, ln, dtype=np.uint8))
                elif kind == 3:
                    at = int(rng.integers(lim)); b[at:at] = b[at:at + int(rng.integers(1, 200))]
                elif kind == 4:                                  # digits -> other digits / huge numbers / signs
                    idx = [i for i in range(min(lim, len(b))) if 48 <= b[i] <= 57]
                    for i in (rng.choice(idx, size=min(len(idx), int(rng.integers(1, 6))), replace=False) if idx else []):
                        rep = rng.choice([b"9", b"0", b"-", b"99999999999", b"1e308", b".", b"e"]); b[int(i):int(i) + 1] = rep
                else:
                    at = int(rng.integers(lim)); del b[at:at + int(rng.integers(1, 100))]
                ext = os.path.splitext(base)[1]
                q = os.path.join(os.path.dirname(base), f"m{k}{ext}")
                open(q, "wb").write(bytes(b)); paths.append(q)
            harness = os.environ.get("PR_MESH_HARNESS")        # a sanitizer build of the importers (see DESIGN): `<harness> files...` prints "ok N refused M"
            for i in range(0, len(paths), 50):
                if harness:
                    pr = subprocess.run([harness] + paths[i:i + 50], capture_output=True, text=True, timeout=300)
                    if pr.returncode == 0 and "ERROR" not in pr.stderr and "runtime error" not in pr.stderr:
                        w = pr.stdout.split(); tot["ok"] += int(w[1]); tot["refused"] += int(w[3]); continue
                    for q in paths[i:i + 50]:
                        p1 = subprocess.run([harness, q], capture_output=True, text=True, timeout=120)
                        if p1.returncode == 0 and "ERROR" not in p1.stderr and "runtime error" not in p1.stderr:
                            w = p1.stdout.split(); tot["ok"] += int(w[1]); tot["refused"] += int(w[3]); continue
                        tot["crashed"] += 1
                        keep = os.path.join("/tmp", "san_" + os.path.basename(base) + "_" + os.path.basename(q))
                        open(keep, "wb").write(open(q, "rb").read())
                        print("SANITIZER", base, "->", keep, "rc", p1.returncode, p1.stderr.strip()[:600], flush=True)
                    continue
                pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + paths[i:i + 50], capture_output=True, text=True, timeout=300)
                try:
                    o = json.loads(pr.stdout.strip().splitlines()[-1]); tot["ok"] += o["ok"]; tot["refused"] += o["refused"]
                except Exception:
                    # find the culprit one by one
                    for q in paths[i:i + 50]:
                        p1 = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", q], capture_output=True, text=True, timeout=120)
                        try:
                            o = json.loads(p1.stdout.strip().splitlines()[-1]); tot["ok"] += o["ok"]; tot["refused"] += o["refused"]
                        except Exception:
                            tot["crashed"] += 1
                            keep = os.path.join("/tmp", "crash_" + os.path.basename(base) + "_" + os.path.basename(q))
                            open(keep, "wb").write(open(q, "rb").read())
                            print("CRASH", base, "->", keep, "rc", p1.returncode, p1.stderr.strip()[-200:], flush=True)
            for q in paths: os.remove(q)
            print(os.path.basename(os.path.dirname(base)) + "/" + os.path.basename(base), dict(tot), flush=True)
    print("total", tot)
    return 1 if tot["crashed"] else 0


if __name__ == "__main__":
    sys.exit(main())
