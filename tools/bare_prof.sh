cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/b64.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from pose_refine_amd import api, synth
api.init(0); api.set_option("solve", 1); api.set_option("pose_groups", 1); api.set_option("graph", 0)
ROOT = os.environ["GRAFT_REPO_ROOT"]
model = api.Model(os.path.join(ROOT, "tests/golden/obj_06.ply"))
K = synth.K_TEST; W, H = 640, 480; proj = api.compute_proj(K, W, H)
depth = api.render_host(model, synth.test_cpp_poses(), W, H, proj)
cloud = api.depth2cloud(api.DeviceVector.from_host(depth[0].reshape(-1).astype(np.int32)), W, H, K).to_host().reshape(-1, 3)
sc = api.Scene_nn().init_Scene_nn_cuda(depth[1], K)
P = 64
host = np.concatenate([cloud + np.float32(0.00004 * i) for i in range(P)]).astype(np.float32)
offs = (np.arange(P + 1) * len(cloud)).astype(np.uint32)
for _ in range(2):
    dev = api.DeviceVector.from_host(host.reshape(-1))
    api.ICP_Point2Plane_batch(dev, offs, sc, api.ICPConvergenceCriteria(0.0, 0.0, 20))
PY
rocprofv3 --kernel-trace -d gpurun_out/bp -o t -- python /tmp/b64.py > /dev/null 2>&1
python - gpurun_out/bp/t_results.db <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, (end-start)/1000.0 from kernels where name like '%nn_%' or name like '%icp_pass%' order by start"))
for key in ("nn_search", "nn_bound", "nn_tree", "icp_pass"):
    v = [r[2] for r in rows if key in r[0]]
    print("%-9s us:" % key, " ".join(f"{x:.0f}" for x in v[-21:]), " sum %.2f ms" % (sum(v[-21:]) / 1e3))
PY
rm -rf gpurun_out/bp
