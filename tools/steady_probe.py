"""Steady-state probe of the fused projective path through the C ABI, without bench.py around it: 5 x 60 pipelined 256-hypothesis steps in one
process (a process repeats to +-0.5 %; boxes of the pool differ by up to 7 %).   tools/steady_probe.py none|sleep|shutdown|shift [option=value ...] [--torch]
shift: like shutdown, with a dummy device allocation of a different size kept across every re-initialisation (does WHERE the buffers land matter?)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--torch" in sys.argv:
    import torch; torch.cuda.set_device(0)
import numpy as np
from pose_refine_amd import api, synth
what = sys.argv[1] if len(sys.argv) > 1 else "shutdown"
api.init(0); api.set_option("solve", 1)
for kv in [a for a in sys.argv[2:] if "=" in a]:
    k_, v_ = kv.split("="); api.set_option(k_, int(v_))
W, H, K = synth.WIDTH, synth.HEIGHT, synth.K_TEST
HR = H + int(os.environ.get("PR_PROBE_EXTRA_ROWS", "0"))     # experiment: a taller render frame (same boxes, another image stride)
model = api.Model(os.path.join(ROOT, "tests/golden/obj_06.ply"))
proj = api.compute_proj(K, W, H)
sd = api.render_host(model, synth.scene_pose()[None], W, H, proj)[0]
if HR != H: proj = api.compute_proj(K, W, HR)
scene = api.Scene_projective().init_Scene_projective_cuda(sd, K)
NP = int(os.environ.get("PR_PROBE_POSES", "256"))
poses = synth.hypotheses(NP)
crit = api.ICPConvergenceCriteria(0.0, 0.0, 20)
res = api.DeviceVector(NP * 18, np.float32)
DELAY = float(os.environ.get("PR_PROBE_SUBMIT_DELAY_US", "0")) * 1e-6     # experiment: busy-wait before every submit (a slow host)
def run(n):
    for k in range(n):
        if DELAY:
            t_ = time.perf_counter()
            while time.perf_counter() - t_ < DELAY: pass
        api.refine_submit(k & 1, model, poses, W, HR, proj, K, scene, crit, results_dev=res.data())
        if k: api.refine_wait((k - 1) & 1)
    api.refine_wait((n - 1) & 1)
out = []; hold = []
for r in range(5):
    run(6)
    t0 = time.perf_counter(); run(60); dt = (time.perf_counter() - t0) / 60
    out.append(NP / dt / 1e3)
    if what == "shutdown": api.shutdown(); api.init(0); api.set_option("solve", 1)
    elif what == "shift":
        api.shutdown(); hold.append(api.DeviceVector((r + 1) * 9_437_184 + 4096 * r, np.float32)); api.init(0); api.set_option("solve", 1)
        res = api.DeviceVector(NP * 18, np.float32)
    elif what == "sleep": time.sleep(0.5)
print(what, " ".join(f"{v:.0f}" for v in out))
