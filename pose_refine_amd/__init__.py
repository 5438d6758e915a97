"""pose_refine_amd -- MI355X-native (gfx950) batched depth render -> point-to-plane ICP.

Only what the hot path needs lives here:
  csrc/      hand-written HIP kernels + the C ABI (include/pose_refine.h)
  _lib.py    ctypes binding of the C ABI
  api.py     host-side mirror of the reference API (cuda_renderer::*, cuda_icp::*, Scene_*)
  synth.py   synthetic benchmark inputs (test.cpp scenario, seeded hypotheses)
  build.py   in-tree hipcc build of lib/libpose_refine_hip.so
"""
__all__ = ["api", "synth", "build"]
