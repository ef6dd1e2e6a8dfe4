#!/usr/bin/env python3
"""Per-wave timeline of the fused ray-cast: build/libdfusion_hip_rtrace.so (tools/build_variant.py rtrace --only dfusion_raycast.hip
-DDF_TRACE_RAYCAST=1) stamps every wave at its start, when its march is done and at its end (100 MHz clock).
tools/trace_raycast.py [CONFIG]"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from dynamicfusion_amd import capi, build as B
B.LIB_PATH = os.path.join(REPO, "build", "libdfusion_hip_rtrace.so"); B._stale = lambda: False
from dynamicfusion_amd import Intr, TsdfVolume, compute_dists, synth, upload_u16
name = sys.argv[1] if len(sys.argv) > 1 else "512"
cfg = synth.CONFIGS[name]; intr = Intr(*cfg.intr)
vol = TsdfVolume(cfg.dims); vol.setSize([cfg.size] * 3); vol.setTruncDist(cfg.trunc_dist); vol.setMaxWeight(cfg.max_weight); vol.setPose(cfg.volume_pose)
vol.setRaycastStepFactor(cfg.raycast_step_factor); vol.setGradientDeltaFactor(cfg.gradient_delta_factor)
for f in range(3):
    vol.integrate(compute_dists(upload_u16(synth.depth_frame(cfg, f)), intr), synth.camera_pose(cfg, f), intr)
cam = synth.camera_pose(cfg, 2)
pts = torch.empty((cfg.rows, cfg.cols, 4), dtype=torch.float32, device="cuda"); nrm = torch.empty_like(pts)
keys = torch.empty((cfg.rows, cfg.cols), dtype=torch.int32, device="cuda")
for _ in range(3): vol.raycast(cam, intr, pts, nrm, keys)
path = os.path.join(REPO, "gpurun_out", "raycast_trace.bin"); os.makedirs(os.path.dirname(path), exist_ok=True)
os.environ["DF_TRACE_RAYCAST_FILE"] = path
vol.raycast(cam, intr, pts, nrm, keys)
del os.environ["DF_TRACE_RAYCAST_FILE"]
t = np.fromfile(path, dtype=np.uint64).reshape(-1, 4).astype(np.int64); os.remove(path)
t = t[t[:, 2] > 0]
t0 = t[:, 0].min(); t = (t[:, :3] - t0) * 0.01
k = keys.cpu().numpy().view(np.uint32)
steps = np.where(k == 0xffffffff, -1, (k >> 1).astype(np.int64))
print("%s: %d waves, launch %.1f us; last wave STARTS at %.1f us" % (name, len(t), t[:, 2].max(), t[:, 0].max()))
march, shade = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1]
for nm, v in (("march", march), ("locate + shade + store", shade), ("wave", t[:, 2] - t[:, 0])):
    print("  %-24s mean %.2f  median %.2f  p90 %.2f  max %.2f us" % (nm, v.mean(), np.median(v), np.percentile(v, 90), v.max()))
print("  event step of the rays that have one: mean %.0f, max %d; rays without: %d" % (steps[steps >= 0].mean(), steps.max(), int((steps < 0).sum())))
T = np.linspace(0, t[:, 2].max(), 11)
for a, b in zip(T[:-1], T[1:]):
    print("  %5.1f-%5.1f us: %5d waves resident, %5d marching" % (a, b, int(((t[:, 0] < b) & (t[:, 2] > a)).sum()), int(((t[:, 0] < b) & (t[:, 1] > a)).sum())))
