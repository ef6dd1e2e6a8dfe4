#!/usr/bin/env python3
"""Time the ray-cast's stages separately (march incl. locate / shade) next to the fused kernel: tools/raycast_stages.py [CONFIG]"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from dynamicfusion_amd import Intr, TsdfVolume, compute_dists, synth, upload_u16
name = sys.argv[1] if len(sys.argv) > 1 else "512"
cfg = synth.CONFIGS[name]; intr = Intr(*cfg.intr)
vol = TsdfVolume(cfg.dims); vol.setSize([cfg.size] * 3); vol.setTruncDist(cfg.trunc_dist); vol.setMaxWeight(cfg.max_weight); vol.setPose(cfg.volume_pose)
vol.setRaycastStepFactor(cfg.raycast_step_factor); vol.setGradientDeltaFactor(cfg.gradient_delta_factor)
for f in range(3):
    vol.integrate(compute_dists(upload_u16(synth.depth_frame(cfg, f)), intr), synth.camera_pose(cfg, f), intr)
cam = synth.camera_pose(cfg, 2)
pts = torch.empty((cfg.rows, cfg.cols, 4), dtype=torch.float32, device="cuda"); nrm = torch.empty_like(pts)
keys = torch.empty((cfg.rows, cfg.cols), dtype=torch.int64, device="cuda")
def timeit(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("fused  %.4f ms" % timeit(lambda: vol.raycast(cam, intr, pts, nrm)))
print("march  %.4f ms" % timeit(lambda: vol.raycast_march(cam, intr, keys)))
vol.raycast_march(cam, intr, keys)
print("shade  %.4f ms" % timeit(lambda: vol.raycast_shade(cam, intr, keys, pts, nrm)))
hit = int((((keys >> 39) & 1) * (keys != 0x7FFFFFFFFFFFFFFF)).sum()); print("hits", hit, "of", cfg.rows * cfg.cols)
