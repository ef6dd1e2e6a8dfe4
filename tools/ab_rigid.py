#!/usr/bin/env python3
"""Timing of the rigid integrate (HIP events, 20 launches per round):  tools/ab_rigid.py [CONFIG]"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
if len(sys.argv) > 2:                      # a variant library (tools/build_variant.py TAG ...)
    from dynamicfusion_amd import build as B
    B.LIB_PATH = os.path.join(REPO, "build", "libdfusion_hip_%s.so" % sys.argv[2]); B._stale = lambda: False
from dynamicfusion_amd import Intr, TsdfVolume, compute_dists, synth, upload_u16
name = sys.argv[1] if len(sys.argv) > 1 else "512"
cfg = synth.CONFIGS[name]; intr = Intr(*cfg.intr)
depth = upload_u16(synth.depth_frame(cfg, 0)); dists = compute_dists(depth, intr); cam = synth.camera_pose(cfg, 1)
vol = TsdfVolume(cfg.dims); vol.setSize([cfg.size] * 3); vol.setTruncDist(cfg.trunc_dist); vol.setMaxWeight(cfg.max_weight); vol.setPose(cfg.volume_pose)
res = []
for rnd in range(6):
    for _ in range(3): vol.integrate(dists, cam, intr, sync=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): vol.integrate(dists, cam, intr, sync=False)
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 20)
print("rigid %s: min %.4f  median %.4f  max %.4f ms" % (name, min(res), float(np.median(res)), max(res)))
