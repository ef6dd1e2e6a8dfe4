#!/usr/bin/env python3
"""A/B timing of the warped-sweep kernel variants on one box (HIP events, interleaved rounds)."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from dynamicfusion_amd import Intr, TsdfVolume, WarpField, compute_dists, synth, upload_u16
name = sys.argv[1] if len(sys.argv) > 1 else "512"
cfg = synth.CONFIGS[name]; intr = Intr(*cfg.intr)
depth = upload_u16(synth.depth_frame(cfg, 0)); dists = compute_dists(depth, intr); cam = synth.camera_pose(cfg, 1)
vol = TsdfVolume(cfg.dims); vol.setSize([cfg.size] * 3); vol.setTruncDist(cfg.trunc_dist); vol.setMaxWeight(cfg.max_weight); vol.setPose(cfg.volume_pose)
pos, sigma = synth.make_nodes(cfg); dq = synth.node_transforms(cfg, 1)
wf = WarpField(k=cfg.k); wf.init(pos, sigma=sigma, transforms=dq); wf.ensure_index(vol, cfg.k)
variants = {"pipelined": {}, "no-depth-pyramid": dict(depth_pyramid=False), "no-zero-skip": dict(zero_skip=False), "batched": dict(pipelined=False), "global-gather": dict(use_lds=False)}
res = {k: [] for k in variants}
for rnd in range(6):
    for k, kw in variants.items():
        for _ in range(3): vol.integrate_warped(dists, cam, intr, wf, sync=False, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): vol.integrate_warped(dists, cam, intr, wf, sync=False, **kw)
        e1.record(); torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / 20)
for k, v in res.items():
    print("%-14s min %.3f  median %.3f  max %.3f ms" % (k, min(v), float(np.median(v)), max(v)))
