#!/usr/bin/env python3
"""The clock a kernel runs at: N launches of the warped integrate at one pose through a chosen build of the library, for a
`rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE` pass (cycles per launch / duration per launch = the clock while it runs).
    tools/clock_probe.py CONFIG [TAG]        TAG: build/libdfusion_hip_TAG.so instead of the product library"""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from dynamicfusion_amd import capi
name = sys.argv[1] if len(sys.argv) > 1 else "512"
if len(sys.argv) > 2:
    capi._lib = capi.load(os.path.join(REPO, "build", "libdfusion_hip_%s.so" % sys.argv[2]), strict=False)
from dynamicfusion_amd import Intr, TsdfVolume, WarpField, compute_dists, synth, upload_u16
cfg = synth.CONFIGS[name]; intr = Intr(*cfg.intr)
pos, sigma = synth.make_nodes(cfg)
wf = WarpField(k=cfg.k); wf.init(pos, sigma=sigma, transforms=synth.node_transforms(cfg, 0))
v = TsdfVolume(cfg.dims); v.setSize([cfg.size] * 3); v.setTruncDist(cfg.trunc_dist); v.setMaxWeight(cfg.max_weight); v.setPose(cfg.volume_pose)
for f in range(5):
    wf.set_transforms(torch.from_numpy(synth.node_transforms(cfg, f)).cuda())
    v.integrate_warped(compute_dists(upload_u16(synth.depth_frame(cfg, f)), intr), synth.camera_pose(cfg, f), intr, wf)
d = compute_dists(upload_u16(synth.depth_frame(cfg, 4)), intr); cam = synth.camera_pose(cfg, 4)
for _ in range(40):
    v.integrate_warped(d, cam, intr, wf, sync=False)
torch.cuda.synchronize()
print("clock_probe done", name, sys.argv[2:] or "product")
