#!/usr/bin/env python3
"""Fixed rigid-integrate workload for rocprofv3 --pmc passes:  tools/pmc_rigid.py CONFIG [TAG] [nosat]
(TAG = a build/libdfusion_hip_TAG.so variant, '-' = the product library); 6 launches on a fused volume."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from dynamicfusion_amd import capi
from dynamicfusion_amd import Intr, TsdfVolume, compute_dists, synth, upload_u16
name = sys.argv[1] if len(sys.argv) > 1 else "512"
tag = sys.argv[2] if len(sys.argv) > 2 else "-"
if tag != "-":
    capi._lib = capi.load(os.path.join(REPO, "build", "libdfusion_hip_%s.so" % tag), strict=False)
FL = capi.DF_RIGID_NO_SAT if (len(sys.argv) > 3 and sys.argv[3] == "nosat") else 0
cfg = synth.CONFIGS[name]; intr = Intr(*cfg.intr); F = 4
dists = [compute_dists(upload_u16(synth.depth_frame(cfg, f)), intr) for f in range(F)]
cams = [synth.camera_pose(cfg, f) for f in range(F)]
vol = TsdfVolume(cfg.dims); vol.setSize([cfg.size] * 3); vol.setTruncDist(cfg.trunc_dist); vol.setMaxWeight(cfg.max_weight); vol.setPose(cfg.volume_pose)
vol.clear()
for i in range(10): vol.integrate(dists[i % F], cams[i % F], intr, sync=False, flags=FL)
torch.cuda.synchronize()
print("done", cfg.name, tag)
