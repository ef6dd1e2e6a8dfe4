#!/usr/bin/env python3
"""Same-box A/B of the ray-cast between the product library and build/libdfusion_hip_TAG.so variants: tools/ab_raycast.py CONFIG TAG..."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from dynamicfusion_amd import capi, build as B
from dynamicfusion_amd import Intr, TsdfVolume, compute_dists, synth, upload_u16
name, tags = sys.argv[1], sys.argv[2:]
libs = {"product": capi.lib()}
for t in tags:
    capi._lib = None
    B.LIB_PATH = os.path.join(REPO, "build", "libdfusion_hip_%s.so" % t); B._stale = lambda: False
    libs[t] = capi.lib()
def use(t): capi._lib = libs[t]
use("product")
cfg = synth.CONFIGS[name]; intr = Intr(*cfg.intr)
vol = TsdfVolume(cfg.dims); vol.setSize([cfg.size] * 3); vol.setTruncDist(cfg.trunc_dist); vol.setMaxWeight(cfg.max_weight); vol.setPose(cfg.volume_pose)
vol.setRaycastStepFactor(cfg.raycast_step_factor); vol.setGradientDeltaFactor(cfg.gradient_delta_factor)
for f in range(3):
    vol.integrate(compute_dists(upload_u16(synth.depth_frame(cfg, f)), intr), synth.camera_pose(cfg, f), intr)
cam = synth.camera_pose(cfg, 2)
pts = torch.empty((cfg.rows, cfg.cols, 4), dtype=torch.float32, device="cuda"); nrm = torch.empty_like(pts)
ref = None
for t in libs:
    use(t); vol.raycast(cam, intr, pts, nrm); torch.cuda.synchronize()
    d = (pts.clone().view(torch.int32), nrm.clone().view(torch.int32))
    if ref is None: ref = d
    else: print("%-10s output %s product" % (t, "==" if torch.equal(d[0], ref[0]) and torch.equal(d[1], ref[1]) else "!="))
res = {t: [] for t in libs}
for rnd in range(6):
    for t in libs:
        use(t)
        for _ in range(3): vol.raycast(cam, intr, pts, nrm)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): vol.raycast(cam, intr, pts, nrm)
        e1.record(); torch.cuda.synchronize()
        res[t].append(e0.elapsed_time(e1) / 20)
for t, v in res.items():
    print("%-10s min %.4f  median %.4f ms" % (t, min(v), float(np.median(v))))
