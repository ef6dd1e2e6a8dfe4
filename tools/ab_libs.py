#!/usr/bin/env python3
"""Same-box, interleaved A/B timing of the warped sweep between the product library and build/libdfusion_hip_TAG.so variants
(tools/build_variant.py):  tools/ab_libs.py CONFIG TAG [TAG ...].  Also reports whether each variant's volume equals the
product's bit for bit after two frames."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from dynamicfusion_amd import capi, build as B
from dynamicfusion_amd import Intr, TsdfVolume, WarpField, compute_dists, synth, upload_u16
name, tags = sys.argv[1], sys.argv[2:]
libs = {"product": capi.lib()}
for t in tags:
    libs[t] = capi.load(os.path.join(REPO, "build", "libdfusion_hip_%s.so" % t), strict=False)
def use(t): capi._lib = libs[t]
use("product")
cfg = synth.CONFIGS[name]; intr = Intr(*cfg.intr)
depth = upload_u16(synth.depth_frame(cfg, 0)); dists = compute_dists(depth, intr)
def mkvol():
    v = TsdfVolume(cfg.dims); v.setSize([cfg.size] * 3); v.setTruncDist(cfg.trunc_dist); v.setMaxWeight(cfg.max_weight); v.setPose(cfg.volume_pose)
    return v
vol = mkvol()
pos, sigma = synth.make_nodes(cfg)
wf = WarpField(k=cfg.k); wf.init(pos, sigma=sigma, transforms=synth.node_transforms(cfg, 1)); wf.ensure_index(vol, cfg.k)
# bit identity after two frames
ref = None
for t in libs:
    use(t); v = mkvol()
    for f in range(2):
        wf.set_transforms(torch.from_numpy(synth.node_transforms(cfg, f)).cuda())
        v.integrate_warped(dists, synth.camera_pose(cfg, f), intr, wf)
    d = v.data().clone()
    if ref is None: ref = d
    else: print("%-12s volume %s product (%d words differ)" % (t, "==" if torch.equal(d, ref) else "!=", int((d != ref).sum())))
    del v
wf.set_transforms(torch.from_numpy(synth.node_transforms(cfg, 1)).cuda()); cam = synth.camera_pose(cfg, 1)
res = {t: [] for t in libs}
for rnd in range(6):
    for t in libs:
        use(t)
        for _ in range(3): vol.integrate_warped(dists, cam, intr, wf, sync=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): vol.integrate_warped(dists, cam, intr, wf, sync=False)
        e1.record(); torch.cuda.synchronize()
        res[t].append(e0.elapsed_time(e1) / 20)
for t, v in res.items():
    print("%-12s min %.3f  median %.3f  max %.3f ms" % (t, min(v), float(np.median(v)), max(v)))
