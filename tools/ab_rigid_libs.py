#!/usr/bin/env python3
"""Same-box, interleaved A/B timing of the RIGID integrate between the product library and build/libdfusion_hip_TAG.so variants
(tools/build_variant.py):  tools/ab_rigid_libs.py CONFIG TAG [TAG ...].  Reports whether each variant's volume equals the product's
bit for bit (3 frames from a cleared volume, then 8 more on the fused one), swept / updated where the build has the counter, and the
time per launch in the bench's own pattern (the frames' camera poses cycled on a fused volume)."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from dynamicfusion_amd import capi
from dynamicfusion_amd import Intr, TsdfVolume, compute_dists, synth, upload_u16
name, tags = sys.argv[1], sys.argv[2:]
libs = {"product": capi.lib()}
nosat = {}                                  # TAG/nosat: the same build with the saturated-sample shortcuts switched off (DF_RIGID_NO_SAT, per call)
for t in tags:
    base = t.split("/")[0]
    if base != "product" and base not in libs:
        libs[base] = capi.load(os.path.join(REPO, "build", "libdfusion_hip_%s.so" % base), strict=False)
    if t.endswith("/nosat"):
        nosat[t] = True; libs[t] = libs[base]
FLAGS = {}
def use(t):
    capi._lib = libs[t]
    FLAGS["f"] = capi.DF_RIGID_NO_SAT if t in nosat else 0
cfg = synth.CONFIGS[name]; intr = Intr(*cfg.intr); F = 4
dists = [compute_dists(upload_u16(synth.depth_frame(cfg, f)), intr) for f in range(F)]
cams = [synth.camera_pose(cfg, f) for f in range(F)]
def mkvol():
    v = TsdfVolume(cfg.dims); v.setSize([cfg.size] * 3); v.setTruncDist(cfg.trunc_dist); v.setMaxWeight(cfg.max_weight); v.setPose(cfg.volume_pose)
    v.clear(); return v
ref = None
for t in libs:
    use(t); v = mkvol()
    n = torch.zeros(2, dtype=torch.int64, device="cuda")
    has_cnt = True
    snaps = []
    for f in range(11):
        v.integrate(dists[f % F], cams[f % F], intr, n_updated=n[:1] if f % 2 == 0 else None, n_swept=n[1:] if f % 2 == 0 else None,
                    flags=FLAGS["f"])     # (both the counting and the plain kernels)
        if f in (0, 2, 10): snaps.append(v.data().clone())
    upd, swept = int(n[0]), int(n[1])
    msg = "updated %d" % upd + (", swept %d = %.3f x updated" % (swept, swept / upd) if has_cnt else "")
    if ref is None: ref = (snaps, upd)
    else:
        same = all(torch.equal(a, b) for a, b in zip(snaps, ref[0])) and upd == ref[1]
        msg += "; volumes after 1 / 3 / 11 frames %s product" % ("==" if same else "!=")
    print("%-10s %s" % (t, msg))
    del v, snaps
vol = mkvol()
use("product")
for f in range(8): vol.integrate(dists[f % F], cams[f % F], intr, sync=False)
res = {t: [] for t in libs}
for rnd in range(8):
    for t in libs:
        use(t)
        for i in range(4): vol.integrate(dists[i % F], cams[i % F], intr, sync=False, flags=FLAGS["f"])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20): vol.integrate(dists[i % F], cams[i % F], intr, sync=False, flags=FLAGS["f"])
        e1.record(); torch.cuda.synchronize()
        res[t].append(e0.elapsed_time(e1) / 20)
for t, v in res.items():
    print("%-10s min %.4f  median %.4f  max %.4f ms" % (t, min(v), float(np.median(v)), max(v)))
