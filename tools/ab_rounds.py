#!/usr/bin/env python3
"""Same-box, interleaved A/B of the warped integrate between the product library and OTHER BUILDS of the library whose handle layout may
differ (an earlier round's sources compiled into build/libdfusion_hip_TAG.so): every library gets its own warp-field handle and volume.

    tools/ab_rounds.py CONFIG [--nodes M] TAG [TAG ...]

Reports (i) bit identity of the volume after 4 moving frames, (ii) the steady state on one pose (HIP events over 20 calls, 6 interleaved
rounds), (iii) the bench's moving camera: 24 consecutive poses, per-library mean of the integrate call."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from dynamicfusion_amd import capi
from dynamicfusion_amd import Intr, TsdfVolume, WarpField, compute_dists, synth, upload_u16

argv = sys.argv[1:]
name = argv.pop(0)
nodes = None
if argv and argv[0] == "--nodes":
    argv.pop(0); nodes = int(argv.pop(0))
tags = argv
libs = {"product": capi.lib()}
for t in tags:
    libs[t] = capi.load(os.path.join(REPO, "build", "libdfusion_hip_%s.so" % t), strict=False)


def use(t):
    capi._lib = libs[t]


base = synth.CONFIGS[name]
cfg = base if nodes is None else synth.Config(base.dims[0], base.size, cols=base.cols, rows=base.rows, nodes=nodes, k=base.k)
intr = Intr(*cfg.intr)
pos, sigma = synth.make_nodes(cfg)
NF = 28
use("product")
dists = [compute_dists(upload_u16(synth.depth_frame(cfg, f)), intr) for f in range(NF)]
dqs = [torch.from_numpy(synth.node_transforms(cfg, f)).cuda() for f in range(NF)]
cams = [synth.camera_pose(cfg, f) for f in range(NF)]


def mkvol():
    v = TsdfVolume(cfg.dims); v.setSize([cfg.size] * 3); v.setTruncDist(cfg.trunc_dist); v.setMaxWeight(cfg.max_weight); v.setPose(cfg.volume_pose)
    return v


state = {}
for t in libs:
    use(t)
    wf = WarpField(k=cfg.k); wf.init(pos, sigma=sigma, transforms=synth.node_transforms(cfg, 0))
    state[t] = (wf, mkvol())

# (i) bit identity after 4 moving frames
ref = None
for t in libs:
    use(t); wf, v = state[t]
    for f in range(4):
        wf.set_transforms(dqs[f]); v.integrate_warped(dists[f], cams[f], intr, wf)
    d = v.data().clone()
    if ref is None: ref = d
    else: print("%-10s volume %s product (%d words differ)" % (t, "==" if torch.equal(d, ref) else "!=", int((d != ref).sum())))
del ref

# (ii) one pose, steady state
res = {t: [] for t in libs}
for rnd in range(6):
    for t in libs:
        use(t); wf, v = state[t]
        wf.set_transforms(dqs[4])
        for _ in range(3): v.integrate_warped(dists[4], cams[4], intr, wf, sync=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): v.integrate_warped(dists[4], cams[4], intr, wf, sync=False)
        e1.record(); torch.cuda.synchronize()
        res[t].append(e0.elapsed_time(e1) / 20)
for t, r in res.items():
    print("static pose  %-10s min %.4f  median %.4f  max %.4f ms" % (t, min(r), float(np.median(r)), max(r)))

# (iii) the moving camera: poses 4 .. NF-1, one integrate each, events around every call; three passes per library, interleaved
mov = {t: [] for t in libs}
for rnd in range(3):
    for t in libs:
        use(t); wf, v = state[t]
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(NF)]
        for f in range(4, NF):
            wf.set_transforms(dqs[f])
            ev[f][0].record(); v.integrate_warped(dists[f], cams[f], intr, wf, sync=False); ev[f][1].record()
        torch.cuda.synchronize()
        mov[t].append(float(np.mean([ev[f][0].elapsed_time(ev[f][1]) for f in range(8, NF)])))
for t, r in mov.items():
    print("moving poses %-10s mean integrate per frame over 20 poses: %s ms" % (t, " ".join("%.4f" % x for x in r)))
