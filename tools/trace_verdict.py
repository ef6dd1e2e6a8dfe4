#!/usr/bin/env python3
"""Per-wave timeline of the warped sweep's verdict pass: build/libdfusion_hip_vtrace.so (tools/build_variant.py vtrace --only
dfusion_warp.hip -DDF_TRACE_VERDICT=1) stamps every wave at its start, after the ball test, after the box test and at its end
(100 MHz clock); prints where the launch's time goes.  tools/trace_verdict.py [CONFIG]"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from dynamicfusion_amd import capi, build as B
B.LIB_PATH = os.path.join(REPO, "build", "libdfusion_hip_vtrace.so"); B._stale = lambda: False
from dynamicfusion_amd import Intr, TsdfVolume, WarpField, compute_dists, synth, upload_u16
name = sys.argv[1] if len(sys.argv) > 1 else "512"
cfg = synth.CONFIGS[name]; intr = Intr(*cfg.intr)
vol = TsdfVolume(cfg.dims); vol.setSize([cfg.size] * 3); vol.setTruncDist(cfg.trunc_dist); vol.setMaxWeight(cfg.max_weight); vol.setPose(cfg.volume_pose)
pos, sigma = synth.make_nodes(cfg)
wf = WarpField(k=cfg.k); wf.init(pos, sigma=sigma, transforms=synth.node_transforms(cfg, 0)); wf.ensure_index(vol, cfg.k)
path = os.path.join(REPO, "gpurun_out", "verdict_trace.bin"); os.makedirs(os.path.dirname(path), exist_ok=True)
for f in range(14):                                    # the bench's monotone sweep; the 14th frame is traced
    dists = compute_dists(upload_u16(synth.depth_frame(cfg, f)), intr)
    wf.set_transforms(torch.from_numpy(synth.node_transforms(cfg, f)).cuda())
    if f == 13: os.environ["DF_TRACE_VERDICT_FILE"] = path
    vol.integrate_warped(dists, synth.camera_pose(cfg, f), intr, wf)
del os.environ["DF_TRACE_VERDICT_FILE"]
t = np.fromfile(path, dtype=np.uint64).reshape(-1, 4).astype(np.int64); os.remove(path)
t = t[t[:, 3] > 0]
t0 = t[:, 0].min(); t = (t - t0) * 0.01
print("%s: %d waves, launch %.1f us (first wave start to last wave end); last wave STARTS at %.1f us" % (name, len(t), t[:, 3].max(), t[:, 0].max()))
life = t[:, 3] - t[:, 0]
print("wave lifetime: mean %.2f  median %.2f  p90 %.2f  p99 %.2f  max %.2f us" % (life.mean(), np.median(life), np.percentile(life, 90), np.percentile(life, 99), life.max()))
ball, box, tail = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
long_ = life > np.percentile(life, 90)
for nm, v in (("start -> ball test done", ball), ("ball -> box test done", box), ("box -> end (lists, atomics)", tail)):
    print("  %-28s all waves mean %.2f max %.2f | slowest 10%% mean %.2f us" % (nm, v.mean(), v.max(), v[long_].mean()))
T = np.linspace(0, t[:, 3].max(), 14)
for a, b in zip(T[:-1], T[1:]):
    print("  %5.1f-%5.1f us: %5d waves resident, %4d start, %4d end" % (a, b, int(((t[:, 0] < b) & (t[:, 3] > a)).sum()), int(((t[:, 0] >= a) & (t[:, 0] < b)).sum()), int(((t[:, 3] >= a) & (t[:, 3] < b)).sum())))
