#!/usr/bin/env python3
"""The frame after a node-set change, for profiling (bench.py's frame_nodes_changed_ms): set_nodes + index + the frame (which, with
on-demand tables, builds the tables of the blocks its launch plan finds alive), then the next two frames timed back to back with
events on the caller's stream (no device-wide synchronisation in between: the third frame's wait for the side stream's model builds is
inside its time, as in a running pipeline).  Usage: tools/nodes_changed.py [CONFIG] [REPEATS] [eager] [--lib TAG]
(TAG: build/libdfusion_hip_TAG.so, tools/build_variant.py)"""
import os, sys, time
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from dynamicfusion_amd import Intr, TsdfVolume, WarpField, capi, compute_dists, synth, upload_u16  # noqa: E402
if "--lib" in sys.argv:
    i = sys.argv.index("--lib"); tag = sys.argv[i + 1]; del sys.argv[i:i + 2]
    capi._lib = capi.load(os.path.join(REPO, "build", "libdfusion_hip_%s.so" % tag), strict=False)

name = sys.argv[1] if len(sys.argv) > 1 else "512"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
on_demand = (sys.argv[3] != "eager") if len(sys.argv) > 3 else True
cfg = synth.CONFIGS[name]; intr = Intr(*cfg.intr)
vol = TsdfVolume(cfg.dims); vol.setSize([cfg.size] * 3); vol.setTruncDist(cfg.trunc_dist); vol.setMaxWeight(cfg.max_weight); vol.setPose(cfg.volume_pose)
vol.setRaycastStepFactor(cfg.raycast_step_factor); vol.setGradientDeltaFactor(cfg.gradient_delta_factor); vol.clear()
pos, sigma = synth.make_nodes(cfg)
wf = WarpField(k=cfg.k, tables_on_demand=on_demand); wf.init(pos, sigma=sigma, transforms=synth.node_transforms(cfg, 0))
dists = compute_dists(upload_u16(synth.depth_frame(cfg, 0)), intr)
cam = synth.camera_pose(cfg, 0)
pts = torch.empty((cfg.rows, cfg.cols, 4), dtype=torch.float32, device="cuda"); nrm = torch.empty_like(pts)
for _ in range(3):
    vol.integrate_warped(dists, cam, intr, wf, sync=False)
ms = []
for _ in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    wf.set_nodes(*wf._keep)
    wf.ensure_index(vol, cfg.k)
    t1 = time.perf_counter()
    vol.integrate_warped(dists, cam, intr, wf, sync=False)
    vol.raycast(cam, intr, pts, nrm)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ms.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3))
    # the next frames: models for the alive blocks (side stream, beside the second frame's sweep), then the steady state
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    for f in range(2):
        vol.integrate_warped(dists, cam, intr, wf, sync=False)
        ev[f + 1].record()
    torch.cuda.synchronize()
    ms.append((0.0, ev[0].elapsed_time(ev[1]))); ms.append((0.0, ev[1].elapsed_time(ev[2])))
print("set_nodes + index / frame (ms):", " ".join("%.2f/%.2f" % m for m in ms))
